#!/usr/bin/env python3
"""torch.profiler view of the bf16 PIR-AT outer step (configs[3]): which ATen ops (with input shapes) still run on the
device, sorted by device time (debug aid for the fp32 islands under autocast)"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "robust-segmentation_amd")
sys.path[:0] = [ROOT, PKG]
import torch  # noqa: E402
import yaml  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from tools import train_rob_seg  # noqa: E402

td = tempfile.mkdtemp()
cfg = yaml.safe_load(open(os.path.join(PKG, "configs", "ade20k_convnext.yaml")))
cfg["MODEL"]["BACKBONE"] = "ConvNeXt-S_CVST"
cfg["TRAIN"].update(N_ITERS=5, BATCH_SIZE=8)
yaml.safe_dump(cfg, open(td + "/cfg.yaml", "w"))
args = ["--cfg", td + "/cfg.yaml", "--synthetic", "16", "--steps", "2", "--warmup", "2", "--batch_size", "8", "--json", td + "/o.json", "--bf16"]
train_rob_seg.main(args)                     # warm (MIOpen Find etc.)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    train_rob_seg.main(["--cfg", td + "/cfg.yaml", "--synthetic", "16", "--steps", "2", "--warmup", "0", "--batch_size", "8",
                        "--json", td + "/o.json", "--bf16"])
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=48,
                                                         max_shapes_column_width=70))
