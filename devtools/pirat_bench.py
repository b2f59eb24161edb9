#!/usr/bin/env python3
"""PIR-AT outer step as BASELINE configs[3] writes it: UperNet-ConvNeXt-S, ADE20K-shaped (C=151), 5-step CE PGD
inner attack, batch 8 per GPU, fp32 vs bf16 autocast.   python devtools/pirat_bench.py"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "robust-segmentation_amd")
sys.path[:0] = [ROOT, PKG]
import yaml  # noqa: E402

from tools import train_rob_seg  # noqa: E402

td = tempfile.mkdtemp()
cfg = yaml.safe_load(open(os.path.join(PKG, "configs", "ade20k_convnext.yaml")))
cfg["MODEL"]["BACKBONE"] = "ConvNeXt-S_CVST"
cfg["TRAIN"].update(N_ITERS=int(os.environ.get("PIRAT_N_ITERS", "5")), BATCH_SIZE=8)
yaml.safe_dump(cfg, open(td + "/cfg.yaml", "w"))
modes = {"fp32": [[]], "bf16": [["--bf16"]]}.get(os.environ.get("PIRAT_MODE", ""), [[], ["--bf16"]])
for flags in modes:
    out = td + "/o.json"
    train_rob_seg.main(["--cfg", td + "/cfg.yaml", "--synthetic", "16", "--steps", "6", "--warmup", "2", "--batch_size", "8", "--json", out] + flags)
