#!/usr/bin/env python3
"""Is tools.infer on UperNet-ConvNeXt-S / C=151 (4 synthetic 512^2 images, batches of 2, 5 iterations) the same table from run
to run (fresh processes), and which switch makes it so?   python devtools/shard_flake.py"""
import os
import subprocess
import sys
import tempfile

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "robust-segmentation_amd")
td = tempfile.mkdtemp()
cfg = yaml.safe_load(open(os.path.join(PKG, "configs", "ade20k_convnext.yaml")))
cfg["SAVE_DIR"] = td + "/"
cfg["MODEL"]["BACKBONE"] = cfg["EVAL"]["BACKBONE"] = "ConvNeXt-S_CVST"
yaml.safe_dump(cfg, open(td + "/cfg.yaml", "w"))
common = ["--cfg", td + "/cfg.yaml", "--eps", "8", "--n_iter", "5", "--synthetic", "4", "--image_size", "512", "--batch_size", "2",
          "--cleanup", "0", "--deterministic"]


def run(tag, env, order=None):
    out = f"{td}/{tag}.pt"
    e = dict(os.environ, PYTHONPATH=PKG + os.pathsep + ROOT, **env)
    r = subprocess.run([sys.executable, "-m", "tools.infer"] + common + ["--dump_stats", out] + (order or []), cwd=PKG, env=e,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return torch.load(out)


for name, env in (("shipped", {}), ("K2u off", {"SEA_FUSE_UPSAMPLE": "0"}), ("eager loop", {"SEA_HIP_GRAPH": "0"}),
                  ("K2u off + eager", {"SEA_FUSE_UPSAMPLE": "0", "SEA_HIP_GRAPH": "0"}), ("PSP on library", {"SEA_WINO_SPLIT_MIN_TILES": "32"}),
                  ("plain dwconv", {"SEA_DWCONV_AB": "16"}), ("wino 2ch", {"SEA_WINO_IN_VEC4": "0"}), ("tap general", {"SEA_TAP_INNER": "0"}),
                  ("MLP unfused", {"SEA_MLP_FUSED": "0"})):
    tabs = [run(f"{name.replace(' ', '_')}_{i}", env) for i in range(3)]
    same = [torch.equal(tabs[0], t) for t in tabs[1:]]
    nd = [(tabs[0] != t).sum().item() for t in tabs[1:]]
    print(f"{name:20s} run-to-run identical: {same}  differing entries: {nd}", flush=True)
