#!/usr/bin/env python3
"""configs[3]: where does the PIR-AT outer step spend its wall time?  Host time to ENQUEUE the inner attack and the outer training
step vs the GPU time (HIP events) of each, bf16 autocast, UperNet-ConvNeXt-S, C = 151, B = 8, 512 x 512, 5 inner PGD steps.
    python devtools/pirat_host_vs_gpu.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd")]
import torch  # noqa: E402
import yaml  # noqa: E402

from semseg.models import UperNetForSemanticSegmentation  # noqa: E402
from tools.train_rob_seg import build_attack  # noqa: E402

cfg = yaml.safe_load(open(os.path.join(ROOT, "robust-segmentation_amd", "configs", "ade20k_convnext.yaml")))
cfg["TRAIN"].update(N_ITERS=5)
C, B = 151, 8
torch.manual_seed(0)
torch.backends.cudnn.benchmark = True
model = UperNetForSemanticSegmentation("ConvNeXt-S_CVST", C, None).cuda()
opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9)
attack = build_attack(dict(cfg["TRAIN"], N_CLS=C))
g = torch.Generator().manual_seed(1)
img = torch.rand(B, 3, 512, 512, generator=g).cuda()
lbl = torch.randint(0, C, (B, 16, 16), generator=g).repeat_interleave(32, 1).repeat_interleave(32, 2).cuda()
rows = []
for i in range(8):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    torch.cuda.synchronize()
    opt.zero_grad(set_to_none=True)
    t0 = time.perf_counter()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        model.eval()
        ev[0].record()
        adv = attack(model, img, lbl)
        ev[1].record()
        t1 = time.perf_counter()
        model.train()
        ev[2].record()
        loss, _ = model(adv, lbl)
    loss.backward()
    opt.step()
    ev[3].record()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    rows.append((1e3 * (t1 - t0), ev[0].elapsed_time(ev[1]), 1e3 * (t2 - t1), ev[2].elapsed_time(ev[3]), 1e3 * (t3 - t0)))
for r in rows[3:]:
    print(f"inner attack: host enqueue {r[0]:6.1f} ms, GPU {r[1]:6.1f} ms | outer step: host enqueue {r[2]:6.1f} ms, GPU {r[3]:6.1f} ms | wall {r[4]:6.1f} ms")
