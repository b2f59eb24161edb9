#!/usr/bin/env python3
"""Launch only the attack-side kernels a few times (for `rocprofv3 --pmc ...` passes).
Cases: K2 fp32 C=21 (16 B/lane) and C=151 (4 B/lane) with gradient, B=8, 512x512, uint8 labels/pred;
K1; plus a plain 16 B/lane device copy of known size for calibrating FETCH_SIZE / WRITE_SIZE."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd"), os.path.join(ROOT, "robust-segmentation_amd", "tools")]
import torch  # noqa: E402

from semseg import _native as N  # noqa: E402
from kernel_bench import loss_case  # noqa: E402

N.lib()
B, H, W = 8, 512, 512
HW = H * W
for C in (21, 151):
    logits, y, w = loss_case(B, C, H, W, torch.float32)
    y8 = y.to(torch.uint8)
    dl = torch.empty_like(logits)
    pred = torch.empty(B, H, W, dtype=torch.uint8, device="cuda")
    ws = N.loss_workspace(B, HW, "cuda")
    for _ in range(5):
        N.loss_fwd_bwd(logits, y8, w, 1, 3, 1.0 / HW, True, pred=pred, workspace=ws, dlogits=dl)
    torch.cuda.synchronize()
    del logits, dl
x = torch.rand(B, 3, H, W, device="cuda")
xa, xo, gr = torch.rand_like(x), torch.rand_like(x), torch.rand_like(x) - 0.5
out = torch.empty_like(x)
step = torch.full((B,), 16 / 255, device="cuda")
for _ in range(5):
    N.apgd_linf_step(x, xa, xo, gr, step, 8 / 255, 0.75, out=out)
# calibration: K5 reads 2 tensors and writes 1 with 16 B/lane: known 3 * 25.17 MB
for _ in range(5):
    N.linf_project(xa, x, 8 / 255, out=out)
torch.cuda.synchronize()
print("pmc cases done")
