"""Times of the M9 stem kernels at the headline size (8 x 3 x 512^2), 20 launches back to back per event pair.
   gpurun -- python devtools/stem_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "robust-segmentation_amd"))
from semseg import _native as N  # noqa: E402

N.lib()
torch.manual_seed(0)
B, H, W = 8, 512, 512
x = torch.rand(B, 3, H, W, device="cuda")
w = torch.randn(48, 3, 3, 3, device="cuda") * 0.2
b = torch.randn(48, device="cuda") * 0.1
g1, be1 = torch.ones(48, device="cuda"), torch.zeros(48, device="cuda")
g2, be2 = torch.ones(96, device="cuda"), torch.zeros(96, device="cuda")
y1, a1 = N.stem_conv1_ln_gelu(x, w, b, g1, be1)
da1 = torch.randn_like(a1)
y2 = torch.randn(B, 128, 128, 96, device="cuda").permute(0, 3, 1, 2)
da2 = torch.randn_like(y2)
dy1 = N.ln_gelu_cl_backward(da1, y1, g1, be1)


def t(name, fn, mb, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"{name:34s} {us:7.1f} us   {mb:6.0f} MB algorithmic = {mb / us:.2f} TB/s")


MB = 1e6
t("stem_conv1_ln_gelu (3->48, LN, GELU)", lambda: N.stem_conv1_ln_gelu(x, w, b, g1, be1), (x.numel() + 2 * y1.numel()) * 4 / MB)
t("stem_conv1 alone", lambda: N.stem_conv1_ln_gelu(x, w, b), (x.numel() + y1.numel()) * 4 / MB)
t("ln_gelu_cl_bwd<48>", lambda: N.ln_gelu_cl_backward(da1, y1, g1, be1), 3 * y1.numel() * 4 / MB)
t("stem_conv1_bwd", lambda: N.stem_conv1_backward(dy1, w, H, W), (x.numel() + y1.numel()) * 4 / MB)
t("ln_gelu_cl_fwd<96>", lambda: N.ln_gelu_cl(y2, g2, be2), 2 * y2.numel() * 4 / MB)
t("ln_gelu_cl_bwd<96>", lambda: N.ln_gelu_cl_backward(da2, y2, g2, be2), 3 * y2.numel() * 4 / MB)
