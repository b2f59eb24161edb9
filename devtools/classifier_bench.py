#!/usr/bin/env python3
"""M9 micro-benchmark: streaming classifier kernels vs the library GEMM the model used before (B=8, 512 ch, 128^2)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd")]
import torch  # noqa: E402

from semseg import _native as N  # noqa: E402


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for B, Cin, H, W, cls in ((8, 512, 128, 128, 21),):
    y = torch.randn(B, Cin, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(cls, Cin, device="cuda") * 0.05
    b = torch.randn(cls, device="cuda")
    g = torch.randn(B, cls, H, W, device="cuda")
    y2 = y.permute(0, 2, 3, 1).reshape(B, H * W, Cin)
    nbytes = y.numel() * 4
    t = timeit(lambda: N.classifier_fwd(y, w, b))
    print(f"fwd own      {t:7.1f} us = {nbytes / t / 1e6:.2f} TB/s")
    t = timeit(lambda: torch.matmul(w, y2.transpose(1, 2)))
    print(f"fwd matmul   {t:7.1f} us = {nbytes / t / 1e6:.2f} TB/s")
    t = timeit(lambda: N.classifier_bwd(g, w, Cin))
    print(f"bwd own      {t:7.1f} us = {nbytes / t / 1e6:.2f} TB/s")
    t = timeit(lambda: torch.matmul(g.reshape(B, cls, H * W).transpose(1, 2), w))
    print(f"bwd matmul   {t:7.1f} us = {nbytes / t / 1e6:.2f} TB/s")
