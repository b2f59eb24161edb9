"""Depthwise 7x7 weight gradient at the four stage shapes of ConvNeXt (B = 8, 512^2 input): libsea_hip M1w vs the library.
   gpurun -- python devtools/dw_wgrad_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "robust-segmentation_amd"))
from semseg import _native as N  # noqa: E402

N.lib()


def t(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for C, S in ((96, 128), (192, 64), (384, 32), (768, 16)):
    x = torch.randn(8, S, S, C, device="cuda")
    gy = torch.randn(8, S, S, C, device="cuda")
    a = t(lambda: N.dwconv7x7_nhwc_weight_grad(x, gy))
    b = t(lambda: (torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2), (C, 1, 7, 7), gy.permute(0, 3, 1, 2), padding=3, groups=C),
                   gy.sum((0, 1, 2))))
    print(f"C={C:4d} {S}x{S}: libsea_hip {a:7.1f} us   library {b:7.1f} us   ({2 * x.numel() * 4 / 1e6:.0f} MB read)")
