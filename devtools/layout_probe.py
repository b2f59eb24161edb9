#!/usr/bin/env python3
"""Diagnostic: memory layouts seen by the ConvNeXt blocks and the effect of forcing NCHW at stage entry."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd"), os.path.join(ROOT, "robust-segmentation_amd", "tools")]
import torch  # noqa: E402

from semseg.models import UperNetForSemanticSegmentation  # noqa: E402
from semseg.models import convnext_upernet as M  # noqa: E402

torch.backends.cudnn.benchmark = True


def timed(fn, n=5, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


torch.manual_seed(0)
model = UperNetForSemanticSegmentation("ConvNeXt-T_CVST", 21, None).eval().cuda()
for p in model.parameters():
    p.requires_grad_(False)
seen = {}
for name, m in model.named_modules():
    if isinstance(m, M.Block):
        def pre(mod, inp, name=name):
            x = inp[0]
            seen[name] = ("nchw" if x.is_contiguous() else "nhwc" if x.is_contiguous(memory_format=torch.channels_last) else "other")
        m.register_forward_pre_hook(pre)
x = torch.rand(8, 3, 512, 512, device="cuda")


def fwdbwd():
    a = x.detach().requires_grad_(True)
    o = model(a)
    return torch.autograd.grad(o, a, torch.ones_like(o))


for mode in ("as-is", "stage-entry-contiguous"):
    M.STAGE_ENTRY_CONTIGUOUS = mode != "as-is"
    for tr in (True, False):
        M.USE_HIP_TRANSPOSE = tr
        seen.clear()
        t = timed(fwdbwd)
        lay = {}
        for k, v in seen.items():
            lay[v] = lay.get(v, 0) + 1
        print(f"{mode:24s} hip_transpose={tr}: fwd+dx-bwd {t:.2f} ms; block input layouts {lay}", flush=True)
