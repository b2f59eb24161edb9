#!/usr/bin/env python3
"""Diagnostic: which convolutions receive non-contiguous gradients in the dx-backward of UperNet
(MIOpen falls back to its naive kernel for non-packed tensors), and how fast is the depthwise kernel."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd"), os.path.join(ROOT, "robust-segmentation_amd", "tools")]
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from semseg import _native as N  # noqa: E402
from semseg.models import UperNetForSemanticSegmentation  # noqa: E402
from semseg.models import convnext_upernet as M  # noqa: E402


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


torch.backends.cudnn.benchmark = True
for C, hw in ((96, 128), (192, 64), (384, 32), (768, 16)):
    x = torch.randn(8, C, hw, hw, device="cuda")
    w = torch.randn(C, 1, 7, 7, device="cuda") * 0.1
    b = torch.randn(C, device="cuda")
    t_hip = timed(lambda: N.dwconv7x7(x, w, b))
    t_hipb = timed(lambda: N.dwconv7x7(x, w, None, flip=True))
    t_lib = timed(lambda: F.conv2d(x, w, b, padding=3, groups=C))
    mb = 2 * x.numel() * 4 / 1e6
    print(f"dwconv C={C} {hw}x{hw}: hip fwd {t_hip * 1e3:7.1f} us ({mb / t_hip / 1e3:6.2f} TB/s)  hip bwd-data {t_hipb * 1e3:7.1f} us   "
          f"MIOpen fwd {t_lib * 1e3:7.1f} us", flush=True)

torch.manual_seed(0)
model = UperNetForSemanticSegmentation("ConvNeXt-T_CVST", 21, None).eval().cuda()
for p in model.parameters():
    p.requires_grad_(False)
seen = []
for name, m in model.named_modules():
    if isinstance(m, torch.nn.Conv2d):
        def hook(mod, gi, go, name=name):
            g = go[0]
            if g is not None and not g.is_contiguous():
                seen.append((name, tuple(g.shape), tuple(g.stride())))
        m.register_full_backward_hook(hook)
x = torch.rand(2, 3, 256, 256, device="cuda", requires_grad=True)
out = model(x)
torch.autograd.grad(out, x, torch.ones_like(out))
print("convs with NON-CONTIGUOUS grad_output:")
for s in seen:
    print("  ", s)

x = torch.rand(8, 3, 512, 512, device="cuda")
def fwdbwd():
    a = x.detach().requires_grad_(True)
    o = model(a)
    return torch.autograd.grad(o, a, torch.ones_like(o))
for flag in (True, False):
    M.USE_HIP_DWCONV = flag
    print(f"UperNet-T B=8 512^2 fwd+dx-bwd, HIP dwconv={flag}: {timed(fwdbwd, n=5, warm=3):.2f} ms", flush=True)
