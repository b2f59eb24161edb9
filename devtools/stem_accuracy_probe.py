"""Accuracy of the stem (fused M9 kernels vs the library path) against float64 on the real model's weights.
   gpurun -- python devtools/stem_accuracy_probe.py"""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "robust-segmentation_amd"))
from semseg.models import UperNetForSemanticSegmentation, convnext_upernet as M  # noqa: E402

torch.manual_seed(0)
model = UperNetForSemanticSegmentation("ConvNeXt-S_CVST", 151, None).eval().cuda()
stem = model.backbone.downsample_layers[0]
for p in stem.parameters():
    p.requires_grad_(False)
ref = copy.deepcopy(stem).double()
g = torch.Generator(device="cuda").manual_seed(1)
for name, x in (("uniform image", torch.rand(2, 3, 512, 512, device="cuda", generator=g)),
                ("8-bit image", torch.randint(0, 256, (2, 3, 512, 512), device="cuda", generator=g).float() / 255),
                ("smooth image", torch.nn.functional.interpolate(torch.rand(2, 3, 32, 32, device="cuda", generator=g), size=512,
                                                                 mode="bilinear"))):
    for gname, scale in (("white gradient", None), ("sparse gradient", 0.999)):
        da = torch.randn(2, 96, 128, 128, device="cuda", generator=g)
        if scale is not None:
            da = da * (torch.rand(2, 1, 128, 128, device="cuda", generator=g) > scale)
        xd = x.double().requires_grad_(True)
        od = ref(xd)
        (gd,) = torch.autograd.grad(od, xd, da.double())
        row = []
        for fused in (True, False):
            old, M.USE_FUSED_STEM = M.USE_FUSED_STEM, fused
            try:
                xs = x.clone().requires_grad_(True)
                o = stem(xs)
                (gx,) = torch.autograd.grad(o, xs, da.contiguous(memory_format=torch.channels_last))
            finally:
                M.USE_FUSED_STEM = old
            row.append(((o.double() - od).norm() / od.norm()).item())
            row.append(((gx.double() - gd).norm() / gd.norm()).item())
        print(f"{name:14s} {gname:16s}: fused out {row[0]:.2e} grad {row[1]:.2e}   library out {row[2]:.2e} grad {row[3]:.2e}")
