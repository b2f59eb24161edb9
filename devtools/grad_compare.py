#!/usr/bin/env python3
"""Diagnostic: input gradients of UperNet-ConvNeXt-T through the model-side fast paths (Winograd M4, fused FPN
bottleneck M6, folded BatchNorm) against the MIOpen / ATen composition, interior vs image border."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd"), os.path.join(ROOT, "robust-segmentation_amd", "tools")]
import torch  # noqa: E402

from semseg.models import UperNetForSemanticSegmentation  # noqa: E402
from semseg.models import convnext_upernet as M  # noqa: E402


def main():
    torch.manual_seed(0)
    model = UperNetForSemanticSegmentation("ConvNeXt-T_CVST", 21, None).eval().cuda()
    for p in model.parameters():
        p.requires_grad_(False)
    B = 4
    x = torch.rand(B, 3, 512, 512, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(1)
    dl = torch.randn(B, 21, 512, 512, device="cuda", generator=g)
    dl = dl * (torch.rand(B, 1, 512, 512, device="cuda", generator=g) < 0.3)  # masked pixels as in mask-ce

    def grad(tile, m6, pw=True, ln=True, nhwc=True):
        M.WINOGRAD_TILE, M.USE_FUSED_FPN_BOTTLENECK, M.USE_GEMM_POINTWISE = tile, m6, pw
        M.USE_HIP_LAYERNORM, M.USE_HIP_UPSAMPLE_NHWC = ln, nhwc
        xi = x.clone().requires_grad_(True)
        y = model(xi)
        (gx,) = torch.autograd.grad(y, xi, dl)
        return y.detach(), gx

    y0, g0 = grad(0, False, False, False, False)
    border = torch.zeros(512, 512, dtype=torch.bool, device="cuda")
    border[:16] = border[-16:] = True
    border[:, :16] = border[:, -16:] = True
    for name, cfg in (("MIOpen again (noise floor)", (0, False, False, False, False)),
                      ("NHWC head + LN + pointwise, no Winograd", (0, False, True, True, True)),
                      ("Winograd F(2x2), no M6", (2, False)), ("Winograd F(4x4), no M6", (4, False)),
                      ("Winograd F(2x2) + M6", (2, True)), ("Winograd F(4x4) + M6 (default)", (4, True))):
        y, gx = grad(*cfg)
        d = gx - g0
        rel = (d.norm() / g0.norm()).item()
        relb = (d[..., border].norm() / g0[..., border].norm()).item()
        reli = (d[..., ~border].norm() / g0[..., ~border].norm()).item()
        sign = ((gx.sign() != g0.sign()).float().mean()).item()
        print(f"{name:42s} logits max|d| {((y - y0).abs().max() / y0.abs().max()).item():.2e}  grad rel-L2 {rel:.2e} "
              f"(border {relb:.2e}, interior {reli:.2e})  max|d|/max|g| {(d.abs().max() / g0.abs().max()).item():.2e}  "
              f"sign flips {sign:.4%}", flush=True)


if __name__ == "__main__":
    main()
