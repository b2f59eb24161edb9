#!/usr/bin/env python3
"""Where does the model's forward / input-gradient time go on MI355X?  (diagnostic, not a test)

Times UperNet-ConvNeXt-T fwd and fwd+dx-bwd at B=8, 512x512 in NCHW and channels_last, and a few
individual conv shapes (forward and backward-data) in both memory formats."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd"), os.path.join(ROOT, "robust-segmentation_amd", "tools")]
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

torch.backends.cudnn.benchmark = True
dev = "cuda"


def timed(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def conv_case(name, cin, cout, k, stride, pad, groups, hw, B=8):
    for fmt_name, fmt in (("nchw", torch.contiguous_format), ("nhwc", torch.channels_last)):
        x = torch.randn(B, cin, hw, hw, device=dev).contiguous(memory_format=fmt).requires_grad_(True)
        w = torch.randn(cout, cin // groups, k, k, device=dev).contiguous(memory_format=fmt) * 0.02
        y = F.conv2d(x, w, None, stride, pad, 1, groups)
        gy = torch.randn_like(y)
        f = timed(lambda: F.conv2d(x, w, None, stride, pad, 1, groups))
        b = timed(lambda: torch.autograd.grad(F.conv2d(x, w, None, stride, pad, 1, groups), x, gy)) - f
        fl = 2 * B * cout * (cin // groups) * k * k * y.shape[2] * y.shape[3] / 1e12
        print(f"{name:34s} {fmt_name}  fwd {f:7.3f} ms ({fl / f * 1e3:6.1f} TF/s)   bwd-data {b:7.3f} ms ({fl / max(b, 1e-6) * 1e3:6.1f} TF/s)",
              flush=True)


def main():
    from semseg.models import UperNetForSemanticSegmentation
    torch.manual_seed(0)
    x = torch.rand(8, 3, 512, 512, device=dev)
    for fmt_name, fmt in (("nchw", torch.contiguous_format), ("channels_last", torch.channels_last)):
        model = UperNetForSemanticSegmentation("ConvNeXt-T_CVST", 21, None).eval().to(dev).to(memory_format=fmt)
        for p in model.parameters():
            p.requires_grad_(False)
        xi = x.contiguous(memory_format=fmt)

        def fwd():
            with torch.no_grad():
                return model(xi)

        def fwdbwd():
            a = xi.detach().requires_grad_(True)
            out = model(a)
            return torch.autograd.grad(out, a, torch.ones_like(out))

        tf = timed(fwd, n=5, warm=3)
        tb = timed(fwdbwd, n=5, warm=3)
        out = fwd()
        print(f"UperNet-ConvNeXt-T B=8 512^2 {fmt_name}: fwd {tf:.2f} ms, fwd+dx-bwd {tb:.2f} ms, logits strides {out.stride()}", flush=True)
        del model
    conv_case("stem conv0 3->48 3x3 s2", 3, 48, 3, 2, 1, 1, 512)
    conv_case("stem conv1 48->96 3x3 s2", 48, 96, 3, 2, 1, 1, 256)
    conv_case("downsample 96->192 2x2 s2", 96, 192, 2, 2, 0, 1, 128)
    conv_case("downsample 192->384 2x2 s2", 192, 384, 2, 2, 0, 1, 64)
    conv_case("downsample 384->768 2x2 s2", 384, 768, 2, 2, 0, 1, 32)
    conv_case("dwconv 96 7x7 @128", 96, 96, 7, 1, 3, 96, 128)
    conv_case("dwconv 192 7x7 @64", 192, 192, 7, 1, 3, 192, 64)
    conv_case("dwconv 384 7x7 @32", 384, 384, 7, 1, 3, 384, 32)
    conv_case("dwconv 768 7x7 @16", 768, 768, 7, 1, 3, 768, 16)
    conv_case("fpn_bottleneck 2048->512 3x3 @128", 2048, 512, 3, 1, 1, 1, 128)
    conv_case("fpn_conv 512->512 3x3 @128", 512, 512, 3, 1, 1, 1, 128)
    conv_case("fpn_conv 512->512 3x3 @64", 512, 512, 3, 1, 1, 1, 64)
    conv_case("bottleneck 2816->512 3x3 @16", 2816, 512, 3, 1, 1, 1, 16)
    conv_case("lateral 96->512 1x1 @128", 96, 512, 1, 1, 0, 1, 128)
    conv_case("classifier 512->21 1x1 @128", 512, 21, 1, 1, 0, 1, 128)


if __name__ == "__main__":
    main()
