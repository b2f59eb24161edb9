#!/usr/bin/env python3
"""Diagnostic: memory layouts inside the UperNet head, A/B of the NHWC up-sampling kernels, and the ATen ops
(with input shapes) that account for the non-convolution time of one forward + input-gradient backward."""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd"), os.path.join(ROOT, "robust-segmentation_amd", "tools")]
import torch  # noqa: E402

from semseg.models import UperNetForSemanticSegmentation  # noqa: E402
from semseg.models import convnext_upernet as M  # noqa: E402

torch.backends.cudnn.benchmark = True


def layout(t):
    if not torch.is_tensor(t) or t.dim() != 4:
        return "-"
    if t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last):
        return "both"
    if t.is_contiguous():
        return "NCHW"
    if t.is_contiguous(memory_format=torch.channels_last):
        return "NHWC"
    return "strided" + str(tuple(t.stride()))


def step(model, x):
    xi = x.clone().requires_grad_(True)
    y = model(xi)
    (g,) = torch.autograd.grad(y.sum(), xi)
    return g


def timed(fn, n=8, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    torch.manual_seed(0)
    model = UperNetForSemanticSegmentation("ConvNeXt-T_CVST", 21, None).eval().cuda()
    for p in model.parameters():
        p.requires_grad_(False)
    x = torch.rand(8, 3, 512, 512, device="cuda")

    hooks = []
    for name, mod in model.decode_head.named_modules():
        if name and len(list(mod.children())) == 0:
            def hook(m, inp, out, name=name):
                print(f"  {name:38s} in {layout(inp[0]):8s} {tuple(inp[0].shape)!s:24s} -> out {layout(out):8s} {tuple(out.shape)}")
            hooks.append(mod.register_forward_hook(hook))
    with torch.no_grad():
        feats = model.backbone(x)
        print("backbone features:", [(tuple(f.shape), layout(f)) for f in feats])
        out = model.decode_head(feats)
        print("head output:", tuple(out.shape), layout(out))
    for h in hooks:
        h.remove()

    for nhwc, tile in ((True, 4), (True, 2), (True, 4)):
        M.USE_HIP_UPSAMPLE_NHWC, M.WINOGRAD_TILE = nhwc, tile  # nhwc=False also disables the fused cat/add
        print(f"NHWC head={nhwc} winograd tile={tile}: {timed(lambda: step(model, x)):.2f} ms / forward+backward",
              flush=True)
    M.USE_HIP_UPSAMPLE_NHWC, M.WINOGRAD_TILE = True, 4
    with torch.no_grad():
        M.WINOGRAD_TILE = 0
        ref = model(x)
        for tile in (2, 4):
            M.WINOGRAD_TILE = tile
            d = (model(x) - ref).abs().max().item()
            print(f"winograd tile={tile}: max |logit diff| vs MIOpen = {d:.3e} (logit scale {ref.abs().max().item():.3f})")
    M.WINOGRAD_TILE = 4

    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for _ in range(3):
            step(model, x)
        torch.cuda.synchronize()
    seen = {}
    for e in prof.events():  # who issues the big layout / slice copies and the device-to-device memcpys?
        big = (e.name in ("aten::copy_", "aten::contiguous", "aten::clone") and e.input_shapes and e.input_shapes[0]
               and len(e.input_shapes[0]) >= 2 and math.prod(e.input_shapes[0]) >= (1 << 22))
        if big or ("emcpy" in e.name and e.cpu_parent is not None):
            chain, q = [], e.cpu_parent
            while q is not None and len(chain) < 5:
                chain.append(q.name + (str(q.input_shapes[0]) if q.input_shapes else ""))
                q = q.cpu_parent
            k = (e.name, str(e.input_shapes[0]) if e.input_shapes else "", " <- ".join(chain))
            seen[k] = seen.get(k, 0) + 1
    for k, v in seen.items():
        print(f"{v:3d}x {k[0]} {k[1]} <- {k[2]}")
    print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=45,
                                                              max_name_column_width=48, max_shapes_column_width=70))


if __name__ == "__main__":
    main()
