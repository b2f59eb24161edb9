#!/usr/bin/env python3
"""A/B of the 128 x 384 one-block-per-CU kernel (sea_gemm_split_pipeline(3)) against the 128 x 128 kernels (0 = single-stage,
1 = ping-pong) on the products of the 32 x 32- and 64 x 64-pixel ConvNeXt stages, with their prologues and split-K, bitwise
comparison.      python devtools/gemm_wide_ab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd")]
import torch  # noqa: E402
from semseg import _native as N  # noqa: E402

CASES = [  # (name, M, K, N, kwargs)
    ("pwconv1 384->1536 @32^2", 8192, 384, 1536, {}),
    ("pwconv2 1536->384 @32^2 (split-K), GELU prologue", 8192, 1536, 384, {"a_gelu": True}),
    ("pwconv2 1536->384 @32^2 (split-K), plain", 8192, 1536, 384, {}),
    ("bwd 1536->384 @32^2 (split-K), GELU' prologue", 8192, 1536, 384, {"gg": True}),
    ("bwd 384->1536 @32^2, per-row scales", 8192, 384, 1536, {"row_amax": True}),
    ("pwconv1 192->768 @64^2", 32768, 192, 768, {}),
    ("pwconv1 768->3072 @16^2", 2048, 768, 3072, {}),
    ("pwconv1 96->384 @128^2", 131072, 96, 384, {}),
]


def timed(fn, reps=30):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


L = N.lib()
for name, M, K, Nn, kw in CASES:
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(Nn, K, device="cuda") / K ** 0.5
    t = torch.randn(M, K, device="cuda")
    Wp = N.gemm_split_pack(W, terms=22)
    kwargs = {k: v for k, v in kw.items() if k != "gg"}
    if kw.get("gg"):
        kwargs["a_gelu_grad_of"] = t
    if "row_amax" not in kwargs:
        kwargs["groups"] = 8
    outs, ts = {}, {}
    for rnd in range(2):
        for pipe in (0, 1, 3):
            L.sea_gemm_split_pipeline(pipe)
            outs[pipe] = torch.empty(M, Nn, device="cuda")
            tt = timed(lambda: N.gemm_split(A, Wp, out=outs[pipe], **kwargs))
            ts[pipe] = min(ts.get(pipe, 1e9), tt)
    same = torch.equal(outs[0], outs[3]) and torch.equal(outs[0], outs[1])
    flop = 2.0 * M * K * Nn * 3
    print(f"{name:52s} M={M:6d} K={K:4d} N={Nn:4d}  single-stage {ts[0]:6.1f} us  ping-pong {ts[1]:6.1f} us  one-block-per-CU {ts[3]:6.1f} us "
          f"({flop / ts[3] / 1e6:5.0f} TF/s)  x{ts[0] / ts[3]:.2f}  bits {'EQUAL' if same else 'DIFFER'}", flush=True)
L.sea_gemm_split_pipeline(2)
