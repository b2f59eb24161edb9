#!/usr/bin/env python3
"""M8 micro-benchmark: the frozen-weight GEMM shapes of one APGD step (UperNet-ConvNeXt-T, B=8, 512x512) through
sea_gemm_split (3 and 2 bf16 terms) vs torch fp32 (hipBLASLt).   python devtools/gemm_split_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd")]

import torch  # noqa: E402

from semseg import _native as N  # noqa: E402

SHAPES = [  # (name, G, M, K, N)
    ("winograd F(4,3) 2048->512 @128^2", 36, 8192, 2048, 512),
    ("winograd F(4,3) 512->512 @128^2", 36, 8192, 512, 512),
    ("winograd F(4,3) 512->512 @64^2", 36, 2048, 512, 512),
    ("winograd F(4,3) 2816->512 @16^2", 36, 128, 2816, 512),
    ("FPN taps 512->9x512 @64^2", 1, 32768, 512, 4608),
    ("FPN taps bwd 9x512->512 @64^2", 1, 32768, 4608, 512),
    ("lateral 96->512 @128^2", 1, 131072, 96, 512),
    ("pwconv1 96->384 @128^2", 1, 131072, 96, 384),
    ("pwconv2 384->96 @128^2", 1, 131072, 384, 96),
    ("pwconv1 192->768 @64^2", 1, 32768, 192, 768),
    ("pwconv2 768->192 @64^2", 1, 32768, 768, 192),
    ("pwconv1 384->1536 @32^2", 1, 8192, 384, 1536),
    ("pwconv2 1536->384 @32^2", 1, 8192, 1536, 384),
    ("pwconv1 768->3072 @16^2", 1, 2048, 768, 3072),
    ("pwconv2 3072->768 @16^2", 1, 2048, 3072, 768),
]


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    tot = {"lib": 0.0, 3: 0.0, 22: 0.0, 2: 0.0}
    for name, G, M, K, Nn in SHAPES:
        A = torch.randn(G, M, K, device="cuda")
        W = torch.randn(G, Nn, K, device="cuda") / K ** 0.5
        Wt = W.transpose(1, 2).contiguous()
        flop = 2.0 * G * M * K * Nn
        t_lib = timed(lambda: torch.bmm(A, Wt) if G > 1 else torch.mm(A[0], Wt[0]))
        row = f"{name:36s} G={G:2d} M={M:6d} K={K:4d} N={Nn:4d}  hipBLASLt fp32 {t_lib * 1e3:8.1f} us {flop / t_lib / 1e9:6.1f} TF/s"
        ref = (A[0, :512].double() @ W[0].double().t())
        floor = 4.0 * G * (M * K + M * Nn) / 5.5e12 * 1e3       # ms: A read once + C written once at 5.5 TB/s
        row += f" | HBM floor {floor * 1e3:7.1f} us"
        amax = A.abs().max().reshape(1).view(torch.int32)
        for terms in (3, 22, 2):
            Wp = N.gemm_split_pack(W, terms=terms)
            out = torch.empty(G, M, Nn, device="cuda")
            t = timed(lambda: N.gemm_split(A, Wp, out=out, amax=amax if terms == 22 else None))
            err = (out[0, :512].double() - ref).abs().max().item() / ref.abs().max().item()
            tot[terms] += t
            row += f" | {terms} terms {t * 1e3:8.1f} us {flop / t / 1e9:6.1f} TF/s err {err:.1e}"
        tot["lib"] += t_lib
        e_lib = ((torch.bmm(A, Wt) if G > 1 else torch.mm(A[0], Wt[0])).reshape(G, M, Nn)[0, :512].double() - ref).abs().max().item() / ref.abs().max().item()
        print(row + f" | lib err {e_lib:.1e}", flush=True)
        del A, W, Wt, out
    print(f"sum over shapes: hipBLASLt {tot['lib']:.2f} ms   bf16x3 {tot[3]:.2f} ms   fp16x2 {tot[22]:.2f} ms   bf16x2 {tot[2]:.2f} ms")


if __name__ == "__main__":
    main()
