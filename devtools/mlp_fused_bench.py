#!/usr/bin/env python3
"""M8f against the launches it replaces, in isolation (interleaved rounds in one process, HIP events, random data):
forward = gemm_split + gemm_split(a_gelu) + residual add; backward = rowmax pass + gemm_split + gemm_split(a_gelu_grad_of).
    python devtools/mlp_fused_bench.py [--rounds 7]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd")]
from semseg import _native as N  # noqa: E402


def timed(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    N.lib()
    g = torch.Generator(device="cuda").manual_seed(1)
    for C, M in ((96, 131072), (192, 32768)):
        H = 4 * C
        x = torch.randn(M, C, generator=g, device="cuda")
        w1 = torch.randn(H, C, generator=g, device="cuda") * 0.05
        b1 = torch.randn(H, generator=g, device="cuda") * 0.1
        w2 = torch.randn(C, H, generator=g, device="cuda") * 0.03
        b2 = torch.randn(C, generator=g, device="cuda") * 0.1
        res = torch.randn(M, C, generator=g, device="cuda")
        gy = torch.randn(M, C, generator=g, device="cuda") * 1e-3
        word = lambda v: torch.tensor([float(v)], dtype=torch.float32, device="cuda").view(torch.int32)  # noqa: E731
        a1 = word(x.abs().max().item() * 1.7)
        a2 = word((x.abs().max() * w1.abs().sum(1) + b1.abs()).max().item())
        mul = (w2.abs().sum(0).max() * 1.13).float().reshape(1)
        P1, P2 = N.gemm_split_pack(w1, terms=22), N.gemm_split_pack(w2, terms=22)
        P2t, P1t = N.gemm_split_pack(w2, trans=True, terms=22), N.gemm_split_pack(w1, trans=True, terms=22)
        t = torch.empty(M, H, device="cuda")
        y, u, gx = torch.empty(M, C, device="cuda"), torch.empty(M, H, device="cuda"), torch.empty(M, C, device="cuda")
        words = torch.empty(M, dtype=torch.int32, device="cuda")

        def fwd_unfused():
            N.gemm_split(x, P1, bias=b1, out=t, amax=a1)
            N.gemm_split(t, P2, bias=b2, out=y, amax=a2, a_gelu=True)
            y.add_(res)

        def fwd_fused():
            N.mlp_fused_forward(x, P1, b1, P2, b2, res, a1, a2, out=y)

        def bwd_unfused():
            N._check(N.lib().sea_absmax_bits(gy.data_ptr(), C, M, C, 1, 0, 1, words.data_ptr(), N._stream()), "absmax")
            N.gemm_split(gy, P2t, out=u, amax=words, amax_rows=1)
            N.gemm_split(u, P1t, out=gx, a_gelu_grad_of=t, amax=words, amax_rows=1, amax_mul=mul)

        def bwd_fused():
            N.mlp_fused_backward(gy, x, P1, b1, P2t, P1t, a1, mul, out=gx)

        cases = [("forward  two GEMMs + add", fwd_unfused), ("forward  fused", fwd_fused),
                 ("backward rowmax + two GEMMs", bwd_unfused), ("backward fused", bwd_fused)]
        for _, f in cases:
            f()
        torch.cuda.synchronize()
        best = {n: [] for n, _ in cases}
        for _ in range(args.rounds):
            for n, f in cases:
                best[n].append(timed(f, args.reps))
        flop_f, flop_b = 2.0 * M * C * H * 2 * 3, 2.0 * M * C * H * 3 * 3        # MFMA flops incl. the three products per pair
        print(f"C={C} M={M} (hidden {H}):")
        for n, _ in cases:
            v = sorted(best[n])
            fl = flop_f if n.startswith("forward") else flop_b
            print(f"   {n:30s} min {v[0]:7.1f} us  median {v[len(v) // 2]:7.1f} us   {fl / v[0] / 1e6:7.0f} TFLOP/s of 16-bit MFMA work"
                  + ("" if "fused" in n else "  (the fused backward also recomputes t: 3 products)"))


if __name__ == "__main__":
    main()
