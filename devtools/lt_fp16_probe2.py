#!/usr/bin/env python3
"""Interleaved hot-state comparison: the library's fp16 GEMM (2-product K' = 2K proxy) vs M8 fp16x2 (3 products) on the
Winograd-domain shape, alternating, 100 back-to-back launches each (so both see the same clock state)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd")]
import torch  # noqa: E402

from semseg import _native as Nn  # noqa: E402


def timeit(fn, n=100):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


G, M, K, N = 36, 8192, 512, 512
A2 = torch.randn(G, M, 2 * K, device="cuda").half()
W2 = torch.randn(G, N, 2 * K, device="cuda").half()
V = torch.randn(G, M, K, device="cuda")
U = torch.randn(G, K, N, device="cuda") / K ** 0.5
Up = Nn.gemm_split_pack(U, trans=True, terms=22)
Up2 = Nn.gemm_split_pack(U, trans=True, terms=2)
Up1 = Nn.gemm_split_pack(U, trans=True, terms=1)
words = torch.full((M,), 0x40800000, dtype=torch.int32, device="cuda")
C32 = torch.empty(G, M, N, device="cuda")
for rnd in range(4):
    t_lib = timeit(lambda: torch.bmm(A2, W2.transpose(1, 2)))
    t_m8 = timeit(lambda: Nn.gemm_split(V, Up, amax=words, amax_rows=1, out=C32))
    t_b2 = timeit(lambda: Nn.gemm_split(V, Up2, out=C32))
    t_b1 = timeit(lambda: Nn.gemm_split(V, Up1, out=C32))
    print(f"round {rnd}: library fp16 2 products {t_lib:.0f} us (x1.5 = {1.5 * t_lib:.0f});  M8 fp16x2 3 products {t_m8:.0f} us;  "
          f"M8 bf16x2 3 products {t_b2:.0f} us;  M8 bf16x1 1 product {t_b1:.0f} us", flush=True)
try:
    t = timeit(lambda: torch.bmm(A2, W2.transpose(1, 2), out_dtype=torch.float32), n=20)
    print(f"library fp16 2 products, fp32 OUTPUT: {t:.0f} us (x1.5 = {1.5 * t:.0f})", flush=True)
except Exception as e:
    print("out_dtype=float32 not available:", type(e).__name__, str(e)[:160])
