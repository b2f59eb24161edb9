#!/usr/bin/env python3
"""How fast is the LIBRARY's fp16 GEMM (fp32 accumulate / fp32 output) on the Winograd-domain shapes?  (decides whether the
pre-split fp16 planes should go through hipBLASLt as one K' = 3K product instead of M8's own kernel)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd")]
import torch  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
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
for dt in (torch.float16,):
    A2 = torch.randn(G, M, 2 * K, device="cuda").to(dt)      # [hi | mid]
    W2 = torch.randn(G, N, 2 * K, device="cuda").to(dt)      # [hi' | hi']
    us = timeit(lambda: torch.bmm(A2, W2.transpose(1, 2)))
    print(f"{dt} bmm 36x8192x1024x512 (16-bit out): {us:.0f} us = {2 * G * M * 2 * K * N / us / 1e6:.0f} TF/s  -> three products at this "
          f"rate: {1.5 * us:.0f} us", flush=True)
    # (the K = 512 product on its own -- 36 x 8192 x 512 x 512, dense fp16 operands or K-strided views alike -- faults the GPU in
    # this image: "Memory access fault by GPU node" inside the library kernel; it is therefore not timed)
# our kernel on the same product for reference
from semseg import _native as Nn  # noqa: E402
V = torch.randn(G, M, K, device="cuda")
U = torch.randn(G, K, N, device="cuda") / K ** 0.5
Up = Nn.gemm_split_pack(U, trans=True, terms=22)
words = torch.full((M,), 0x40800000, dtype=torch.int32, device="cuda")
us = timeit(lambda: Nn.gemm_split(V, Up, amax=words, amax_rows=1))
print(f"M8 fp16x2 (3 products) 36x8192x512x512: {us:.0f} us = {3 * 2 * G * M * K * N / us / 1e6:.0f} TF/s 16-bit-equivalent")
