#!/usr/bin/env python3
"""One GEMM shape through sea_gemm_split, a few launches (profiling target: rocprofv3 ... -- python3 devtools/gemm_split_case.py)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd")]
import torch  # noqa: E402
from semseg import _native as N  # noqa: E402

G, M, K, Nn = [int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (36, 8192, 512, 512))]
terms = int(sys.argv[5]) if len(sys.argv) > 5 else 3
A = torch.randn(G, M, K, device="cuda")
W = torch.randn(G, Nn, K, device="cuda") / K ** 0.5
Wp = N.gemm_split_pack(W, terms=terms)
out = torch.empty(G, M, Nn, device="cuda")
for _ in range(3):
    N.gemm_split(A, Wp, out=out)
torch.cuda.synchronize()
