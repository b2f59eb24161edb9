#!/usr/bin/env python3
"""Where does the time of M8's largest launch go?  Timing-only builds of the fp16 x 2 ping-pong kernel with parts knocked
out (csrc/gemm_split_pp.hip, -DSEA_GEMM_KNOCKOUT -> devtools/_knock/libknock.so, built by `--build` in the build container):
    1 no C stores   2 no A loads   4 no W loads (zero-record descriptors)   8 no MFMAs   16 no LDS staging writes
    python devtools/gemm_knockout.py --build          (CPU, cross-compiles)
    python devtools/gemm_knockout.py [G M K N]        (GPU)"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "robust-segmentation_amd")
SO = os.path.join(ROOT, "devtools", "_knock", "libknock.so")

if "--build" in sys.argv:
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-shared", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize",
                    "-DSEA_GEMM_KNOCKOUT", os.path.join(PKG, "csrc", "gemm_split_pp.hip"), "-o", SO], check=True)
    print(SO)
    sys.exit(0)

sys.path[:0] = [ROOT, PKG]
import torch  # noqa: E402
from semseg import _native as N  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("-")]
G, M, K, Nn = [int(v) for v in (args[:4] if len(args) >= 4 else (36, 8192, 512, 512))]
L = C.CDLL(SO)
L.sea_gemm_pp_knockout.restype = C.c_int
L.sea_gemm_pp_knockout.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
A = torch.randn(G, M, K, device="cuda")
W = torch.randn(G, Nn, K, device="cuda") / K ** 0.5
Wp = N.gemm_split_pack(W, terms=22)
out = torch.empty(G, M, Nn, device="cuda")
amax = torch.empty(M, dtype=torch.int32, device="cuda")
N.lib().sea_absmax_bits(N._p(A[0]), K, M, K, 1, 0, 1, N._p(amax), N._stream())


def launch(ko):
    rc = L.sea_gemm_pp_knockout(A.data_ptr(), K, Wp.data.data_ptr(), out.data_ptr(), Nn, M, Nn, K, G, M * K, Wp.stride, M * Nn,
                                amax.data_ptr(), 1, ko, N._stream())
    assert rc == 0, rc


def timed(ko, reps=20):
    launch(ko)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch(ko)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


NAMES = {1: "no C stores", 2: "no A loads", 4: "no W loads", 8: "no MFMA", 16: "no LDS writes", 64: "nt C stores", 128: "no operand split (VALU)"}
print(f"G={G} M={M} K={K} N={Nn}  fp16x2 ping-pong kernel, 20 launches back to back, two rounds")
KOS = [int(a[5:]) for a in sys.argv if a.startswith('--ko=')] or [0, 64, 1, 2, 4, 6, 8, 72, 16, 7, 9, 24, 14, 15, 25, 31]
for ko in KOS:
    name = " + ".join(v for b, v in NAMES.items() if ko & b) or "full kernel"
    t = [timed(ko) for _ in range(2)]
    print(f"ko={ko:2d}  {name:60s} {t[0]:8.1f} / {t[1]:8.1f} us", flush=True)
