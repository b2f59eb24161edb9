#!/usr/bin/env python3
"""Why does the largest M8 launch take ~19 % longer inside the attack loop than back to back?  The launch timed (HIP events around
each launch) (a) back to back, (b) behind a 1 GiB streaming copy (cold L2 / Infinity Cache / TLB, no matrix load), (c) behind
a burst of OTHER matrix work of ~10 ms (the clock / power state of the loop, warm or cold caches).
    python devtools/gemm_inloop_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd")]
import torch  # noqa: E402
from semseg import _native as N  # noqa: E402

G, M, K, Nn = 36, 8192, 512, 512
A = torch.randn(G, M, K, device="cuda")
W = torch.randn(G, Nn, K, device="cuda") / K ** 0.5
Wp = N.gemm_split_pack(W, terms=22)
out = torch.empty(G, M, Nn, device="cuda")
amax = torch.empty(M, dtype=torch.int32, device="cuda")
N.lib().sea_absmax_bits(N._p(A[0]), K, M, K, 1, 0, 1, N._p(amax), N._stream())
src = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device="cuda").normal_()
dst = torch.empty_like(src)
A2 = torch.randn(8, 8192, 1024, device="cuda")
W2p = N.gemm_split_pack(torch.randn(8, 1024, 1024, device="cuda") / 32, terms=22)
out2 = torch.empty(8, 8192, 1024, device="cuda")


def gemm():
    N.gemm_split(A, Wp, out=out, amax=amax, amax_rows=1)


def other_matrix_work():          # ~10 ms of different GEMMs (other buffers)
    for _ in range(40):
        N.gemm_split(A2, W2p, out=out2, groups=1)


def timed(before, reps=15):
    ts = []
    for _ in range(reps):
        if before is not None:
            before()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gemm()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0], ts[-1]


gemm()
torch.cuda.synchronize()
for name, before in (("back to back (previous launch = the same GEMM)", gemm),
                     ("behind a 1 GiB streaming copy", lambda: dst.copy_(src)),
                     ("behind ~10 ms of other matrix work", other_matrix_work),
                     ("behind the copy AND the matrix work", lambda: (other_matrix_work(), dst.copy_(src))),
                     ("behind the matrix work, then the copy last", lambda: (dst.copy_(src), other_matrix_work()))):
    med, lo, hi = timed(before)
    print(f"{name:52s} median {med:7.1f} us   (min {lo:7.1f}, max {hi:7.1f})", flush=True)
