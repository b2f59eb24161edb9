#!/usr/bin/env python3
"""split-K GEMM chain: eager vs HIP-graph replay, bitwise, many replays (debug aid)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd")]
import torch  # noqa: E402
from semseg import _native as N  # noqa: E402

terms = int(sys.argv[1]) if len(sys.argv) > 1 else 2
g = torch.Generator(device="cuda").manual_seed(1)
A = torch.randn(2048, 384, generator=g, device="cuda")
W1 = torch.randn(1536, 384, generator=g, device="cuda") / 384 ** 0.5
W2 = torch.randn(384, 1536, generator=g, device="cuda") / 1536 ** 0.5
W3 = torch.randn(768, 384, generator=g, device="cuda") / 384 ** 0.5     # K = 384: nkb 12 -> no split
P1, P2, P3 = (N.gemm_split_pack(w, terms=terms) for w in (W1, W2, W3))


def chain(x):
    outs = []
    for _ in range(6):
        h = N.gemm_split(x, P1)
        x = N.gemm_split(torch.nn.functional.gelu(h), P2)       # split-K (K = 1536, 48 tiles)
        outs.append(x)
    outs.append(N.gemm_split(x, P3))
    return outs


ref = [t.clone() for t in chain(A)]
for _ in range(3):
    again = chain(A)
    assert all(torch.equal(a, b) for a, b in zip(ref, again)), "eager not reproducible"
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    chain(A)
torch.cuda.current_stream().wait_stream(s)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr, stream=s):
    outs = chain(A)
bad = 0
for it in range(300):
    gr.replay()
    torch.cuda.synchronize()
    ok = [torch.equal(a, b) for a, b in zip(ref, outs)]
    if not all(ok):
        bad += 1
        if bad <= 3:
            i = ok.index(False)
            d = (ref[i] - outs[i]).abs()
            print(f"replay {it}: output {i} differs, {int((d > 0).sum())} elements, max {d.max().item():.3e}, rows {sorted(set((d > 0).nonzero()[:, 0].tolist()))[:8]}")
print(f"terms {terms} KSPLIT={N.KSPLIT}: {bad} / 300 replays differ from eager")
