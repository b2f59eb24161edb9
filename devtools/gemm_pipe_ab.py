#!/usr/bin/env python3
"""M8 A/B on one lease: the single-stage K loop (sea_gemm_split_pipeline(0)) against the ping-pong pipeline (1) on the
frozen-weight GEMM shapes of one APGD step (UperNet-ConvNeXt-T, B = 8, 512 x 512), alternating, 3 rounds of `reps` launches
each, bitwise comparison of the outputs.      python devtools/gemm_pipe_ab.py [terms ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd")]

import torch  # noqa: E402

from semseg import _native as N  # noqa: E402

SHAPES = [  # (name, G, M, K, N, launches per APGD step)
    ("winograd 512->512 @128^2 (x3/step)", 36, 8192, 512, 512),
    ("winograd 2048->512 fwd / 512->1024.. @128^2", 36, 8192, 1024, 512),
    ("winograd 512->512 @64^2", 36, 2048, 512, 512),
    ("winograd 512->512 @32^2", 36, 512, 512, 512),
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


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    L = N.lib()
    terms_list = [int(v) for v in sys.argv[1:]] or [22, 1]
    tot = {}
    for name, G, M, K, Nn in SHAPES:
        A = torch.randn(G, M, K, device="cuda")
        W = torch.randn(G, Nn, K, device="cuda") / K ** 0.5
        amax = torch.empty(M, dtype=torch.int32, device="cuda")
        L.sea_absmax_bits(N._p(A[0]), K, M, K, 1, 0, 1, N._p(amax), N._stream())
        flop = 2.0 * G * M * K * Nn
        reps = max(5, min(50, int(3e4 / (flop / 4e8 / 1e3))))
        for terms in terms_list:
            Wp = N.gemm_split_pack(W, terms=terms)
            outs = [torch.empty(G, M, Nn, device="cuda") for _ in range(2)]
            kw = dict(amax=amax, amax_rows=1) if terms == 22 else {}
            pipes = (0, 1, 3) if (Nn % 256 == 0 and terms in (22, 2)) else (0, 1)
            outs.append(torch.empty(G, M, Nn, device="cuda"))
            ts = {0: [], 1: [], 3: []}
            for rnd in range(3):
                for pipe in pipes:
                    L.sea_gemm_split_pipeline(pipe)
                    ts[pipe].append(timed(lambda: N.gemm_split(A, Wp, out=outs[min(pipe, 2)], **kw), reps))
            same = torch.equal(outs[0], outs[1]) and (3 not in pipes or torch.equal(outs[0], outs[2]))
            prod = 3 if terms in (22, 2) else 1
            t0, t1 = min(ts[0]), min(ts[1])
            tot[terms, 0] = tot.get((terms, 0), 0) + t0
            tot[terms, 1] = tot.get((terms, 1), 0) + t1
            print(f"{name:44s} G={G:2d} M={M:6d} K={K:4d} N={Nn:4d} terms={terms:2d}  single-stage "
                  f"{'/'.join(f'{t:7.1f}' for t in ts[0])} us ({prod * flop / t0 / 1e6:6.0f} TF/s)   ping-pong "
                  f"{'/'.join(f'{t:7.1f}' for t in ts[1])} us ({prod * flop / t1 / 1e6:6.0f} TF/s)   x{t0 / t1:.2f}  "
                  + (f"256x256 {'/'.join(f'{t:7.1f}' for t in ts[3])} us ({prod * flop / min(ts[3]) / 1e6:6.0f} TF/s) x{t0 / min(ts[3]):.2f}  " if ts[3] else "") +
                  f"bits {'EQUAL' if same else 'DIFFER'}", flush=True)
            del outs
        del A, W
    L.sea_gemm_split_pipeline(1)
    for terms in terms_list:
        print(f"sum over shapes, terms {terms}: single-stage {tot[terms, 0] / 1e3:.2f} ms   ping-pong {tot[terms, 1] / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
