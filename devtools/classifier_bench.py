#!/usr/bin/env python3
"""M10 (classifier kernels, csrc/classifier.hip) against the library matmul on the headline shape (8 x 128 x 128 pixels,
512 -> 21): hot (one buffer) and cold (a ring larger than the Infinity Cache).

    python devtools/classifier_bench.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd")]

import torch  # noqa: E402

from semseg import _native as N  # noqa: E402


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / n)
    return sorted(ts)[3]


def main():
    B, P, K = 8, 128 * 128, 512
    for cls in (21, 32):
        ys = [torch.randn(B * P, K, device="cuda") for _ in range(4)]
        w = torch.randn(cls, K, device="cuda") * 0.05
        b = torch.randn(cls, device="cuda")
        gs = [torch.randn(B, cls, P, device="cuda") for _ in range(4)]
        mb = B * P * K * 4 / 1e6
        for name, f, gfn in (("own", lambda y: N.classifier_forward(y, w, b, B, P), lambda g: N.classifier_backward(g, w)),
                             ("library", lambda y: torch.matmul(w, y.view(B, P, K).transpose(1, 2)) + b.view(1, -1, 1),
                              lambda g: torch.matmul(g.transpose(1, 2), w))):
            hot_f, hot_b = timed(lambda: f(ys[0]), 10), timed(lambda: gfn(gs[0]), 10)
            cold_f = timed(lambda: [f(y) for y in ys], 1) / 4
            cold_b = timed(lambda: [gfn(g) for g in gs], 1) / 4
            print(f"M10 classifier cls={cls:3d} {name:8s} fwd hot {hot_f * 1e3:6.1f} us cold {cold_f * 1e3:6.1f} us ({mb / cold_f / 1e3:5.2f} TB/s)"
                  f"   bwd hot {hot_b * 1e3:6.1f} us cold {cold_b * 1e3:6.1f} us ({mb / cold_b / 1e3:5.2f} TB/s)", flush=True)


if __name__ == "__main__":
    main()
