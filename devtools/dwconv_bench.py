#!/usr/bin/env python3
"""M1 (depthwise 7x7, NHWC) A/B: two output rows per lane (default) vs one row, XCD-aware vs plain block order, on the four
ConvNeXt-T stage shapes.

    python devtools/dwconv_bench.py

`hot` repeats one buffer pair (what the layer sees in the attack loop: its input was just written by the previous
layer); `cold` walks a ring of buffers larger than the Infinity Cache.  Both orders must give identical bits.
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
    B = 8
    for Cc, hw in ((96, 128), (192, 64), (384, 32), (768, 16)):
        nset = max(2, int(1.6e9 // (8 * B * hw * hw * Cc)))
        xs = [torch.randn(B, hw, hw, Cc, device="cuda") for _ in range(nset)]
        wt = torch.randn(49, Cc, device="cuda") * 0.1
        bb = torch.randn(Cc, device="cuda")
        for name, bias, fl in (("fwd", bb, 0), ("bwd-data", None, 1)):
            def run(x, ab):      # A/B switches travel through the environment (SEA_DWCONV_AB, read per call here)
                os.environ["SEA_DWCONV_AB"] = str(ab)
                return N.dwconv7x7_nhwc(x, wt, bias, flip=bool(fl))
            ya = run(xs[0], 0)
            same = torch.equal(ya, run(xs[0], 2))
            mb = 8 * xs[0].numel() / 1e6
            same = same and torch.equal(ya, run(xs[0], 4)) and torch.equal(ya, run(xs[0], 16)) and torch.equal(ya, run(xs[0], 20))
            # (bit 16: the plain kernels; without it the software-pipelined ones of round 6)
            same = same and torch.equal(ya, run(xs[0], 32 + 8)) and torch.equal(ya, run(xs[0], 32 + 4)) and torch.equal(ya, run(xs[0], 8))
            # (bit 32: filter rows from global memory instead of LDS)
            for order, f in (("pipe2", 8), ("pipe1", 4), ("p2-glb", 40), ("p1-glb", 36), ("2row", 24), ("1row", 20)):
                hot = timed(lambda: run(xs[0], f), 20)

                def ring():
                    for x in xs:
                        run(x, f)
                cold = timed(ring, 1) / nset
                print(f"M1 dwconv NHWC C={Cc:4d} {hw:3d}x{hw:<3d} {name:8s} {order:6s}  hot {hot * 1e3:7.1f} us {mb / hot / 1e3:6.2f} TB/s"
                      f"   cold {cold * 1e3:7.1f} us {mb / cold / 1e3:6.2f} TB/s   identical={same}", flush=True)


if __name__ == "__main__":
    main()
