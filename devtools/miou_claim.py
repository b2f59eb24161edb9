#!/usr/bin/env python3
"""Full-size SEA evaluation (3 losses x 300 iterations, eps 4 and 8) of a class-balanced synthetic set under the three
convolution modes of the UperNet head (Winograd F(4x4), F(2x2), MIOpen): is the final worst-case mIoU / aAcc the
same to 0.05 percentage points?  Writes one JSON (-> profiles/r2_miou_claim.json).

    python devtools/miou_claim.py --n 16 --out gpurun_out/r2_miou_claim.json
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "robust-segmentation_amd")
sys.path[:0] = [ROOT, PKG]

import torch  # noqa: E402

from semseg.models import UperNetForSemanticSegmentation, convnext_upernet as M  # noqa: E402
from semseg.utils.utils import VOC_WTS  # noqa: E402
from tools.synth import balance_classes, sea_evaluate  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--n_iter", type=int, default=300)
    ap.add_argument("--eps", type=float, nargs="+", default=[4.0, 8.0])
    ap.add_argument("--tiles", type=int, nargs="+", default=[4, 2, 0])
    ap.add_argument("--repeat", type=int, default=1, help="runs per mode (the MIOpen mode is not reproducible run to run)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    C = 21
    torch.manual_seed(0)
    model = UperNetForSemanticSegmentation("ConvNeXt-T_CVST", C, None).eval().cuda()
    for p in model.parameters():
        p.requires_grad_(False)
    images = torch.rand(args.n, 3, 512, 512, generator=torch.Generator().manual_seed(1234))
    share = balance_classes(model, images)      # ONE model, ONE label set for every mode
    with torch.no_grad():
        labels = torch.cat([model(images[i:i + 4].cuda()).max(1)[1].cpu() for i in range(0, args.n, 4)])
    print(f"class shares of the labels: min {float(share.min()):.4f} max {float(share.max()):.4f}", flush=True)
    w = torch.tensor(VOC_WTS)
    rows = []
    names = {4: "winograd F(4x4)", 2: "winograd F(2x2)", 0: "MIOpen"}
    for eps in args.eps:
        for tile in args.tiles:
            for rep in range(args.repeat):
                M.WINOGRAD_TILE = tile
                torch.cuda.synchronize()
                t0 = time.time()
                _, acc, miou = sea_evaluate(model, images, labels, w, eps / 255.0, args.n_iter, batch=8)
                torch.cuda.synchronize()
                dt = time.time() - t0
                rows.append({"eps": eps, "conv_mode": names[tile], "run": rep, "worst_aAcc_pct": 100 * acc, "worst_mIoU_pct": 100 * miou,
                             "n_images": args.n, "n_iter": args.n_iter,
                             "image_iterations_per_s": args.n * 3 * args.n_iter / dt})
                print(json.dumps(rows[-1]), flush=True)
    for eps in args.eps:
        sub = [r for r in rows if r["eps"] == eps]
        for k in ("worst_aAcc_pct", "worst_mIoU_pct"):
            vals = [r[k] for r in sub]
            print(f"eps {eps}: {k} spread over conv modes / runs = {max(vals) - min(vals):.4f} points ({[round(v, 4) for v in vals]})")
    if args.out:
        json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
