#!/usr/bin/env python3
"""Noise-free A/B of the model-side arithmetic modes on the FULL SEA protocol (BASELINE configs[1] as written: 3 losses x
300 iterations in stages 90 / 90 / 120, eps = 4/255 and 8/255, 512 x 512, batches of 8; reference tools/infer.py:332-408,
semseg/attacker.py:691-695).  At 512 x 512 the device path is bitwise reproducible run to run, so any difference between
two rows of the output IS the arithmetic mode, not chance.  Also the sustained-rate measurement: wall time per batch of
900 steps.

    python devtools/sea_modes_512.py --n 32 --out gpurun_out/r4_sea_modes_512.json

Modes (module switches of semseg.models.convnext_upernet; the packed weights are cached per mode):
    shipped           forward fp16x2, input gradient fp16x2 with per-row scales (22 significant bits both ways)
    bwd_bf16x2        round 3's default: input gradient with two bf16 terms (16 bits)
    bwd_bf16x3        input gradient with three bf16 terms (the fp32 operand exactly, six products)
    all_bf16x3        three bf16 terms everywhere (exact operands, scale-free)
    hipblaslt_fp32    hipBLASLt fp32 GEMMs, fp32 Winograd F(4x4) (rounds 1-2)
    miopen_fp32       hipBLASLt fp32 GEMMs, MIOpen 3x3 convolutions (no own arithmetic in the head; NOT reproducible
                      run to run: MIOpen kernels accumulate with atomics)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "robust-segmentation_amd")
sys.path[:0] = [ROOT, PKG]

import torch  # noqa: E402

from semseg.models import UperNetForSemanticSegmentation, convnext_upernet as M  # noqa: E402
from semseg.utils.utils import VOC_WTS  # noqa: E402
from tools.synth import balance_classes, sea_evaluate  # noqa: E402

MODES = {
    "shipped": dict(GEMM_TERMS=22, GEMM_TERMS_BWD=22, WINOGRAD_TILE=4),
    "bwd_bf16x2": dict(GEMM_TERMS=22, GEMM_TERMS_BWD=2, WINOGRAD_TILE=4),
    "bwd_bf16x3": dict(GEMM_TERMS=22, GEMM_TERMS_BWD=3, WINOGRAD_TILE=4),
    "all_bf16x3": dict(GEMM_TERMS=3, GEMM_TERMS_BWD=3, WINOGRAD_TILE=4),
    "hipblaslt_fp32": dict(GEMM_TERMS=0, GEMM_TERMS_BWD=3, WINOGRAD_TILE=4),
    "miopen_fp32": dict(GEMM_TERMS=0, GEMM_TERMS_BWD=3, WINOGRAD_TILE=0),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--n_iter", type=int, default=300)
    ap.add_argument("--eps", type=float, nargs="+", default=[4.0, 8.0])
    ap.add_argument("--modes", nargs="+", default=list(MODES))
    ap.add_argument("--repeat", type=int, default=1, help="runs per mode (a second run shows reproducibility)")
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
    rows, preds = [], {}
    for eps in args.eps:
        for name in args.modes:
            for rep in range(args.repeat):
                for k, v in MODES[name].items():
                    setattr(M, k, v)
                torch.cuda.synchronize()
                t0 = time.time()
                p, acc, miou = sea_evaluate(model, images, labels, w, eps / 255.0, args.n_iter, batch=8)
                torch.cuda.synchronize()
                dt = time.time() - t0
                batches = -(-args.n // 8) * 3
                row = {"eps": eps, "mode": name, "run": rep, **MODES[name], "worst_aAcc_pct": 100 * acc,
                       "worst_mIoU_pct": 100 * miou, "n_images": args.n, "n_iter": args.n_iter,
                       "seconds": dt, "seconds_per_batch_of_8_per_attack": dt / batches,
                       "ms_per_step_sustained": 1e3 * dt / (batches * args.n_iter),
                       "image_iterations_per_s": args.n * 3 * args.n_iter / dt}
                key = (eps, "shipped")
                if name == "shipped" and rep == 0:
                    preds[key] = p
                if key in preds:
                    row["adv_argmax_pixels_differing_from_shipped"] = float((p != preds[key]).float().mean())
                rows.append(row)
                print(json.dumps(row), flush=True)
    for eps in args.eps:
        sub = [r for r in rows if r["eps"] == eps]
        for k in ("worst_aAcc_pct", "worst_mIoU_pct"):
            vals = [r[k] for r in sub]
            each = ", ".join("{} {:.4f}".format(r["mode"], r[k]) for r in sub)
            print(f"eps {eps}: {k} spread over the modes = {max(vals) - min(vals):.4f} points ({each})")
    if args.out:
        json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
