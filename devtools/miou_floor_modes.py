#!/usr/bin/env python3
"""How far is ANY GPU implementation from the CPU reference, image by image?  Full SEA (3 x 100 iterations, eps 8/255) on
parts of tests/golden/miou_ref under several arithmetic modes of the device model -- including stock PyTorch-ROCm fp32
(hipBLASLt GEMMs + MIOpen convolutions: no Winograd, no operand splitting: "the reference run on the GPU") -- against the
reference's tables, next to the reference's own re-runs.  Prints per-image |diff| statistics of the worst-case accuracy.

    python devtools/miou_floor_modes.py [--parts 0 1 2 3]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from scipy import stats  # noqa: E402

import miou_ref as R  # noqa: E402
from semseg.models import UperNetForSemanticSegmentation, convnext_upernet as M  # noqa: E402
from semseg.utils.utils import VOC_WTS  # noqa: E402
from tools.synth import sea_evaluate  # noqa: E402

MODES = {
    "shipped (fp16x2 both ways, Winograd F(4x4))": dict(GEMM_TERMS=22, GEMM_TERMS_BWD=22, WINOGRAD_TILE=4),
    "bf16x3 everywhere (exact operands), Winograd F(4x4)": dict(GEMM_TERMS=3, GEMM_TERMS_BWD=3, WINOGRAD_TILE=4),
    "hipBLASLt fp32 + fp32 Winograd F(2x2)": dict(GEMM_TERMS=0, GEMM_TERMS_BWD=3, WINOGRAD_TILE=2),
    "stock PyTorch-ROCm fp32: hipBLASLt + MIOpen convolutions": dict(GEMM_TERMS=0, GEMM_TERMS_BWD=3, WINOGRAD_TILE=0),
}


def per_image(inter, union):
    n = inter.shape[1]
    return R.worst_case(inter, union, torch.full((n,), R.SIZE * R.SIZE))[2].double()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--parts", type=int, nargs="+", default=[0, 1, 2, 3])
    ap.add_argument("--eps255", type=int, default=8)
    args = ap.parse_args()
    torch.manual_seed(0)
    model = UperNetForSemanticSegmentation("ConvNeXt-T_CVST", R.C, None).eval().cuda()
    with torch.no_grad():
        model.decode_head.classifier.bias.copy_(R.bias().cuda())
    M.WINOGRAD_MIN_PIXELS = 16
    w = torch.tensor(VOC_WTS)
    plist = [(p, d) for p, d in R.parts(args.eps255) if p in args.parts]
    ref = per_image(torch.cat([torch.from_numpy(d["ints"]).long() for _, d in plist], 1),
                    torch.cat([torch.from_numpy(d["unions"]).long() for _, d in plist], 1))
    rows = {}
    for name, sw in MODES.items():
        for k, v in sw.items():
            setattr(M, k, v)
        ti, tu = [], []
        for part, d in plist:
            images, labels = R.part_images(part), torch.from_numpy(d["labels"]).long()

            def noise_fn(idx, a, part=part):
                return [torch.stack([R.start_noise(part * R.PART + j, a, st) for j in idx]).cuda() for st in range(3)]
            t = {}
            sea_evaluate(model, images, labels, w, args.eps255 / 255.0, int(d["n_iter"]), batch=16, losses=R.LOSSES, noise_fn=noise_fn,
                         tables=t)
            ti.append(t["inter"])
            tu.append(t["union"])
        rows[name] = per_image(torch.cat(ti, 1), torch.cat(tu, 1)) - ref
        dd = rows[name]
        print(f"{name:62s} vs reference: mean {dd.mean():+.3f}  median|d| {dd.abs().median():.3f}  mean|d| {dd.abs().mean():.3f}  "
              f"sd {dd.std():.3f}   ({dd.numel()} images)", flush=True)
    for tag in ("_t3", "_nomkldnn"):
        prim = dict(R.parts(args.eps255))
        rer = [(p, d) for p, d in R.parts(args.eps255, tag) if p in prim]
        if not rer:
            continue
        a = per_image(torch.cat([torch.from_numpy(prim[p]["ints"]).long() for p, _ in rer], 1),
                      torch.cat([torch.from_numpy(prim[p]["unions"]).long() for p, _ in rer], 1))
        b = per_image(torch.cat([torch.from_numpy(d["ints"]).long() for _, d in rer], 1),
                      torch.cat([torch.from_numpy(d["unions"]).long() for _, d in rer], 1))
        dd = b - a
        print(f"{'reference re-run ' + tag:62s} vs reference: mean {dd.mean():+.3f}  median|d| {dd.abs().median():.3f}  mean|d| "
              f"{dd.abs().mean():.3f}  sd {dd.std():.3f}   ({dd.numel()} images)")
        for name, d1 in rows.items():
            p = stats.mannwhitneyu(d1.abs().numpy(), dd.abs().numpy(), alternative="greater").pvalue
            print(f"    Mann-Whitney one-sided ({name[:40]} deviates more than this re-run): p = {p:.4f}")
    names = list(rows)
    for i in range(len(names)):
        for j in range(i + 1, len(names)):
            dd = rows[names[i]] - rows[names[j]]
            print(f"device mode vs device mode: {names[i][:34]} | {names[j][:34]}: median|d| {dd.abs().median():.3f} mean|d| {dd.abs().mean():.3f}")
    s0 = rows[names[0]]
    for name in names[1:]:
        p = stats.mannwhitneyu(s0.abs().numpy(), rows[name].abs().numpy(), alternative="greater").pvalue
        print(f"Mann-Whitney one-sided (shipped deviates more from the reference than '{name[:44]}'): p = {p:.4f}")


if __name__ == "__main__":
    main()
