"""The reference's worst-case SEA tables (tests/golden/miou_ref/, oracle/gen_miou_reference.py): the REAL
nmndeep/Robust-Segmentation run on CPU (apgd_largereps x 3 losses, 100 iterations, then evalSEA) on parts of 64
synthetic 128x128 images, UperNet-ConvNeXt-T with the build's seeded weights and a fitted classifier bias.  This module
restates the generator's inputs (images, random starts) so that the device path runs on exactly the same problem."""
import glob
import os
import random

import numpy as np
import torch

from conftest import GOLDEN

DIR = os.path.join(GOLDEN, "miou_ref")
PART, SIZE, C = 64, 128, 21
LOSSES = ("mask-ce-bal", "mask-ce-avg", "js-avg")


def part_images(part):
    return torch.rand(PART, 3, SIZE, SIZE, generator=torch.Generator().manual_seed(1234 + part))


def start_noise(image, attack, stage):
    g = torch.Generator().manual_seed(225 * 1000003 + image * 64 + attack * 8 + stage)
    return torch.rand(3, SIZE, SIZE, generator=g)


def parts(eps255, suffix=""):
    """[(part index, fixture dict)] of every committed part for this radius.  ``suffix``: "" = the primary runs (4 threads,
    3 x 100 iterations); "_t3" = RE-RUNS of the same parts by the same unmodified reference with 3 threads (the
    reference-vs-reference noise floor); "_it300" = runs at the protocol's full length, 3 x 300 iterations"""
    out = []
    for f in sorted(glob.glob(os.path.join(DIR, f"part_[0-9][0-9]_eps{int(eps255)}{suffix}.npz"))):
        d = np.load(f)
        out.append((int(d["part"]), {k: d[k] for k in d.files}))
    return out


def bias():
    return torch.from_numpy(np.load(os.path.join(DIR, "model.npz"))["bias"])


def worst_case(inter, union, valid):
    """(worst-case aAcc, worst-case mIoU) in PERCENT + the per-image worst-case accuracies, reference arithmetic
    (tools/worse_only.py:279-334, 351-422) through the product's host-side bookkeeping"""
    from tools.worse_only import worst_acc_from_counts, worst_miou_from_tables
    inter, union, valid = (torch.as_tensor(np.asarray(t)).long() for t in (inter, union, valid))
    worst, _, mat = worst_acc_from_counts(inter.sum(-1), valid)
    st = random.getstate()
    random.seed(225)
    miou, _, _ = worst_miou_from_tables(inter, union)
    random.setstate(st)
    return 100.0 * worst, 100.0 * miou, 100.0 * mat.min(0)[0]
