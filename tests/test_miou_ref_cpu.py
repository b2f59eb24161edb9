"""CPU checks of the reference's worst-case tables (tests/golden/miou_ref): the product's host-side worst-case
bookkeeping (tools/worse_only.py: worst_acc_from_counts + the C++ greedy K9) reproduces the numbers the REAL
reference's evalSEA computed from the same per-image tables, for every committed part."""
import numpy as np
import pytest

import miou_ref as R


@pytest.mark.parametrize("eps255", [8, 4])
def test_host_bookkeeping_reproduces_the_reference_numbers(eps255):
    ps = R.parts(eps255)
    if not ps:
        pytest.skip(f"no reference part committed for eps {eps255}/255")
    for part, d in ps:
        valid = np.full(R.PART, R.SIZE * R.SIZE)
        assert (d["ints"].sum(-1) == d["correct"]).all()          # no ignored pixels: intersections = correct pixels
        acc, miou, _ = R.worst_case(d["ints"], d["unions"], valid)
        assert acc == pytest.approx(100.0 * float(d["worst_Acc"]), abs=1e-4), (part, acc, d["worst_Acc"])
        assert miou == pytest.approx(100.0 * float(d["final_miou"]), abs=1e-9), (part, miou, d["final_miou"])
        assert int(d["n_iter"]) >= 100 and d["labels"].shape == (R.PART, R.SIZE, R.SIZE)
