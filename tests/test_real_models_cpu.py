"""Pins the CPU oracle on the REAL models of the BASELINE configs against goldens from the real reference
(oracle/gen_goldens.py --config1-pins / --config3 / --config4).  Same CPU convolutions on both sides, so the
tolerances are tight.  Bounded to one loss per model to keep the CPU suite at a few minutes."""
import os

import pytest
import torch

from oracle import sea_oracle as O
from real_models import CASES, EPS, setup, stage_noises


def _pins(case, loss, tags=("p1",)):
    g, model, x, x1, y, w, C = setup(case)
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    mode = O.MODE_BY_NAME[loss]
    for tag, xp in (("p0", x), ("p1", x1)):
        if tag not in tags:
            continue
        key = f"{tag}_{loss.replace('-', '_')}"
        xin = xp.clone().requires_grad_(True)
        logits = model(xin)
        st = O.loss_fwd_bwd(logits.detach(), y, w, mode, O.MODE_CE, with_grad=True)
        (gr,) = torch.autograd.grad(logits, [xin], grad_outputs=st["dlogits"])
        torch.testing.assert_close(st["loss_img"], g[key + "_img"], rtol=2e-5, atol=1e-7)
        torch.testing.assert_close(st["track_img"], g[tag + "_ce_img"], rtol=2e-5, atol=1e-7)
        assert (st["n_correct"] - g[tag + "_n_correct"]).abs().max() <= 2
        got, ref = gr.flatten()[g["grad_idx"]], g[key + "_grad"]
        assert ((got - ref).norm() / ref.norm()).item() <= 1e-3
    return g, model, x, y, w, C


@pytest.mark.parametrize("case,loss", [("upernet_t", "js-avg"), ("segmenter", "mask-ce-bal"), ("upernet_s", "mask-ce-avg")])
def test_oracle_step_pins_on_real_models(case, loss):
    g, model, x, y, w, C = _pins(case, loss)
    with torch.no_grad():
        got = model(x).flatten()[g["logit_idx"]]
    # the build's model IS the reference's model: same state-dict, logits equal to rounding
    assert (got - g["logit_samples"]).abs().max() <= 2e-5 * float(g["logit_absmax"])


def test_oracle_five_step_largereps_segmenter():
    g, model, x, x1, y, w, C = setup("segmenter")
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    xa, _, acc = O.apgd_largereps(model, x.clone(), y, w, eps=EPS, n_iter=5, use_rs=True, loss="mask-ce-bal",
                                  track_loss="ce-avg", early_stop=True, noises=stage_noises(x))
    assert (acc - g["mask_ce_bal_acc"]).abs().max() <= 5e-4
    frac = ((xa.flatten()[g["idx"]] - g["mask_ce_bal_x_adv_samples"]).abs() > 1e-6).float().mean().item()
    assert frac <= 0.02, frac
