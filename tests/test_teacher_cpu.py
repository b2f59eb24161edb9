"""CPU checks of the per-step goldens of the real reference (tests/golden/t1_*.npz): the oracle's L-inf arithmetic
rebuilds every iterate of the reference's real-model trajectories from the stored recipes and gradient signs, and the
oracle's loss restatement on the build's model (CPU) reproduces the reference's per-step loss / counts / gradient sign
on the headline model.  Pins the oracle on full-size trajectories; no GPU."""
import os

import pytest
import torch

import teacher as T
from real_models import build_model
from oracle import sea_oracle as O


@pytest.mark.parametrize("case,loss", [(c, l) for c in sorted(T.RUNS) for l in T.RUNS[c]])
def test_recipes_rebuild_every_reference_iterate(case, loss):
    g = T.load(case, loss)
    x = T.image()
    xs = T.replay_apgd(g, x)                # asserts the stored samples of every iterate
    assert len(xs) == int(g["n_evals"]) == sum(n + 1 for n in O.largereps_schedule(int(g["n_iter"]), float(g["eps"]))[0])
    for e, xe in enumerate(xs):
        eps_s = T.stage_eps(g, e)
        assert (xe - x).abs().max() <= eps_s + 1e-7 and xe.min() >= 0 and xe.max() <= 1
    # the fixture exercises the controller: at least one step-size halving with a jump back to the best point
    kinds = [g[f"e{e}_recipe"].tolist() for e in range(len(xs))]
    if int(g["n_iter"]) >= 10:
        assert any(k[0] == 2 and k[3] >= 1 for k in kinds)


def test_pgd_recipe_rebuilds_every_reference_iterate():
    g = T.load_golden("t1_upernet_s_pgd")
    xs, deltas, x_adv = T.replay_pgd(g, T.image())
    assert len(xs) == 5 and all(d.abs().max() <= float(g["eps"]) for d in deltas)


def test_oracle_step_on_the_headline_model_matches_the_reference_per_step():
    """UperNet-ConvNeXt-T, mask-ce-bal: evaluations 0 and 4 (a stage start) through the oracle on the build's model."""
    from semseg.utils.utils import VOC_WTS
    torch.set_num_threads(min(os.cpu_count() or 1, 8))
    g = T.load("upernet_t", "mask-ce-bal")
    x = T.image()
    xs = T.replay_apgd(g, x)
    model = build_model("upernet", "ConvNeXt-T_CVST", 21)
    y, w = g["y"].long(), torch.tensor(VOC_WTS)
    for e in (0, 4):
        mode = O.MODE_BY_NAME["mask-ce-bal"]
        logits, grad = O._model_logits_and_grad(model, xs[e], y, w, mode, want_grad=True)
        r = O.loss_fwd_bwd(logits, y, w, mode, O.MODE_CE, with_grad=False)
        torch.testing.assert_close(r["loss_img"], g[f"e{e}_li"], rtol=1e-4, atol=1e-7)
        torch.testing.assert_close(r["track_img"], g[f"e{e}_ce"], rtol=1e-4, atol=1e-7)
        assert abs(int(r["n_correct"]) - int(g[f"e{e}_n_correct"])) <= int(g[f"e{e}_n_near"])
        sign_ref, lvl = T.unpack_gradient(g, e, x.shape)
        bad = (torch.sign(grad) != sign_ref) & (lvl >= 2)
        assert bad.sum().item() / int((lvl >= 2).sum()) <= 1e-3
