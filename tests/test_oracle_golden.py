"""Pins the CPU oracle (oracle/sea_oracle.py) to golden vectors produced by the real reference
(oracle/gen_goldens.py, run in the build container).  CPU only; runs everywhere."""
import glob
import os
import random

import pytest
import torch

from conftest import GOLDEN, load_golden
from oracle import sea_oracle as O
from oracle.tiny_models import PointwiseNet, TinyConvNet

LOSSES = ("mask-ce-avg", "mask-ce-bal", "js-avg")


@pytest.mark.parametrize("C", [5, 21, 151])
def test_g1_losses_and_grads(C):
    g = load_golden(f"g1_losses_C{C}")
    logits, y, w = g["logits"], g["y"], g["w"]
    mask_bg = (y != -1).float()
    for name, mode in (("mask-ce-avg", 0), ("mask-ce-bal", 1), ("js-avg", 2), ("ce", 3)):
        key = name.replace("-", "_")
        px = O.pixel_losses(logits, y, w, mode)
        torch.testing.assert_close(px, g[key + "_px"], rtol=2e-5, atol=2e-6)
        img = O.pixel_to_img_loss(px, mask_bg)
        torch.testing.assert_close(img, g[key + "_img"], rtol=1e-5, atol=1e-7)
        gr = O.pixel_loss_grad(logits, y, w, mode)
        torch.testing.assert_close(gr, g[key + "_grad"], rtol=1e-4, atol=2e-8)  # grads are O(1/HW)=4e-3; JS autograd in the reference cancels
    st0 = O.loss_fwd_bwd(logits, y, w, 0, 3, with_grad=False, ignored_correct=False)
    st1 = O.loss_fwd_bwd(logits, y, w, 0, 3, with_grad=False, ignored_correct=True)
    assert torch.equal(st0["pred"], g["pred"])
    assert torch.equal(st0["acc_img"], g["acc_step0"])
    assert torch.equal(st1["acc_img"], g["acc_loop"])
    torch.testing.assert_close(st1["track_img"], g["ce_img"], rtol=1e-5, atol=1e-7)


def test_g1_argmax_first_max():
    g = load_golden("g1_argmax_ties")
    assert torch.equal(O.argmax_first(g["z"].view(3, 4, 1, 1)).view(3), g["arg"])


def test_g2_linf_bit_exact():
    g = load_golden("g2_linf")
    for ci in range(5):
        p = lambda k: g[f"c{ci}_{k}"]  # noqa: E731
        out = O.apgd_linf_step(p("x"), p("x_adv"), p("x_old"), p("grad"), p("step"), p("eps"), p("a"))
        assert torch.equal(out, p("out"))
        assert torch.equal(O.linf_random_start(p("x"), p("u"), p("eps")), p("rs"))
        assert torch.equal(O.linf_project(p("zz"), p("x"), p("eps")), p("proj"))
        assert torch.equal(O.pgd_linf_step(p("x"), p("delta"), p("grad"), p("alpha"), p("eps")), p("delta_out"))


@pytest.mark.parametrize("C", [5, 21])
def test_g3_counts_and_metrics(C):
    g = load_golden(f"g3_counts_C{C}")
    pred, y = g["pred"].clone(), g["y"]
    m_acc, a_acc, m_iou = O.compute_iou_acc(pred, y, C)
    assert torch.equal(pred, g["pred_after"])  # in-place ignore overwrite (attacker.py:20)
    assert m_acc.item() == pytest.approx(g["m_acc"], rel=1e-6)
    assert a_acc.item() == pytest.approx(g["a_acc"], rel=1e-6)
    assert m_iou.item() == pytest.approx(g["m_iou"], rel=1e-6)
    hist = O.confusion_matrix(g["pred"], y, C)
    assert torch.equal(hist.float(), g["hist"])
    m = O.metrics_from_hist(hist)
    import numpy as np
    np.testing.assert_allclose(np.array(m["ious"]), g["ious"].numpy(), rtol=0, atol=1e-9, equal_nan=True)
    np.testing.assert_allclose(np.array(m["acc"]), g["acc"].numpy(), rtol=0, atol=1e-9, equal_nan=True)
    np.testing.assert_allclose(np.array(m["f1"]), g["f1"].numpy(), rtol=0, atol=1e-9, equal_nan=True)
    assert m["miou"] == g["miou"] and m["macc"] == g["macc"] and m["mf1"] == g["mf1"]
    assert float(m["aacc"]) == pytest.approx(g["aacc"])
    # counts are consistent with the confusion matrix
    inter, pc, tc = O.class_counts(g["pred"], y, C)
    assert torch.equal(inter, hist.diag()) and torch.equal(tc, hist.sum(1)) and torch.equal(pc, hist.sum(0))


def _g4_files():
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "g4_apgd_*_C*.npz")))


@pytest.mark.parametrize("name", _g4_files())
def test_g4_apgd_train_trajectory(name):
    g = load_golden(name)
    _, _, netname, Cs, loss, n = name.split("_")
    C, n_iter = int(Cs[1:]), int(n)
    net = (TinyConvNet if netname == "conv" else PointwiseNet)(C, seed=C)
    xb, acc, lb, xba = O.apgd_train(net, g["x"], g["y"], "Linf", g["eps"], n_iter=n_iter, use_rs=False, loss=loss,
                                    track_loss="ce-avg", x_init=g["x_init"], weights=g["w"], early_stop=True)
    assert torch.equal(acc, g["acc"])
    torch.testing.assert_close(lb, g["loss_best"], rtol=1e-5, atol=1e-6)
    # trajectories are sign-sensitive; demand agreement on all but a vanishing fraction of values
    for got, ref in ((xb, g["x_best"]), (xba, g["x_best_adv"])):
        frac = ((got - ref).abs() > 1e-6).float().mean().item()
        assert frac <= 0.002, frac


def _g14_files():
    # (the 25-iteration runs of the convolutional net and the 10-iteration runs of the point-wise one: half the fixtures, all of
    # the code paths; the GPU suite runs all 24)
    names = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "g14_apgd_l2_*_C*.npz")))
    return [n for n in names if ("_conv_" in n) == n.endswith("_25")]


@pytest.mark.parametrize("name", _g14_files())
def test_g14_apgd_train_l2_trajectory(name):
    """the L2 branch of apgd_train (reference semseg/attacker.py:412-436), fixtures written by oracle/gen_l2_goldens.py"""
    g = load_golden(name)
    _, _, _, netname, Cs, loss, n = name.split("_")
    C, n_iter = int(Cs[1:]), int(n)
    net = (TinyConvNet if netname == "conv" else PointwiseNet)(C, seed=C)
    xb, acc, lb, xba = O.apgd_train(net, g["x"], g["y"], "L2", g["eps"], n_iter=n_iter, use_rs=False, loss=loss,
                                    track_loss="ce-avg", x_init=g["x_init"], weights=g["w"], early_stop=True)
    assert torch.equal(acc, g["acc"])
    torch.testing.assert_close(lb, g["loss_best"], rtol=1e-5, atol=1e-6)
    for got, ref in ((xb, g["x_best"]), (xba, g["x_best_adv"])):
        # the step uses the gradient's VALUES (L-inf: its signs), and the oracle's closed-form loss gradients differ from the
        # reference's autograd in the last bits: the iterates agree to rounding level, not bit for bit
        err = (got - ref).abs()
        print(f"{name}: max |x - reference| {err.max().item():.2e}, fraction above 1e-5: {(err > 1e-5).float().mean().item():.4f}")
        assert (err > 1e-5).float().mean().item() <= 0.002 and err.max().item() <= 5e-3
        assert ((got - g["x"]).flatten(1).norm(dim=1) <= float(g["eps"]) * (1 + 1e-5)).all()


EARLY = {
    "a": (TinyConvNet, dict(seed=4, gain=3.0), "js-avg"),
    "b": (TinyConvNet, dict(seed=5, gain=3.0), "mask-ce-avg"),
    "c": (PointwiseNet, dict(seed=3, gain=8.0, bias=0.0), "mask-ce-avg"),
}


@pytest.mark.parametrize("tag", sorted(EARLY))
def test_g4_early_stop(tag):
    g = load_golden(f"g4_earlystop_{tag}")
    Net, kw, loss = EARLY[tag]
    net = Net(5, **kw)
    tr = {}
    xb, acc, lb, xba = O.apgd_train(net, g["x"], g["y"], "Linf", g["eps"], n_iter=int(g["n_iter"]), loss=loss,
                                    track_loss="ce-avg", early_stop=True, trace=tr)
    # the reference ran 1 + n_done forwards: same number of loop iterations before the break
    assert tr["n_done"] + 1 == int(g["n_forward"]) and tr["n_done"] < int(g["n_iter"])
    assert acc.sum() == 0 and torch.equal(acc, g["acc"])
    torch.testing.assert_close(lb, g["loss_best"], rtol=1e-5, atol=1e-6)
    for got, ref in ((xb, g["x_best"]), (xba, g["x_best_adv"])):
        assert ((got - ref).abs() > 1e-6).float().mean().item() <= 0.01


@pytest.mark.parametrize("C,loss", [(c, l) for c in (5, 21) for l in LOSSES])
def test_g5_largereps(C, loss):
    g = load_golden(f"g5_largereps_C{C}_{loss}")
    net = TinyConvNet(C, seed=C + 50)
    torch.manual_seed(int(g["seed"]))
    xa, _, acc = O.apgd_largereps(net, g["x"].clone(), g["y"], g["w"], eps=g["eps"], n_iter=int(g["n_iter"]),
                                  use_rs=True, loss=loss, track_loss="ce-avg", early_stop=True)
    assert torch.equal(acc, g["acc"])
    frac = ((xa - g["x_adv"]).abs() > 1e-6).float().mean().item()
    assert frac <= 0.002, frac
    assert (xa - g["x"]).abs().max() <= g["eps"] + 1e-7


def test_g6_pgd():
    g = load_golden("g6_pgd")
    net = TinyConvNet(21, seed=9)
    torch.manual_seed(int(g["seed"]))
    xa1, logits1 = O.pgd_attack_1(net, g["x"], g["y"], epsilon=4.0 / 255, alpha=1e-2, num_iter=5, los="pgd")
    frac = ((xa1 - g["x_adv_1"]).abs() > 1e-6).float().mean().item()
    assert frac <= 0.002
    torch.testing.assert_close(logits1, g["logits_1"], rtol=1e-3, atol=1e-3)
    for los, key in (("mask-ce-avg", "x_adv_mce"), ("js-avg", "x_adv_js")):
        xa = O.pgd_attack(net, g["x"], g["y"], eps=4.0 / 255, alpha=1e-2, num_iter=5, los=los)
        frac = ((xa - g[key]).abs() > 1e-6).float().mean().item()
        assert frac <= 0.002, (los, frac)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_g7_evalsea(tag):
    g = load_golden(f"g7_evalsea_{tag}")
    C, bs = int(g["n_cls"]), int(g["bs"])
    preds, tgt = g["preds"], g["targets"]
    wa, indiv, _ = O.worst_case_acc(preds, tgt, C, bs=bs)
    assert wa == pytest.approx(g["worst_Acc"], rel=1e-6)
    torch.testing.assert_close(indiv, g["worst_Acc_indiv"], rtol=1e-6, atol=0)
    ints, unis = O.per_image_tables(preds, tgt, C)
    assert torch.equal(ints, g["ints"]) and torch.equal(unis, g["unions"])
    final, sel, rounds = O.worst_case_miou(ints, unis, rng=random.Random(225))
    assert final == g["final_miou"]  # exact: same float64 arithmetic, same shuffle stream


# ---------------------------------------------------------------------------------------------------
# round-2 fixtures: remaining API surface (oracle/gen_goldens.py --extras)
# ---------------------------------------------------------------------------------------------------
def test_g12_val_losses_table():
    """semseg/val.py:121-127: pgd (scalar mean CE), mask-ce-avg / js-avg (per-image means), l2-loss."""
    g = load_golden("g12_val_losses")
    logits, y = g["logits"], g["y"]
    for name, other in (("pgd", y), ("mask-ce-avg", y), ("js-avg", y), ("l2-loss", g["l2_other"])):
        key = name.replace("-", "_")
        z = logits.clone().requires_grad_(True)
        val = O._val_loss(z, other, name)
        torch.testing.assert_close(val.detach(), torch.as_tensor(g[key]), rtol=1e-5, atol=1e-6)
        # the mask of mask-ce-avg is a constant of the autograd graph in the reference too (val.py:113)
        (gr,) = torch.autograd.grad(val.sum(), [z])
        torch.testing.assert_close(gr, g[key + "_grad"], rtol=1e-4, atol=2e-8)


def test_g12_js_div_general_arguments():
    g = load_golden("g12_js_div_general")
    logits, y = g["logits"], g["y"]
    torch.testing.assert_close(O.js_div_general(logits, y), g["full"], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(O.js_div_general(logits, y, red_dim=(1)), g["sum_c"], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(O.js_div_general(logits, y, red_dim=(1, 2, 3)), g["sum_chw"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(O.js_div_general(torch.softmax(logits, 1), y, softmax_output=True, red_dim=(1)),
                               g["from_probs"], rtol=1e-5, atol=1e-7)
    # the class-summed general form equals the closed form the fused kernel uses
    torch.testing.assert_close(O.pixel_losses(logits, y, None, O.MODE_JS), g["sum_c"], rtol=1e-4, atol=1e-6)
    z = logits.clone().requires_grad_(True)
    (gr,) = torch.autograd.grad(O.js_div_general(z, y).sum(), [z])
    torch.testing.assert_close(gr, g["full_grad"], rtol=1e-4, atol=1e-7)
    with pytest.raises(ValueError):
        O.js_div_general(logits, y, reduction="sum")          # legal only when every pixel is ignored
    out = O.js_div_general(logits, torch.full_like(y, -1), reduction="sum")
    assert out.shape == g["allign_sum"].shape and torch.equal(out, g["allign_sum"])


def test_g12_apgd_restarts():
    g = load_golden("g12_apgd_restarts")
    net = TinyConvNet(5, seed=12, gain=3.0)
    noises = [g[f"noise_{i}"] for i in range(int(g["n_noise"]))]
    xa, acc_last, acc = O.apgd_restarts(net, g["x"], g["y"], eps=g["eps"], n_iter=int(g["n_iter"]), loss="mask-ce-avg",
                                        n_restarts=int(g["n_restarts"]), early_stop=True, track_loss="ce-avg",
                                        use_rs=True, noises=noises)
    assert torch.equal(acc, g["acc"]) and torch.equal(acc_last, g["acc_last"])
    assert noises[1].shape[0] == 3          # the image that reached zero accuracy dropped out after restart 1
    assert ((xa - g["x_adv"]).abs() > 1e-6).float().mean() <= 0.01


@pytest.mark.parametrize("tag", ["a", "b"])
def test_g12_eval_performance(tag):
    g = load_golden(f"g12_eval_performance_{tag}")
    C = int(g["n_cls"])
    net = TinyConvNet(C, seed=int(g["seed_net"]))
    loader, s = [], 0
    for n in g["sizes"].tolist():
        loader.append((g["images"][s:s + n], g["targets"][s:s + n], "n"))
        s += n
    stats, l_out = O.eval_performance(net, loader, n_batches=int(g["n_batches"]), n_cls=C)
    assert torch.equal(l_out, g["l_output"])
    for k in ("mAcc", "aAcc", "mIoU"):
        assert stats[k] == pytest.approx(g[k], rel=1e-6)


def _g13_files(kind):
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, f"g13_ctrl_{kind}_*.npz")))


@pytest.mark.parametrize("name", _g13_files("train") + _g13_files("largereps") + _g13_files("neartie"))
def test_g13_controller_exact_at_stage_lengths(name):
    """the oracle's controller at SEA's real stage lengths (n_iter 90 / 120, apgd_largereps(300)): with the reference's
    gradient signs injected (tests/teacher.py:SignInjector) every one of the 91 / 121 / 303 iterates equals the reference's
    bit for bit -- all halvings, jumps back to the best point, best-adversarial copies and stage hand-overs included"""
    import teacher as T
    g = load_golden(name)
    loss = name.split("_")[3]
    net = PointwiseNet(21, seed=int(g["net_seed"]))
    inj = T.SignInjector(net, T.sign_planes(g, g["x"].shape, "cpu")).eval()
    n = int(g["n_evals"])
    if "train" in name or "neartie" in name:   # (neartie: a run with a loss comparison within 1e-5 of a tie: the oracle's float
        # arithmetic is the reference's, so it reproduces that run exactly too)
        xb, acc, lb, xba = O.apgd_train(inj, g["x"], g["y"], "Linf", float(g["eps"]), n_iter=int(g["n_iter"]), use_rs=False,
                                        loss=loss, track_loss="ce-avg", x_init=g["x_init"], weights=g["w"], early_stop=True)
        assert torch.equal(xb, g["x_best"]) and torch.equal(xba, g["x_best_adv"]) and torch.equal(acc, g["acc"])
    else:
        torch.manual_seed(int(g["seed"]))
        xa, _, acc = O.apgd_largereps(inj, g["x"].clone(), g["y"], g["w"], eps=float(g["eps"]), n_iter=300, use_rs=True,
                                      loss=loss, track_loss="ce-avg", early_stop=True)
        assert torch.equal(xa, g["x_adv"]) and torch.equal(acc, g["acc"])
    assert int(inj.counter) == n
    bad = (inj.chk[:n] != g["chk"]).any(1).nonzero().flatten().tolist()
    assert not bad, f"iterate differs first at evaluation {bad[0]}"
