"""The device-side controller of the attack (K7 step-size control, K4 conditional copies, the HIP-graph replay, the stage
schedule) pinned EXACTLY (`-m gpu`), at SEA's real stage lengths and on the real models.

The reference's runs were recorded through a wrapper model (oracle/gen_controller_goldens.py: the point-wise net at
n_iter = 90, 120 and apgd_largereps(300) = stages 90 / 90 / 120; oracle/gen_teacher_goldens.py: the three real models at
512 x 512).  Here the PRODUCT's apgd_train / apgd_largereps run unchanged (graph replay on where n_iter >= 12) on the same
model behind tests/teacher.py:SignInjector, which replaces the gradient that reaches the iterate by the reference's
sign(g) of that evaluation.  K1 is bit-exact given the signs (tests/test_teacher_forced_gpu.py), so every iterate the
product feeds to the model must equal the reference's bit for bit -- which it does if and only if every step-size
halving (attacker.py:528-551), every jump back to the best point (547-548), every best-adversarial copy (494-495) and
every stage hand-over (683-690) is the reference's decision.  Loss comparisons of the reference that come within 1e-4
relative of a tie are printed: a device loss may differ by 2e-6 relative (fast exp / log), so a decision that close to
a tie could legitimately go the other way; the generator refused seeds with ties closer than 2e-5.
"""
import glob
import os

import pytest
import torch

import teacher as T
from conftest import GOLDEN, load_golden
from oracle.tiny_models import PointwiseNet

pytestmark = pytest.mark.gpu


def _near_ties(track, n_iters, tol=1e-4):
    """[(evaluation, image, kind, relative gap)] of the reference's loss comparisons closer than `tol` (exact ties excluded)"""
    out, e0 = [], 0
    for n in n_iters:
        L = track[e0:e0 + n + 1]
        best = L[0].clone()
        for j in range(1, n + 1):
            for kind, other in (("vs best", best), ("vs previous", L[j - 1])):
                d = (L[j] - other).abs() / other.abs().clamp_min(1e-30)
                for b in ((d > 0) & (d < tol)).nonzero().flatten().tolist():
                    out.append((e0 + j, b, kind, float(d[b])))
            best = torch.maximum(best, L[j])
        e0 += n + 1
    return out


def _compare_checksums(name, inj, want, n_evals):
    got = inj.chk[:n_evals].cpu()
    bad = (got != want).any(1).nonzero().flatten().tolist()
    assert int(inj.counter.item()) == n_evals, (name, int(inj.counter.item()), n_evals)
    assert not bad, f"{name}: iterate differs from the reference's first at evaluation {bad[0]} (of {n_evals}); all: {bad[:12]}"


def _g13(kind):
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, f"g13_ctrl_{kind}_*.npz")))


@pytest.mark.parametrize("graph", [True, False])
@pytest.mark.parametrize("name", _g13("train"))
def test_apgd_train_controller_exact_at_stage_lengths(name, graph):
    """apgd_train at n_iter = 90 and 120, three losses: 91 / 121 iterates per image equal to the reference's, bit for bit"""
    from semseg import attacker as A
    g = load_golden(name)
    loss, n_iter = name.split("_")[3], int(g["n_iter"])
    net = PointwiseNet(21, seed=int(g["net_seed"])).cuda()
    inj = T.SignInjector(net, T.sign_planes(g, g["x"].shape, "cuda")).eval()
    ties = _near_ties(g["ce"], [n_iter])
    old, A.USE_HIP_GRAPH = A.USE_HIP_GRAPH, graph
    try:
        xb, acc, lb, xba = A.apgd_train(inj, g["x"].cuda(), g["y"].cuda(), "Linf", float(g["eps"]), n_iter=n_iter, use_rs=False,
                                        loss=loss, track_loss="ce-avg", x_init=g["x_init"].cuda(), num_classes=21,
                                        weights=g["w"].cuda(), early_stop=True)
    finally:
        A.USE_HIP_GRAPH = old
    print(f"\n[{name} graph={graph}] reference loss comparisons within 1e-4 of a tie: {ties or 'none'} "
          f"(smallest gap of the run {float(g['min_gap']):.1e})")
    _compare_checksums(name, inj, g["chk"], n_iter + 1)
    assert torch.equal(xb.cpu(), g["x_best"]) and torch.equal(xba.cpu(), g["x_best_adv"])
    assert torch.equal(acc.cpu(), g["acc"])
    torch.testing.assert_close(lb.cpu(), g["loss_best"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("graph", [True, False])
@pytest.mark.parametrize("name", _g13("neartie"))
def test_controller_at_a_near_tie(name, graph):
    """The other g13 fixtures were seed-selected to keep every decisive loss comparison at least 2e-5 (relative) away from a
    tie.  This one was selected the OTHER way (oracle/gen_controller_goldens.py --near-tie): the reference's run contains a
    comparison within 1e-5 of a tie, closer than the device loss kernels' own error budget.  What can be asserted there: the
    product reproduces every iterate of the reference anyway, or it leaves the reference's trajectory at an evaluation AFTER
    the first such comparison and nowhere before (a decision at evaluation j shows in the iterates from j + 1 on, at the
    latest at the next checkpoint)."""
    from semseg import attacker as A
    g = load_golden(name)
    loss, n_iter = name.split("_")[3], int(g["n_iter"])
    ties = _near_ties(g["ce"], [n_iter], tol=1e-5)
    assert ties and float(g["min_gap"]) < 1e-5, "the fixture no longer contains a near-tie"
    net = PointwiseNet(21, seed=int(g["net_seed"])).cuda()
    inj = T.SignInjector(net, T.sign_planes(g, g["x"].shape, "cuda")).eval()
    old, A.USE_HIP_GRAPH = A.USE_HIP_GRAPH, graph
    try:
        xb, acc, lb, xba = A.apgd_train(inj, g["x"].cuda(), g["y"].cuda(), "Linf", float(g["eps"]), n_iter=n_iter, use_rs=False,
                                        loss=loss, track_loss="ce-avg", x_init=g["x_init"].cuda(), num_classes=21,
                                        weights=g["w"].cuda(), early_stop=True)
    finally:
        A.USE_HIP_GRAPH = old
    got = inj.chk[:n_iter + 1].cpu()
    bad = (got != g["chk"]).any(1).nonzero().flatten().tolist()
    first_tie = min(t[0] for t in ties)
    print(f"\n[{name} graph={graph}] reference loss comparisons within 1e-5 of a tie: {ties}; "
          + ("the product reproduces all iterates" if not bad else f"the product leaves the reference's iterates at evaluation {bad[0]}"))
    if bad:
        assert bad[0] > first_tie, (bad[0], first_tie)
    else:
        assert torch.equal(xb.cpu(), g["x_best"]) and torch.equal(xba.cpu(), g["x_best_adv"]) and torch.equal(acc.cpu(), g["acc"])


@pytest.mark.parametrize("name", _g13("largereps"))
def test_apgd_largereps_300_controller_exact(name):
    """the full 300-iteration schedule (stages 90 / 90 / 120 at radii 2 eps, 1.5 eps, eps; graph replay on): all 303
    iterates, the two stage hand-overs (projection of the best-accuracy iterate) and the returned image are the reference's"""
    from semseg import attacker as A
    g = load_golden(name)
    loss = name.split("_")[3]
    net = PointwiseNet(21, seed=int(g["net_seed"])).cuda()
    inj = T.SignInjector(net, T.sign_planes(g, g["x"].shape, "cuda")).eval()
    torch.manual_seed(int(g["seed"]))
    noises = [torch.rand_like(g["x"]) for _ in range(3)]       # the reference drew rand_like(x) once per stage
    assert A.USE_HIP_GRAPH
    xa, _, acc = A.apgd_largereps(inj, g["x"].cuda().clone(), g["y"].cuda(), g["w"].cuda(), norm="Linf", eps=float(g["eps"]),
                                  n_iter=300, n_restarts=1, use_rs=True, loss=loss, verbose=False, track_loss="ce-avg",
                                  log_path=None, num_classes=21, early_stop=True, noises=noises)
    print(f"\n[{name}] reference loss comparisons within 1e-4 of a tie: {_near_ties(g['ce'], [90, 90, 120]) or 'none'}")
    _compare_checksums(name, inj, g["chk"], 303)
    assert torch.equal(xa.cpu(), g["x_adv"]) and torch.equal(acc.cpu(), g["acc"])


@pytest.mark.parametrize("case,loss", [(c, l) for c in sorted(T.RUNS) for l in T.RUNS[c]])
def test_apgd_largereps_controller_exact_on_the_real_models(case, loss):
    """the product's apgd_largereps on UperNet-ConvNeXt-T / -S and Segmenter ViT-S at 512 x 512 with the reference's
    gradient signs: every iterate of the reference's 5- / 10-step runs (stage starts, halvings and jumps back to the best
    point included), the hand-overs between the stages and the returned image, bit for bit"""
    from real_models import CASES, build_model
    from semseg import attacker as A
    from semseg.utils.utils import ADE_WTS, VOC_WTS
    _, kind, backbone, C = CASES[case]
    g = T.load(case, loss)
    x = T.image()
    xs = T.replay_apgd(g, x)                                     # the reference's iterates, rebuilt on the CPU
    n = int(g["n_evals"])
    want = torch.stack([T.checksum(v) for v in xs])
    model = build_model(kind, backbone, C).cuda()
    inj = T.SignInjector(model, T.sign_planes(g, x.shape, "cuda")).eval()
    w = torch.tensor(VOC_WTS if C == 21 else ADE_WTS).cuda()
    torch.manual_seed(int(g["seed"]))
    noises = [torch.rand_like(x) for _ in range(3)]
    n_iters, _ = A.largereps_schedule(int(g["n_iter"]), float(g["eps"]))
    xa, _, acc = A.apgd_largereps(inj, x.cuda().clone(), g["y"].long().cuda(), w, norm="Linf", eps=float(g["eps"]),
                                  n_iter=int(g["n_iter"]), n_restarts=1, use_rs=True, loss=loss, verbose=False,
                                  track_loss="ce-avg", log_path=None, num_classes=C, early_stop=True, noises=noises)
    ties = _near_ties(torch.stack([g[f"e{e}_ce"] for e in range(n)]), n_iters)
    print(f"\n[{case} {loss}] recipes {[[int(v) for v in g[f'e{e}_recipe'][:4]] for e in range(n)]}; reference loss "
          f"comparisons within 1e-4 of a tie: {ties or 'none'}")
    _compare_checksums(f"{case} {loss}", inj, want, n)
    assert torch.equal(xa.cpu(), xs[int(g["final_eval"])]), "the returned image is not the reference's best-accuracy iterate"
    # accuracy of the returned image: the reference's, up to its near-tie pixels (arg-max rounding)
    e = int(g["final_eval"])
    assert abs(float(acc.item()) - float(g["acc"].item())) <= int(g[f"e{e}_n_near"]) / (512.0 * 512.0) + 1e-7


@pytest.mark.parametrize("loss", ["mask-ce-bal", "js-avg"])
def test_apgd_largereps_controller_exact_on_upernet_t_with_graph_replay(loss):
    """UperNet-ConvNeXt-T at 512 x 512, apgd_largereps(n_iter = 40) = stages 12 / 12 / 16: every stage is long enough for the
    product's HIP-graph replay, so THIS test runs the captured graphs (K1 in place, device-side loop index and checkpoint
    table, K7 / K4 inside graph B) on the real model against the reference's 43 iterates, bit for bit.  The product's losses
    on the real model differ from the reference's by up to 7e-5 relative (teacher-forced measurements), so a decision of
    the reference that hangs on a comparison closer than that could legitimately go the other way: the fixture records the
    smallest gap of the run, printed here."""
    name = f"g13_ctrl_real_upernet_t_{loss}_40"
    if not os.path.exists(os.path.join(GOLDEN, name + ".npz")):
        pytest.skip("no real-model controller fixture for this loss (oracle/gen_controller_goldens.py --real <loss>)")
    from real_models import build_model
    from semseg import attacker as A
    from semseg.utils.utils import VOC_WTS
    g = load_golden(name)
    x = T.image()
    n = int(g["n_evals"])
    model = build_model("upernet", "ConvNeXt-T_CVST", 21).cuda()
    inj = T.SignInjector(model, T.sign_planes(g, x.shape, "cuda")).eval()
    torch.manual_seed(int(g["seed"]))
    noises = [torch.rand_like(x) for _ in range(3)]
    assert A.USE_HIP_GRAPH and min(A.largereps_schedule(int(g["n_iter"]), 1.0)[0]) >= A.GRAPH_MIN_ITER
    xa, _, acc = A.apgd_largereps(inj, x.cuda().clone(), g["y"].long().cuda(), torch.tensor(VOC_WTS).cuda(), norm="Linf",
                                  eps=float(g["eps"]), n_iter=int(g["n_iter"]), n_restarts=1, use_rs=True, loss=loss,
                                  verbose=False, track_loss="ce-avg", log_path=None, num_classes=21, early_stop=True, noises=noises)
    ties = _near_ties(g["ce"], A.largereps_schedule(int(g["n_iter"]), 1.0)[0])
    print(f"\n[{name}] smallest relative gap of a reference loss comparison {float(g['min_gap']):.2e}; within 1e-4 of a tie: "
          f"{ties or 'none'}")
    _compare_checksums(name, inj, g["chk"], n)
    assert torch.equal(T.checksum(xa.cpu()), g["x_adv_chk"]), "the returned image is not the reference's"
    last = n - 1
    assert abs(float(acc.item()) - float(g["acc"].item())) <= int(g["n_near"][last].max()) / (512.0 * 512.0) + 1e-7
