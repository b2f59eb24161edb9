"""Device-resident APGD / PGD drivers against the reference's golden trajectories (`-m gpu`).

The tiny models come from oracle/tiny_models.py (test infrastructure): PointwiseNet evaluates to
bit-identical logits on CPU and GPU, TinyConvNet goes through MIOpen and differs in the last bits,
so trajectories are compared like the oracle-vs-reference tests do: integer accuracy counts exact,
float losses to 1e-4, iterates equal on all but a vanishing fraction of elements.
"""
import glob
import os

import pytest
import torch

from conftest import GOLDEN, load_golden
from oracle.tiny_models import PointwiseNet, TinyConvNet

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    from semseg import attacker
    from semseg import _native
    _native.lib()
    return attacker


def _g4_files():
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "g4_apgd_*_C*.npz")))


def _frac_diff(a, b, tol=1e-6):
    return ((a.cpu() - b).abs() > tol).float().mean().item()


@pytest.mark.parametrize("name", _g4_files())
def test_apgd_train_golden_trajectory(A, name):
    g = load_golden(name)
    _, _, netname, Cs, loss, n = name.split("_")
    C, n_iter = int(Cs[1:]), int(n)
    net = (TinyConvNet if netname == "conv" else PointwiseNet)(C, seed=C).cuda()
    xb, acc, lb, xba = A.apgd_train(net, g["x"].cuda(), g["y"].cuda(), "Linf", g["eps"], n_iter=n_iter, use_rs=False,
                                    loss=loss, track_loss="ce-avg", x_init=g["x_init"].cuda(), num_classes=C,
                                    weights=g["w"].cuda(), early_stop=True, logger=A.Logger(None))
    exact = netname == "pw"
    if exact:
        assert torch.equal(acc.cpu(), g["acc"])
    else:
        assert (acc.cpu() - g["acc"]).abs().max() <= 2.0 / 256  # at most two pixels of a 16x16 image flip
    torch.testing.assert_close(lb.cpu(), g["loss_best"], rtol=1e-4, atol=1e-4)
    lim = 0.002 if exact else 0.02
    assert _frac_diff(xb, g["x_best"]) <= lim
    assert _frac_diff(xba, g["x_best_adv"]) <= lim
    assert (xba.cpu() - g["x"]).abs().max() <= g["eps"] + 1e-7


EARLY = {
    "a": (TinyConvNet, dict(seed=4, gain=3.0), "js-avg"),
    "b": (TinyConvNet, dict(seed=5, gain=3.0), "mask-ce-avg"),
    "c": (PointwiseNet, dict(seed=3, gain=8.0, bias=0.0), "mask-ce-avg"),
}


class _Counting(torch.nn.Module):
    def __init__(self, net):
        super().__init__()
        self.net, self.calls = net, 0

    def forward(self, x):
        self.calls += 1
        return self.net(x)


@pytest.mark.parametrize("tag", sorted(EARLY))
def test_apgd_early_stop_is_frozen_on_device(A, tag):
    g = load_golden(f"g4_earlystop_{tag}")
    Net, kw, loss = EARLY[tag]
    net = _Counting(Net(5, **kw).cuda()).eval()
    n_iter = int(g["n_iter"])
    xb, acc, lb, xba = A.apgd_train(net, g["x"].cuda(), g["y"].cuda(), "Linf", g["eps"], n_iter=n_iter, loss=loss,
                                    track_loss="ce-avg", early_stop=True, num_classes=5, poll_every=2)
    assert acc.sum().item() == 0 and torch.equal(acc.cpu(), g["acc"])
    torch.testing.assert_close(lb.cpu(), g["loss_best"], rtol=1e-4, atol=1e-4)
    assert _frac_diff(xb, g["x_best"]) <= 0.02 and _frac_diff(xba, g["x_best_adv"]) <= 0.02
    # the host left the loop early (it may run a few extra, frozen, iterations past the reference's break)
    assert net.calls <= min(n_iter + 1, int(g["n_forward"]) + 6)
    # never polling gives the same outputs: state is frozen on the device once the flag is up
    net2 = Net(5, **kw).cuda()
    xb2, acc2, lb2, xba2 = A.apgd_train(net2, g["x"].cuda(), g["y"].cuda(), "Linf", g["eps"], n_iter=n_iter, loss=loss,
                                        track_loss="ce-avg", early_stop=True, num_classes=5, poll_every=10 ** 6)
    assert torch.equal(xb, xb2) and torch.equal(xba, xba2) and torch.equal(lb, lb2) and torch.equal(acc, acc2)


@pytest.mark.parametrize("C,loss", [(c, l) for c in (5, 21) for l in ("mask-ce-avg", "mask-ce-bal", "js-avg")])
def test_apgd_largereps_golden(A, C, loss):
    g = load_golden(f"g5_largereps_C{C}_{loss}")
    net = TinyConvNet(C, seed=C + 50).cuda()
    # the reference drew torch.rand_like(x) once per stage from the CPU generator seeded with `seed`
    torch.manual_seed(int(g["seed"]))
    noises = [torch.rand_like(g["x"]) for _ in range(3)]
    xa, _, acc = A.apgd_largereps(net, g["x"].cuda().clone(), g["y"].cuda(), g["w"].cuda(), norm="Linf", eps=g["eps"],
                                  n_iter=int(g["n_iter"]), n_restarts=1, use_rs=True, loss=loss, verbose=False,
                                  track_loss="ce-avg", log_path=None, num_classes=C, early_stop=True, noises=noises)
    assert (acc.cpu() - g["acc"]).abs().max() <= 2.0 / 256
    assert _frac_diff(xa, g["x_adv"]) <= 0.03
    assert (xa.cpu() - g["x"]).abs().max() <= g["eps"] + 1e-7


def test_pgd_attacks_golden():
    from semseg import val as V
    g = load_golden("g6_pgd")
    net = TinyConvNet(21, seed=9).cuda()
    torch.manual_seed(int(g["seed"]))
    delta0 = torch.zeros_like(g["x"]).uniform_(-4.0 / 255, 4.0 / 255)  # what the reference drew on CPU
    xa1, logits1, _ = V.Pgd_Attack_1(epsilon=4.0 / 255, alpha=1e-2, num_iter=5, los="pgd").adv_attack(
        net, g["x"].cuda(), g["y"].cuda(), delta0=delta0.cuda())
    assert _frac_diff(xa1, g["x_adv_1"]) <= 0.02
    torch.testing.assert_close(logits1.cpu(), g["logits_1"], rtol=1e-2, atol=5e-2)
    for los, key in (("mask-ce-avg", "x_adv_mce"), ("js-avg", "x_adv_js")):
        xa, _, _ = V.Pgd_Attack(eps=4.0 / 255, alpha=1e-2, num_iter=5, los=los).adv_attack(net, g["x"].cuda(), g["y"].cuda())
        assert _frac_diff(xa, g[key]) <= 0.02, los
    # the broken call sites of the reference are handled explicitly
    V.Pgd_Attack(epsilon=4.0 / 255, los="mask-ce-avg")
    with pytest.raises(ValueError):
        V.Pgd_Attack(los="pgd")


def test_no_cpu_fallback(A):
    from semseg import _native
    net = PointwiseNet(5, seed=1)
    x = torch.rand(1, 3, 8, 8)
    y = torch.zeros(1, 8, 8, dtype=torch.int64)
    with pytest.raises(_native.SeaNativeError):
        A.apgd_train(net, x, y, "Linf", 4.0 / 255, n_iter=2, loss="mask-ce-avg")


def test_apgd_restarts_api(A):
    """cold API of the reference (attacker.py:574-659): runs, respects the eps-ball, never increases accuracy"""
    net = TinyConvNet(5, seed=2).cuda()
    g = torch.Generator().manual_seed(3)
    x = torch.rand(3, 3, 16, 16, generator=g).cuda()
    with torch.no_grad():
        y = net(x).max(1)[1]
    x_adv, _, acc = A.apgd_restarts(net, x, y, norm="Linf", eps=8 / 255, n_iter=5, loss="mask-ce-avg", n_restarts=2,
                                    track_loss="ce-avg")
    assert (x_adv - x).abs().max() <= 8 / 255 + 1e-6 and x_adv.min() >= 0 and x_adv.max() <= 1
    assert (acc <= 1.0).all() and acc.mean() < 1.0


def test_val_evaluate_and_metrics_on_device():
    from oracle import sea_oracle as O
    from semseg import val as V
    net = TinyConvNet(7, seed=5).cuda()
    g = torch.Generator().manual_seed(8)
    batches = []
    for _ in range(3):
        x = torch.rand(2, 3, 24, 20, generator=g)
        y = torch.randint(0, 7, (2, 24, 20), generator=g)
        y[torch.rand(y.shape, generator=g) < 0.1] = -1
        batches.append((x, y))
    cla_acc, macc, aacc, f1, mf1, ious, miou = V.evaluate(net, batches, "cuda", 7)
    hist = torch.zeros(7, 7, dtype=torch.int64)
    with torch.no_grad():
        for x, y in batches:
            hist += O.confusion_matrix(net(x.cuda()).max(1)[1].cpu(), y, 7)
    ref = O.metrics_from_hist(hist)
    assert miou == ref["miou"] and macc == ref["macc"] and mf1 == ref["mf1"]
    assert ious == ref["ious"] and cla_acc == ref["acc"] and f1 == ref["f1"]
    assert float(aacc) == pytest.approx(float(ref["aacc"]))


def test_pgd_attack_under_bf16_autocast():
    """PIR-AT config 4 runs the inner PGD under bf16 autocast: K2 takes the bf16 logits natively"""
    from semseg.val import Pgd_Attack_1
    net = TinyConvNet(21, seed=9).cuda()
    g = torch.Generator().manual_seed(4)
    x = torch.rand(2, 3, 32, 32, generator=g).cuda()
    with torch.no_grad():
        y = net(x).max(1)[1]
    with torch.autocast("cuda", dtype=torch.bfloat16):
        x_adv, logits, _ = Pgd_Attack_1(epsilon=4 / 255, alpha=1e-2, num_iter=3, los="pgd").adv_attack(net, x, y)
    assert logits.dtype == torch.bfloat16 and x_adv.dtype == torch.float32
    assert (x_adv - x).abs().max() <= 4 / 255 + 1e-6
    with torch.no_grad():
        acc0 = (net(x).max(1)[1] == y).float().mean()
        acc1 = (net(x_adv).max(1)[1] == y).float().mean()
    assert acc1 < acc0


@pytest.mark.parametrize("loss", ["mask-ce-bal", "js-avg"])
def test_hip_graph_replay_is_bitwise_the_eager_loop(loss):
    """ApgdRun's HIP-graph mode (two captured graphs around the eager K2 launch, loop index and checkpoint schedule in
    device memory, K1 in place) against the eager loop on the real UperNet-ConvNeXt-T: same kernels, same order ->
    every output identical bit for bit, including runs with step-size halvings, restarts from the best point and the
    early-stop freeze."""
    from semseg import attacker as A
    from semseg.models import UperNetForSemanticSegmentation
    from semseg.utils.utils import VOC_WTS
    torch.manual_seed(0)
    model = UperNetForSemanticSegmentation("ConvNeXt-T_CVST", 21, None).eval().cuda()
    # 512 x 512: the size at which every kernel of the model path is bitwise reproducible (smaller maps send the PSP
    # bottleneck to a MIOpen kernel that accumulates with atomics, HISTORY §4b)
    x = torch.rand(2, 3, 512, 512, generator=torch.Generator().manual_seed(5)).cuda()
    with torch.no_grad():
        y = model(x).max(1)[1]
    y[0, :4] = -1                                               # some ignored pixels
    w = torch.tensor(VOC_WTS).cuda()
    noise = torch.rand(x.shape, generator=torch.Generator().manual_seed(6)).cuda()
    outs = []
    for graph in (False, True, False):
        old, A.USE_HIP_GRAPH = A.USE_HIP_GRAPH, graph
        try:
            outs.append(A.apgd_train(model, x, y, "Linf", 8.0 / 255, n_iter=30, use_rs=True, loss=loss, early_stop=True,
                                     track_loss="ce-avg", num_classes=21, weights=w, noise=noise, return_pred=True))
        finally:
            A.USE_HIP_GRAPH = old
    for a, c in zip(outs[0], outs[2]):
        assert torch.equal(a, c), "the eager loop itself is not reproducible here"
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    assert (outs[0][3] - x).abs().max() <= 8.0 / 255 + 1e-6


def test_hip_graph_replay_with_the_fused_upsample_loss_is_bitwise_the_eager_loop():
    """Round 6: K2u (loss fused with the model's final x4 up-sampling, lanes = classes) is the default from 96 classes on, so
    the graph mode has to carry it: graph A captures `forward_lowres`, the eager launch between the graphs is K2u, graph B
    back-propagates from the low-resolution gradient.  UperNet-ConvNeXt-T with 151 classes at 256 x 256, graph replay
    against the eager loop bit for bit, and both against the unfused path to the tolerance of two correct arithmetics."""
    from semseg import attacker as A
    from semseg.models import UperNetForSemanticSegmentation
    from semseg.utils.utils import ADE_WTS
    torch.manual_seed(0)
    model = UperNetForSemanticSegmentation("ConvNeXt-T_CVST", 151, None).eval().cuda()
    x = torch.rand(2, 3, 512, 512, generator=torch.Generator().manual_seed(5)).cuda()
    with torch.no_grad():
        y = model(x).max(1)[1]
    y[0, :4] = -1
    w = torch.tensor(ADE_WTS).cuda()[:151]
    noise = torch.rand(x.shape, generator=torch.Generator().manual_seed(6)).cuda()
    runs = {}
    seen = []
    real = A.N.loss_fwd_bwd_upsampled
    A.N.loss_fwd_bwd_upsampled = lambda *a, **k: (seen.append(1), real(*a, **k))[1]
    old = (A.USE_HIP_GRAPH, A.FUSE_UPSAMPLE)
    try:
        for tag, graph, fuse in (("eager", False, "auto"), ("graph", True, "auto"), ("unfused", True, False)):
            A.USE_HIP_GRAPH, A.FUSE_UPSAMPLE = graph, fuse
            A.release_graph_cache(model)
            n0 = len(seen)
            runs[tag] = A.apgd_train(model, x, y, "Linf", 8.0 / 255, n_iter=20, use_rs=True, loss="mask-ce-bal", early_stop=True,
                                     track_loss="ce-avg", num_classes=151, weights=w, noise=noise, return_pred=True)
            assert (len(seen) > n0) == (fuse == "auto"), "the default did not pick K2u at 151 classes" if fuse == "auto" else "K2u ran although forbidden"
    finally:
        A.USE_HIP_GRAPH, A.FUSE_UPSAMPLE = old
        A.N.loss_fwd_bwd_upsampled = real
        A.release_graph_cache(model)
    for a, b in zip(runs["eager"], runs["graph"]):
        assert torch.equal(a, b)
    # fused vs unfused: the same attack up to rounding (the interpolated logits are bit-identical, the loss sums are not)
    assert (runs["graph"][1] - runs["unfused"][1]).abs().max().item() <= 0.02
    torch.testing.assert_close(runs["graph"][2], runs["unfused"][2], rtol=2e-2, atol=1e-3)


@pytest.mark.parametrize("where", ["forward", "backward"])
def test_hip_graph_capture_failure_falls_back_to_the_eager_loop(where, capfd):
    """A model that cannot be captured (a host synchronisation inside its forward, or inside its backward) must keep
    working: ApgdRun notices the failed capture, reports it once and continues eagerly with the same kernels -- the
    result equals the run with graphs switched off, bit for bit, and later graph users are unaffected."""
    from oracle.tiny_models import make_labels
    from semseg import attacker as A

    class _Sync(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, in_fwd):
            ctx.in_fwd = in_fwd
            if in_fwd:
                float(x.flatten()[0].item())          # host sync: illegal during capture
            return x.view_as(x)

        @staticmethod
        def backward(ctx, g):
            if not ctx.in_fwd:
                float(g.flatten()[0].item())
            return g, None

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net = PointwiseNet(21, seed=3)      # explicit element-wise ops: bitwise reproducible on any box

        def forward(self, x):
            return self.net(_Sync.apply(x, where == "forward"))

    model = Net().eval().cuda()
    g = torch.Generator().manual_seed(9)
    x = torch.rand(2, 3, 32, 32, generator=g)
    y = make_labels(model.net.cpu(), x, ignore_frac=0.03, flip_frac=0.1, seed=2).cuda()
    model.cuda()
    x = x.cuda()
    w = torch.rand(21, generator=g).cuda()
    outs = []
    for graph in (False, True):
        old, A.USE_HIP_GRAPH = A.USE_HIP_GRAPH, graph
        try:
            outs.append(A.apgd_train(model, x, y, "Linf", 8.0 / 255, n_iter=20, loss="mask-ce-bal", early_stop=True,
                                     track_loss="ce-avg", num_classes=21, weights=w, return_pred=True))
        finally:
            A.USE_HIP_GRAPH = old
    assert "continuing with the eager loop" in capfd.readouterr().err
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    assert torch.cuda.current_stream() == torch.cuda.default_stream()
    # a capturable model still gets its graphs afterwards
    run = A.ApgdRun(model.net, x, y, 8.0 / 255, 20, "mask-ce-bal", "ce-avg", True, 21, w, x.clone())
    run.start()
    for i in range(4):
        run.step(i)
    assert run.graphs is not None
    run.release_graphs()


def test_hip_graph_mode_on_images_whose_size_is_not_a_multiple_of_four(capfd):
    """PASCAL-VOC evaluates 473 x 473 crops (configs/pascalvoc_convnext.yaml): 3 x 473 x 473 floats per image is odd, so the
    per-image bases of the iterate buffers are not 16-byte aligned.  The replayable K1 has a one-float-per-lane form for that
    (round 3 only had the float4 form: the capture raised); graph replay == the eager loop bit for bit, without a fallback."""
    from oracle.tiny_models import make_labels
    from semseg import attacker as A
    net = PointwiseNet(21, seed=5)
    g = torch.Generator().manual_seed(19)
    x = torch.rand(3, 3, 7, 9, generator=g)
    y = make_labels(net, x, ignore_frac=0.05, flip_frac=0.1, seed=2).cuda()
    net, x = net.cuda(), x.cuda()
    w = torch.rand(21, generator=g).cuda()
    outs = []
    for graph in (False, True):
        old, A.USE_HIP_GRAPH = A.USE_HIP_GRAPH, graph
        try:
            outs.append(A.apgd_train(net, x, y, "Linf", 8.0 / 255, n_iter=30, loss="mask-ce-bal", early_stop=True,
                                     track_loss="ce-avg", num_classes=21, weights=w, return_pred=True))
        finally:
            A.USE_HIP_GRAPH = old
    assert "continuing with the eager loop" not in capfd.readouterr().err
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


def test_one_captured_graph_pair_serves_every_stage_loss_and_batch(A):
    """Round 5: the radius, the run length and the checkpoint table are device state (sea_apgd_linf_step_graph_dev /
    sea_apgd_track_graph_dev) and the buffers a captured graph addresses live in a slot of the model: ONE capture serves the
    nine apgd_train calls of a SEA batch (three losses x three stages of different radius and length, reference
    attacker.py:691-728, tools/infer.py:338-370) and the next batch of the same shape -- with the bits of a capture per run."""
    from oracle.tiny_models import make_labels
    net = PointwiseNet(21, seed=11)
    g = torch.Generator().manual_seed(23)
    xs = [torch.rand(3, 3, 16, 20, generator=g) for _ in range(2)]
    ys = [make_labels(net, x, ignore_frac=0.04, flip_frac=0.1, seed=3 + i).cuda() for i, x in enumerate(xs)]
    net = net.cuda()
    xs = [x.cuda() for x in xs]
    w = torch.rand(21, generator=g).cuda()
    noises = [[torch.rand(3, 3, 16, 20, generator=g).cuda() for _ in range(3)] for _ in range(6)]

    captures = []
    real = A.ApgdRun._capture

    def counting(self, i):
        captures.append((self.n_iter, self.eps))
        return real(self, i)

    def evaluate():
        outs, k = [], 0
        for x, y in zip(xs, ys):
            for loss in ("mask-ce-bal", "mask-ce-avg", "js-avg"):
                outs.append(A.apgd_largereps(net, x, y, w, eps=4.0 / 255, n_iter=60, loss=loss, use_rs=True, early_stop=True,
                                             track_loss="ce-avg", num_classes=21, noises=noises[k], return_pred=True))
                k += 1
        return outs

    A.ApgdRun._capture = counting
    old = A.GRAPH_CACHE
    try:
        A.release_graph_cache()
        A.GRAPH_CACHE = False
        per_run = evaluate()
        n_per_run = len(captures)
        captures.clear()
        A.GRAPH_CACHE = True
        cached = evaluate()
        n_cached = len(captures)
        again = evaluate()                       # the slot (and its pair) survive the evaluation
    finally:
        A.ApgdRun._capture, A.GRAPH_CACHE = real, old
    assert n_per_run == 18 and n_cached == 1 and len(captures) == 1, (n_per_run, n_cached, len(captures))
    for a, b, c in zip(per_run, cached, again):
        for t, u, v in zip(a, b, c):
            if t is None:
                continue
            assert torch.equal(t, u) and torch.equal(t, v)
    # results are copies: a later run does not overwrite what an earlier one handed out
    assert not torch.equal(cached[0][0], cached[1][0])
    A.release_graph_cache(net)
    assert net not in A._GRAPH_SLOTS


@pytest.mark.parametrize("amp", [False, True])
def test_inner_pgd_graph_survives_weight_updates_bitwise(amp):
    """PIR-AT's inner PGD (semseg/val.py:Pgd_Attack_1, reference val.py:181-218 + tools/train_rob_seg.py:326-352): iteration 0
    eager, the rest replayed from ONE captured graph, across optimizer-style in-place weight updates -- every weight-derived
    cache refreshes in place, so the graph captured in the first outer step serves the later ones.  Bitwise the eager loop in
    every outer step (fp32 and under bf16 autocast), and no second capture."""
    from semseg import val as V
    from semseg.models import UperNetForSemanticSegmentation
    torch.manual_seed(0)
    model = UperNetForSemanticSegmentation("ConvNeXt-T_CVST", 21, None).cuda().eval()
    g = torch.Generator().manual_seed(3)
    X = torch.rand(2, 3, 512, 512, generator=g).cuda()
    y = torch.randint(0, 21, (2, 16, 16), generator=g).repeat_interleave(32, 1).repeat_interleave(32, 2).cuda()
    atk = V.Pgd_Attack_1(epsilon=4 / 255, alpha=1e-2, num_iter=5, los="pgd")
    V.release_pgd_graphs()
    graphs = []
    for outer in range(3):
        delta0 = (torch.rand(X.shape, generator=g).cuda() * 2 - 1) * (4 / 255)
        outs = {}
        for mode in (True, False):
            old, V.PGD_GRAPH = V.PGD_GRAPH, mode
            try:
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                    outs[mode] = atk.adv_attack(model, X, y, delta0=delta0)
            finally:
                V.PGD_GRAPH = old
        assert torch.equal(outs[True][0], outs[False][0]), f"outer step {outer}: graph replay != eager loop"
        assert torch.equal(outs[True][1].float(), outs[False][1].float())
        assert (outs[True][0] - X).abs().max() <= 4 / 255 + 1e-6
        (slot,) = V._PGD_SLOTS[model].values()
        assert slot.graph is not None and not slot.failed
        graphs.append(slot.graph)
        with torch.no_grad():                       # what an optimizer step does: every parameter and BatchNorm buffer, in place
            for p in model.parameters():
                p.add_(torch.randn(p.shape, generator=torch.Generator(device="cuda").manual_seed(outer), device="cuda") * 1e-3 * p.abs().mean())
            for b in model.buffers():
                if b.dtype.is_floating_point:
                    b.mul_(1.01)
    assert graphs[0] is graphs[1] is graphs[2], "the inner PGD was captured again after a weight update"
    V.release_pgd_graphs(model)
