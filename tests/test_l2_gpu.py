"""K1': the L2 branch of apgd_train (csrc/l2_kernels.hip), `-m gpu`.  Reference: semseg/attacker.py:412-436 (step) with
autoattack.other_utils.L2_norm (attacker.py:6); fixtures g14_* written by oracle/gen_l2_goldens.py from the reference itself."""
import glob
import os

import pytest
import torch

from conftest import GOLDEN, load_golden
from oracle import sea_oracle as O
from oracle.tiny_models import PointwiseNet, TinyConvNet

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def N():
    from semseg import _native
    _native.lib()
    return _native


@pytest.mark.parametrize("shape,a", [((3, 3, 16, 16), 0.75), ((2, 3, 37, 41), 1.0), ((8, 3, 512, 512), 0.75)])
def test_l2_step_matches_the_oracle(N, shape, a):
    g = torch.Generator().manual_seed(shape[-1])
    x = torch.rand(shape, generator=g)
    x_old = (x + 0.02 * torch.randn(shape, generator=g)).clamp(0, 1)
    x_adv = (x + 0.03 * torch.randn(shape, generator=g)).clamp(0, 1)
    grad = torch.randn(shape, generator=g) * torch.rand(shape[0], 1, 1, 1, generator=g) * 1e-3
    step = torch.rand(shape[0], generator=g) * 2.0
    eps = 1.5
    ref = O.apgd_l2_step(x, x_adv, x_old, grad, step, eps, a)
    got = N.apgd_l2_step(x.cuda(), x_adv.cuda(), x_old.cuda(), grad.cuda(), step.cuda(), eps, a)
    # element-wise arithmetic is the reference's op for op; the three norms are sums in another order than ATen's (last bits)
    err = (got.cpu() - ref).abs().max().item()
    print(f"{shape}: max |device - oracle| {err:.2e}")
    assert err <= 2e-6
    assert ((got.cpu() - x).flatten(1).norm(dim=1) <= eps * (1 + 1e-5)).all() and got.min() >= 0 and got.max() <= 1
    assert torch.equal(got, N.apgd_l2_step(x.cuda(), x_adv.cuda(), x_old.cuda(), grad.cuda(), step.cuda(), eps, a))   # reproducible


def _g14():
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "g14_apgd_l2_*_C*.npz")))


@pytest.mark.parametrize("name", _g14())
def test_apgd_train_l2_golden_trajectory(name):
    """the product's apgd_train(norm='L2') against the reference's own run (semseg/attacker.py:260-571)"""
    from semseg import attacker as A
    g = load_golden(name)
    _, _, _, netname, Cs, loss, n = name.split("_")
    C, n_iter = int(Cs[1:]), int(n)
    net = (TinyConvNet if netname == "conv" else PointwiseNet)(C, seed=C).cuda()
    xb, acc, lb, xba = A.apgd_train(net, g["x"].cuda(), g["y"].cuda(), "L2", float(g["eps"]), n_iter=n_iter, use_rs=False,
                                    loss=loss, track_loss="ce-avg", x_init=g["x_init"].cuda(), num_classes=C,
                                    weights=g["w"].cuda(), early_stop=True)
    torch.testing.assert_close(lb.cpu(), g["loss_best"], rtol=1e-4, atol=1e-6)
    assert (acc.cpu() - g["acc"]).abs().max().item() <= 2.0 / 256      # at most a tie pixel or two of 16 x 16
    for got, ref in ((xb, g["x_best"]), (xba, g["x_best_adv"])):
        err = (got.cpu() - ref).abs()
        assert (err > 1e-4).float().mean().item() <= 0.01, (name, err.max().item())


def test_l2_surface(N):
    """what the drop-in surface does for the norms the reference's apgd_train accepts"""
    from semseg import attacker as A
    net = PointwiseNet(5, seed=1).cuda()
    x = torch.rand(2, 3, 8, 8).cuda()
    y = torch.zeros(2, 8, 8, dtype=torch.long).cuda()
    with pytest.raises(NotImplementedError):
        A.apgd_train(net, x, y, "L1", 1.0, n_iter=2, loss="mask-ce-avg", num_classes=5)
    with pytest.raises(ValueError):
        A.apgd_train(net, x, y, "L2", 1.0, n_iter=2, loss="mask-ce-avg", num_classes=5, use_rs=True)
    out = A.apgd_train(net, x, y, "L2", 1.0, n_iter=3, loss="mask-ce-avg", num_classes=5)
    assert len(out) == 4 and ((out[0] - x).flatten(1).norm(dim=1) <= 1.0 + 1e-5).all()
