"""Replay of the per-step goldens of the REAL reference (tests/golden/t1_*.npz, oracle/gen_teacher_goldens.py).

The fixtures hold, per model evaluation of the reference's attack, a recipe that rebuilds the reference's iterate
from earlier ones with the oracle's L-inf arithmetic, plus the sign / magnitude level of every input-gradient
element.  `replay_apgd` / `replay_pgd` rebuild ALL iterates on the CPU (checked against the stored samples), so a
test can feed the reference's iterate k to the device path."""
import numpy as np
import torch

from conftest import load_golden
from oracle import sea_oracle as O

RUNS = {   # case -> losses with a teacher fixture (oracle/gen_teacher_goldens.py:CASES)
    "upernet_t": ("mask-ce-bal", "mask-ce-avg", "js-avg"),
    "segmenter": ("mask-ce-avg", "js-avg"),
    "upernet_s": ("mask-ce-bal",),
}


def load(case, loss):
    return load_golden(f"t1_{case}_{loss.replace('-', '_')}")


def image():
    """the ONE 512x512 image of the teacher fixtures (= image 0 of the two-image real-model goldens)"""
    return torch.rand(2, 3, 512, 512, generator=torch.Generator().manual_seed(1234))[:1].clone()


def unpack_gradient(g, e, shape):
    """-> (sign (float32: -1, 0, +1), level (uint8: how many of 1e-4, 1e-3, 1e-2 x max|g| the element exceeds))"""
    n = int(np.prod(shape))
    neg = torch.from_numpy(np.unpackbits(g[f"e{e}_neg"].numpy())[:n]).bool()
    sign = torch.where(neg, -1.0, 1.0)
    sign[g[f"e{e}_zeros"].long()] = 0.0
    lo = torch.from_numpy(np.unpackbits(g[f"e{e}_lvl_lo"].numpy())[:n])
    hi = torch.from_numpy(np.unpackbits(g[f"e{e}_lvl_hi"].numpy())[:n])
    return sign.view(shape), (lo + 2 * hi).view(shape)


def unpack_mask(packed, shape):
    n = int(np.prod(shape))
    return torch.from_numpy(np.unpackbits(packed.numpy())[:n]).bool().view(shape)


def stage_eps(g, e):
    n_iters, epss = O.largereps_schedule(int(g["n_iter"]), float(g["eps"]))
    return epss[int(g[f"e{e}_stage"])]


def replay_apgd(g, x):
    """All iterates the reference fed to the model, rebuilt on the CPU and checked against the stored samples."""
    torch.manual_seed(int(g["seed"]))
    noises = [torch.rand_like(x) for _ in range(3)]      # the reference drew rand_like(x) once per stage
    xs = []
    for e in range(int(g["n_evals"])):
        kind, b, o, m, a = g[f"e{e}_recipe"].tolist()
        eps_s = stage_eps(g, e)
        if kind == 0:
            xe = O.linf_random_start(x, noises[0], eps_s).clamp(0.0, 1.0)
        elif kind == 1:
            xe = O.linf_project(xs[int(b)], x, eps_s).clamp(0.0, 1.0)
        else:
            sign, _ = unpack_gradient(g, int(b), x.shape)
            step = torch.full((1,), 2.0 * eps_s) / (2.0 ** int(m))
            xe = O.apgd_linf_step(x, xs[int(b)], xs[int(o)], sign, step, eps_s, a)
        assert torch.equal(xe.flatten()[g["sample_idx"].long()], g[f"e{e}_x_samples"]), f"replay differs at evaluation {e}"
        xs.append(xe)
    return xs


def replay_pgd(g, x):
    """(inputs X + delta_t the reference fed to the model, the deltas, the returned x_adv)"""
    eps, alpha = float(g["eps"]), float(g["alpha"])
    # the generator served the reference's `delta.uniform_(-eps, eps)` from torch.rand (host-independent arithmetic)
    delta = eps * (2 * torch.rand(x.shape, generator=torch.Generator().manual_seed(int(g["seed"]))) - 1)
    xs, deltas = [], []
    for e in range(int(g["n_evals"])):
        xs.append(x + delta)
        deltas.append(delta)
        assert torch.equal(xs[-1].flatten()[g["sample_idx"].long()], g[f"e{e}_x_samples"]), f"PGD replay differs at step {e}"
        sign, _ = unpack_gradient(g, e, x.shape)
        delta = O.pgd_linf_step(x, delta, sign, alpha, eps)
    deltas.append(delta)
    x_adv = (x + delta).clamp(0.0, 1.0)
    assert torch.equal(x_adv.flatten()[g["sample_idx"].long()], g["x_adv_samples"])
    return xs, deltas, x_adv
