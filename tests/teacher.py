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


# ---------------------------------------------------------------------------------------------------------------------
# exact controller tests (tests/test_controller_exact_gpu.py): the product's attack with the reference's gradient signs
# ---------------------------------------------------------------------------------------------------------------------
def checksum(x):
    """per-image sum of the float32 bit patterns as int64 (oracle/gen_controller_goldens.py:checksum)"""
    return x.contiguous().view(torch.int32).to(torch.int64).flatten(1).sum(1)


class _InjectSign(torch.autograd.Function):
    """identity whose backward DISCARDS the incoming gradient and returns the reference's sign plane of this evaluation"""

    @staticmethod
    def forward(ctx, x, signs, idx):
        ctx.save_for_backward(signs, idx)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        signs, idx = ctx.saved_tensors
        return signs.index_select(0, idx)[0].to(g.dtype), None, None


class SignInjector(torch.nn.Module):
    """Thin wrapper around the attacked model for the exact controller tests.  Every forward (a) stores the per-image
    checksum of the iterate it was given in row `evaluation index` of ``chk`` and (b) makes the gradient that flows back to
    the iterate the REFERENCE's sign(g) of that evaluation (``signs``: (E, *x.shape) int8 on the device, rows without a
    gradient unused).  The evaluation counter lives in device memory and is advanced by the forward itself, so the wrapper
    behaves the same inside the product's captured HIP graphs (where this Python code runs once, at capture) as in its
    eager iterations.  The model's own forward / backward still run: losses, accuracies and all controller decisions are
    the product's."""

    def __init__(self, model, signs):
        super().__init__()
        self.model, self.signs = model, signs
        self.E = signs.shape[0]
        dev = signs.device
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        self.chk = torch.zeros(self.E + 1, signs.shape[1], dtype=torch.int64, device=dev)   # row E: overflow sink

    def forward(self, x):
        with torch.no_grad():
            idx = self.counter.clamp(max=self.E)
            self.counter += 1
            self.chk.index_copy_(0, idx, checksum(x.detach()).unsqueeze(0))
        if torch.is_grad_enabled() and x.requires_grad:
            x = _InjectSign.apply(x, self.signs, idx.clamp(max=self.E - 1))
        return self.model(x)


def sign_planes(g, shape, device):
    """(E, *shape) int8 sign planes of a g13 / t1 fixture (zeros where the evaluation took no gradient)"""
    E = int(g["n_evals"])
    n = int(np.prod(shape))
    out = torch.zeros(E, *shape, dtype=torch.int8)
    for e in range(E):
        if f"e{e}_neg" not in g:
            continue
        if f"e{e}_zero" in g:        # g13: two bit planes
            neg = torch.from_numpy(np.unpackbits(g[f"e{e}_neg"].numpy())[:n]).bool()
            zero = torch.from_numpy(np.unpackbits(g[f"e{e}_zero"].numpy())[:n]).bool()
            s = torch.where(neg, -1, 1).to(torch.int8)
            s[zero] = 0
            out[e] = s.view(shape)
        else:                         # t1: bit plane + list of exact zeros
            out[e] = unpack_gradient(g, e, shape)[0].to(torch.int8)
    return out.to(device)
