"""Tiny deterministic segmentation "models" used by the golden generator and the parity tests.

TEST INFRASTRUCTURE ONLY (same rule as sea_oracle.py).  Both nets map (B,3,H,W) in [0,1] to
(B,C,H,W) logits, have their weights derived from a seed by closed-form integer arithmetic (no RNG
stream dependence), and are cheap enough that the reference's APGD runs in milliseconds on CPU.
"""
from __future__ import annotations

import torch
from torch import nn


def _det_weights(shape, seed: int, scale: float) -> torch.Tensor:
    """Deterministic pseudo-random weights in [-scale, scale]: multiples of scale/64 so that every
    value (and sums of a few of them) is exactly representable in float32."""
    n = 1
    for s in shape:
        n *= s
    idx = torch.arange(n, dtype=torch.int64)
    v = ((idx * 2654435761 + seed * 40503 + 12345) % 129) - 64  # integers in [-64, 64]
    return (v.to(torch.float32) * (scale / 64.0)).reshape(shape)


class TinyConvNet(nn.Module):
    """conv3x3(3->8) -> tanh -> conv1x1(8->C).  Spatial mixing makes it a miniature of the real
    segmentation nets: input gradients couple neighbouring pixels."""

    def __init__(self, n_cls: int, seed: int = 0, gain: float = 6.0):
        super().__init__()
        self.c1 = nn.Conv2d(3, 8, 3, padding=1)
        self.c2 = nn.Conv2d(8, n_cls, 1)
        with torch.no_grad():
            self.c1.weight.copy_(_det_weights(self.c1.weight.shape, seed + 1, 1.0))
            self.c1.bias.copy_(_det_weights(self.c1.bias.shape, seed + 2, 0.5))
            self.c2.weight.copy_(_det_weights(self.c2.weight.shape, seed + 3, gain))
            self.c2.bias.copy_(_det_weights(self.c2.bias.shape, seed + 4, 0.5))
        self.eval()

    def forward(self, x):
        return self.c2(torch.tanh(self.c1(2.0 * x - 1.0)))


class PointwiseNet(nn.Module):
    """logits[c] = gain * sum_k W[c,k] * (x_k - 0.5) + b[c], evaluated with an explicit, fixed
    sequence of elementwise multiply/add ops so that CPU and GPU produce bit-identical logits
    (no library convolution, no reduction-order freedom)."""

    def __init__(self, n_cls: int, seed: int = 0, gain: float = 8.0, bias: float = 0.25):
        super().__init__()
        self.register_buffer("W", _det_weights((n_cls, 3), seed + 7, gain))
        self.register_buffer("b", _det_weights((n_cls,), seed + 9, bias))
        self.n_cls = n_cls
        self.eval()

    def forward(self, x):
        xc = x - 0.5
        outs = []
        for c in range(self.n_cls):
            t = xc[:, 0] * self.W[c, 0]
            t = t + xc[:, 1] * self.W[c, 1]
            t = t + xc[:, 2] * self.W[c, 2]
            outs.append(t + self.b[c])
        return torch.stack(outs, dim=1)


def make_labels(model, x, ignore_frac: float = 0.05, flip_frac: float = 0.1, seed: int = 0):
    """Labels = clean argmax of the model, a fraction flipped to a random class (so that the
    mask-ce mask is not all ones) and a fraction set to the ignore label -1."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        y = model(x).max(1)[1]
    n_cls = model(x[:1]).shape[1]
    r = torch.rand(y.shape, generator=g)
    rnd = torch.randint(0, n_cls, y.shape, generator=g)
    y = torch.where(r < flip_frac, rnd, y)
    if ignore_frac > 0:
        r2 = torch.rand(y.shape, generator=g)
        y = torch.where(r2 < ignore_frac, torch.full_like(y, -1), y)
    return y
