import torch
from torch import nn

trunc_normal_ = nn.init.trunc_normal_


class DropPath(nn.Module):
    """Stochastic depth; identity in eval mode or when p == 0."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = float(drop_prob or 0.0)

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        mask = x.new_empty(shape).bernoulli_(keep)
        return x * mask / keep


def to_2tuple(v):
    return v if isinstance(v, (tuple, list)) else (v, v)
