"""Sharded SEA evaluation statistics: the only cross-rank exchange of the whole evaluation.

Images are independent, so rank r attacks the images ``r, r+world, r+2*world, ...`` and accumulates
integer statistics locally.  At the end ONE ``all_reduce(SUM)`` (RCCL over xGMI when the tensors live
on GPUs, gloo on CPU in the tests) of a single packed int64 buffer makes every rank hold the global
tables; rank 0 then runs the worst-case bookkeeping (K8/K9) on the host.  The reference has no
distributed evaluation (run_infer.sh pins one GPU); this is the build's addition.

Buffer layout (int64, zero initialised, each rank writes only its own image slots):
    [ clean  inter | pred_cnt | tgt_cnt                      ]  3*C
    [ per-attack dataset totals inter | pred_cnt | tgt_cnt   ]  A*3*C
    [ per-image tables inter (A,N,C) | union (A,N,C)         ]  2*A*N*C
    [ per-image correct (A,N) | valid (N)                    ]  A*N + N
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.distributed as dist


def shard_indices(n_images: int, rank: int, world: int):
    """Global indices of the images rank `rank` owns (round robin: balances a ragged tail)."""
    return list(range(rank, n_images, world))


@dataclass
class SeaStats:
    A: int  # number of attacks (3 for SEA)
    N: int  # GLOBAL number of images
    C: int  # classes
    device: torch.device = torch.device("cpu")

    def __post_init__(self):
        A, N, C = self.A, self.N, self.C
        self._sizes = [3 * C, A * 3 * C, A * N * C, A * N * C, A * N, N]
        self.buf = torch.zeros(sum(self._sizes), dtype=torch.int64, device=self.device)
        o = [0]
        for s in self._sizes:
            o.append(o[-1] + s)
        v = lambda i, shape: self.buf[o[i]:o[i + 1]].view(shape)  # noqa: E731
        self.clean = v(0, (3, C))
        self.attack_totals = v(1, (A, 3, C))
        self.inter = v(2, (A, N, C))
        self.union = v(3, (A, N, C))
        self.correct = v(4, (A, N))
        self.valid = v(5, (N,))

    # ---- local accumulation (all arguments are integer tensors on self.device) -----------------
    def add_clean(self, inter, pred_cnt, tgt_cnt):
        self.clean[0] += inter
        self.clean[1] += pred_cnt
        self.clean[2] += tgt_cnt

    def add_attack_batch(self, a: int, global_idx, inter, pred_cnt, tgt_cnt):
        """Per-image (b,C) counts of one attacked batch, predictions masked at ignored pixels: that is what
        both consumers see in the reference, the dataset totals (tools/infer.py:88-116) and the per-image tables
        of evalSEA, whose input logs were masked in place by eval_performance (infer.py:88-90; worse_only.py:49-66
        itself does not mask)."""
        idx = torch.as_tensor(global_idx, device=self.buf.device)
        self.attack_totals[a, 0] += inter.sum(0)
        self.attack_totals[a, 1] += pred_cnt.sum(0)
        self.attack_totals[a, 2] += tgt_cnt.sum(0)
        self.inter[a, idx] = inter
        self.union[a, idx] = tgt_cnt + pred_cnt - inter
        self.correct[a, idx] = inter.sum(-1)
        self.valid[idx] = tgt_cnt.sum(-1)

    # ---- the one collective ----------------------------------------------------------------------
    def all_reduce(self):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            if self.buf.is_cuda and dist.get_backend() == "gloo":  # single-GPU test set-ups: reduce through the host
                host = self.buf.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
                self.buf.copy_(host)
            else:
                dist.all_reduce(self.buf, op=dist.ReduceOp.SUM)
        return self

    def cpu(self) -> "SeaStats":
        out = SeaStats(self.A, self.N, self.C, torch.device("cpu"))
        out.buf.copy_(self.buf)
        return out
