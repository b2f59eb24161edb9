"""Worst-case SEA bookkeeping over the three attacks (counterpart of tools/worse_only.py).

``evalSEA`` keeps the reference's constructor, ``worse_case_eval(bs, n_batches)``,
``worst_case_miou()`` and ``saveDict`` keys.  Per-image tables come from the LDS-histogram kernel K3
(one launch per attack instead of ~2*n_cls Python iterations per image), the greedy attack selection
runs in host C++ (K9, bit-exact with the reference incl. the ``random.shuffle`` stream), the worst-case
accuracy matrix (K8) is a few float32 tensor ops on the host.
"""
from __future__ import annotations

import os
import random

import numpy as np
import torch

from semseg import _native as N

SEED = 225
random.seed(SEED)
np.random.seed(SEED)


# ---- host arithmetic (K8 / K9), usable from integer tables alone -------------------------------------
def worst_acc_from_counts(correct: torch.Tensor, valid: torch.Tensor):
    """worst_Acc, worst_Acc_indiv and the (A,N) matrix (tools/worse_only.py:398-408).
    correct (A,N), valid (N): integer counts.  float32 ratio like the reference."""
    mat = correct.float().cpu() / valid.float().cpu().unsqueeze(0)
    return mat.min(0)[0].mean().item(), mat.mean(-1), mat


def worst_miou_from_tables(inter: torch.Tensor, union: torch.Tensor, n_rounds: int = 1000):
    """final_miou (a fraction) and the selected attack per image (tools/worse_only.py:279-334) through
    the host C++ greedy.  Uses and advances Python's global ``random`` state like the reference."""
    st = random.getstate()
    miou, sel, rounds, new_mt = N.worst_miou_greedy(inter.float().cpu(), union.float().cpu(), st[1], n_rounds)
    random.setstate((st[0], new_mt, st[2]))
    return miou, sel, rounds


def per_image_tables(preds: torch.Tensor, targets: torch.Tensor, n_cls: int):
    """inter, union (N,C) int64 for one attack's argmax maps (update_fn_indiv, worse_only.py:49-66): the
    prediction is NOT masked at ignored pixels here."""
    inter, pc, tc = N.class_counts(preds.contiguous(), targets.contiguous(), n_cls, per_image=True, mask_pred=False)
    return inter, tc + pc - inter, tc


def _targets_of(val_data):
    if isinstance(val_data, torch.Tensor):
        return val_data
    return torch.stack([torch.as_tensor(val_data[i][1]) for i in range(len(val_data))])


class evalSEA:
    """Worst-case SEA evaluation across the attacks (constructor as tools/worse_only.py:143-166).

    ``val_data`` is a dataset yielding ``(img, target, name)`` or directly an (N,H,W) target tensor;
    ``l_outs`` is the list of per-attack argmax logs (N,H,W)."""

    def __init__(self, val_data, l_outs, eps, n_cls, addendum, saveDir, saveDict, modelName, device="cuda"):
        self.val_data, self.l_output = val_data, l_outs
        self.eps, self.addendum, self.saveDir, self.saveDict, self.modelName = eps, addendum, saveDir, saveDict, modelName
        self.n_cls = n_cls
        self.device = device
        self.los_pairs = ["mask-ce-bal", "mask-ce-avg", "js-avg"]
        self._tables = None

    def _load_logs(self):
        if not self.l_output:
            self.l_output = [torch.load(os.path.join(self.saveDir, "argmax-logs", f"{self.modelName}_{l}_{self.eps}.pt"))
                             for l in self.los_pairs]
        return self.l_output

    def _compute_tables(self):
        if self._tables is None:
            tgt = _targets_of(self.val_data).to(self.device)
            inters, unions = [], []
            for p in self._load_logs():
                i, u, tc = per_image_tables(p.to(self.device), tgt, self.n_cls)
                inters.append(i)
                unions.append(u)
            self._tables = (torch.stack(inters).cpu(), torch.stack(unions).cpu(), tc.sum(-1).cpu())
        return self._tables

    def worse_case_eval(self, bs=16, n_batches=-1, compat_slicing=False):
        """Worst-case aAcc over the attacks.  The reference slices the predictions of loader batch i as
        ``[i*BS : i*BS+BS]`` with BS the current batch size (worse_only.py:374-378), which mis-aligns the
        last partial batch when N % bs != 0; ``compat_slicing=True`` reproduces that."""
        inter, _, valid = self._compute_tables()
        n = inter.shape[1] if n_batches in (-1, None) else min(inter.shape[1], n_batches * bs)
        correct = inter.sum(-1)
        if compat_slicing and n % bs != 0:
            tgt = _targets_of(self.val_data)[:n].to(self.device)
            logs = torch.stack(self._load_logs())
            rows = []
            for i, s in enumerate(range(0, n, bs)):
                BS = min(bs, n - s)
                rows.append(logs[:, i * BS: i * BS + BS])
            pa = torch.cat(rows, dim=1).to(self.device)
            correct = torch.stack([N.class_counts(pa[a].contiguous(), tgt.contiguous(), self.n_cls, True, False)[0].sum(-1)
                                   for a in range(pa.shape[0])]).cpu()
        worst, indiv, _ = worst_acc_from_counts(correct[:, :n], valid[:n])
        print("SEA evaluated Acc", worst)
        self.saveDict["worst_Acc"] = worst
        self.saveDict["worst_Acc_indiv"] = indiv

    def worst_case_miou(self):
        inter, union, _ = self._compute_tables()
        os.makedirs(os.path.join(self.saveDir, "test_results"), exist_ok=True)
        torch.save({"run_int_imwise": inter.float(), "run_union_imwise": union.float()},
                   os.path.join(self.saveDir, "test_results", f"stats_{self.addendum}_{self.eps}.pt"))
        miou, sel, _ = worst_miou_from_tables(inter, union)
        self.saveDict["seed"] = SEED
        self.saveDict["final_miou"] = miou
        self.selected_attack = sel
        print("SEA Evaluation complete, saved-dict:")
        print(self.saveDict)
