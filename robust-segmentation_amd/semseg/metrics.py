"""Confusion-matrix metrics (counterpart of semseg/metrics.py:21-60).

``update`` fuses argmax + bincount(target*K+pred) into two HIP kernels (argmax via K2 without
gradient, LDS-privatised histogram K3) and keeps the histogram in int64 - the reference accumulates
float32 and loses exactness beyond 2^24 pixels per cell.
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import _native as N


class Metrics:
    def __init__(self, num_classes: int, ignore_label: int, device) -> None:
        self.ignore_label = ignore_label
        self.num_classes = num_classes
        self.device = torch.device(device)
        self._hist = torch.zeros(num_classes, num_classes, dtype=torch.int64, device=device)

    @property
    def hist(self) -> Tensor:
        return self._hist.float()

    def update(self, pred: Tensor, target: Tensor) -> None:
        """pred: (B,K,H,W) scores (argmax taken here) or an integer (B,H,W) prediction map."""
        if self.ignore_label not in (-1, 255):
            target = torch.where(target == self.ignore_label, torch.full_like(target, -1), target)
        target = target.contiguous()
        if pred.dim() == 4:
            B, K, H, W = pred.shape
            dt = torch.uint8 if K <= 255 else torch.int16
            arg = torch.empty(B, H, W, dtype=dt, device=pred.device)
            N.loss_fwd_bwd(pred.detach(), target, None, 3, 3, 0.0, want_grad=False, pred=arg)
            pred = arg
        if target.dtype == torch.uint8 and self.ignore_label == 255:
            pass  # 255 already means ignore for byte labels
        N.confusion(pred, target, self.num_classes, self._hist)

    def compute_iou(self):
        hist = self.hist
        ious = hist.diag() / (hist.sum(0) + hist.sum(1) - hist.diag())
        miou = ious[~ious.isnan()].mean().item()
        return (ious * 100).cpu().numpy().round(2).tolist(), round(miou * 100, 2)

    def compute_f1(self):
        hist = self.hist
        f1 = 2 * hist.diag() / (hist.sum(0) + hist.sum(1))
        mf1 = f1[~f1.isnan()].mean().item()
        return (f1 * 100).cpu().numpy().round(2).tolist(), round(mf1 * 100, 2)

    def compute_pixel_acc(self):
        hist = self.hist
        acc = hist.diag() / hist.sum(1)
        aAcc = hist.diag().sum() / hist.sum()
        macc = acc[~acc.isnan()].mean().item()
        return (acc * 100).cpu().numpy().round(2).tolist(), round(macc * 100, 2), (aAcc * 100).cpu().numpy().round(2)
