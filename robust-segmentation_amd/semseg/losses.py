"""Training-time losses (counterpart of semseg/losses.py:6-109): plain PyTorch modules used by the
PIR-AT OUTER step; not on the attack hot path, kept for ``get_loss`` API parity."""
from __future__ import annotations

import torch
from torch import Tensor, nn
from torch.nn import functional as F

__all__ = ["CrossEntropy", "OhemCrossEntropy", "Dice"]


class _AuxWeighted(nn.Module):
    aux_weights: list

    def _forward(self, preds, labels):
        raise NotImplementedError

    def forward(self, preds, labels: Tensor) -> Tensor:
        if isinstance(preds, tuple):
            return sum(w * self._forward(p, labels) for p, w in zip(preds, self.aux_weights))
        return self._forward(preds, labels)


class CrossEntropy(_AuxWeighted):
    def __init__(self, ignore_label: int = 255, weight: Tensor = None, aux_weights=(1, 0.4, 0.4)) -> None:
        super().__init__()
        self.aux_weights = list(aux_weights)
        self.criterion = nn.CrossEntropyLoss(weight=weight, ignore_index=ignore_label)

    def _forward(self, preds, labels):
        return self.criterion(preds, labels)


class OhemCrossEntropy(_AuxWeighted):
    def __init__(self, ignore_label: int = 255, weight: Tensor = None, thresh: float = 0.7, aux_weights=(1, 1)) -> None:
        super().__init__()
        self.ignore_label = ignore_label
        self.aux_weights = list(aux_weights)
        self.thresh = -torch.log(torch.tensor(thresh, dtype=torch.float))
        self.criterion = nn.CrossEntropyLoss(weight=weight, ignore_index=ignore_label, reduction="none")

    def _forward(self, preds, labels):
        n_min = labels[labels != self.ignore_label].numel() // 16
        loss = self.criterion(preds, labels).view(-1)
        hard = loss[loss > self.thresh]
        if hard.numel() < n_min:
            hard, _ = loss.topk(n_min)
        return torch.mean(hard)


class Dice(_AuxWeighted):
    def __init__(self, delta: float = 0.5, aux_weights=(1, 0.4, 0.4)):
        super().__init__()
        self.delta = delta
        self.aux_weights = list(aux_weights)

    def _forward(self, preds, labels):
        k = preds.shape[1]
        onehot = F.one_hot(labels, k).permute(0, 3, 1, 2)
        tp = torch.sum(onehot * preds, dim=(2, 3))
        fn = torch.sum(onehot * (1 - preds), dim=(2, 3))
        fp = torch.sum((1 - onehot) * preds, dim=(2, 3))
        score = (tp + 1e-6) / (tp + self.delta * fn + (1 - self.delta) * fp + 1e-6)
        return (torch.sum(1 - score, dim=-1) / k).mean()


def get_loss(loss_fn_name: str = "CrossEntropy", ignore_label: int = 255, cls_weights: Tensor = None):
    assert loss_fn_name in __all__, f"Unavailable loss function name >> {loss_fn_name}.\nAvailable loss functions: {__all__}"
    if loss_fn_name == "Dice":
        return Dice()
    return {"CrossEntropy": CrossEntropy, "OhemCrossEntropy": OhemCrossEntropy}[loss_fn_name](ignore_label, cls_weights)
