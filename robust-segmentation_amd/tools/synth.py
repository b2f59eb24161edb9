"""Synthetic evaluation sets for a random-init model (datasets are out of scope of this build).

`balance_classes` makes such a set WELL CONDITIONED for mIoU: with seeded random weights a segmentation net predicts
a handful of classes, so most of the C per-class IoUs are 0/0 or rest on a few stray pixels and the mean over
classes jumps by whole points when one pixel flips.  Fitting the bias of the final classifier so that the clean
prediction spreads evenly over all C classes (labels = that prediction, clean accuracy 100 %) gives every class
thousands of pixels; the network, its gradients and the attack are untouched otherwise.

`sea_evaluate` is the evaluation loop of tools/infer.py on tensors (three SEA attacks, per-image random starts,
worst case over the attacks) for callers that compare implementations on the SAME model, images and labels."""
from __future__ import annotations

import random

import torch


def _classifier(model):
    head = getattr(model, "decode_head", None)
    if head is not None and hasattr(head, "classifier"):
        return head.classifier
    raise ValueError("balance_classes needs a model with decode_head.classifier (UperNet)")


@torch.no_grad()
def balance_classes(model, images, batch: int = 4, min_frac: float = 0.005):
    """Fit ``decode_head.classifier.bias`` in place so that the clean prediction on ``images`` (host (N,3,H,W)) gives
    every class about the same number of pixels; returns the achieved class shares (C,).

    Balanced assignment by its dual: the bias is the price vector of an entropic optimal-transport problem between
    pixels and C equal-capacity classes (Sinkhorn column scaling on softmax((L + b) / tau)), annealed in tau until the
    hard arg-max shares follow the soft ones."""
    conv = _classifier(model)
    dev = conv.weight.device
    if conv.bias is None:
        conv.bias = torch.nn.Parameter(torch.zeros(conv.out_channels, device=dev))
    conv.bias.zero_()
    lows = []
    for i in range(0, images.shape[0], batch):
        low = model.forward_lowres(images[i:i + batch].to(dev))
        lows.append((low[0] if isinstance(low, (tuple, list)) else low).float())
    L = torch.cat(lows).permute(1, 0, 2, 3).flatten(1).double()      # (C, pixels) logits before the final upsample
    C = L.shape[0]
    b = -L.mean(1)
    sd = float((L - L.mean(1, keepdim=True)).std())
    share = None
    for tau in (0.5 * sd, 0.2 * sd, 0.08 * sd, 0.03 * sd, 0.012 * sd):
        for _ in range(300):
            soft = torch.softmax((L + b[:, None]) / tau, 0).mean(1)
            b = b - tau * torch.log(soft * C).clamp(-4.0, 4.0)
        share = torch.bincount((L + b[:, None]).argmax(0), minlength=C).double() / L.shape[1]
        if float(share.min()) >= max(min_frac, 0.5 / C):
            break
    b = b - b.mean()
    conv.bias.copy_(b.float())
    return share.float().cpu()


def image_noises(idx, a: int, shape, device, seed: int = 225):
    """uniform draws of the three random starts of apgd_largereps, one stream per (image, attack) (tools/infer.py)"""
    out = [torch.empty(len(idx), *shape, device=device) for _ in range(3)]
    gen = torch.Generator(device=device)
    for j, gi in enumerate(idx):
        gen.manual_seed(seed * 1000003 + int(gi) * 7 + a)
        for st in range(3):
            out[st][j] = torch.rand(*shape, generator=gen, device=device)
    return out


def sea_evaluate(model, images, labels, weights, eps: float, n_iter: int, batch: int = 8,
                 losses=("mask-ce-bal", "mask-ce-avg", "js-avg"), noise_fn=None, tables=None):
    """Full SEA evaluation of host tensors ``images`` (N,3,H,W), ``labels`` (N,H,W) on the model's device.  Returns
    ``(preds (A,N,H,W) int64 on the host, worst-case aAcc, worst-case mIoU)`` with the reference's arithmetic
    (tools/worse_only.py:279-334, 351-422).  ``noise_fn(idx, a)`` supplies the three random-start draws of a batch
    (default: the per-image device streams of tools/infer.py); a dict passed as ``tables`` receives the per-attack
    per-image ``inter`` / ``union`` (A,N,C) and ``valid`` (N) counts the worst-case numbers were computed from."""
    from semseg import _native as N
    from semseg import attacker
    from tools.worse_only import worst_acc_from_counts, worst_miou_from_tables
    dev = next(model.parameters()).device
    n, C = images.shape[0], int(weights.numel())
    preds = torch.empty(len(losses), n, *labels.shape[1:], dtype=torch.int64)
    inter = torch.zeros(len(losses), n, C, dtype=torch.int64)
    union = torch.zeros_like(inter)
    valid = torch.zeros(n, dtype=torch.int64)
    for a, loss in enumerate(losses):
        for i in range(0, n, batch):
            idx = list(range(i, min(i + batch, n)))
            x, y = images[idx].to(dev), labels[idx].to(dev).contiguous()
            _, _, _, p = attacker.apgd_largereps(model, x, y, weights.to(dev), norm="Linf", eps=eps, n_iter=n_iter, use_rs=True,
                                                 loss=loss, track_loss="ce-avg", early_stop=True, num_classes=C,
                                                 return_pred=True,
                                                 noises=(noise_fn(idx, a) if noise_fn is not None
                                                         else image_noises(idx, a, tuple(x.shape[1:]), dev)))
            im, pm, tc = N.class_counts(p, y, C, per_image=True, mask_pred=True)
            inter[a, idx], union[a, idx], valid[idx] = im.cpu(), (tc + pm - im).cpu(), tc.sum(-1).cpu()
            pl = p.long()
            pl[y == -1] = -1
            preds[a, idx] = pl.cpu()
    attacker.release_graph_cache(model)      # the captured pair and its activation pool (several GB) do not outlive the evaluation
    if tables is not None:
        tables.update(inter=inter, union=union, valid=valid)
    worst, _, _ = worst_acc_from_counts(inter.sum(-1), valid)
    st = random.getstate()
    random.seed(225)
    miou, _, _ = worst_miou_from_tables(inter, union)
    random.setstate(st)
    return preds, worst, miou
