"""CPU oracle for the SEA / PIR-AT attack hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (plain PyTorch-CPU tensor ops, float32 unless a
function says otherwise) of the algorithms in the reference repository
nmndeep/Robust-Segmentation.  It exists so that the hand-written HIP kernels and the device-resident
drivers in ``robust-segmentation_amd/`` can be checked for parity.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the product
package never does (it raises when the HIP library is missing instead of falling back here).

Parity status: PINNED.  Every function below is checked in ``tests/test_oracle_golden.py`` against
golden vectors that ``oracle/gen_goldens.py`` produced by importing the real reference from
``/root/reference`` (torch 2.10 CPU) in the build container; the vectors live in ``tests/golden``.

Each function cites the reference lines (relative to the reference repo root) it restates.
"""
from __future__ import annotations

import math
import random
import statistics
from typing import Optional, Sequence

import torch

LN2 = math.log(2.0)

# loss-mode ids shared with include/sea_hip.h
MODE_MASK_CE = 0      # "mask-ce-avg"
MODE_MASK_CE_BAL = 1  # "mask-ce-bal"
MODE_JS = 2           # "js-avg"
MODE_CE = 3           # "ce" / "ce-avg"
MODE_BY_NAME = {"mask-ce-avg": 0, "mask-ce-bal": 1, "js-avg": 2, "ce": 3, "ce-avg": 3}


# --------------------------------------------------------------------------------------------------
# K1 / K5 / K6 : elementwise L-inf arithmetic (bit-exact contract)
# --------------------------------------------------------------------------------------------------
def _box(z: torch.Tensor, x: torch.Tensor, eps: float) -> torch.Tensor:
    """min(max(z, x-eps), x+eps) then clip to [0,1]  (semseg/attacker.py:397-399)."""
    return torch.clamp(torch.min(torch.max(z, x - eps), x + eps), 0.0, 1.0)


def apgd_linf_step(x, x_adv, x_old, grad, step, eps: float, a: float) -> torch.Tensor:
    """One APGD L-inf update with momentum (semseg/attacker.py:389-410, 456).

    ``x_old`` is the iterate before ``x_adv``; ``step`` is the per-image step size (B,) or
    (B,1,1,1).  Returns the new iterate.  The caller performs ``x_old <- x_adv`` (line 390).
    """
    step = step.view(-1, 1, 1, 1)
    g2 = x_adv - x_old
    z = x_adv + step * torch.sign(grad)
    z = _box(z, x, eps)
    z = x_adv + (z - x_adv) * a + g2 * (1 - a)
    return _box(z, x, eps)


def l2_norm(t: torch.Tensor) -> torch.Tensor:
    """per-image L2 norm, kept broadcastable: autoattack.other_utils.L2_norm(t, keepdim=True) (autoattack@a3922004:
    ``(x ** 2).view(x.shape[0], -1).sum(-1).sqrt()``), the helper semseg/attacker.py:6 imports"""
    return (t ** 2).reshape(t.shape[0], -1).sum(-1).sqrt().view(-1, *[1] * (t.dim() - 1))


def apgd_l2_step(x, x_adv, x_old, grad, step, eps: float, a: float) -> torch.Tensor:
    """One APGD L2 update with momentum (semseg/attacker.py:412-436, 456): a normalised gradient step, projection onto
    the eps-ball around x and [0, 1], the momentum combination, projection again."""
    step = step.view(-1, 1, 1, 1)
    g2 = x_adv - x_old

    def project(z):
        d = z - x
        n = l2_norm(d)
        return torch.clamp(x + d / (n + 1e-12) * torch.min(eps * torch.ones_like(x), n), 0.0, 1.0)

    z = project(x_adv + step * grad / (l2_norm(grad) + 1e-12))
    z = x_adv + (z - x_adv) * a + g2 * (1 - a)
    return project(z)


def linf_random_start(x, u, eps: float) -> torch.Tensor:
    """clip(x + eps*(2u-1), 0, 1) with u ~ U[0,1) supplied by the caller (semseg/attacker.py:293-294)."""
    t = 2 * u - 1
    return (x + eps * t).clamp(0.0, 1.0)


def linf_project(z, x, eps: float) -> torch.Tensor:
    """clip(x + clip(z-x, -eps, eps), 0, 1): stage re-projection (semseg/attacker.py:683-690)."""
    return (x + (z - x).clamp(-eps, eps)).clamp(0.0, 1.0)


def pgd_linf_step(X, delta, grad, alpha: float, eps: float) -> torch.Tensor:
    """PIR-AT inner PGD update of the perturbation (semseg/val.py:209-214 and 168-172)."""
    d = delta + alpha * torch.sign(grad)
    d = (X + d).clamp(0.0, 1.0) - X
    return d.clamp(-eps, eps)


# --------------------------------------------------------------------------------------------------
# K2 : per-pixel SEA losses, their logit gradients, accuracy and argmax
# --------------------------------------------------------------------------------------------------
def argmax_first(logits: torch.Tensor) -> torch.Tensor:
    """Index of the first maximum over the class dim (torch.max semantics, attacker.py:145, 370)."""
    return logits.max(1)[1]


def pixel_losses(logits, y, weights=None, mode: int = MODE_MASK_CE, dtype=torch.float32):
    """Per-pixel loss map (B,H,W) for one of the four modes.

    mask-ce-avg  semseg/attacker.py:143-152   1[argmax=y, y!=-1] * (lse - z_y)
    mask-ce-bal  semseg/attacker.py:155-173   same * w[y]
    js-avg       semseg/attacker.py:187-234   1[y!=-1] * JS(softmax(z) || onehot(y)) (closed form in p_y)
    ce / ce-avg  semseg/attacker.py:252-253   1[y!=-1] * (lse - z_y)
    """
    z = logits.to(dtype)
    valid = y != -1
    ys = torch.where(valid, y, torch.zeros_like(y))
    lse = torch.logsumexp(z, dim=1)
    zy = z.gather(1, ys.unsqueeze(1)).squeeze(1)
    ce = lse - zy
    if mode == MODE_CE:
        return torch.where(valid, ce, torch.zeros_like(ce))
    if mode in (MODE_MASK_CE, MODE_MASK_CE_BAL):
        m = valid & (argmax_first(logits) == y)
        out = torch.where(m, ce, torch.zeros_like(ce))
        if mode == MODE_MASK_CE_BAL:
            out = out * weights.to(dtype)[ys]
        return out
    if mode == MODE_JS:
        logp = zy - lse                      # ln p_y  (<= 0)
        py = torch.exp(logp)
        js = LN2 + 0.5 * (py * logp - (1.0 + py) * torch.log1p(py))
        return torch.where(valid, js, torch.zeros_like(js))
    raise ValueError(mode)


def pixel_loss_grad(logits, y, weights=None, mode: int = MODE_MASK_CE, dtype=torch.float32):
    """d(sum_b mean_px(mask_bg * loss)) / d logits, closed form (autograd of
    semseg/attacker.py:347-350 / 462-469).  Upstream factor is mask_bg/(H*W) per pixel."""
    z = logits.to(dtype)
    B, C = z.shape[:2]
    hw = z.shape[2] * z.shape[3]
    valid = y != -1
    ys = torch.where(valid, y, torch.zeros_like(y))
    p = torch.softmax(z, dim=1)
    onehot = torch.zeros_like(p).scatter_(1, ys.unsqueeze(1), 1.0)
    if mode in (MODE_MASK_CE, MODE_MASK_CE_BAL, MODE_CE):
        if mode == MODE_CE:
            coef = valid.to(dtype)
        else:
            coef = (valid & (argmax_first(logits) == y)).to(dtype)
        if mode == MODE_MASK_CE_BAL:
            coef = coef * weights.to(dtype)[ys]
        g = (p - onehot) * coef.unsqueeze(1)
    elif mode == MODE_JS:
        lse = torch.logsumexp(z, dim=1)
        logp = z.gather(1, ys.unsqueeze(1)).squeeze(1) - lse
        py = torch.exp(logp)
        coef = 0.5 * (logp - torch.log1p(py)) * py * valid.to(dtype)
        g = (onehot - p) * coef.unsqueeze(1)
    else:
        raise ValueError(mode)
    return g / hw


def pixel_to_img_loss(loss, mask_background=None):
    """mean over ALL pixels of mask*loss (semseg/attacker.py:237-240)."""
    if mask_background is not None:
        loss = mask_background * loss
    return loss.reshape(loss.shape[0], -1).mean(-1)


def loss_fwd_bwd(logits, y, weights, mode, track_mode=None, with_grad=True, ignored_correct=True):
    """Everything the fused HIP kernel K2 produces for one batch of logits.

    Returns dict(dlogits, loss_img, track_img, n_correct, acc_img, pred):
      loss_img / track_img  per-image mean of the attack / tracking loss (attacker.py:462-464, 473-474)
      n_correct             #pixels with argmax==y among valid pixels (int64)
      acc_img               per-image pixel accuracy; ignored pixels count as correct inside the loop
                            (attacker.py:485-490) but as wrong at step 0 (attacker.py:370-371)
      pred                  argmax map, int64 (attacker.py:373, 495)
    """
    B = logits.shape[0]
    hw = logits.shape[2] * logits.shape[3]
    mask_bg = (y != -1).float()
    pred = argmax_first(logits)
    li = pixel_to_img_loss(pixel_losses(logits, y, weights, mode), mask_bg)
    if track_mode is None or track_mode == mode:
        ti = li.clone()
    else:
        ti = pixel_to_img_loss(pixel_losses(logits, y, weights, track_mode), mask_bg)
    ok = (pred == y)
    n_correct = ok.reshape(B, -1).sum(-1)
    n_ign = (y == -1).reshape(B, -1).sum(-1)
    cnt = n_correct + (n_ign if ignored_correct else 0)
    acc = cnt.float() / float(hw)
    out = dict(loss_img=li, track_img=ti, n_correct=n_correct, acc_img=acc, pred=pred, dlogits=None)
    if with_grad:
        out["dlogits"] = pixel_loss_grad(logits, y, weights, mode)
    return out


def loss_fwd_bwd_upsampled(low, y, weights, mode, track_mode=None, with_grad=True, ignored_correct=True):
    """K2u: bilinear upsample of the low-res logits to the label resolution
    (semseg/models/uperforseg.py:416-418, segmenter.py:228) followed by everything K2 computes; the
    gradient is returned w.r.t. the LOW-RES logits (autograd through F.interpolate)."""
    lo = low.detach().clone().requires_grad_(with_grad)
    hi = torch.nn.functional.interpolate(lo, size=tuple(y.shape[-2:]), mode="bilinear", align_corners=False)
    out = loss_fwd_bwd(hi.detach(), y, weights, mode, track_mode, with_grad=with_grad, ignored_correct=ignored_correct)
    out["logits_hi"] = hi.detach()
    if with_grad:
        (out["dlow"],) = torch.autograd.grad(hi, [lo], grad_outputs=out["dlogits"])
    return out


# --------------------------------------------------------------------------------------------------
# K3 : per-class counts / confusion matrix
# --------------------------------------------------------------------------------------------------
def class_counts(pred, y, n_cls: int, per_image: bool = False, mask_pred: bool = True):
    """Integer per-class statistics.

    inter[c]    #{pred==y==c}
    tgt_cnt[c]  #{y==c}
    pred_cnt[c] #{pred==c}; with mask_pred the prediction is first overwritten by the ignore label
                wherever y==-1 (semseg/attacker.py:20, tools/infer.py:90); tools/worse_only.py:49-66
                does NOT do that overwrite (mask_pred=False).
    union = tgt_cnt + pred_cnt - inter  (attacker.py:41-45, infer.py:112-116, worse_only.py:60-64).
    Shapes (C,) or (B,C); int64.
    """
    B = pred.shape[0]
    p = pred.reshape(B, -1)
    t = y.reshape(B, -1)
    if mask_pred:
        p = torch.where(t == -1, torch.full_like(p, -1), p)
    inter = torch.zeros(B, n_cls, dtype=torch.int64)
    tc = torch.zeros(B, n_cls, dtype=torch.int64)
    pc = torch.zeros(B, n_cls, dtype=torch.int64)
    for b in range(B):
        tb, pb = t[b], p[b]
        vt = (tb >= 0) & (tb < n_cls)
        tc[b] = torch.bincount(tb[vt], minlength=n_cls)[:n_cls]
        vp = (pb >= 0) & (pb < n_cls)
        pc[b] = torch.bincount(pb[vp], minlength=n_cls)[:n_cls]
        eq = vt & (pb == tb)
        inter[b] = torch.bincount(tb[eq], minlength=n_cls)[:n_cls]
    if per_image:
        return inter, pc, tc
    return inter.sum(0), pc.sum(0), tc.sum(0)


def confusion_matrix(pred, y, n_cls: int, ignore_label: int = -1):
    """hist[t, p] over pixels with y != ignore (semseg/metrics.py:27-33); int64 (K,K)."""
    keep = y != ignore_label
    idx = y[keep] * n_cls + pred[keep]
    return torch.bincount(idx, minlength=n_cls * n_cls)[: n_cls * n_cls].view(n_cls, n_cls)


def compute_iou_acc(pred, y, n_cls: int):
    """(m_acc, a_acc, m_iou) as float32 0-dim tensors (semseg/attacker.py:9-52).

    The reference accumulates counts in float32 (exact below 2^24 per class); it also overwrites
    ``pred`` with -1 at ignored pixels in place (line 20) -- mirrored here.
    """
    pred[y == -1] = -1
    inter, pc, tc = class_counts(pred, y, n_cls, per_image=False, mask_pred=False)
    inter, pc, tc = inter.float(), pc.float(), tc.float()
    union = tc + pc - inter
    ind = tc > 0
    m_acc = (inter[ind] / tc[ind]).mean()
    a_acc = inter.sum() / tc.sum()
    ind = union > 0
    m_iou = (inter[ind] / union[ind]).mean()
    return m_acc, a_acc, m_iou


def metrics_from_hist(hist: torch.Tensor):
    """IoU / F1 / pixel-acc summaries of a float32 confusion matrix (semseg/metrics.py:35-60)."""
    hist = hist.float()
    d = hist.diag()
    ious = d / (hist.sum(0) + hist.sum(1) - d)
    miou = ious[~ious.isnan()].mean().item()
    f1 = 2 * d / (hist.sum(0) + hist.sum(1))
    mf1 = f1[~f1.isnan()].mean().item()
    acc = d / hist.sum(1)
    aacc = d.sum() / hist.sum()
    macc = acc[~acc.isnan()].mean().item()
    return dict(
        ious=(ious * 100).numpy().round(2).tolist(), miou=round(miou * 100, 2),
        f1=(f1 * 100).numpy().round(2).tolist(), mf1=round(mf1 * 100, 2),
        acc=(acc * 100).numpy().round(2).tolist(), macc=round(macc * 100, 2),
        aacc=(aacc * 100).numpy().round(2),
    )


# --------------------------------------------------------------------------------------------------
# K7 : step-size controller
# --------------------------------------------------------------------------------------------------
def apgd_schedule(n_iter: int):
    """(k0, k_min, k_decr) for Linf/L2 (semseg/attacker.py:323-329)."""
    return max(int(0.22 * n_iter), 1), max(int(0.06 * n_iter), 1), max(int(0.03 * n_iter), 1)


def checkpoints(n_iter: int):
    """Iterations i (0-based) at which the step-size check fires and the window k used there.

    The schedule is data independent (semseg/attacker.py:528-551): counter3 counts iterations since
    the last check, a check happens when it reaches k, afterwards k <- max(k - decr, k_min).
    """
    k, kmin, dec = apgd_schedule(n_iter)
    out, c3 = {}, 0
    for i in range(n_iter):
        c3 += 1
        if c3 == k:
            out[i] = k
            c3 = 0
            k = max(k - dec, kmin)
    return out


def check_oscillation(loss_steps, j: int, k: int, k3: float = 0.75):
    """1.0 where the tracked loss increased in at most k3*k of the last k steps
    (semseg/attacker.py:243-248).  Row indices wrap like Python negative indices."""
    n = loss_steps.shape[0]
    t = torch.zeros(loss_steps.shape[1])
    for c in range(k):
        t += (loss_steps[(j - c) % n] > loss_steps[(j - c - 1) % n]).float()
    return (t <= k * k3 * torch.ones_like(t)).float()


# --------------------------------------------------------------------------------------------------
# a1 / a2 : APGD drivers
# --------------------------------------------------------------------------------------------------
def _model_logits_and_grad(model, x_adv, y, weights, mode, want_grad=True):
    xin = x_adv.detach().clone().requires_grad_(want_grad)
    logits = model(xin)
    if want_grad:
        dl = pixel_loss_grad(logits.detach(), y, weights, mode)
        (g,) = torch.autograd.grad(logits, [xin], grad_outputs=dl)
        return logits.detach(), g.detach()
    return logits.detach(), None


def apgd_train(model, x, y, norm="Linf", eps=8.0 / 255, n_iter=10, use_rs=False, loss="mask-ce-avg",
               early_stop=False, track_loss=None, x_init=None, weights=None, noise=None, trace=None):
    """One APGD run, L-inf or L2 (semseg/attacker.py:260-571; the L1 branch, lines 437-454 and 553-566,
    is not restated: no shipped entry point reaches it).  The reference has a random start for L-inf only
    (lines 291-294): ``use_rs`` with L2 needs ``x_init``, as there.

    ``noise`` replaces ``torch.rand_like(x)`` (line 293) so device and CPU runs can share it; when
    None and use_rs, the global torch CPU generator is consumed exactly like the reference.
    ``trace`` (optional dict) receives loss_steps / step sizes / acc per step for the tests.
    Returns (x_best, acc, loss_best, x_best_adv).
    """
    assert norm in ("Linf", "L2")
    assert not model.training
    assert norm == "Linf" or not use_rs or x_init is not None
    mode = MODE_BY_NAME[loss]
    tmode = MODE_BY_NAME[track_loss] if track_loss is not None else mode
    B = x.shape[0]
    if not use_rs or norm != "Linf":
        x_adv = x.clone()
    else:
        u = torch.rand_like(x) if noise is None else noise
        x_adv = linf_random_start(x, u, eps)
    if x_init is not None:
        x_adv = x_init.clone()
    x_adv = x_adv.clamp(0.0, 1.0)
    x_best = x_adv.clone()
    x_best_adv = x_adv.clone()
    loss_steps = torch.zeros(n_iter, B)
    step = 2.0 * eps * torch.ones(B)
    cps = checkpoints(n_iter)

    logits, grad = _model_logits_and_grad(model, x_adv, y, weights, mode)
    st = loss_fwd_bwd(logits, y, weights, mode, tmode, with_grad=False, ignored_correct=False)
    acc = st["acc_img"].clone()
    pred_best = st["pred"].clone()
    loss_best = st["track_img"].clone()
    loss_best_last = loss_best.clone()
    reduced_last = torch.ones(B)
    grad_best = grad.clone()
    x_old = x_adv.clone()
    steps_hist = []

    for i in range(n_iter):
        a = 0.75 if i > 0 else 1.0
        x_new = (apgd_linf_step if norm == "Linf" else apgd_l2_step)(x, x_adv, x_old, grad, step, eps, a)
        x_old = x_adv
        x_adv = x_new
        want = i < n_iter - 1
        logits, g = _model_logits_and_grad(model, x_adv, y, weights, mode, want_grad=want)
        if want:
            grad = g
        st = loss_fwd_bwd(logits, y, weights, mode, tmode, with_grad=False, ignored_correct=True)
        avg_acc = st["acc_img"]
        ind = avg_acc <= acc
        acc = torch.min(acc, avg_acc)
        x_best_adv = torch.where(ind.view(-1, 1, 1, 1), x_adv, x_best_adv)
        pred_best = torch.where(ind.view(-1, 1, 1), st["pred"], pred_best)

        y1 = st["track_img"]
        loss_steps[i] = y1
        ind = y1 > loss_best
        v4 = ind.view(-1, 1, 1, 1)
        x_best = torch.where(v4, x_adv, x_best)
        grad_best = torch.where(v4, grad, grad_best)
        loss_best = torch.where(ind, y1, loss_best)

        if i in cps:
            k = cps[i]
            fl = check_oscillation(loss_steps, i, k)
            no_impr = (1.0 - reduced_last) * (loss_best_last >= loss_best).float()
            fl = torch.max(fl, no_impr)
            reduced_last = fl.clone()
            loss_best_last = loss_best.clone()
            r = fl > 0
            step = torch.where(r, step / 2.0, step)
            r4 = r.view(-1, 1, 1, 1)
            x_adv = torch.where(r4, x_best, x_adv)
            grad = torch.where(r4, grad_best, grad)
        steps_hist.append(step.clone())
        if early_stop and acc.sum() == 0:
            break
    if trace is not None:
        trace.update(loss_steps=loss_steps, steps=torch.stack(steps_hist) if steps_hist else None,
                     pred_best=pred_best, n_done=i + 1 if n_iter else 0)
    return x_best, acc, loss_best, x_best_adv


def apgd_restarts(model, x, y, eps=8.0 / 255, n_iter=10, loss="mask-ce-avg", n_restarts=1, early_stop=False,
                  track_loss=None, use_rs=False, noises=None):
    """APGD with restarts on the images whose pixel accuracy is still > 0 (semseg/attacker.py:574-659).
    ``noises[r]`` replaces the random start of restart r (it has the shape of the active sub-batch).
    Returns (x_adv, acc of the last apgd_train call, acc)."""
    B = x.shape[0]
    acc = torch.ones(B)
    x_adv = x.clone()
    acc_last = None
    for r in range(n_restarts):
        ind = acc > 0
        if acc.sum() > 0:
            _, acc_last, _, cand = apgd_train(model, x[ind], y[ind], eps=eps, n_iter=n_iter, use_rs=use_rs, loss=loss,
                                              early_stop=early_stop, track_loss=track_loss,
                                              noise=None if noises is None else noises[r])
            with torch.no_grad():
                ok = model(cand).max(1)[1] == y[ind]
            ok[y[ind] == -1] = True
            acc_c = ok.float().reshape(cand.shape[0], -1).mean(-1)
            upd = acc_c < acc[ind]
            rows = torch.nonzero(ind).flatten()[upd]
            x_adv[rows] = cand[upd]
            acc[rows] = acc_c[upd]
    return x_adv, acc_last, acc


def largereps_schedule(n_iter: int, eps: float):
    """Stage lengths and radii of the 3-stage schedule (semseg/attacker.py:693-695)."""
    n_iters = [int(c * n_iter) for c in (0.3, 0.3)]
    n_iters.append(n_iter - sum(n_iters))
    return n_iters, [c * eps for c in (2, 1.5, 1)]


def apgd_largereps(model, x, y, weights, norm="Linf", eps=8.0 / 255, n_iter=10, loss="mask-ce-avg",
                   early_stop=False, track_loss=None, use_rs=False, noises=None):
    """3-stage radius schedule around apgd_train (semseg/attacker.py:662-728).
    Returns (x_adv, None, acc) where x_adv is the lowest-accuracy iterate of the last stage."""
    n_iters, epss = largereps_schedule(n_iter, eps)
    x_init, acc = None, torch.ones(x.shape[0])
    for s, (it, e) in enumerate(zip(n_iters, epss)):
        if x_init is not None:
            x_init = linf_project(x_init, x, e)
        _, acc, _, x_init = apgd_train(
            model, x, y, norm=norm, eps=e, n_iter=it, use_rs=use_rs, loss=loss, early_stop=early_stop,
            track_loss=track_loss, x_init=x_init, weights=weights,
            noise=None if noises is None else noises[s])
    return x_init, None, acc


# --------------------------------------------------------------------------------------------------
# a13 : PIR-AT inner PGD
# --------------------------------------------------------------------------------------------------
def _val_loss(logits, y, los: str):
    """The `losses` table of semseg/val.py:104-127 (no ignore handling there)."""
    B = logits.shape[0]
    if los == "l2-loss":   # here `y` is a tensor of the logits' shape (semseg/val.py:125)
        return ((logits - y) ** 2).reshape(B, -1).sum(-1)
    lse = torch.logsumexp(logits, 1)
    zy = logits.gather(1, y.unsqueeze(1)).squeeze(1)
    ce = lse - zy
    if los == "pgd":
        return ce.mean()
    if los == "mask-ce-avg":
        m = (argmax_first(logits) == y).float()
        return (m * ce).reshape(B, -1).mean(-1)
    if los == "js-avg":
        logp = zy - lse
        py = torch.exp(logp)
        js = LN2 + 0.5 * (py * logp - (1.0 + py) * torch.log1p(py))
        return js.reshape(B, -1).mean(-1)
    raise ValueError(los)


def js_div_general(p, q, softmax_output=False, reduction="none", red_dim=None):
    """js_div_fn for arbitrary arguments (semseg/attacker.py:187-226), written out element-wise:
    0.5 * (KL(p || m) + KL(onehot || m)) per class with m = (p + onehot) / 2, the 0*log(0) = 0 convention of
    F.kl_div for zero targets, ignored pixels zeroed, optional sum over ``red_dim``.  ``reduction="sum"`` is
    only legal when every pixel is ignored (line 209); the result is then that scalar times the (all-zero)
    mask, shape (B,1,H,W)."""
    prob = p if softmax_output else torch.softmax(p, 1)
    keep = q != -1
    if reduction != "none" and keep.sum() > 0:
        raise ValueError("Incompatible setup.")
    ys = torch.where(keep, q, torch.zeros_like(q))
    onehot = torch.zeros_like(prob).scatter_(1, ys.unsqueeze(1), 1.0)
    log_m = torch.log((prob + onehot) / 2)
    term = 0.5 * ((torch.xlogy(prob, prob) - prob * log_m) + (torch.xlogy(onehot, onehot) - onehot * log_m))
    if reduction == "sum":
        term = term.sum()
    elif reduction == "mean":
        term = term.mean()
    out = keep.unsqueeze(1).to(prob.dtype) * term
    if red_dim is not None:
        out = out.sum(dim=red_dim)
    return out


def pgd_attack_1(model, X, y, epsilon=4.0 / 255, alpha=1e-2, num_iter=2, los="pgd", delta0=None):
    """Pgd_Attack_1.adv_attack (semseg/val.py:181-218): random start, model sees X+delta unclamped.
    ``delta0`` replaces ``delta.uniform_(-eps, eps)``.  Returns (x_adv, last logits)."""
    delta = delta0.clone() if delta0 is not None else torch.zeros_like(X).uniform_(-epsilon, epsilon)
    logits = None
    for _ in range(num_iter):
        d = delta.clone().requires_grad_(True)
        logits = model(X + d)
        loss = _val_loss(logits, y, los).sum()
        (g,) = torch.autograd.grad(loss, [d])
        delta = pgd_linf_step(X, delta, g, alpha, epsilon)
    return (X + delta).clamp(0.0, 1.0).detach(), None if logits is None else logits.detach()


def pgd_attack(model, X, y, eps=4.0 / 255, alpha=1e-2, num_iter=2, los="mask-ce-avg"):
    """Pgd_Attack.adv_attack (semseg/val.py:130-178): zero start, clamped input, keeps the delta
    *after* the step for images whose pre-step loss was >= the running best (lines 158-175)."""
    delta = torch.zeros_like(X)
    best = torch.zeros(X.shape[0])
    best_delta = torch.zeros_like(X)
    for _ in range(num_iter):
        d = delta.clone().requires_grad_(True)
        logits = model((X + d).clamp(0.0, 1.0))
        loss = _val_loss(logits, y, los)
        ind = loss.detach() >= best
        best = torch.where(ind, loss.detach(), best)
        (g,) = torch.autograd.grad(loss.sum(), [d])
        delta = pgd_linf_step(X, delta, g, alpha, eps)
        best_delta = torch.where(ind.view(-1, 1, 1, 1), delta, best_delta)
    return (X + best_delta).clamp(0.0, 1.0).detach()


# --------------------------------------------------------------------------------------------------
# a17-a19 : dataset-level statistics and worst-case bookkeeping
# --------------------------------------------------------------------------------------------------
def eval_stats_from_counts(inter, pred_cnt, tgt_cnt):
    """{mAcc, aAcc, mIoU} from accumulated per-class counts (tools/infer.py:93-118, 131).
    Counts are converted to float32 as in the reference (which accumulates float32)."""
    inter, pc, tc = inter.float(), pred_cnt.float(), tgt_cnt.float()
    union = tc + pc - inter
    ind = tc > 0
    m_acc = (inter[ind] / tc[ind]).mean()
    a_acc = inter.sum() / tc.sum()
    ind = union > 0
    m_iou = (inter[ind] / union[ind]).mean()
    return {"mAcc": m_acc.item(), "aAcc": a_acc.item(), "mIoU": m_iou.item()}


def eval_performance(model, loader, n_batches: int = -1, n_cls: int = 21):
    """Clean / adversarial evaluation pass (tools/infer.py:56-133): ``loader`` yields (input, target, ...);
    returns ({mAcc, aAcc, mIoU}, concatenated argmax maps (N,H,W) int64).  The returned maps carry the ignore
    label at ignored pixels because the reference overwrites ``pred`` in place after appending it (lines 88-90)."""
    inter = torch.zeros(n_cls, dtype=torch.int64)
    pc = torch.zeros(n_cls, dtype=torch.int64)
    tc = torch.zeros(n_cls, dtype=torch.int64)
    outs = []
    for i, vals in enumerate(loader):
        inp, target = vals[0], vals[1]
        with torch.no_grad():
            pred = model(inp).max(1)[1]
        pred[target == -1] = -1
        outs.append(pred)
        a, b, c = class_counts(pred, target, n_cls, per_image=False, mask_pred=False)
        inter += a
        pc += b
        tc += c
        if i + 1 == n_batches:
            break
    return eval_stats_from_counts(inter, pc, tc), torch.cat(outs)


def worst_case_acc(preds: torch.Tensor, targets: torch.Tensor, n_cls: int, bs: Optional[int] = None):
    """Worst-case aAcc over attacks (tools/worse_only.py:351-422).

    preds (A,N,H,W) int64, targets (N,H,W).  Returns (worst_Acc float, worst_Acc_indiv (A,),
    matrix (A,N) float32).  With ``bs`` given, the reference's batch slicing is reproduced
    literally: predictions for loader batch i are taken from ``[i*BS : i*BS+BS]`` with BS the
    CURRENT batch's size (worse_only.py:374-378), which mis-aligns the final partial batch when
    N % bs != 0.  With ``bs=None`` images are aligned correctly."""
    A, N = preds.shape[:2]
    if bs is None:
        pa = preds
    else:
        rows = []
        for i, s in enumerate(range(0, N, bs)):
            BS = min(bs, N - s)
            rows.append(preds[:, i * BS: i * BS + BS])
        pa = torch.cat(rows, dim=1)
    valid = (targets >= 0) & (targets < n_cls)
    n_valid = valid.reshape(N, -1).sum(-1).float()
    corr = ((pa == targets.unsqueeze(0)) & valid.unsqueeze(0)).reshape(A, N, -1).sum(-1).float()
    mat = corr / n_valid.unsqueeze(0)
    return mat.min(0)[0].mean().item(), mat.mean(-1), mat


def per_image_tables(preds: torch.Tensor, targets: torch.Tensor, n_cls: int):
    """cons_ints / cons_unions (A,N,C) float32 (tools/worse_only.py:200-234, 49-66)."""
    A, N = preds.shape[:2]
    ints = torch.zeros(A, N, n_cls)
    unis = torch.zeros(A, N, n_cls)
    for a in range(A):
        inter, pc, tc = class_counts(preds[a], targets, n_cls, per_image=True, mask_pred=False)
        ints[a] = inter.float()
        unis[a] = (tc + pc - inter).float()
    return ints, unis


def _miou(inters: Sequence[float], unions: Sequence[float]) -> float:
    """tools/worse_only.py:69-76: mean over classes with union != 0 of int/union."""
    vals = [a / b for a, b in zip(inters, unions) if b != 0]
    return statistics.mean(vals)


def _miou_sub(run_i, run_u, d_i, d_u):
    """tools/worse_only.py:79-93: candidate totals + mean of int/(union+1e-8); classes whose OLD
    union is zero are dropped from the returned (compacted) lists."""
    iou, uni, vals = [], [], []
    for a, b, c, d in zip(run_i, run_u, d_i, d_u):
        if b == 0:
            continue
        iou.append(a + c)
        uni.append(b + d)
        vals.append(iou[-1] / (uni[-1] + 1e-8))
    return statistics.mean(vals), iou, uni


def worst_case_miou(ints: torch.Tensor, unis: torch.Tensor, orders=None, n_rounds: int = 1000,
                    rng: Optional[random.Random] = None):
    """Greedy per-image attack selection minimising mIoU (tools/worse_only.py:279-334).

    ints/unis (A,N,C) float32 tables.  ``orders`` optionally supplies the per-round image orders
    (list of lists); otherwise ``rng.shuffle`` (or the global ``random``) is used as in the
    reference.  All quirks are reproduced: the stale threshold inside the per-image attack loop,
    the +1e-8 only in the candidate, the compaction of never-seen classes (zip truncation), the
    float32 table arithmetic, float(…) of float32 sums, ``statistics.mean``.
    Returns (final_miou, selected list, rounds run).
    """
    A, N, C = ints.shape
    # running totals start from attack 0, accumulated image by image in float32 (worse_only.py:241-250, 30-46)
    run_i_t = torch.zeros(C)
    run_u_t = torch.zeros(C)
    for n in range(N):
        run_i_t += ints[0, n]
        run_u_t += unis[0, n]
    run_i = [v.item() for v in run_i_t]
    run_u = [v.item() for v in run_u_t]
    final = _miou(run_i, run_u)
    sel = [0] * N
    prev_best = 10
    shuffle = (rng.shuffle if rng is not None else random.shuffle)
    rounds = 0
    for r in range(n_rounds):
        rounds += 1
        if orders is not None:
            order = list(orders[r])
        else:
            order = list(range(N))
            shuffle(order)
        for idx in order:
            for a in range(A):
                # float32 tensor arithmetic exactly like torch.tensor(list) + table differences
                est_i = torch.tensor(run_i)
                est_u = torch.tensor(run_u)
                d_i = ints[a, idx] - ints[sel[idx], idx]
                d_u = unis[a, idx] - unis[sel[idx], idx]
                est, new_i, new_u = _miou_sub(
                    [v.item() for v in est_i], [v.item() for v in est_u],
                    [v.item() for v in d_i], [v.item() for v in d_u])
                if est < final:
                    sel[idx] = a
                    run_i, run_u = new_i, new_u
            final = _miou([torch.tensor(v).item() for v in run_i], [torch.tensor(v).item() for v in run_u])
        if prev_best - final <= 1e-6:
            break
        prev_best = final
    return final, sel, rounds
