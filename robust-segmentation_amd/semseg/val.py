"""PIR-AT inner PGD attacks and confusion-matrix evaluation (counterpart of semseg/val.py).

``Pgd_Attack_1`` / ``Pgd_Attack`` keep the reference's constructor and ``adv_attack`` signatures
(val.py:130-218).  Each inner step is: model forward, ONE fused loss/gradient kernel on the logits
(K2), input-gradient backward, ONE fused sign-step/projection kernel (K6).  Unlike the reference the
backward is ``autograd.grad`` w.r.t. the input only: no parameter ``.grad`` is touched and, under DDP,
no gradient all-reduce fires inside the inner loop.
"""
from __future__ import annotations

import os
import sys
import weakref

import torch

from . import _native as N
from . import attacker as _A
from .attacker import _FrozenParameters
from .metrics import Metrics

__all__ = ["Pgd_Attack", "Pgd_Attack_1", "evaluate", "losses", "js_loss", "masked_cross_entropy"]


def js_loss(p, q, reduction="mean"):
    """val.py:103-108: JS divergence summed over classes, per-image mean by default."""
    return _A.js_loss(p, q, reduction=reduction)


def masked_cross_entropy(pred, target):
    """val.py:111-117: per-image mean of the cross-entropy of the correctly classified pixels."""
    return _A.masked_cross_entropy(pred, target, reduction="mean")


class _ValLoss:
    """One entry of the `losses` table (val.py:121-127): a differentiable callable ``fn(logits, y)`` like the
    reference's, plus what the PGD classes below need to run it as ONE fused K2 launch instead of through
    autograd: ``mode`` (K2 loss id, None = no fused form) and ``per_image`` (False: scalar mean over the batch)."""

    def __init__(self, fn, mode, per_image):
        self.fn, self.mode, self.per_image = fn, mode, per_image

    def __call__(self, x, y):
        return self.fn(x, y)


# none of these knows ignore labels (F.cross_entropy's default ignore_index is -100 there)
losses = {
    "pgd": _ValLoss(lambda x, y: _A._ce(x, y).mean(), 3, False),
    "mask-ce-avg": _ValLoss(masked_cross_entropy, 0, True),
    "js-avg": _ValLoss(js_loss, 2, True),
    "l2-loss": _ValLoss(lambda x, y: ((x - y) ** 2).view(x.shape[0], -1).sum(-1), None, True),
}


def _labels(y):
    return y.long().contiguous() if y.dtype not in (torch.int64, torch.int32, torch.int16, torch.uint8) else y.contiguous()


def _fwd_grad(model, x_in, y, los, scale, ws, out, dlogits):
    """logits = model(x_in); returns (input gradient of sum of the loss, K2 stats, logits).  Losses without a
    fused form (``l2-loss``) go through autograd on the callable."""
    x_in = x_in.detach().requires_grad_(True)
    with torch.enable_grad(), _FrozenParameters(model):  # gradient w.r.t. the input only (val.py:150, 201)
        logits = model(x_in)
        if los.mode is None:
            li = los(logits, y)
            (g,) = torch.autograd.grad(li.sum(), [x_in])
            return g.contiguous(), dict(dlogits=None, loss_img=li.detach()), logits.detach()
    r = N.loss_fwd_bwd(logits.detach(), y, None, los.mode, los.mode, scale, want_grad=True, workspace=ws, out=out,
                       dlogits=dlogits)
    (g,) = torch.autograd.grad(logits, [x_in], grad_outputs=r["dlogits"])
    return g.contiguous(), r, logits.detach()


# ---- HIP-graph replay of the inner PGD (round 5; opt-in: SEA_PGD_GRAPH=1) ----------------------------------------------------
# PIR-AT runs this attack once per outer step on weights the optimizer has just changed.  A captured iteration used to be
# impossible because every weight-derived cache (packed weights, folded BatchNorm, Winograd-domain filters, analytic bounds)
# was re-created at a new address after each optimizer step.  Those caches now refresh IN PLACE
# (models/convnext_upernet.py:_stable), so: iteration 0 of every attack runs eagerly -- which re-derives every cache the
# forward and the backward touch, through the ordinary code path -- and iterations 1 .. n - 1 replay ONE captured graph
# (forward, fused loss / gradient kernel, input-gradient backward, sign step) over persistent buffers.  The graph is valid for
# one set of parameter objects / addresses and for one `_native.CACHE_EPOCH` (bumped whenever a cache value could NOT keep its
# address); anything else captures again.  Bitwise the eager loop across weight updates, fp32 and bf16 (tests/test_attack_gpu.py).
# MEASURED (BASELINE configs[3]: UperNet-ConvNeXt-S, B = 8, 512 x 512, bf16 autocast, 5 steps; devtools/pirat_host_vs_gpu.py,
# profiles/r5_pirat_host_vs_gpu.log): the host time to enqueue the attack drops from 84-103 ms to 24-27 ms per outer step,
# the attack's GPU time stays at 98-104 ms and the outer step at 194-197 ms -- at this size the eager loop was GPU-bound
# with the host just keeping pace, so the replay buys host time (eight ranks on one host), not throughput.  Hence opt-in.
PGD_GRAPH = os.environ.get("SEA_PGD_GRAPH", "0") == "1"
PGD_GRAPH_MIN_ITER = 3
_PGD_SLOTS = weakref.WeakKeyDictionary()     # model -> {key: _PgdSlot}


class _PgdSlot:
    def __init__(self, X, y):
        self.X, self.delta, self.x_in = torch.empty_like(X), torch.empty_like(X), torch.empty_like(X)
        self.y = torch.empty_like(y)
        B, HW = X.shape[0], X.shape[-2] * X.shape[-1]
        self.ws = N.loss_workspace(B, HW, X.device)
        self.out = tuple(torch.empty(B, dtype=d, device=X.device) for d in (torch.float32, torch.float32, torch.int32))
        self.dl = None
        self.graph = self.logits = self.pin = None
        self.epoch = self.wkey = None
        self.warm = False          # one eager iteration has run on the capture stream (library state per stream)
        self.failed = False


def _param_key(model):
    """what a captured inner-PGD graph is valid for besides `_native.CACHE_EPOCH`: the parameter / buffer tensors (identity and
    address; their CONTENTS may change, the caches refresh in place), which of them require a gradient (a frozen layer takes
    other kernels), and the process-global arithmetic state incl. ``model.training`` (attacker._arith_signature)"""
    ts = list(model.parameters()) + list(model.buffers())
    return hash((tuple((id(t), t.data_ptr(), bool(t.requires_grad)) for t in ts), _A._arith_signature(model)))


def release_pgd_graphs(model=None):
    """drop the captured inner-PGD graphs (and their activation pools) of ``model``, or of every model"""
    for m in ([model] if model is not None else list(_PGD_SLOTS.keys())):
        _PGD_SLOTS.pop(m, None)


class Pgd_Attack_1:
    """Random-start PGD, model sees X+delta unclamped (val.py:181-218)."""

    def __init__(self, epsilon=4.0 / 255.0, alpha=1e-2, num_iter=2, los="pgd"):
        self.epsilon, self.alpha, self.num_iter, self.los_name = epsilon, alpha, num_iter, los
        self.loss_fn = losses[los]
        self.mode, self.per_image = self.loss_fn.mode, self.loss_fn.per_image

    def adv_attack(self, model, X, y, delta0=None):
        model.eval()
        X = X.detach().contiguous().float()
        B, HW = X.shape[0], X.shape[-2] * X.shape[-1]
        y = _labels(y) if self.mode is not None else y
        delta = torch.zeros_like(X).uniform_(-self.epsilon, self.epsilon) if delta0 is None else delta0.clone()
        # F.cross_entropy(x, y) is the mean over all B*H*W pixels (val.py:122); the others are per-image means
        scale = 1.0 / (HW if self.per_image else B * HW)
        ws = N.loss_workspace(B, HW, X.device)
        out = tuple(torch.empty(B, dtype=d, device=X.device) for d in (torch.float32, torch.float32, torch.int32))
        if (PGD_GRAPH and self.mode is not None and self.num_iter >= PGD_GRAPH_MIN_ITER and X.is_cuda
                and isinstance(model, torch.nn.Module)):
            res = self._adv_attack_graph(model, X, y, delta, scale)
            if res is not None:
                return res
        x_in = X + delta
        logits, dl = None, None
        for _ in range(self.num_iter):
            g, r, logits = _fwd_grad(model, x_in, y, self.loss_fn, scale, ws, out, dl)
            dl = r["dlogits"]
            N.pgd_linf_step(X, delta, g, float(self.alpha), float(self.epsilon), delta_out=delta, x_in_out=x_in,
                            clamp_input=False)
        x_adv = (X + delta).clamp_(0.0, 1.0)
        return x_adv.detach(), logits, None

    def _adv_attack_graph(self, model, X, y, delta, scale):
        """iteration 0 eager (refreshes the weight-derived caches in place), the rest replayed from one captured graph; returns
        None when this model / shape cannot be captured (the caller then runs the eager loop from the same start)"""
        try:
            slots = _PGD_SLOTS.setdefault(model, {})
        except TypeError:
            return None
        ac = (torch.is_autocast_enabled(), torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else None)
        key = (tuple(X.shape), tuple(y.shape), y.dtype, X.device.index, self.mode, self.per_image, float(self.alpha),
               float(self.epsilon), ac)
        sl = slots.get(key)
        if sl is None:
            sl = slots[key] = _PgdSlot(X, y)
        if sl.failed:
            return None
        sl.X.copy_(X)
        sl.y.copy_(y)
        sl.delta.copy_(delta)
        torch.add(sl.X, sl.delta, out=sl.x_in)
        alpha, eps = float(self.alpha), float(self.epsilon)

        def iteration():
            g, r, logits = _fwd_grad(model, sl.x_in, sl.y, self.loss_fn, scale, sl.ws, sl.out, sl.dl)
            sl.dl = r["dlogits"]
            N.pgd_linf_step(sl.X, sl.delta, g, alpha, eps, delta_out=sl.delta, x_in_out=sl.x_in, clamp_input=False)
            return logits

        logits = iteration()                     # eager: every cache the forward and the backward use is current again
        wkey = _param_key(model)
        if sl.graph is not None and (sl.epoch != N.CACHE_EPOCH[0] or sl.wkey != wkey):
            sl.graph = sl.logits = sl.pin = None      # some cache moved, or other parameter tensors: capture again
        it = 1
        cur = torch.cuda.current_stream()
        gs = _A._capture_stream(X.device)
        if sl.graph is None:
            if not sl.warm:                      # per-stream library state must exist before a capture starts
                gs.wait_stream(cur)
                with torch.cuda.stream(gs):
                    logits = iteration()
                cur.wait_stream(gs)
                sl.warm = True
                it += 1
            if it < self.num_iter:
                graph = torch.cuda.CUDAGraph()
                try:
                    with torch.cuda.graph(graph, stream=gs):
                        sl.logits = iteration()
                except Exception as exc:         # a model that cannot be captured keeps the eager loop, from here on
                    torch.cuda.set_stream(cur)
                    torch.cuda.synchronize()
                    sl.failed = True
                    print(f"[sea] HIP-graph capture of the inner PGD failed ({type(exc).__name__}: "
                          f"{str(exc).splitlines()[0][:160]}); continuing with the eager loop", file=sys.stderr)
                    for _ in range(it, self.num_iter):
                        logits = iteration()
                    return (sl.X + sl.delta).clamp_(0.0, 1.0).detach(), logits, None
                sl.graph, sl.epoch, sl.wkey = graph, N.CACHE_EPOCH[0], wkey
                sl.pin = N.ksplit_workspace_pin(X.device)
        for _ in range(it, self.num_iter):
            sl.graph.replay()
            logits = sl.logits
        x_adv = (sl.X + sl.delta).clamp_(0.0, 1.0)
        return x_adv.detach(), (logits.clone() if logits is not None else None), None


class Pgd_Attack:
    """Zero-start PGD on the clamped input that keeps, per image, the perturbation obtained from the
    step taken at its highest loss so far (val.py:130-178).  ``epsilon=`` is accepted as an alias of
    ``eps=`` (tools/train_rob_seg.py:295-300 passes it)."""

    def __init__(self, eps=4.0 / 255.0, alpha=1e-2, num_iter=2, los="pgd", epsilon=None):
        self.epsilon = eps if epsilon is None else epsilon
        self.alpha, self.num_iter, self.los_name = alpha, num_iter, los
        self.loss_fn = losses[los]
        self.mode, self.per_image = self.loss_fn.mode, self.loss_fn.per_image
        if not self.per_image:
            # the reference indexes a scalar loss per image here and raises IndexError
            raise ValueError("Pgd_Attack needs a per-image loss ('mask-ce-avg' or 'js-avg'); use Pgd_Attack_1 for 'pgd'")

    def adv_attack(self, model, X, y, wt=None):
        model.eval()
        X = X.detach().contiguous().float()
        B, HW = X.shape[0], X.shape[-2] * X.shape[-1]
        # val.py:152-155 hands `y.long()` to EVERY loss, 'l2-loss' included (float targets are truncated there)
        y = _labels(y) if self.mode is not None else y.long()
        delta = torch.zeros_like(X)
        best_delta = torch.zeros_like(X)
        best = torch.zeros(B, device=X.device)
        ws = N.loss_workspace(B, HW, X.device)
        out = tuple(torch.empty(B, dtype=d, device=X.device) for d in (torch.float32, torch.float32, torch.int32))
        x_in = X.clamp(0.0, 1.0)
        dl = None
        for _ in range(self.num_iter):
            g, r, _ = _fwd_grad(model, x_in, y, self.loss_fn, 1.0 / HW, ws, out, dl)
            dl = r["dlogits"]
            loss = r["loss_img"] if "loss_img" in r else r["loss_sum"] / HW
            ind = loss >= best
            best = torch.where(ind, loss, best)
            N.pgd_linf_step(X, delta, g, float(self.alpha), float(self.epsilon), delta_out=delta, x_in_out=x_in,
                            clamp_input=True)
            best_delta = torch.where(ind.view(-1, 1, 1, 1), delta, best_delta)
        return (X + best_delta).clamp_(0.0, 1.0).detach(), None, None


@torch.no_grad()
def evaluate(model, dataloader, device, cls, n_batches=-1):
    """Confusion-matrix evaluation loop (val.py:14-32); same 7-tuple."""
    model.eval()
    metrics = Metrics(cls, -1, device)
    for i, batch in enumerate(dataloader):
        images, labels = batch[0].to(device), batch[1].to(device)
        metrics.update(model(images), labels)  # argmax(softmax(z)) == argmax(z): the softmax pass is dropped
        if i + 1 == n_batches:
            break
    ious, miou = metrics.compute_iou()
    cla_acc, macc, aacc = metrics.compute_pixel_acc()
    f1, mf1 = metrics.compute_f1()
    return cla_acc, macc, aacc, f1, mf1, ious, miou
