"""UperNet + ConvNeXt: the convolutional model family attacked by SEA / trained by PIR-AT.

Own implementation for PyTorch-ROCm.  The attack hot path (semseg.attacker) only needs ``model(x) ->
(B,C,H,W) logits``; this file exists so that the benchmark runs the architecture BASELINE.json names
and so that published checkpoints load: parameter/buffer names and shapes follow the state-dict
schema of the reference (semseg/models/uperforseg.py:382-404, backbones/convnext_orig.py:88-175;
SURVEY Appendix B) exactly -- ``load_state_dict(strict=True)`` works both ways.

The model stays on MIOpen / CK / hipBLASLt (MFMA GEMMs and convolutions) except for the depthwise
7x7 stencil, which goes through libsea_hip (M1, include/sea_hip.h) because the library runs it ~9x
below what an HBM-bound stencil allows.  Forward semantics mirror the reference: logits at 1/4 resolution, bilinear x4 to the input size
(uperforseg.py:415-418), eval mode returns logits only, train mode returns (loss, logits) with the
0.4-weighted auxiliary CE (uperforseg.py:421-439).
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F
from torch import nn

# variant -> (depths, dims, aux-head input channels, stochastic-depth rate); convnext_orig.py:88-100
CONVNEXT_SETTINGS = {
    "T": ([3, 3, 9, 3], [96, 192, 384, 768], 384, 0.4),
    "T_CVST": ([3, 3, 9, 3], [96, 192, 384, 768], 384, 0.4),
    "T_CVST_ROB": ([3, 3, 9, 3], [96, 192, 384, 768], 384, 0.4),
    "S_CVST": ([3, 3, 27, 3], [96, 192, 384, 768], 384, 0.3),
    "S_CVST_ROB": ([3, 3, 27, 3], [96, 192, 384, 768], 384, 0.3),
    "B": ([3, 3, 27, 3], [128, 256, 512, 1024], 512, 0.4),
}


# The custom autograd Functions below hand raw fp32 pointers to libsea_hip and call torch GEMMs in between.  Under
# autocast (PIR-AT's inner PGD with TRAIN.AMP) they are fp32 islands: floating-point inputs are cast to fp32 and
# autocast is off inside forward AND backward (a bf16 torch.bmm result handed to an fp32 kernel reads past its buffer).
# Same contract as torch.amp.custom_fwd(cast_inputs=torch.float32) / custom_bwd, without their per-call context
# managers when autocast is off (about 200 Function calls per attack step: ~1.5 ms of host time per step, which made
# ConvNeXt-S launch-bound).
def _fp32_fwd(fn):
    def forward(ctx, *args):
        if torch.is_autocast_enabled():
            args = [a.float() if (torch.is_tensor(a) and a.is_floating_point() and a.dtype != torch.float32) else a
                    for a in args]
            with torch.autocast("cuda", enabled=False):
                return fn(ctx, *args)
        return fn(ctx, *args)
    return forward


def _fp32_bwd(fn):
    def backward(ctx, *grads):
        if torch.is_autocast_enabled():
            grads = [g.float() if (torch.is_tensor(g) and g.dtype != torch.float32) else g for g in grads]
            with torch.autocast("cuda", enabled=False):
                return fn(ctx, *grads)
        return fn(ctx, *grads)
    return backward


USE_HIP_LAYERNORM = True


class _LayerNormHip(torch.autograd.Function):
    """LayerNorm over the last dim through libsea_hip M5 (frozen affine parameters: input gradient only)."""

    @staticmethod
    @_fp32_fwd
    def forward(ctx, x, weight, bias, eps):
        from .. import _native as N
        y, mean, rstd = N.layernorm(x, weight, bias, eps)
        ctx.save_for_backward(x, weight, mean, rstd)
        return y

    @staticmethod
    @_fp32_bwd
    def backward(ctx, g):
        from .. import _native as N
        x, weight, mean, rstd = ctx.saved_tensors
        return N.layernorm_backward(g.contiguous(), x, weight, mean, rstd), None, None, None


def _layer_norm(x, dim, weight, bias, eps):
    if (USE_HIP_LAYERNORM and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and dim % 4 == 0
            and dim <= 1024 and x.shape[-1] == dim and not weight.requires_grad and not bias.requires_grad):
        # also under autocast: LayerNorm is an fp32 op there too, and the kernel is called with raw fp32 pointers
        return _LayerNormHip.apply(x, weight, bias, eps)
    return F.layer_norm(x, (dim,), weight, bias, eps)


class LayerNorm(nn.Module):
    """LayerNorm over the channel dim for NHWC ("channels_last") or NCHW ("channels_first") tensors."""

    def __init__(self, dim: int, eps: float = 1e-6, data_format: str = "channels_last"):
        super().__init__()
        if data_format not in ("channels_last", "channels_first"):
            raise NotImplementedError(data_format)
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))
        self.eps, self.data_format, self.dim = eps, data_format, dim

    def forward(self, x):
        if self.data_format == "channels_last":
            return _layer_norm(x, self.dim, self.weight, self.bias, self.eps)
        xt = _ToNHWC.apply(x) if _fast_layout_ok(x) else x.permute(0, 2, 3, 1)
        y = _layer_norm(xt, self.dim, self.weight, self.bias, self.eps)
        y = y.permute(0, 3, 1, 2)
        return y.contiguous() if LN_CONTIGUOUS else y


class StochasticDepth(nn.Module):
    def __init__(self, p: float):
        super().__init__()
        self.p = float(p)

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        keep = 1.0 - self.p
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x * mask / keep


class _DwConv7x7(torch.autograd.Function):
    """ConvNeXt's depthwise 7x7 through the hand-written stencil kernel (libsea_hip M1): forward and
    input gradient.  The weight/bias gradients (only needed when training) use PyTorch."""

    @staticmethod
    @_fp32_fwd
    def forward(ctx, x, weight, bias):
        from .. import _native as N
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return N.dwconv7x7(x.contiguous(), weight.contiguous(), bias, flip=False)

    @staticmethod
    @_fp32_bwd
    def backward(ctx, gy):
        from .. import _native as N
        x, weight = ctx.saved_tensors
        gx = gw = gb = None
        gy = gy.contiguous()
        if ctx.needs_input_grad[0]:
            gx = N.dwconv7x7(gy, weight.contiguous(), None, flip=True)
        if ctx.needs_input_grad[1]:
            gw = torch.nn.grad.conv2d_weight(x, weight.shape, gy, padding=3, groups=x.shape[1])
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum((0, 2, 3))
        return gx, gw, gb


USE_HIP_DW_WGRAD = os.environ.get("SEA_HIP_DW_WGRAD", "1") != "0"


def _dw_weight_grad(x, weight, gy, want_bias):
    """weight (and bias) gradient of the NHWC depthwise 7x7: libsea_hip M1w, or the library (SEA_HIP_DW_WGRAD=0)"""
    from .. import _native as N
    if USE_HIP_DW_WGRAD and x.is_cuda and x.dtype == torch.float32 and gy.dtype == torch.float32 and x.shape[-1] % 4 == 0:
        return N.dwconv7x7_nhwc_weight_grad(x.contiguous(), gy, want_bias)
    gw = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2), weight.shape, gy.permute(0, 3, 1, 2), padding=3, groups=x.shape[-1])
    return gw, (gy.sum((0, 1, 2)) if want_bias else None)


class _DwConv7x7NHWC(torch.autograd.Function):
    """Same layer on a (B,H,W,C) contiguous tensor (channels_last trunk): nothing in the block changes
    layout any more."""

    @staticmethod
    @_fp32_fwd
    def forward(ctx, x, weight, bias, wt):
        from .. import _native as N
        ctx.save_for_backward(x, weight, wt)
        ctx.has_bias = bias is not None
        return N.dwconv7x7_nhwc(x, wt, bias, flip=False)

    @staticmethod
    @_fp32_bwd
    def backward(ctx, gy):
        from .. import _native as N
        x, weight, wt = ctx.saved_tensors
        gx = gw = gb = None
        gy = gy.contiguous()
        if ctx.needs_input_grad[0]:
            gx = N.dwconv7x7_nhwc(gy, wt, None, flip=True)
        if ctx.needs_input_grad[1]:
            gw, gb = _dw_weight_grad(x, weight, gy, ctx.has_bias and ctx.needs_input_grad[2])
        elif ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum((0, 1, 2))
        return gx, gw, gb, None


class _DwConv7x7NHWCSkip(torch.autograd.Function):
    """The depthwise layer of a residual block together with the block's skip connection: returns ``(skip, y)`` with
    ``skip`` an alias of the input.  The block computes ``skip + branch(y)``; in the backward the skip gradient
    arrives here next to the branch gradient and is added inside the backward-data kernel (``y = conv(g) + g_skip``,
    added after the taps: bitwise what autograd's separate accumulation kernel would give) instead of in an
    element-wise pass of its own (reference block: convnext_orig.py:75-86)."""

    @staticmethod
    @_fp32_fwd
    def forward(ctx, x, weight, bias, wt):
        from .. import _native as N
        ctx.save_for_backward(x, weight, wt)
        ctx.has_bias = bias is not None
        return x.view_as(x), N.dwconv7x7_nhwc(x, wt, bias, flip=False)

    @staticmethod
    @_fp32_bwd
    def backward(ctx, g_skip, gy):
        from .. import _native as N
        x, weight, wt = ctx.saved_tensors
        gx = gw = gb = None
        if gy is None:
            return g_skip, None, None, None
        gy = gy.contiguous()
        if ctx.needs_input_grad[0]:
            add = None
            if g_skip is not None:
                add = g_skip if (g_skip.is_contiguous() and g_skip.dtype == torch.float32) else g_skip.float().contiguous()
            gx = N.dwconv7x7_nhwc(gy, wt, None, flip=True, addend=add)
        if ctx.needs_input_grad[1]:
            gw, gb = _dw_weight_grad(x, weight, gy, ctx.has_bias and ctx.needs_input_grad[2])
        elif ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum((0, 1, 2))
        return gx, gw, gb, None


def _tkey(*ts):
    """Cache key of weights derived from the tensors ``ts``: object identity, storage address, in-place version and
    device of every source.  The identity term covers a parameter that was REPLACED by a fresh tensor which happens
    to reuse a freed address at version 0 (``module.weight = nn.Parameter(...)``, a re-materialised model), and the
    caches copied along by ``copy.deepcopy`` (new parameter objects -> new key); ``.to()`` changes address / device."""
    return tuple((id(t), t.data_ptr(), t._version, t.device.index) for t in ts if t is not None)


def _stable(old, new):
    """Store a re-derived cache value at the OLD value's device address when shapes allow: a tensor is copied into the
    old tensor (same address, in-place version bumped, so caches derived from IT notice), a packed weight into the old packed
    buffers, lists / tuples element-wise.  A captured HIP graph over the model (semseg/val.py: the inner PGD of PIR-AT, whose
    weights change every outer step) then stays valid after a weight update as long as one eager pass has refreshed the
    caches; `_native.CACHE_EPOCH` counts the values that could NOT keep their address."""
    from .. import _native as N
    if isinstance(new, N.PackedWeight):
        if isinstance(old, N.PackedWeight) and old.refresh_from(new):
            return old
        N.CACHE_EPOCH[0] += 1
        return new
    if isinstance(new, torch.Tensor):
        if (isinstance(old, torch.Tensor) and old is not new and old.shape == new.shape and old.dtype == new.dtype
                and old.device == new.device and old.stride() == new.stride()):
            old.copy_(new)
            return old
        N.CACHE_EPOCH[0] += 1
        return new
    if isinstance(new, (list, tuple)):
        olds = old if isinstance(old, (list, tuple)) and len(old) == len(new) else [None] * len(new)
        return type(new)(_stable(o, n) for o, n in zip(olds, new))
    return new


# M8 (csrc/gemm_split.hip): the frozen-weight GEMMs of the attacked model on the bf16 matrix cores by operand splitting.
#   3: every fp32 operand as three bf16 terms, six MFMA products -> fp32-level accuracy
#   2: two terms, three products (16 significant bits per operand; 64x finer than the TF32 convolutions the reference's
#      own GPU runs use by default)
#  22: fp16 x 2 (hi + mid = 22 significant bits, three products; operands scaled per tensor / per weight row by powers
#      of two because fp16 has 5 exponent bits; the activation's scale comes from a device-side max|A| pass): measured
#      error vs float64 <= the fp32 GEMM's own, at half the matrix-core work of 3 -> the default of the FORWARD products
#   0: hipBLASLt fp32 (torch.mm / bmm / addmm), as in rounds 1-2
GEMM_TERMS = int(os.environ.get("SEA_GEMM_TERMS", "22"))
# Under bf16 autocast (PIR-AT's inner PGD with TRAIN.AMP, BASELINE configs[3]) the decode head's fp32 islands run M8 with
# TWO terms: 16 significant bits per operand (twice bf16's) at three MFMA products instead of six.
GEMM_TERMS_AUTOCAST = int(os.environ.get("SEA_GEMM_TERMS_AUTOCAST", "2"))
# Round 4: the WHOLE attack forward / backward of a frozen model under bf16 autocast as one fp32-I/O island on M8 with ONE
# bf16 term per operand (SEA_AUTOCAST_ISLAND=model): exactly the operand precision autocast asks for (bf16 operands, fp32
# accumulate), but with fp32 activations between the layers (no autocast cast kernels, no 16-bit round trips), the fused
# prologues / epilogues of the fp32 attack path and its host cost.  "head" = round 3's behaviour (only the decode head is an
# island, with two terms; the trunk's nn.Linear run through autocast's bf16 library GEMMs).
AUTOCAST_ISLAND = os.environ.get("SEA_AUTOCAST_ISLAND", "model")
GEMM_TERMS_AUTOCAST_MODEL = int(os.environ.get("SEA_GEMM_TERMS_AUTOCAST_MODEL", "1"))
# Terms of the INPUT-GRADIENT products.  Default 22 (round 4): fp16 x 2 like the forward -- 22 significant bits per operand,
# error <= the fp32 GEMM's own -- so that the whole evaluation is fp32-equivalent, as the reference's is.  A gradient
# operand's rows (pixels) span many orders of magnitude, so its power-of-two scales are PER ROW: exact row maxima
# (sea_absmax_bits, one word per row), the Winograd transform's per-tile words, or a row bound carried through the GEMM in
# between (rowmax(g) ||W||_1, _FrozenMlp).  2 = two bf16 terms (16 significant bits, no scales; rounds 3's default: the
# attack only uses sign(gradient), attacker.py:396), 3 = three bf16 terms (exact split, six products).
GEMM_TERMS_BWD = int(os.environ.get("SEA_GEMM_TERMS_BWD", "22"))
GEMM_MIN_ROWS = 1024   # below, a 128-row tile grid cannot fill the chip: hipBLASLt's split-K kernels win
_TERMS_OVERRIDE = [None]


def _terms():
    """number of bf16 terms M8 uses right now (a Function records it in forward and reuses it in backward)"""
    return GEMM_TERMS if _TERMS_OVERRIDE[0] is None else _TERMS_OVERRIDE[0]


def _bwd_terms(fwd_terms):
    """terms of the input-gradient products of a Function whose forward ran with ``fwd_terms``"""
    if fwd_terms == 22:                      # fp16 x 2 forward: fp16 x 2 with per-row scales, or the scale-free bf16 terms
        return GEMM_TERMS_BWD if GEMM_TERMS_BWD in (2, 3, 22) else 3
    return min(fwd_terms, GEMM_TERMS_BWD) if fwd_terms in (2, 3) and GEMM_TERMS_BWD in (2, 3) else fwd_terms


class _gemm_terms:
    def __init__(self, terms):
        self.terms = terms

    def __enter__(self):
        self.old, _TERMS_OVERRIDE[0] = _TERMS_OVERRIDE[0], self.terms

    def __exit__(self, *exc):
        _TERMS_OVERRIDE[0] = self.old
        return False


def _split_ok(x2d, K, terms=None):
    return ((_terms() if terms is None else terms) in (1, 2, 3, 22) and x2d.is_cuda and x2d.dtype == torch.float32 and x2d.dim() == 2 and K % 32 == 0
            and x2d.stride(1) == 1 and x2d.stride(0) % 4 == 0 and x2d.data_ptr() % 16 == 0
            and x2d.shape[0] >= GEMM_MIN_ROWS and not torch.is_autocast_enabled())


def _packed(w, cache, name, trans, terms):
    """the packed (pre-split) image of the frozen weight w, cached in ``cache`` under ``name`` (per number of terms) until
    w changes"""
    from .. import _native as N
    name = f"{name}_t{terms}"
    key = (_tkey(w), trans)
    if cache.get(name + "_key") != key:
        with torch.no_grad():
            cache[name] = _stable(cache.get(name), N.gemm_split_pack(w.detach(), trans=trans, terms=terms))
        cache[name + "_key"] = key
    return cache[name]


def _frozen_mm(x2d, w, cache, name, trans=False, bias=None, relu=False, terms=None, amax=None, out_amax=None, groups=1,
               row_amax=False):
    """act(x2d @ W^T + bias) for a FROZEN weight: w is (N, K), or (K, N) with ``trans``.  ``groups`` = number of images
    the rows of x2d belong to (image-major): without a supplied ``amax`` the fp16 x 2 mode scales every image by its own
    maximum, so that an image's result does not depend on its batch partners; ``row_amax``: every ROW by its own (the
    input-gradient products: a gradient's rows span many orders of magnitude)."""
    K = w.shape[0] if trans else w.shape[1]
    terms = _terms() if terms is None else terms
    if _split_ok(x2d, K, terms):
        from .. import _native as N
        if terms != 22:
            amax = out_amax = None
        return N.gemm_split(x2d, _packed(w, cache, name, trans, terms), bias=bias, relu=relu, amax=amax, out_amax=out_amax,
                            groups=groups, row_amax=row_amax)
    wt = w if trans else w.t()
    y = torch.addmm(bias, x2d, wt) if bias is not None else x2d @ wt
    return torch.relu_(y) if relu else y


def _ln_bound_word(norm, cache):
    """float bits (int32 device word) of sqrt(C) max|w| + max|b|: an upper bound of |LayerNorm(x)| for ANY x
    (|(x - mean) * rstd| <= sqrt(C - 1)), used as the fp16 x 2 activation scale of the GEMM that consumes the
    LayerNorm's output: no pass over the activations.  Cached until the affine parameters change."""
    key = _tkey(norm.weight, norm.bias)
    if cache.get("ln_bound_key") != key:
        with torch.no_grad():
            bnd = norm.weight.numel() ** 0.5 * norm.weight.abs().max() + norm.bias.abs().max()
            cache.update(ln_bound_key=key, ln_bound=_stable(cache.get("ln_bound"), bnd.float().reshape(1).contiguous().view(torch.int32)))
    return cache["ln_bound"]


def _linear_bound_word(in_word, w, b, cache):
    """float bits of max_n (bound(x) ||W_n||_1 + |b_n|): an upper bound of |x W^T + b| for any x with |x| <= bound(x)
    (``in_word``: float bits of that bound), hence of GELU / ReLU of it and of any convex combination of its rows (attention).
    The fp16 x 2 scale of the NEXT GEMM without a pass over the activations and without any dependence on the batch.
    Cached until the weights change."""
    key = (_tkey(w, b), in_word._version)      # (the input word may be refreshed in place: its version is part of the key)
    # (the input word is compared by identity and kept referenced: a re-derived word can then never reuse its address)
    if cache.get("lin_bound_key") != key or cache.get("lin_bound_in") is not in_word:
        with torch.no_grad():
            bnd = in_word.view(torch.float32) * w.abs().sum(1)
            if b is not None:
                bnd = bnd + b.abs()
            cache.update(lin_bound_key=key, lin_bound_in=in_word,
                         lin_bound=_stable(cache.get("lin_bound"), bnd.max().float().reshape(1).contiguous().view(torch.int32)))
    return cache["lin_bound"]


def _l1_bound(w, cache, dim, factor=1.0):
    """ONE float32 in device memory: factor * max over the other dim of sum_dim |w|, rounded up (cached until the weight
    changes; derived on the device, so a weight that changes every step -- PIR-AT -- costs no host round trip).  For w of
    shape (N, K): dim = 0 bounds the input-gradient product u = g w row by row,
    |u[r][k]| = |sum_n g[r][n] w[n][k]| <= rowmax(g[r]) * max_k sum_n |w[n][k]|; dim = 1 bounds the forward product x w^T."""
    key = (_tkey(w), dim, factor)
    if cache.get("l1_key") != key:
        with torch.no_grad():
            cache.update(l1_key=key, l1=_stable(cache.get("l1"), (w.detach().abs().sum(dim).max() * (factor * (1.0 + 1e-6))).float().reshape(1)))
    return cache["l1"]


class _FrozenLinear(torch.autograd.Function):
    """F.linear(x, w, b) for frozen (w, b), input gradient only: both directions through ``_frozen_mm``."""

    @staticmethod
    @_fp32_fwd
    def forward(ctx, x, w, b, cache, amax=None, out_amax=None):
        ctx.w, ctx.cache, ctx.shape, ctx.terms = w, cache, x.shape, _terms()
        x2 = x.reshape(-1, x.shape[-1])
        return _frozen_mm(x2, w, cache, "lin_fwd", bias=b, amax=amax, out_amax=out_amax,
                          groups=x.shape[0] if x.dim() > 2 else 1).view(*x.shape[:-1], w.shape[0])

    @staticmethod
    @_fp32_bwd
    def backward(ctx, gy):
        g2 = gy.reshape(-1, gy.shape[-1])
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        return (_frozen_mm(g2, ctx.w, ctx.cache, "lin_bwd", trans=True, terms=_bwd_terms(ctx.terms), row_amax=True,
                           groups=ctx.shape[0] if len(ctx.shape) > 2 else 1).view(ctx.shape), None, None, None, None, None)


def _linear_frozen(mod_cache, x, w, b, amax=None, out_amax=None):
    """F.linear through M8 when the weights are frozen and the shape qualifies; plain F.linear otherwise.
    ``amax`` / ``out_amax``: device words for the fp16 x 2 mode (see _native.gemm_split): an upper bound of max|x| supplied
    by the producer instead of a pass over x, and a pre-zeroed word that receives max|output|."""
    if (_terms() in (1, 2, 3, 22) and x.is_cuda and x.dtype == torch.float32 and not w.requires_grad
            and (b is None or not b.requires_grad) and w.shape[1] % 32 == 0 and not torch.is_autocast_enabled()
            and x.is_contiguous() and x.numel() // x.shape[-1] >= GEMM_MIN_ROWS):
        return _FrozenLinear.apply(x, w, b, mod_cache, amax, out_amax)
    return F.linear(x, w, b)


# Element-wise neighbours of a block's MLP inside its GEMMs (sea_gemm_split_fused).  Bit mask (env SEA_FUSE_MLP; -1 = the
# two projections as separate autograd nodes with ATen's GELU between them):
#   8 = GELU' applied to the A tile of the first projection's input-gradient GEMM while it is staged (prologue): removes
#       the GELU-backward pass and one write + read of the 4C-wide gradient.  Used where that GEMM has at most
#       FUSE_PROLOGUE_MAX_NBLOCKS column blocks (every column block re-stages A and would re-evaluate erf).  DEFAULT.
#   1 = GELU in the first forward GEMM's epilogue, 2 = GELU' in the second backward GEMM's epilogue, 4 = residual add in
#       the second forward GEMM's epilogue: MEASURED SLOWER on MI355X and off (UperNet-ConvNeXt-T, B=8, 512x512, ms per
#       APGD step: none 18.16, 1: 18.24, 4: 18.59, 2: 18.97, 7: 19.32): the separate ATen passes stream from the 256 MiB
#       Infinity Cache at full occupancy, while the same loads / erf evaluations in the epilogue of a short-K GEMM run at 3
#       waves per SIMD behind the accumulators (64 scalar accesses per lane).  Kept as tested options of the C ABI.
#  16 = GELU applied to the A tile of the second projection's forward GEMM while it is staged: GELU(t) is never written.
FUSE_MLP = int(os.environ.get("SEA_FUSE_MLP", "24"))
RELU_GATE_PROLOGUE = os.environ.get("SEA_RELU_GATE_PROLOGUE", "1") != "0"   # A/B: 1x1 ConvModule backward (see _PointwiseRelu)
REDUCE_ADD = os.environ.get("SEA_REDUCE_ADD", "1") != "0"   # A/B: the residual in the split-K reduce pass
FUSE_PROLOGUE_MAX_NBLOCKS = int(os.environ.get("SEA_FUSE_PRO_NB", "3"))


def _mlp_fusable(x, w1, b1, w2, b2):
    """both projections of a block's MLP qualify for M8 (frozen weights, fp32 HIP tensor, shapes): the pair runs as two
    GEMMs with GELU, GELU' and the residual add in their epilogues (_FrozenMlp)"""
    return (FUSE_MLP >= 0 and _terms() in (1, 2, 3, 22) and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
            and not torch.is_autocast_enabled() and x.numel() // x.shape[-1] >= GEMM_MIN_ROWS
            and w1.shape[1] % 32 == 0 and w2.shape[1] % 32 == 0 and w2.shape[0] == x.shape[-1]
            and not any(t is not None and t.requires_grad for t in (w1, b1, w2, b2)))


class _FrozenMlp(torch.autograd.Function):
    """res + W2 GELU(W1 x + b1) + b2 for frozen (W1, b1, W2, b2) (the MLP of a ConvNeXt / ViT block, reference
    convnext_orig.py:38-58, vit_encoder.py:41-60), input gradient only.  Two M8 GEMMs each way.  By default (FUSE_MLP
    bits 8 + 16) the element-wise neighbours are PROLOGUES of the consuming GEMMs: the second forward GEMM reads the
    pre-activation t and applies GELU to its A tile while staging it, the first projection's input-gradient GEMM reads the
    incoming gradient and t and multiplies by GELU'(t) the same way; the residual rides in the split-K reduce pass where the
    second forward GEMM is split.  Only t is kept for the backward; GELU(t) and g * GELU'(t) are never written.  The
    epilogue variants (bits 1, 2, 4) were measured slower and are off."""

    @staticmethod
    @_fp32_fwd
    def forward(ctx, x, w1, b1, w2, b2, res, caches, a1, a2, ln=None):
        from .. import _native as N
        terms = _terms()
        x2 = x.reshape(-1, x.shape[-1])
        ctx.ln = None
        if terms != 22:
            a1 = a2 = None
        fuse, nb = int(FUSE_MLP), (x.shape[0] if x.dim() > 2 else 1)
        r2 = None if res is None else res.reshape(-1, res.shape[-1])
        ctx.w, ctx.b1, ctx.caches, ctx.shape, ctx.terms, ctx.has_res = (w1, w2), b1, caches, x.shape, terms, res is not None
        ctx.fuse, ctx.a1 = fuse, a1
        # M8f (csrc/mlp_fused.hip): both projections, GELU and the residual in ONE kernel whose hidden tile never leaves the CU
        # -- where the two-GEMM form is bound by the 4C-wide tensor's HBM round trips (C = 96 / 192).  Same bits as the pair
        # below; only x2 is kept for the backward, which recomputes t in its own fused kernel.
        rows_ok = lambda t: t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0   # noqa: E731
        if (terms == 22 and fuse & 16 and a1 is not None and a2 is not None and a1.numel() == 1 and a2.numel() == 1
                and b1 is not None and N.mlp_fused_ok(w2.shape[0], w1.shape[0]) and rows_ok(x2)
                and (r2 is None or rows_ok(r2))):
            # ``ln`` = (weight, bias, eps) of the block's LayerNorm: x is then its INPUT and the normalisation (and, backward, its
            # input gradient) runs inside the kernels too, with the LayerNorm kernel's own arithmetic (see _mlp_takes_layernorm)
            y = N.mlp_fused_forward(x2, _packed(w1, caches[0], "lin_fwd", False, terms), b1,
                                    _packed(w2, caches[1], "lin_fwd", False, terms), b2, r2, a1, a2, ln=ln)
            ctx.kernel_fused, ctx.ln = True, ln
            ctx.save_for_backward(x2)
            return y.view(x.shape)
        if ln is not None:
            raise RuntimeError("_FrozenMlp: a LayerNorm was handed in but the fused kernel does not apply (see _mlp_takes_layernorm)")
        ctx.kernel_fused = False
        t = torch.empty(x2.shape[0], w1.shape[0], dtype=torch.float32, device=x.device)
        pro = bool(fuse & 16) and -(-w2.shape[0] // 128) <= FUSE_PROLOGUE_MAX_NBLOCKS and not fuse & 1
        if pro:
            N.gemm_split(x2, _packed(w1, caches[0], "lin_fwd", False, terms), bias=b1, out=t, amax=a1, groups=nb)
            h = t
        elif fuse & 1:
            h = torch.empty_like(t)
            N.gemm_split(x2, _packed(w1, caches[0], "lin_fwd", False, terms), bias=b1, out=t, gelu_out=h, amax=a1, groups=nb)
        else:
            N.gemm_split(x2, _packed(w1, caches[0], "lin_fwd", False, terms), bias=b1, out=t, amax=a1, groups=nb)
            h = F.gelu(t)
        # the residual rides in the split-K reduce pass where the second GEMM is split (deep stages); in a GEMM epilogue
        # it was measured slower than the separate add (bit 4)
        split2 = REDUCE_ADD and N._ksplit(x2.shape[0], w2.shape[0], w2.shape[1]) > 1
        add_in = r2 is not None and r2.is_contiguous() and (split2 or (bool(fuse & 4) and not pro))
        y = N.gemm_split(t if pro else h, _packed(w2, caches[1], "lin_fwd", False, terms), bias=b2, amax=a2, groups=nb,
                         a_gelu=pro, addend=r2 if add_in else None)
        if r2 is not None and not add_in:
            y += r2
        ctx.save_for_backward(t)
        return y.view(x.shape)

    @staticmethod
    @_fp32_bwd
    def backward(ctx, g):
        from .. import _native as N
        (t,) = ctx.saved_tensors
        terms = _bwd_terms(ctx.terms)
        g2 = g.reshape(-1, g.shape[-1])
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        nb = ctx.shape[0] if len(ctx.shape) > 2 else 1
        if ctx.kernel_fused:
            x2 = t                       # (the fused forward saved its INPUT: t is recomputed)
            p1f = _packed(ctx.w[0], ctx.caches[0], "lin_fwd", False, ctx.terms)
            if terms == 22 and g2.data_ptr() % 16 == 0:
                gx = N.mlp_fused_backward(g2, x2, p1f, ctx.b1, _packed(ctx.w[1], ctx.caches[1], "lin_bwd", True, terms),
                                          _packed(ctx.w[0], ctx.caches[0], "lin_bwd", True, terms), ctx.a1,
                                          _l1_bound(ctx.w[1], ctx.caches[1], dim=0, factor=1.13), ln=ctx.ln)
                return gx.view(ctx.shape), None, None, None, None, (g if ctx.has_res else None), None, None, None, None
            if ctx.ln is not None:                                               # (the LayerNorm lived in the forward kernel: redo it here)
                xin = x2
                x2, ln_mean, ln_rstd = N.layernorm(xin, ctx.ln[0], ctx.ln[1], ctx.ln[2])
            t = N.gemm_split(x2, p1f, bias=ctx.b1, amax=ctx.a1, groups=nb)      # another backward arithmetic: the pair below
        p2, p1 = (_packed(ctx.w[1], ctx.caches[1], "lin_bwd", True, terms), _packed(ctx.w[0], ctx.caches[0], "lin_bwd", True, terms))
        # fp16 x 2: ONE pass for the per-row maxima of g; the second product's operand u = g W2 (times GELU' <= 1.13) is
        # bounded row by row through the first: |u[r][n]| <= rowmax(g[r]) * max_n ||W2[n]||_1  (no pass over the 4C-wide u)
        rows = {}
        if terms == 22:
            words, _ = N._amax_words(g2.unsqueeze(0), g2.shape[0], g2.shape[1], 1, 0, nb, per_row=True)
            rows = dict(amax=words, amax_rows=1)
            mul = dict(amax_mul=_l1_bound(ctx.w[1], ctx.caches[1], dim=0, factor=1.13), **rows)
        else:
            mul = {}
        if ctx.fuse & 2:
            gx = N.gemm_split(N.gemm_split(g2, p2, gelu_grad_of=t, groups=nb, **rows), p1, groups=nb, **mul)
        elif ctx.fuse & 8 and terms in (1, 2, 22) and -(-ctx.w[0].shape[1] // 128) <= FUSE_PROLOGUE_MAX_NBLOCKS:
            gx = N.gemm_split(N.gemm_split(g2, p2, groups=nb, **rows), p1, a_gelu_grad_of=t, groups=nb, **mul)
        else:
            gx = N.gemm_split(torch.ops.aten.gelu_backward(N.gemm_split(g2, p2, groups=nb, **rows), t), p1, groups=nb, **mul)
        if ctx.kernel_fused and ctx.ln is not None:
            gx = N.layernorm_backward(gx.contiguous(), xin, ctx.ln[0], ln_mean, ln_rstd)
        return gx.view(ctx.shape), None, None, None, None, (g if ctx.has_res else None), None, None, None, None


def _mlp_takes_layernorm(y, norm, w1, b1, w2, a1, a2):
    """the block's LayerNorm can ride inside the fused MLP kernels (M8f): the conditions under which _FrozenMlp.forward takes
    sea_mlp_fused_fwd, plus frozen, aligned LayerNorm parameters"""
    from .. import _native as N
    return (FUSE_LN_INTO_MLP and _terms() == 22 and int(FUSE_MLP) & 16 and a1 is not None and a2 is not None and a1.numel() == 1
            and a2.numel() == 1 and b1 is not None and N.mlp_fused_ok(w2.shape[0], w1.shape[0]) and y.is_contiguous()
            and y.data_ptr() % 16 == 0 and norm.weight is not None and norm.bias is not None
            and not (norm.weight.requires_grad or norm.bias.requires_grad) and norm.weight.is_contiguous()
            and norm.bias.is_contiguous() and norm.weight.data_ptr() % 16 == 0 and norm.bias.data_ptr() % 16 == 0
            and USE_HIP_LAYERNORM and _bwd_terms(22) == 22)


FUSE_LN_INTO_MLP = os.environ.get("SEA_MLP_FUSE_LN", "1") != "0"


def _taps_major(conv: nn.Conv2d):
    """(49, C) copy of the depthwise filter bank, cached until the weight changes."""
    w = conv.weight
    key = _tkey(w)
    cached = getattr(conv, "_sea_wt", None)
    if cached is None or cached[0] != key:
        cached = (key, _stable(None if cached is None else cached[1], w.detach().reshape(w.shape[0], 49).t().contiguous()))
        conv._sea_wt = cached
    return cached[1]


def depthwise7x7(conv: nn.Conv2d, x):
    """Route fp32 HIP tensors through the stencil kernel, everything else through nn.Conv2d."""
    if x.is_cuda and x.dtype == torch.float32 and conv.weight.dtype == torch.float32 and USE_HIP_DWCONV:
        return _DwConv7x7.apply(x, conv.weight, conv.bias)
    return conv(x)


USE_HIP_DWCONV = True
# Materialising NCHW after the channels-first LayerNorms was measured SLOWER (+8 ms per APGD step at B=8,
# 512x512: MIOpen then picks slower NCHW algorithms for the stride-2 convolutions), so the permuted view stays.
LN_CONTIGUOUS = os.environ.get("SEA_LN_CONTIGUOUS", "0") != "0"


class _ToNHWC(torch.autograd.Function):
    """(B,C,H,W) -> contiguous (B,H,W,C) through the LDS-tiled transpose (libsea_hip M3); backward is the
    opposite transpose."""

    @staticmethod
    @_fp32_fwd
    def forward(ctx, x):
        from .. import _native as N
        return N.nchw_to_nhwc(x)

    @staticmethod
    @_fp32_bwd
    def backward(ctx, g):
        from .. import _native as N
        return N.nhwc_to_nchw(g.contiguous())


class _ScaleResidual(torch.autograd.Function):
    """out(NCHW) = x(NCHW) + gamma[c] * y(NHWC)^T: layer scale, permute and residual add of the ConvNeXt
    block in one pass; backward sends gamma[c] * g^T to the branch and g unchanged to the trunk."""

    @staticmethod
    @_fp32_fwd
    def forward(ctx, x, y, gamma):
        from .. import _native as N
        ctx.save_for_backward(y if (gamma is not None and gamma.requires_grad) else None, gamma)
        return N.nhwc_to_nchw(y, gamma, x)

    @staticmethod
    @_fp32_bwd
    def backward(ctx, g):
        from .. import _native as N
        y, gamma = ctx.saved_tensors
        g = g.contiguous()
        gx = g if ctx.needs_input_grad[0] else None
        gy = N.nchw_to_nhwc(g, gamma) if ctx.needs_input_grad[1] else None
        gg = None
        if gamma is not None and ctx.needs_input_grad[2]:
            gg = (N.nchw_to_nhwc(g) * y).sum((0, 1, 2))
        return gx, gy, gg


USE_HIP_TRANSPOSE = True
STAGE_ENTRY_CONTIGUOUS = False  # the trunk arrives channels_last from MIOpen; Block has an all-NHWC path for that


def _fast_layout_ok(x):
    return (USE_HIP_TRANSPOSE and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
            and x.shape[1] % 4 == 0 and (x.shape[2] * x.shape[3]) % 4 == 0)


# The stem through libsea_hip M9 (env SEA_FUSED_STEM=0: the library path): conv1 + LayerNorm + GELU in one kernel, a
# LayerNorm + GELU kernel (statistics recomputed in the backward), and a direct input-gradient kernel for the 3-channel
# convolution; NHWC from the first convolution on (the 48 -> 96 one stays a library call, fastest in that layout, and the
# trunk's blocks run NHWC).  For frozen parameters only (no weight gradients are computed).
USE_FUSED_STEM = os.environ.get("SEA_FUSED_STEM", "1") != "0"


class _StemConv1LnGelu(torch.autograd.Function):
    """GELU(LN_c(conv2d(x, w, b, stride 2, padding 1))) for the 3 -> 48 convolution of the stem (convnext_orig.py:22-24);
    x NCHW, result in channels_last memory"""

    @staticmethod
    @_fp32_fwd
    def forward(ctx, x, weight, bias, gamma, beta, eps):
        from .. import _native as N
        y, a = N.stem_conv1_ln_gelu(x.contiguous(), weight, bias, gamma, beta, eps)
        ctx.save_for_backward(y, weight, gamma, beta)
        ctx.eps, ctx.hw = eps, (x.shape[2], x.shape[3])
        return a

    @staticmethod
    @_fp32_bwd
    def backward(ctx, da):
        from .. import _native as N
        y, weight, gamma, beta = ctx.saved_tensors
        if not (da.is_contiguous() or da.is_contiguous(memory_format=torch.channels_last)):
            da = da.contiguous(memory_format=torch.channels_last)
        dy = N.ln_gelu_cl_backward(da, y, gamma, beta, ctx.eps)
        return N.stem_conv1_backward(dy, weight, *ctx.hw), None, None, None, None, None


class _LnGeluCL(torch.autograd.Function):
    """GELU(LayerNorm over the channels) of a tensor in channels_last memory, frozen affine parameters
    (convnext_orig.py:31-32); the result stays channels_last (the trunk's blocks run NHWC: Block.forward)"""

    @staticmethod
    @_fp32_fwd
    def forward(ctx, y, gamma, beta, eps):
        from .. import _native as N
        y = y.contiguous(memory_format=torch.channels_last)
        ctx.save_for_backward(y, gamma, beta)
        ctx.eps = eps
        return N.ln_gelu_cl(y, gamma, beta, eps)

    @staticmethod
    @_fp32_bwd
    def backward(ctx, da):
        from .. import _native as N
        y, gamma, beta = ctx.saved_tensors
        if not (da.is_contiguous() or da.is_contiguous(memory_format=torch.channels_last)):
            da = da.contiguous()
        return N.ln_gelu_cl_backward(da, y, gamma, beta, ctx.eps), None, None, None


class ConvStem(nn.Module):
    """"CVST" stem: two stride-2 3x3 convs (3->48->96) each followed by channel LN + GELU."""

    def __init__(self, width: int = 48):
        super().__init__()
        self.stem = nn.Sequential(
            nn.Conv2d(3, width, 3, stride=2, padding=1), LayerNorm(width, data_format="channels_first"), nn.GELU(),
            nn.Conv2d(width, 2 * width, 3, stride=2, padding=1), LayerNorm(2 * width, data_format="channels_first"),
            nn.GELU())

    def _fused_ok(self, x):
        from .. import _native as N
        c1, n1, c2, n2 = self.stem[0], self.stem[1], self.stem[3], self.stem[4]
        return (USE_FUSED_STEM and x.is_cuda and x.dim() == 4 and x.shape[1] == 3 and x.dtype == torch.float32
                and c1.out_channels in N.STEM_CONV1_CHANNELS and c2.out_channels in N.LN_GELU_CL_CHANNELS
                and not any(p.requires_grad for p in self.parameters())
                and isinstance(self.stem[2], nn.GELU) and self.stem[2].approximate == "none"
                and isinstance(self.stem[5], nn.GELU) and self.stem[5].approximate == "none"
                and n1.data_format == "channels_first" and n2.data_format == "channels_first")

    def forward(self, x):
        if not self._fused_ok(x):
            return self.stem(x)
        c1, n1, c2, n2 = self.stem[0], self.stem[1], self.stem[3], self.stem[4]
        a1 = _StemConv1LnGelu.apply(x, c1.weight, c1.bias, n1.weight, n1.bias, n1.eps)
        return _LnGeluCL.apply(c2(a1), n2.weight, n2.bias, n2.eps)


class Block(nn.Module):
    """7x7 depthwise conv -> LN -> Linear(4x) -> GELU -> Linear -> layer scale -> residual."""

    def __init__(self, dim: int, drop_path: float = 0.0, layer_scale_init_value: float = 1.0):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 7, padding=3, groups=dim)
        self.norm = LayerNorm(dim)
        self.pwconv1 = nn.Linear(dim, 4 * dim)
        self.act = nn.GELU()
        self.pwconv2 = nn.Linear(4 * dim, dim)
        self.gamma = nn.Parameter(layer_scale_init_value * torch.ones(dim)) if layer_scale_init_value > 0 else None
        self.drop_path = StochasticDepth(drop_path) if drop_path > 0 else nn.Identity()

    def forward(self, x):
        no_drop = isinstance(self.drop_path, nn.Identity) or not self.training
        if (USE_HIP_DWCONV and no_drop and x.is_cuda and x.dtype == torch.float32 and x.shape[1] % 4 == 0
                and x.shape[1] <= 1024 and not x.is_contiguous()
                and x.is_contiguous(memory_format=torch.channels_last)):
            # channels_last trunk (what MIOpen's NHWC convolutions hand us): the whole block stays in NHWC
            xn = x.permute(0, 2, 3, 1)  # contiguous (B,H,W,C) view
            # (skip, y): the skip gradient is added inside the depthwise backward kernel
            xn, y = _DwConv7x7NHWCSkip.apply(xn, self.dwconv.weight, self.dwconv.bias, _taps_major(self.dwconv))
            gc = self.__dict__.setdefault("_gemm_cache", ({}, {}))
            # fp16 x 2 GEMMs take their activation scale from analytic bounds (no pass over the activations, no dependence on
            # the batch): sqrt(C) max|w| + max|b| for the LayerNorm's output, and its image under the first projection for
            # GELU's (|GELU(t)| <= |t|)
            a1 = a2 = None
            from .. import _native as N
            if (_terms() == 22 and N.AMAX_FROM_PRODUCERS
                    and not (self.norm.weight.requires_grad or self.norm.bias.requires_grad
                             or self.pwconv1.weight.requires_grad or self.pwconv1.bias.requires_grad)):
                a1 = _ln_bound_word(self.norm, gc[0])
                a2 = _linear_bound_word(a1, self.pwconv1.weight, self.pwconv1.bias, gc[0])
            w2, b2, g = self.pwconv2.weight, self.pwconv2.bias, self.gamma
            folded = g is not None and not (w2.requires_grad or g.requires_grad or (b2 is not None and b2.requires_grad))
            if folded:
                # frozen weights: the layer scale is folded into the second projection (one kernel less each way)
                key = _tkey(w2, b2, g)
                cache = self.__dict__.setdefault("_fold_cache", {})
                if cache.get("key") != key:
                    with torch.no_grad():
                        cache.update(key=key, w=_stable(cache.get("w"), (w2 * g[:, None]).contiguous()),
                                     b=None if b2 is None else _stable(cache.get("b"), b2 * g))
                gelu_ok = isinstance(self.act, nn.GELU) and self.act.approximate == "none"
                if (gelu_ok and _mlp_fusable(y, self.pwconv1.weight, self.pwconv1.bias, cache["w"], cache["b"])
                        and _mlp_takes_layernorm(y, self.norm, self.pwconv1.weight, self.pwconv1.bias, cache["w"], a1, a2)
                        and xn.is_contiguous() and xn.data_ptr() % 16 == 0):
                    # LayerNorm, both projections, GELU and the residual in ONE kernel per direction (M8f)
                    return _FrozenMlp.apply(y, self.pwconv1.weight, self.pwconv1.bias, cache["w"], cache["b"], xn, gc, a1, a2,
                                            (self.norm.weight, self.norm.bias, self.norm.eps)).permute(0, 3, 1, 2)
                yn = self.norm(y)
                if gelu_ok and _mlp_fusable(yn, self.pwconv1.weight, self.pwconv1.bias, cache["w"], cache["b"]):
                    # GELU, GELU' and the residual add in the GEMM epilogues
                    return _FrozenMlp.apply(yn, self.pwconv1.weight, self.pwconv1.bias, cache["w"], cache["b"], xn, gc, a1,
                                            a2).permute(0, 3, 1, 2)
                y = self.act(_linear_frozen(gc[0], yn, self.pwconv1.weight, self.pwconv1.bias, amax=a1))
                y = _linear_frozen(gc[1], y, cache["w"], cache["b"], amax=a2)
            else:
                y = self.act(_linear_frozen(gc[0], self.norm(y), self.pwconv1.weight, self.pwconv1.bias, amax=a1))
                y = self.pwconv2(y)
                if g is not None:
                    y = g * y
            return (xn + y).permute(0, 3, 1, 2)
        y = depthwise7x7(self.dwconv, x)
        if no_drop and _fast_layout_ok(x) and y.is_contiguous():
            # NCHW -> NHWC and back through the tiled transposes, layer scale + residual fused into the second
            y = self.pwconv2(self.act(self.pwconv1(self.norm(_ToNHWC.apply(y)))))
            return _ScaleResidual.apply(x, y, self.gamma)
        y = y.permute(0, 2, 3, 1)
        y = self.pwconv2(self.act(self.pwconv1(self.norm(y))))
        if self.gamma is not None:
            y = self.gamma * y
        return x + self.drop_path(y.permute(0, 3, 1, 2))


class _PatchConv2x2(torch.autograd.Function):
    """2x2 / stride-2 convolution with frozen weights (the three down-sampling layers of the trunk,
    reference convnext_orig.py:118-124) as ONE GEMM on the 2x2 patches: y[b,i,j,:] = W (Cout, 2*2*C) . patch(b,i,j).
    MIOpen's kernel for this shape accumulates with atomics: its output differs in the last bit from run to run,
    which made whole attacks irreproducible (and dependent on how images were batched or sharded).  The GEMM is
    bitwise reproducible; input and output are channels_last, the patch gather is one strided copy each way."""

    @staticmethod
    def forward(ctx, x, weight, bias, cache, amax=None):
        B, C, H, W = x.shape
        key = _tkey(weight)
        if cache.get("key") != key:
            cache.update(key=key, w=_stable(cache.get("w"), weight.detach().permute(0, 2, 3, 1).reshape(weight.shape[0], 4 * C).contiguous()))
        wr = cache["w"]                                                        # (Cout, (di, dj, c))
        from .. import _native as N
        xn = x.permute(0, 2, 3, 1)
        if xn.is_contiguous() and xn.dtype == torch.float32 and C % 4 == 0:
            patches = N.patch2x2(xn)                                           # one 16-byte-per-lane gather pass
        else:
            patches = xn.reshape(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, 4 * C)
        y = _frozen_mm(patches, wr, cache, "patch_fwd", bias=bias, amax=amax, groups=B)
        ctx.wr, ctx.shape, ctx.cache, ctx.terms = wr, (B, C, H, W), cache, _terms()
        # under autocast the GEMM ran (and returned) bf16: the trunk's residual stream stays fp32
        return y.float().view(B, H // 2, W // 2, -1).permute(0, 3, 1, 2)       # channels_last (B,Cout,H/2,W/2)

    @staticmethod
    def backward(ctx, gy):
        B, C, H, W = ctx.shape
        g = gy.permute(0, 2, 3, 1).reshape(-1, gy.shape[1])
        rows = _frozen_mm(g if g.is_contiguous() else g.contiguous(), ctx.wr, ctx.cache, "patch_bwd", trans=True,
                          terms=_bwd_terms(ctx.terms), groups=B, row_amax=True).float()
        if C % 4 == 0 and rows.is_contiguous():
            from .. import _native as N
            gp = N.unpatch2x2(rows, B, H, W)
        else:
            gp = rows.view(B, H // 2, W // 2, 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)
        return gp.permute(0, 3, 1, 2), None, None, None, None


def _patch_conv_ok(conv, x):
    return (USE_HIP_DWCONV and isinstance(conv, nn.Conv2d) and conv.kernel_size == (2, 2) and conv.stride == (2, 2)
            and conv.padding == (0, 0) and conv.groups == 1 and x.is_cuda and x.dtype == torch.float32
            and conv.weight.dtype == torch.float32
            and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
            and not (conv.weight.requires_grad or (conv.bias is not None and conv.bias.requires_grad)))


class ConvNeXt(nn.Module):
    def __init__(self, strr: str, in_chans: int = 3, layer_scale_init_value: float = 1.0, out_indices=(0, 1, 2, 3)):
        super().__init__()
        if strr not in CONVNEXT_SETTINGS:
            raise AssertionError(f"ConvNeXt model name should be in {list(CONVNEXT_SETTINGS)}")
        depths, dims, _, dpr = CONVNEXT_SETTINGS[strr]
        self.variant = strr
        self.downsample_layers = nn.ModuleList()
        if "CVST" in strr:
            self.downsample_layers.append(ConvStem())
        else:
            self.downsample_layers.append(nn.Sequential(
                nn.Conv2d(in_chans, dims[0], 4, stride=4), LayerNorm(dims[0], data_format="channels_first")))
        for i in range(3):
            self.downsample_layers.append(nn.Sequential(
                LayerNorm(dims[i], data_format="channels_first"), nn.Conv2d(dims[i], dims[i + 1], 2, stride=2)))
        rates = torch.linspace(0, dpr, sum(depths)).tolist()
        self.stages = nn.ModuleList()
        k = 0
        for i in range(4):
            self.stages.append(nn.Sequential(*[
                Block(dims[i], rates[k + j], layer_scale_init_value) for j in range(depths[i])]))
            k += depths[i]
        self.out_indices = tuple(out_indices)
        for i in range(4):
            self.add_module(f"norm{i}", LayerNorm(dims[i], data_format="channels_first"))
        self.apply(self._init)

    @staticmethod
    def _init(m):
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)

    def forward(self, x):
        feats = []
        for i in range(4):
            ds = self.downsample_layers[i]
            if i > 0 and _patch_conv_ok(ds[1], x):
                pc = ds[1].__dict__.setdefault("_patch_cache", {})
                from .. import _native as N
                ln_ok = _terms() == 22 and N.AMAX_FROM_PRODUCERS and not (ds[0].weight.requires_grad or ds[0].bias.requires_grad)
                x = _PatchConv2x2.apply(ds[0](x), ds[1].weight, ds[1].bias, pc, _ln_bound_word(ds[0], pc) if ln_ok else None)
            else:
                x = ds(x)
            if STAGE_ENTRY_CONTIGUOUS and not x.is_contiguous():
                x = x.contiguous()  # the blocks' stencil / transpose kernels want NCHW planes
            x = self.stages[i](x)
            if i in self.out_indices:
                feats.append(getattr(self, f"norm{i}")(x))
        return tuple(feats)


# Winograd tile for the head's large 3x3 convolutions (env SEA_WINOGRAD): 4 = F(4x4,3x3), 4x fewer
# multiplications than a direct convolution; 2 = F(2x2,3x3), 2.25x fewer; 0 = MIOpen's implicit-GEMM kernels.
# All fp32.  Measured on MI355X (B=8, 512x512, ConvNeXt-T head, random init): APGD step 66.9 ms (0) / 44.4 ms
# (2) / 34.3 ms (4); max |logit difference| against the MIOpen path 4.9e-7 (2) / 1.0e-6 (4) at logit scale
# 0.25; unit-test error bounds vs float64: 2e-5 (2, same as direct fp32) / 3e-4 (4).  The config-#1 reference
# parity test (tests/test_config1_parity.py) passes at the same tolerances for all three settings.
WINOGRAD_TILE = int(os.environ.get("SEA_WINOGRAD", "4"))
WINOGRAD_MIN_PIXELS = 16 * 16  # the PSP bottleneck (2816 -> 512 at 16x16) still wins 3x; below, few tiles per GEMM


class _WinoConv3x3(torch.autograd.Function):
    """act(scale[c] * conv2d(x, w, padding=1) + shift[c]) for frozen 3x3 filters: libsea_hip Winograd transforms
    around a hipBLASLt batched GEMM, forward and input gradient (the filters get no gradient: attack-time
    weights are frozen).  With ``relu`` the eval-mode BatchNorm (scale, shift) and the ReLU of a ConvModule run
    in the output transform, and their backward in the input transform of the gradient convolution."""

    @staticmethod
    @_fp32_fwd
    def forward(ctx, x, weight, m, cache, scale, shift, relu):
        from .. import _native as N
        key = (_tkey(weight), m)
        if cache.get("key") != key:
            # (the backward image is refreshed lazily, in place: `bwd_key` says which weights it was derived from)
            cache.update(key=key, fwd=_stable(cache.get("fwd"), N.wino_filter(weight.contiguous(), m, False)))
        ctx.cache, ctx.m, ctx.weight, ctx.relu, ctx.terms = cache, m, weight, relu, _terms()
        y = N.wino_conv3x3_cl(_dense_cl(x), cache["fwd"], m, bias=shift, scale=scale, relu=relu, gemm_terms=ctx.terms)
        if relu:
            ctx.save_for_backward(y, scale)
        return y

    @staticmethod
    @_fp32_bwd
    def backward(ctx, gy):
        from .. import _native as N
        cache = ctx.cache
        if cache.get("bwd") is None or cache.get("bwd_key") != cache.get("key"):
            cache.update(bwd=_stable(cache.get("bwd"), N.wino_filter(ctx.weight.contiguous(), ctx.m, True)), bwd_key=cache.get("key"))
        gate, gscale = ctx.saved_tensors if ctx.relu else (None, None)
        g = gy if N.cl_pixel_stride(gy) is not None else gy.contiguous(memory_format=_CL)  # slices read in place
        gx = N.wino_conv3x3_cl(g, cache["bwd"], ctx.m, gate=gate, gate_scale=gscale, gemm_terms=_bwd_terms(ctx.terms))
        return gx, None, None, None, None, None, None


def _folded_bn(bn, conv_bias, cache):
    """eval-mode BatchNorm as y = scale[c] * x + shift[c] (cached until a buffer / parameter changes)"""
    ts = (bn.weight, bn.bias, bn.running_mean, bn.running_var, conv_bias)
    key = _tkey(*ts)
    if cache.get("bn_key") != key:
        with torch.no_grad():
            scale = torch.rsqrt(bn.running_var + bn.eps)
            if bn.weight is not None:
                scale = scale * bn.weight
            shift = -bn.running_mean * scale
            if bn.bias is not None:
                shift = shift + bn.bias
            if conv_bias is not None:
                shift = shift + scale * conv_bias
        cache.update(bn_key=key, scale=_stable(cache.get("scale"), scale.float().contiguous()),
                     shift=_stable(cache.get("shift"), shift.float().contiguous()))
    return cache["scale"], cache["shift"]


def _wino_ok(conv, x):
    return (WINOGRAD_TILE in (2, 4) and x.is_cuda and x.dtype == torch.float32 and conv.kernel_size == (3, 3)
            and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
            and conv.padding_mode == "zeros" and not conv.weight.requires_grad
            and (conv.bias is None or not conv.bias.requires_grad) and conv.in_channels % 4 == 0
            and conv.out_channels % 4 == 0 and x.shape[2] * x.shape[3] >= WINOGRAD_MIN_PIXELS
            and not torch.is_autocast_enabled())


USE_GEMM_POINTWISE = True


def _pointwise_ok(mod, x):
    conv, bn = mod.conv, mod.batch_norm
    return (USE_GEMM_POINTWISE and x.is_cuda and x.dtype == torch.float32 and conv.kernel_size == (1, 1)
            and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.groups == 1 and not bn.training
            and bn.track_running_stats and isinstance(mod.activation, nn.ReLU)
            and not any(p.requires_grad for p in mod.parameters()) and not torch.is_autocast_enabled()
            and x.is_contiguous(memory_format=torch.channels_last))


class _PointwiseRelu(torch.autograd.Function):
    """relu(x2d @ W^T + shift) for frozen (W, shift) through M8 (bias + ReLU in the GEMM's epilogue); the backward gates
    the incoming gradient with the output and runs the transposed product."""

    @staticmethod
    def forward(ctx, x2, w, shift, cache, groups=1):
        y = _frozen_mm(x2, w, cache, "pw_fwd", bias=shift, relu=True, groups=groups)
        ctx.save_for_backward(y)
        ctx.w, ctx.cache, ctx.terms, ctx.groups = w, cache, _terms(), groups
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        terms = _bwd_terms(ctx.terms)
        if (RELU_GATE_PROLOGUE and terms in (1, 2, 22) and gy.is_contiguous() and gy.dtype == torch.float32
                and _split_ok(gy, ctx.w.shape[0], terms)):
            # the ReLU gate is applied to the A tile of the input-gradient GEMM while it is staged: no masked copy of gy
            from .. import _native as N
            return (N.gemm_split(gy, _packed(ctx.w, ctx.cache, "pw_bwd", True, terms), a_relu_gate=y, groups=ctx.groups,
                                 row_amax=True), None, None, None, None)
        g = torch.where(y > 0, gy, torch.zeros((), dtype=gy.dtype, device=gy.device))
        return (_frozen_mm(g, ctx.w, ctx.cache, "pw_bwd", trans=True, terms=terms, groups=ctx.groups, row_amax=True), None, None,
                None, None)


class ConvModule(nn.Module):
    """bias-free conv + BatchNorm + ReLU (uperforseg.py:119-146)."""

    def __init__(self, cin, cout, kernel_size, padding=0, dilation=1):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size, padding=padding, dilation=dilation, bias=False)
        self.batch_norm = nn.BatchNorm2d(cout)
        self.activation = nn.ReLU()

    def _pointwise(self, x):
        """1x1 ConvModule on a channels_last tensor with frozen weights and eval-mode BatchNorm: one hipBLASLt GEMM
        (the BatchNorm folded into its weights and bias) + ReLU; no MIOpen layout transposes, no BatchNorm kernels."""
        if not hasattr(self, "_wino_cache"):
            object.__setattr__(self, "_wino_cache", {})
        cache, w = self._wino_cache, self.conv.weight
        scale, shift = _folded_bn(self.batch_norm, self.conv.bias, cache)
        key = (_tkey(w), cache["bn_key"])
        if cache.get("pw_key") != key:
            with torch.no_grad():
                cache.update(pw_key=key, pw_w=_stable(cache.get("pw_w"), (w.view(w.shape[0], -1) * scale[:, None]).contiguous()))
        B, _, H, W = x.shape
        # 2-D GEMM on the (pixels, Cin) view; ReLU in place on the GEMM's own output (an in-place op on a VIEW of it
        # would make autograd clone / copy whole tensors in CopySlices)
        x2 = x.permute(0, 2, 3, 1).reshape(B * H * W, -1)
        if _split_ok(x2, x2.shape[1]):
            y = _PointwiseRelu.apply(x2, cache["pw_w"], shift, cache, B)
        else:
            y = torch.relu_(torch.addmm(shift, x2, cache["pw_w"].t()))
        return y.view(B, H, W, -1).permute(0, 3, 1, 2)

    def forward(self, x):
        if _pointwise_ok(self, x):
            return self._pointwise(x)
        if _wino_ok(self.conv, x):
            if not hasattr(self, "_wino_cache"):
                object.__setattr__(self, "_wino_cache", {})
            bn = self.batch_norm
            if (not bn.training and bn.track_running_stats and isinstance(self.activation, nn.ReLU)
                    and not any(p.requires_grad for p in bn.parameters())):
                scale, shift = _folded_bn(bn, self.conv.bias, self._wino_cache)
                return _WinoConv3x3.apply(x, self.conv.weight, WINOGRAD_TILE, self._wino_cache, scale, shift, True)
            y = _WinoConv3x3.apply(x, self.conv.weight, WINOGRAD_TILE, self._wino_cache, None, self.conv.bias, False)
        else:
            y = self.conv(x)
        return self.activation(self.batch_norm(y))


class _UpsampleBilinear(torch.autograd.Function):
    """Bilinear up-sampling through libsea_hip M2 (forward: one streaming write; backward: deterministic
    gather).  Channels_last inputs (what the head's MIOpen convolutions hand over) stay channels_last: the
    NHWC kernels run lanes along C, so no layout copy surrounds the op."""

    @staticmethod
    @_fp32_fwd
    def forward(ctx, x, size):
        from .. import _native as N
        ctx.in_size = tuple(x.shape[2:])
        ctx.cl = USE_HIP_UPSAMPLE_NHWC and N._is_cl(x)
        if ctx.cl:
            return N.upsample_bilinear_cl(x, size)
        return N.upsample_bilinear(x.contiguous(), size)

    @staticmethod
    @_fp32_bwd
    def backward(ctx, gy):
        from .. import _native as N
        if ctx.cl:
            return N.upsample_bilinear_backward_cl(gy.contiguous(memory_format=torch.channels_last), ctx.in_size), None
        return N.upsample_bilinear_backward(gy.contiguous(), ctx.in_size), None


USE_HIP_UPSAMPLE_NHWC = True
USE_HIP_UPSAMPLE = True


def _up(x, size):
    size = tuple(int(v) for v in size)
    if USE_HIP_UPSAMPLE and x.is_cuda and size[0] >= x.shape[2] and size[1] >= x.shape[3]:
        if x.dtype == torch.float32:
            return _UpsampleBilinear.apply(x, size)
        if x.dtype in (torch.bfloat16, torch.float16) and torch.is_autocast_enabled():
            # 16-bit logits of a training forward under autocast (PIR-AT's outer step): M2 in fp32 on the small input instead
            # of ATen's bilinear kernels on the large output (151 channels x4: 4.0 ms backward vs 0.25 ms); the losses that
            # follow run in fp32 under autocast anyway
            return _UpsampleBilinear.apply(x.float(), size)
    return F.interpolate(x, size=size, mode="bilinear", align_corners=False)


_CL = torch.channels_last


def _dense_cl(t):
    from .. import _native as N
    return t if N.cl_pixel_stride(t) == t.shape[1] else t.contiguous(memory_format=_CL)


class _UpAddCL(torch.autograd.Function):
    """res + up(x) in one pass over channels_last tensors (the FPN top-down add)."""

    @staticmethod
    @_fp32_fwd
    def forward(ctx, x, res):
        from .. import _native as N
        ctx.in_size = tuple(x.shape[2:])
        return N.upsample_bilinear_cl(_dense_cl(x), res.shape[2:], residual=_dense_cl(res))

    @staticmethod
    @_fp32_bwd
    def backward(ctx, gy):
        from .. import _native as N
        g = gy if N.cl_pixel_stride(gy) is not None else gy.contiguous(memory_format=_CL)
        gx = N.upsample_bilinear_backward_cl(g, ctx.in_size) if ctx.needs_input_grad[0] else None
        return gx, gy


class _UpCatCL(torch.autograd.Function):
    """torch.cat([up(t) for t in ts], 1) for channels_last tensors without materialising the up-sampled maps:
    each one is written straight into its channel slice of the concatenation buffer, and the backward
    gathers each gradient straight out of the matching slice of the buffer's gradient."""

    @staticmethod
    @_fp32_fwd
    def forward(ctx, size, *ts):
        from .. import _native as N
        B, (H, W) = ts[0].shape[0], size
        buf = torch.empty(B, sum(t.shape[1] for t in ts), H, W, dtype=torch.float32, device=ts[0].device,
                          memory_format=_CL)
        off = 0
        for t in ts:
            sl = buf[:, off:off + t.shape[1]]
            if tuple(t.shape[2:]) == (H, W):
                sl.copy_(t)
            else:
                N.upsample_bilinear_cl(_dense_cl(t), size, out=sl)
            off += t.shape[1]
        ctx.shapes = [tuple(t.shape) for t in ts]
        return buf

    @staticmethod
    @_fp32_bwd
    def backward(ctx, g):
        from .. import _native as N
        if N.cl_pixel_stride(g) != g.shape[1]:
            g = g.contiguous(memory_format=_CL)
        grads, off = [None], 0
        for i, shp in enumerate(ctx.shapes):
            sl = g[:, off:off + shp[1]]
            off += shp[1]
            if not ctx.needs_input_grad[i + 1]:
                grads.append(None)
            elif shp[2:] == tuple(g.shape[2:]):
                grads.append(sl)
            else:
                grads.append(N.upsample_bilinear_backward_cl(sl, shp[2:]))
        return tuple(grads)


def _cl_fusable(ts, size):
    from .. import _native as N
    return (USE_HIP_UPSAMPLE and USE_HIP_UPSAMPLE_NHWC and N._is_cl(ts[0])
            and all(t.is_cuda and t.dtype == torch.float32 and t.shape[1] % 4 == 0 and t.shape[2] <= size[0]
                    and t.shape[3] <= size[1] for t in ts))


def _up_add(x, res):
    """res + bilinear_up(x -> res's size)"""
    size = tuple(int(v) for v in res.shape[2:])
    if _cl_fusable([res, x], size):
        return _UpAddCL.apply(x, res)
    return res + _up(x, size)


def _up_cat(ts, size):
    """torch.cat([bilinear_up(t -> size) for t in ts], dim=1) (tensors already at `size` are copied)"""
    size = tuple(int(v) for v in size)
    if _cl_fusable(ts, size):
        return _UpCatCL.apply(size, *ts)
    return torch.cat([t if tuple(t.shape[2:]) == size else _up(t, size) for t in ts], dim=1)


# M6: inputs of the FPN bottleneck that are >= x3 up-samplings are not up-sampled at all: their nine 3x3 taps run
# as one GEMM at the coarse resolution (channel mixing commutes with bilinear interpolation) and a gather adds the
# shifted interpolations into the Winograd output transform.  ConvNeXt-T, B=8, 512x512: the Winograd GEMM of the
# bottleneck shrinks from K=2048 to K=1024 input channels.
USE_FUSED_FPN_BOTTLENECK = True
LOWRES_MIN_FACTOR = 3.0


def _fpn_bottleneck_forward(ctx, m, cache, weight, scale, shift, fs, first_grad_index):
    """forward of _FpnBottleneck / _FpnBottleneckClassify: returns y = relu(bn(conv3x3(cat(...)))) and fills ``ctx`` for
    _fpn_bottleneck_backward; ``first_grad_index`` = position of fs[0] among the Function's inputs"""
    from .. import _native as N
    B, (H, W), Cout = fs[0].shape[0], fs[0].shape[2:], weight.shape[0]
    chans = [f.shape[1] for f in fs]
    offs = [sum(chans[:i]) for i in range(len(fs))]
    hi = [i for i, f in enumerate(fs) if i == 0 or H / f.shape[2] < LOWRES_MIN_FACTOR or W / f.shape[3] < LOWRES_MIN_FACTOR]
    lo = [i for i in range(len(fs)) if i not in hi]
    key = (_tkey(weight), m, tuple(hi), tuple(chans))
    if cache.get("fpn_key") != key:
        w_hi = torch.cat([weight[:, offs[i]:offs[i] + chans[i]] for i in hi], 1).contiguous()
        cache.update(fpn_key=key, fpn_fwd=_stable(cache.get("fpn_fwd"), N.wino_filter(w_hi, m, False)),
                     fpn_bwd=_stable(cache.get("fpn_bwd"), N.wino_filter(w_hi, m, True)),
                     fpn_lo=_stable(cache.get("fpn_lo"), [weight[:, offs[i]:offs[i] + chans[i]].permute(2, 3, 0, 1)
                                                          .reshape(9 * Cout, chans[i]).contiguous() for i in lo]))
        # rows of fpn_lo (tap, cout): F.linear -> (B,h,w,9*Cout)
    # the fine inputs are transformed side by side into the Winograd domain: no concatenation buffer
    xs = [_dense_cl(fs[i]) if tuple(fs[i].shape[2:]) == (H, W) else N.upsample_bilinear_cl(_dense_cl(fs[i]), (H, W))
          for i in hi]
    extra = None
    for j, i in enumerate(lo):
        f = _dense_cl(fs[i])
        G = _frozen_mm(f.permute(0, 2, 3, 1).reshape(-1, f.shape[1]), cache["fpn_lo"][j], cache,
                       f"fpn_lo_fwd{j}", groups=B).view(B, f.shape[2], f.shape[3], 9, Cout)
        extra = N.tap_gather(G, (H, W), extra)
    ctx.terms = _terms()
    y = N.wino_conv3x3_cl(xs, cache["fpn_fwd"], m, bias=shift, scale=scale, relu=True, addend=extra,
                          gemm_terms=ctx.terms)
    ctx.cache, ctx.m, ctx.hi, ctx.lo, ctx.chans, ctx.first = cache, m, hi, lo, chans, first_grad_index
    ctx.shapes = [tuple(f.shape) for f in fs]
    return y


def _fpn_bottleneck_backward(ctx, gz, y):
    """input gradients of the FPN bottleneck from gz = the gradient at the convolution's output (the ReLU gate and the folded
    BatchNorm scale already applied); y: the saved output (its shape)"""
    from .. import _native as N
    cache, chans, shapes = ctx.cache, ctx.chans, ctx.shapes
    B, _, H, W = y.shape
    grads = [None] * len(shapes)
    if any(ctx.needs_input_grad[ctx.first + i] for i in ctx.hi):
        gbuf = N.wino_conv3x3_cl(gz, cache["fpn_bwd"], ctx.m, gemm_terms=_bwd_terms(ctx.terms))
        off = 0
        for i in ctx.hi:
            sl = gbuf[:, off:off + chans[i]]
            off += chans[i]
            if ctx.needs_input_grad[ctx.first + i]:
                grads[i] = sl if shapes[i][2:] == (H, W) else N.upsample_bilinear_backward_cl(sl, shapes[i][2:])
    for j, i in enumerate(ctx.lo):
        if ctx.needs_input_grad[ctx.first + i]:
            h, w = shapes[i][2:]
            dG = N.tap_gather_backward(gz, (h, w))
            grads[i] = _frozen_mm(dG.view(B * h * w, -1), cache["fpn_lo"][j], cache, f"fpn_lo_bwd{j}", trans=True,
                                  terms=_bwd_terms(ctx.terms), groups=B, row_amax=True).view(B, h, w, chans[i]).permute(0, 3, 1, 2)
    return grads


class _FpnBottleneck(torch.autograd.Function):
    """relu(bn(conv3x3(cat([f0, up(f1), ..., up(fn)])))) for frozen weights / eval-mode BatchNorm, input grads only."""

    @staticmethod
    @_fp32_fwd
    def forward(ctx, m, cache, weight, scale, shift, *fs):
        y = _fpn_bottleneck_forward(ctx, m, cache, weight, scale, shift, fs, 5)
        ctx.save_for_backward(y, scale)
        return y

    @staticmethod
    @_fp32_bwd
    def backward(ctx, gy):
        from .. import _native as N
        y, scale = ctx.saved_tensors
        gz = N.gate_scale(_dense_cl(gy), y, scale)
        return (None, None, None, None, None, *_fpn_bottleneck_backward(ctx, gz, y))


class _FpnBottleneckClassify(torch.autograd.Function):
    """_FpnBottleneck followed by the head's classifier (uperforseg.py:296-304 + 262) as ONE autograd node, for at most 32
    classes (M10): the classifier's input-gradient kernel applies the bottleneck's ReLU gate and BatchNorm scale on the way
    out, so the 0.8 GB gate pass between the two backward steps disappears; same bits as the two nodes."""

    @staticmethod
    @_fp32_fwd
    def forward(ctx, m, cache, weight, scale, shift, cls_w2d, cls_bias, *fs):
        from .. import _native as N
        y = _fpn_bottleneck_forward(ctx, m, cache, weight, scale, shift, fs, 7)
        B, Cout, H, W = y.shape
        ctx.save_for_backward(y, scale)
        ctx.cls_w2d = cls_w2d
        return N.classifier_forward(y.permute(0, 2, 3, 1).reshape(B * H * W, Cout), cls_w2d, cls_bias, B, H * W).view(B, -1, H, W)

    @staticmethod
    @_fp32_bwd
    def backward(ctx, g):
        from .. import _native as N
        y, scale = ctx.saved_tensors
        B, Cout, H, W = y.shape
        gz = N.classifier_backward(g.reshape(B, g.shape[1], H * W).contiguous(), ctx.cls_w2d,
                                   gate=y.permute(0, 2, 3, 1).reshape(B * H * W, Cout), gate_scale=scale)
        gz = gz.view(B, H, W, Cout).permute(0, 3, 1, 2)
        return (None, None, None, None, None, None, None, *_fpn_bottleneck_backward(ctx, gz, y))


def _fpn_fusable(mod, outs):
    bn = mod.batch_norm
    size = tuple(outs[0].shape[2:])
    return (USE_FUSED_FPN_BOTTLENECK and _wino_ok(mod.conv, outs[0]) and mod.conv.bias is None and not bn.training
            and bn.track_running_stats and isinstance(mod.activation, nn.ReLU)
            and not any(p.requires_grad for p in bn.parameters()) and _cl_fusable(outs, size)
            and mod.conv.in_channels == sum(o.shape[1] for o in outs)
            and any(size[0] / o.shape[2] >= LOWRES_MIN_FACTOR and size[1] / o.shape[3] >= LOWRES_MIN_FACTOR
                    for o in outs[1:]))


USE_HIP_ADAPTIVE_POOL = os.environ.get("SEA_HIP_POOL", "1") != "0"


class _AdaptivePoolCL(torch.autograd.Function):
    """nn.AdaptiveAvgPool2d(s) of a dense channels_last map through libsea_hip (no parameters: training and attack alike)"""

    @staticmethod
    @_fp32_fwd
    def forward(ctx, x, oh, ow):
        from .. import _native as N
        ctx.hw = (x.shape[2], x.shape[3])
        return N.adaptive_avg_pool_nhwc(x, oh, ow)

    @staticmethod
    @_fp32_bwd
    def backward(ctx, g):
        from .. import _native as N
        return N.adaptive_avg_pool_nhwc_backward(g, *ctx.hw), None, None


def _adaptive_pool(pool: nn.AdaptiveAvgPool2d, x):
    from .. import _native as N
    s = pool.output_size
    oh, ow = (s, s) if isinstance(s, int) else s
    if (USE_HIP_ADAPTIVE_POOL and x.is_cuda and x.dtype == torch.float32 and oh is not None and ow is not None
            and oh <= x.shape[2] and ow <= x.shape[3] and N.cl_pixel_stride(x) == x.shape[1]):
        return _AdaptivePoolCL.apply(x, oh, ow)
    return pool(x)


class PyramidPooling(nn.Module):
    """Children "0".."3", each Sequential-like [AdaptiveAvgPool2d(s), ConvModule] with children "0","1"."""

    def __init__(self, scales, cin, cout):
        super().__init__()
        for i, s in enumerate(scales):
            blk = nn.Module()
            blk.add_module("0", nn.AdaptiveAvgPool2d(s))
            blk.add_module("1", ConvModule(cin, cout, 1))
            self.add_module(str(i), blk)
        self.n = len(scales)

    def pooled(self, x):
        """the pooled + projected maps at their own (1, 2, 3, 6) resolutions"""
        return [getattr(getattr(self, str(i)), "1")(_adaptive_pool(getattr(getattr(self, str(i)), "0"), x)) for i in range(self.n)]

    def forward(self, x):
        return [_up(p, x.shape[2:]) for p in self.pooled(x)]


class _ClassifierGemm(torch.autograd.Function):
    """logits (B,cls,H,W) NCHW = W (cls,Cin) . y_b^T for a dense channels_last y (B,Cin,H,W), frozen weights.
    The backward computes the input gradient directly in y's channels_last layout, g_b^T (P,cls) . W (cls,Cin):
    letting autograd differentiate the transposed-view matmul instead costs a 268 MB transposing copy per step."""

    @staticmethod
    @_fp32_fwd
    def forward(ctx, y, w2d, bias):
        from .. import _native as N
        B, Cin, H, W = y.shape
        ctx.w2d, ctx.shape = w2d, (B, Cin, H, W)
        ctx.own = N.classifier_ok(H * W, Cin, w2d.shape[0]) and w2d.is_contiguous() and (bias is None or bias.is_contiguous())
        if ctx.own:     # M10: at most 32 classes on the fp32 matrix cores, y read once (csrc/classifier.hip)
            rows = y.permute(0, 2, 3, 1).reshape(B * H * W, Cin)
            return N.classifier_forward(rows, w2d, bias, B, H * W).view(B, w2d.shape[0], H, W)
        out = torch.matmul(w2d, y.permute(0, 2, 3, 1).reshape(B, H * W, Cin).transpose(1, 2))
        if bias is not None:
            out += bias.view(1, -1, 1)
        return out.view(B, w2d.shape[0], H, W)

    @staticmethod
    @_fp32_bwd
    def backward(ctx, g):
        from .. import _native as N
        B, Cin, H, W = ctx.shape
        if ctx.own:
            gy = N.classifier_backward(g.reshape(B, g.shape[1], H * W).contiguous(), ctx.w2d)
        else:
            gy = torch.matmul(g.reshape(B, g.shape[1], H * W).transpose(1, 2), ctx.w2d)      # (B, P, Cin) contiguous
        return gy.view(B, H, W, Cin).permute(0, 3, 1, 2), None, None


FUSE_CLASSIFIER_GATE = os.environ.get("SEA_FUSE_CLS_GATE", "1") != "0"   # A/B: _FpnBottleneckClassify


def _classifier_fusable(conv: nn.Conv2d, cin, f0):
    """the classifier behind the fused FPN bottleneck can ride in its autograd node (M10 kernels, frozen fp32 weights)"""
    from .. import _native as N
    return (FUSE_CLASSIFIER_GATE and conv.kernel_size == (1, 1) and conv.in_channels == cin and not torch.is_autocast_enabled()
            and conv.weight.dtype == torch.float32 and not conv.weight.requires_grad
            and (conv.bias is None or not conv.bias.requires_grad) and conv.weight.is_contiguous()
            and N.classifier_ok(f0.shape[2] * f0.shape[3], cin, conv.out_channels))


def _classify(conv: nn.Conv2d, y):
    """The head's final 1x1 convolution (uperforseg.py:262).  For frozen fp32 weights and a dense channels_last
    input it runs as one batched GEMM  W (cls, Cin) @ y_b^T (Cin, H*W)  whose output IS the NCHW logit tensor the
    loss kernel wants (no layout copy), bitwise reproducible; MIOpen's kernel for this shape is not (run-to-run
    differences in the last bit of the logits)."""
    from .. import _native as N
    if (y.is_cuda and y.dtype == torch.float32 and conv.kernel_size == (1, 1) and not torch.is_autocast_enabled()
            and not conv.weight.requires_grad and (conv.bias is None or not conv.bias.requires_grad)
            and N.cl_pixel_stride(y) == y.shape[1]):
        return _ClassifierGemm.apply(y, conv.weight.view(conv.out_channels, y.shape[1]), conv.bias)
    return conv(y)


class UperNetHead(nn.Module):
    def __init__(self, in_channels, cls, channels: int = 512, pool_scales=(1, 2, 3, 6)):
        super().__init__()
        self.in_channels = list(in_channels)
        self.classifier = nn.Conv2d(channels, cls, 1)
        self.psp_modules = PyramidPooling(pool_scales, self.in_channels[-1], channels)
        self.bottleneck = ConvModule(self.in_channels[-1] + len(pool_scales) * channels, channels, 3, padding=1)
        self.lateral_convs = nn.ModuleList(ConvModule(c, channels, 1) for c in self.in_channels[:-1])
        self.fpn_convs = nn.ModuleList(ConvModule(channels, channels, 3, padding=1) for _ in self.in_channels[:-1])
        self.fpn_bottleneck = ConvModule(len(self.in_channels) * channels, channels, 3, padding=1)

    def init_weights(self):
        self.apply(_head_init)

    def forward(self, feats):
        top = feats[-1]
        lat = [conv(feats[i]) for i, conv in enumerate(self.lateral_convs)]
        lat.append(self.bottleneck(_up_cat([top] + self.psp_modules.pooled(top), top.shape[2:])))
        for i in range(len(lat) - 1, 0, -1):
            lat[i - 1] = _up_add(lat[i], lat[i - 1])
        outs = [self.fpn_convs[i](lat[i]) for i in range(len(lat) - 1)] + [lat[-1]]
        neck = self.fpn_bottleneck
        if _fpn_fusable(neck, outs):
            if not hasattr(neck, "_wino_cache"):
                object.__setattr__(neck, "_wino_cache", {})
            scale, shift = _folded_bn(neck.batch_norm, None, neck._wino_cache)
            cls = self.classifier
            if _classifier_fusable(cls, neck.conv.out_channels, outs[0]):
                return _FpnBottleneckClassify.apply(WINOGRAD_TILE, neck._wino_cache, neck.conv.weight, scale, shift,
                                                    cls.weight.view(cls.out_channels, -1), cls.bias, *outs)
            y = _FpnBottleneck.apply(WINOGRAD_TILE, neck._wino_cache, neck.conv.weight, scale, shift, *outs)
        else:
            y = neck(_up_cat(outs, outs[0].shape[2:]))
        return _classify(self.classifier, y)


class UperNetFCNHead(nn.Module):
    def __init__(self, in_channels: int = 384, cls: int = 150, in_index: int = 2, channels: int = 256):
        super().__init__()
        self.in_index = in_index
        self.convs = nn.Sequential(ConvModule(in_channels, channels, 3, padding=1))
        self.classifier = nn.Conv2d(channels, cls, 1)

    def init_weights(self):
        self.apply(_head_init)

    def forward(self, feats):
        return self.classifier(self.convs(feats[self.in_index]))


def _head_init(m):
    if isinstance(m, (nn.Linear, nn.Conv2d)):
        m.weight.data.normal_(mean=0.0, std=0.02)
        if m.bias is not None:
            m.bias.data.zero_()


class UperNetForSemanticSegmentation(nn.Module):
    """``UperNetForSemanticSegmentation("ConvNeXt-T_CVST", n_cls, pretrained)`` (uperforseg.py:382-404)."""

    def __init__(self, backbone: str = "ConvNeXt-T_CVST", n_cls: int = 150, pretrained=None):
        super().__init__()
        _, variant = backbone.split("-")
        self.backbone = ConvNeXt(variant)
        dims, aux_in = CONVNEXT_SETTINGS[variant][1], CONVNEXT_SETTINGS[variant][2]
        self.decode_head = UperNetHead(dims, n_cls)
        self.auxiliary_head = UperNetFCNHead(aux_in, n_cls)
        if pretrained is not None:
            load_backbone_checkpoint(self.backbone, pretrained)
            self.decode_head.init_weights()
            self.auxiliary_head.init_weights()

    def _head_logits(self, feats):
        """Decode head.  Under autocast with FROZEN weights (the attack's forward: PIR-AT's inner PGD with TRAIN.AMP,
        BASELINE configs[3]) the head runs with autocast switched off: its fp32 fast paths (Winograd-domain 3x3
        convolutions at a quarter of the multiplications, NHWC upsample / concat kernels, folded-BatchNorm GEMMs) beat
        the bf16 library composition, while the trunk keeps autocast and runs its MLP GEMMs in bf16.  The islands' GEMMs
        run M8 with two bf16 terms there (16 significant bits per operand, three MFMA products)."""
        if torch.is_autocast_enabled() and feats[0].dtype == torch.float32 and not self.decode_head.classifier.weight.requires_grad:
            with torch.autocast("cuda", enabled=False), _gemm_terms(GEMM_TERMS_AUTOCAST if GEMM_TERMS else 0):
                return self.decode_head(feats)
        return self.decode_head(feats)

    def forward_lowres(self, input):
        """(logits at 1/4 resolution, output size): semseg.attacker fuses the final bilinear upsample
        into its loss kernel (K2u) when a model offers this hook."""
        return self._head_logits(self.backbone(input)), tuple(input.shape[2:])

    def _attack_island(self, input, lbl):
        """bf16 autocast + frozen weights + eval mode = the attack's forward (PIR-AT inner PGD, BASELINE configs[3])"""
        return (AUTOCAST_ISLAND == "model" and lbl is None and not self.training and torch.is_autocast_enabled() and input.is_cuda
                and input.dtype == torch.float32 and GEMM_TERMS != 0
                and not any(p.requires_grad for p in self.parameters()))       # EVERY weight frozen, not a sample of them

    def forward(self, input, lbl=None):
        if self._attack_island(input, lbl):
            with torch.autocast("cuda", enabled=False), _gemm_terms(GEMM_TERMS_AUTOCAST_MODEL):
                return _up(self.decode_head(self.backbone(input)).contiguous(), input.shape[2:])
        feats = self.backbone(input)
        logits = _up(self._head_logits(feats).contiguous(), input.shape[2:])  # NCHW logits for K2
        loss = None
        if lbl is not None:
            aux = _up(self.auxiliary_head(feats).contiguous(), input.shape[2:])
            loss = F.cross_entropy(logits, lbl, ignore_index=-1) + 0.4 * F.cross_entropy(aux, lbl, ignore_index=-1)
        if self.training:
            return loss, logits
        return logits


def load_backbone_checkpoint(backbone: ConvNeXt, path: str) -> None:
    """Import an ImageNet ConvNeXt checkpoint into the backbone (convnext_orig.py:190-307): either the
    official layout ({"model": {...}}, plain ConvNeXt) or the conv-stem robust-ImageNet layout
    (``stem.stem.N``, ``stages.L.downsample.P``, ``stages.J.blocks.K.{conv_dw,norm,mlp.fc1,mlp.fc2}``)."""
    ckpt = torch.load(path, map_location="cpu")
    sd = backbone.state_dict()
    new = {}
    if "CVST" not in backbone.variant:
        src = ckpt["model"]
        for k in sd:
            if k.startswith(("downsample_layers.", "stages.")) and k in src:
                new[k] = src[k]
    else:
        src = {k.replace("module.", "").replace("base_model.", ""): v for k, v in ckpt.items()}
        ren = {"dwconv": "conv_dw", "pwconv1": "mlp.fc1", "pwconv2": "mlp.fc2"}
        for k in sd:
            p = k.split(".")
            if p[0] == "downsample_layers" and p[1] == "0":
                new[k] = src[f"stem.stem.{p[3]}.{p[4]}"]
            elif p[0] == "downsample_layers":
                new[k] = src[f"stages.{p[1]}.downsample.{p[2]}.{p[3]}"]
            elif p[0] == "stages":
                tail = ".".join([ren.get(p[3], p[3])] + p[4:])
                new[k] = src[f"stages.{p[1]}.blocks.{p[2]}.{tail}"]
    missing = [k for k in sd if k.startswith(("downsample_layers.", "stages.")) and k not in new]
    if missing:
        raise KeyError(f"backbone checkpoint lacks {missing[:4]}...")
    backbone.load_state_dict({**sd, **new})
