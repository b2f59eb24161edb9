"""ctypes binding of libsea_hip.so (include/sea_hip.h) + thin torch-tensor wrappers.

PyTorch is plumbing here: it owns device memory and the stream; every kernel is launched through
the C ABI with raw pointers.  There is NO fallback: if the library is missing or a tensor is not on a
HIP device the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch  # must be imported before the library so that ITS libamdhip64 (same SONAME) is the one bound

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("SEA_HIP_LIB", os.path.join(_PKG, "lib", "libsea_hip.so"))

MODE_BY_NAME = {"mask-ce-avg": 0, "mask-ce-bal": 1, "js-avg": 2, "ce": 3, "ce-avg": 3}
DTYPE_CODE = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
LAYOUT_NCHW, LAYOUT_NHWC = 0, 1

_lib = None

_vp, _i, _i64, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
_SIGS = {
    "sea_abi_version": (C.c_int, []),
    "sea_build_info": (C.c_char_p, []),
    "sea_fastdiv_magic": (_i, [C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "sea_apgd_linf_step": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _i, _i64, _vp]),
    "sea_apgd_l2_workspace_bytes": (_i64, [_i]),
    "sea_apgd_l2_step": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _i, _i64, _vp]),
    "sea_linf_random_start": (_i, [_vp, _vp, _f, _vp, _i64, _vp]),
    "sea_linf_project": (_i, [_vp, _vp, _f, _vp, _i64, _vp]),
    "sea_pgd_linf_step": (_i, [_vp, _vp, _vp, _f, _f, _vp, _vp, _i, _i64, _vp]),
    "sea_loss_workspace_bytes": (_sz, [_i, _i64]),
    "sea_loss_fwd_bwd": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i64, _f, _vp, _vp, _i, _vp, _vp, _sz,
                              _vp, _vp, _vp, _vp]),
    "sea_loss_fwd_bwd_tuned": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i64, _f, _vp, _vp, _i, _vp, _vp,
                                    _sz, _vp, _vp, _vp, _vp, _i]),
    "sea_loss_upsampled_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "sea_loss_fwd_bwd_upsampled": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _i, _vp, _sz, _vp,
                                        _vp, _vp, _vp]),
    "sea_class_counts": (_i, [_vp, _i, _vp, _i, _i, _i, _i64, _i, _i, _vp, _vp, _vp, _vp]),
    "sea_confusion": (_i, [_vp, _i, _vp, _i, _i64, _i, _vp, _vp]),
    "sea_apgd_track": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                            _vp, _vp, _vp, _vp]),
    "sea_apgd_linf_step_graph": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _vp, _i, _i64, _vp]),
    "sea_apgd_track_graph": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                  _vp, _vp, _vp]),
    "sea_apgd_linf_step_graph_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _vp]),
    "sea_apgd_track_graph_dev": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                      _vp, _vp, _vp]),
    "sea_select_copy": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _i64, _vp]),
    "sea_count_ignored": (_i, [_vp, _i, _i, _i64, _vp, _vp]),
    "sea_worst_miou_greedy": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "sea_dwconv7x7": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sea_dwconv7x7_nhwc": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sea_dwconv7x7_nhwc_add": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sea_nchw_to_nhwc": (_i, [_vp, _vp, _vp, _i, _i, _i64, _vp]),
    "sea_nhwc_to_nchw": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i64, _vp]),
    "sea_upsample_bilinear_fwd": (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _vp]),
    "sea_upsample_bilinear_bwd": (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _vp]),
    "sea_layernorm_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _f, _vp]),
    "sea_layernorm_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _vp]),
    "sea_dwconv7x7_nhwc_wgrad_workspace": (_i64, [_i, _i, _i]),
    "sea_dwconv7x7_nhwc_wgrad": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "sea_adaptive_avg_pool_nhwc_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "sea_adaptive_avg_pool_nhwc_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "sea_stem_conv1_ln_gelu": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "sea_stem_conv1_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "sea_ln_gelu_cl_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i64, _f, _vp]),
    "sea_ln_gelu_cl_bwd": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i64, _f, _vp]),
    "sea_wino_tiles": (_i64, [_i, _i, _i, _i]),
    "sea_wino_input_transform": (_i, [_vp, _i64, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _vp]),
    "sea_wino_filter_transform": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "sea_wino_output_transform": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "sea_tap_gather_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "sea_tap_gather_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "sea_gate_scale": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _vp]),
    "sea_upsample_bilinear_nhwc_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i64, _vp]),
    "sea_upsample_bilinear_nhwc_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i64, _vp]),
    "sea_patch2x2": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sea_attention_fwd": (_i, [_vp, _vp, _vp, _i64, _i64, _i64, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    "sea_attention_bwd": (_i, [_vp, _vp, _vp, _i64, _i64, _i64, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                               _i64, _i64, _i64, _vp]),
    "sea_attention_bwd_terms": (_i, [_vp, _vp, _vp, _i64, _i64, _i64, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                     _i64, _i64, _i64, _i, _vp]),
    "sea_attention_fwd_terms": (_i, [_vp, _vp, _vp, _i64, _i64, _i64, _i, _i, _i, _i, _f, _vp, _vp, _i, _vp]),
    "sea_attention_fwd_f16": (_i, [_vp, _vp, _vp, _i64, _i64, _i64, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "sea_attention_bwd_f16": (_i, [_vp, _vp, _vp, _i64, _i64, _i64, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                   _i64, _i64, _i64, _vp]),
    "sea_absmax_bits": (_i, [_vp, _i64, _i, _i, _i, _i64, _i, _vp, _vp]),
    "sea_gemm_split_f16": (_i, [_vp, _i64, _vp, _vp, _i64, _vp, _i, _i, _i, _i, _i, _i64, _i64, _i64, _vp, _i, _vp, _vp]),
    "sea_wino_input_transform_amax": (_i, [_vp, _i64, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _vp, _vp]),
    "sea_gemm_split_fused": (_i, [_vp, _i64, _vp, _vp, _i64, _vp, _i, _i, _i, _i, _i, _i, _i64, _i64, _i64, _vp, _i, _vp, _vp, _vp]),
    "sea_gemm_splitk_reduce": (_i, [_vp, _i, _i, _i, _vp, _vp, _i64, _i, _vp, _i64, _vp, _vp]),
    "sea_gemm_split_packed_bytes": (_i64, [_i, _i, _i]),
    "sea_gemm_split_pack": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _vp]),
    "sea_gemm_split": (_i, [_vp, _i64, _vp, _vp, _i64, _vp, _i, _i, _i, _i, _i, _i, _i64, _i64, _i64, _vp]),
    "sea_mlp_fused_supported": (_i, [_i, _i]),
    "sea_classifier_supported": (_i, [_i, _i, _i]),
    "sea_classifier_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "sea_classifier_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "sea_probe_gelu_mismatches": (_i, [_vp, _vp]),
    "sea_probe_ln_rows": (_i, [_vp, _vp, _vp, _f, _i, _i, _vp, _vp, _vp, _vp]),
    "sea_mlp_fused_stamps": (_i, [_i, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "sea_mlp_fused_fwd": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i, _i, _i, _vp, _vp, _vp]),
    "sea_mlp_fused_bwd": (_i, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp, _vp, _vp]),
    "sea_ln_mlp_fused_fwd": (_i, [_vp, _i64, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i, _i, _i, _vp, _vp, _vp]),
    "sea_ln_mlp_fused_bwd": (_i, [_vp, _i64, _vp, _i64, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp, _vp, _vp]),
    "sea_gemm_split_mfma_shape": (_i, [_i]),
    "sea_gemm_split_pipeline": (_i, [_i]),
    "sea_probe_stream_copy": (_i, [_vp, _vp, _sz, _i, _vp]),
    "sea_probe_stream_read": (_i, [_vp, _vp, _sz, _vp]),
}
EXPORTS = tuple(_SIGS)


class SeaNativeError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle.  Raises SeaNativeError if the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SeaNativeError(
                f"libsea_hip.so not found at {LIB_PATH}: build it with "
                "`python robust-segmentation_amd/build_native.py` (there is no CPU fallback)")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(h, name)  # AttributeError here = ABI mismatch
            fn.restype, fn.argtypes = res, args
        _lib = h
    return _lib


def _check(rc: int, what: str):
    if rc != 0:
        raise SeaNativeError(f"{what} failed with hipError {rc}")


def _dev(*ts):
    """Every kernel is launched on the CURRENT stream of the CURRENT device, so all tensors must live there."""
    cur = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise SeaNativeError("libsea_hip works on HIP device tensors only (no CPU fallback); got a CPU tensor")
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise SeaNativeError(f"tensor on {t.device} but the current device is cuda:{cur}: call "
                                 "torch.cuda.set_device (or use `with torch.cuda.device(...)`) before the launch")


def _stream() -> int:
    # raw handle of the current stream of the current device (no Stream object: this runs ~600 times per attack step)
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


# A/B knob for in-loop measurements: SEA_K2_FORCE=<int> is passed as `force_vec` to every K2 launch that does not set
# one itself (bits 4-7: tuning variant, see csrc/loss_kernels.hip)
_K2_FORCE = int(os.environ.get("SEA_K2_FORCE", "0"), 0)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise SeaNativeError("expected a contiguous float32 tensor")
    return t


_INT_BYTES = {torch.int64: 8, torch.int32: 4, torch.int16: 2, torch.uint8: 1}


def int_bytes(t: torch.Tensor) -> int:
    try:
        return _INT_BYTES[t.dtype]
    except KeyError:
        raise SeaNativeError(f"unsupported label/index dtype {t.dtype}") from None


# ------------------------------------------------------------------------------------------------ K1/K5/K6
def apgd_linf_step(x, x_adv, x_old, grad, step_b, eps: float, a: float, out=None):
    _dev(x, x_adv, x_old, grad, step_b)
    B = x.shape[0]
    out = torch.empty_like(x) if out is None else out
    _check(lib().sea_apgd_linf_step(_p(_f32c(x)), _p(_f32c(x_adv)), _p(_f32c(x_old)), _p(_f32c(grad)),
                                    _p(_f32c(step_b)), eps, a, _p(_f32c(out)), B, x[0].numel(), _stream()),
           "sea_apgd_linf_step")
    return out


def apgd_l2_step(x, x_adv, x_old, grad, step_b, eps: float, a: float, out=None, workspace=None):
    """One APGD L2 update with momentum (reference semseg/attacker.py:412-436): four streaming passes, three deterministic
    per-image norms.  ``workspace``: sea_apgd_l2_workspace_bytes(B) bytes of device scratch (allocated here when None)."""
    _dev(x, x_adv, x_old, grad, step_b, out, workspace)
    B = x.shape[0]
    out = torch.empty_like(x) if out is None else out
    if workspace is None:
        workspace = torch.empty(lib().sea_apgd_l2_workspace_bytes(B) // 8, dtype=torch.float64, device=x.device)
    _check(lib().sea_apgd_l2_step(_p(_f32c(x)), _p(_f32c(x_adv)), _p(_f32c(x_old)), _p(_f32c(grad)), _p(_f32c(step_b)), eps, a,
                                  _p(_f32c(out)), _p(workspace), B, x[0].numel(), _stream()), "sea_apgd_l2_step")
    return out


def linf_random_start(x, u, eps: float, out=None):
    _dev(x, u)
    out = torch.empty_like(x) if out is None else out
    _check(lib().sea_linf_random_start(_p(_f32c(x)), _p(_f32c(u)), eps, _p(_f32c(out)), x.numel(), _stream()),
           "sea_linf_random_start")
    return out


def linf_project(z, x, eps: float, out=None):
    _dev(z, x)
    out = torch.empty_like(x) if out is None else out
    _check(lib().sea_linf_project(_p(_f32c(z)), _p(_f32c(x)), eps, _p(_f32c(out)), x.numel(), _stream()),
           "sea_linf_project")
    return out


def pgd_linf_step(X, delta, grad, alpha: float, eps: float, delta_out=None, x_in_out=None, clamp_input=False):
    _dev(X, delta, grad)
    delta_out = torch.empty_like(delta) if delta_out is None else delta_out
    _check(lib().sea_pgd_linf_step(_p(_f32c(X)), _p(_f32c(delta)), _p(_f32c(grad)), alpha, eps, _p(_f32c(delta_out)),
                                   _p(x_in_out), int(clamp_input), X.numel(), _stream()), "sea_pgd_linf_step")
    return delta_out


# ------------------------------------------------------------------------------------------------ K2
def logits_layout(logits: torch.Tensor):
    """(tensor usable by the kernel, layout code).  NCHW-contiguous and channels_last are both native."""
    if logits.dim() != 4:
        raise SeaNativeError("logits must be (B,C,H,W)")
    if logits.is_contiguous():
        return logits, LAYOUT_NCHW
    # the channels_last kernel stages 256 pixels x C classes in LDS (160 KB): beyond 159 classes go through NCHW
    if logits.is_contiguous(memory_format=torch.channels_last) and 256 * (logits.shape[1] | 1) * 4 <= 160 * 1024 - 64:
        return logits, LAYOUT_NHWC
    return logits.contiguous(), LAYOUT_NCHW


def loss_workspace(B: int, HW: int, device) -> torch.Tensor:
    n = lib().sea_loss_workspace_bytes(B, HW)
    return torch.empty(n, dtype=torch.uint8, device=device)


def loss_fwd_bwd(logits, y, weights, mode: int, track_mode: int, grad_scale: float, want_grad: bool = True,
                 pred=None, loss_px=None, workspace=None, out=None, dlogits=None, force_vec: int = 0,
                 defer: bool = False):
    """Run K2.  Returns dict(dlogits, loss_sum, track_sum, n_correct, pred, workspace).  With ``defer`` the
    second reduction stage is left to ``apgd_track`` (the sums are then None)."""
    _dev(logits, y, weights, pred, loss_px)
    logits, layout = logits_layout(logits)
    B, Cc, H, W = logits.shape
    HW = H * W
    if y.shape != (B, H, W) or not y.is_contiguous():
        raise SeaNativeError("labels must be a contiguous (B,H,W) tensor")
    dev = logits.device
    if workspace is None:
        workspace = loss_workspace(B, HW, dev)
    if defer:
        out = (None, None, None)
    elif out is None:
        out = (torch.empty(B, dtype=torch.float32, device=dev), torch.empty(B, dtype=torch.float32, device=dev),
               torch.empty(B, dtype=torch.int32, device=dev))
    if want_grad and dlogits is None:
        dlogits = torch.empty_like(logits)  # preserves the memory format
    if not want_grad:
        dlogits = None
    if weights is not None:
        weights = _f32c(weights)
        if weights.numel() != Cc:
            raise SeaNativeError("class weights must have C entries")
    L = lib()
    args = [_p(logits), DTYPE_CODE[logits.dtype], layout, _p(y), int_bytes(y), _p(weights), mode, track_mode, B, Cc,
            HW, grad_scale, _p(dlogits), _p(pred), 0 if pred is None else int_bytes(pred), _p(loss_px),
            _p(workspace), workspace.numel(), _p(out[0]), _p(out[1]), _p(out[2]), _stream()]
    if not force_vec and _K2_FORCE:
        force_vec = _K2_FORCE
    if force_vec:
        _check(L.sea_loss_fwd_bwd_tuned(*args, force_vec), "sea_loss_fwd_bwd_tuned")
    else:
        _check(L.sea_loss_fwd_bwd(*args), "sea_loss_fwd_bwd")
    return dict(dlogits=dlogits, loss_sum=out[0], track_sum=out[1], n_correct=out[2], pred=pred,
                workspace=workspace if defer else None)


def loss_fwd_bwd_upsampled(low, y, weights, mode: int, track_mode: int, grad_scale: float, want_grad: bool = True,
                           pred=None, workspace=None, out=None, dlow=None):
    """Run K2u on LOW-RES logits (B,C,h,w); labels / pred are at the full resolution of `y`."""
    _dev(low, y, weights, pred)
    low = _f32c(low)
    B, Cc, h, w = low.shape
    H, W = y.shape[-2:]
    if y.shape != (B, H, W) or not y.is_contiguous():
        raise SeaNativeError("labels must be a contiguous (B,H,W) tensor")
    dev = low.device
    L = lib()
    if workspace is None:
        nb = L.sea_loss_upsampled_workspace_bytes(B, Cc, h, w, H, W)
        if nb == 0:
            raise SeaNativeError("no K2u tiling for this shape")
        workspace = torch.empty(nb, dtype=torch.uint8, device=dev)
    if out is None:
        out = (torch.empty(B, dtype=torch.float32, device=dev), torch.empty(B, dtype=torch.float32, device=dev),
               torch.empty(B, dtype=torch.int32, device=dev))
    if want_grad and dlow is None:
        dlow = torch.empty_like(low)
    if not want_grad:
        dlow = None
    if weights is not None:
        weights = _f32c(weights)
    _check(L.sea_loss_fwd_bwd_upsampled(_p(low), _p(y), int_bytes(y), _p(weights), mode, track_mode, B, Cc, h, w, H, W,
                                        grad_scale, _p(dlow), _p(pred), 0 if pred is None else int_bytes(pred),
                                        _p(workspace), workspace.numel(), _p(out[0]), _p(out[1]), _p(out[2]),
                                        _stream()), "sea_loss_fwd_bwd_upsampled")
    return dict(dlogits=dlow, loss_sum=out[0], track_sum=out[1], n_correct=out[2], pred=pred)


# ------------------------------------------------------------------------------------------------ K3
def class_counts(pred, y, n_cls: int, per_image: bool = False, mask_pred: bool = True, out=None):
    """(inter, pred_cnt, tgt_cnt) int64; accumulates into `out` when given."""
    _dev(pred, y)
    B = pred.shape[0]
    HW = pred[0].numel()
    shape = (B, n_cls) if per_image else (n_cls,)
    if out is None:
        out = tuple(torch.zeros(shape, dtype=torch.int64, device=pred.device) for _ in range(3))
    if not (pred.is_contiguous() and y.is_contiguous()):
        raise SeaNativeError("pred / labels must be contiguous")
    _check(lib().sea_class_counts(_p(pred), int_bytes(pred), _p(y), int_bytes(y), B, n_cls, HW, int(mask_pred),
                                  int(per_image), _p(out[0]), _p(out[1]), _p(out[2]), _stream()), "sea_class_counts")
    return out


def confusion(pred, y, n_cls: int, hist=None):
    _dev(pred, y)
    if hist is None:
        hist = torch.zeros(n_cls, n_cls, dtype=torch.int64, device=pred.device)
    _check(lib().sea_confusion(_p(pred.contiguous()), int_bytes(pred), _p(y.contiguous()), int_bytes(y), pred.numel(),
                               n_cls, _p(hist), _stream()), "sea_confusion")
    return hist


def count_ignored(y, out=None):
    _dev(y)
    B = y.shape[0]
    out = torch.empty(B, dtype=torch.int32, device=y.device) if out is None else out
    _check(lib().sea_count_ignored(_p(y), int_bytes(y), B, y[0].numel(), _p(out), _stream()), "sea_count_ignored")
    return out


# ------------------------------------------------------------------------------------------------ K4/K7
def apgd_track(stats, n_ignored, HW: int, it: int, n_iter: int, check_k: int, early_stop: bool, init: bool, st):
    """`st` is the ApgdState of semseg.attacker (device buffers)."""
    _check(lib().sea_apgd_track(_p(stats["loss_sum"]), _p(stats["track_sum"]), _p(stats["n_correct"]),
                                _p(n_ignored), st.B, HW, it, n_iter, check_k, int(early_stop), int(init),
                                _p(st.acc_cnt), _p(st.acc), _p(st.loss_best), _p(st.loss_best_last),
                                _p(st.reduced_last), _p(st.step), _p(st.loss_steps), _p(st.flags), _p(st.done),
                                _p(stats.get("workspace")), _stream()), "sea_apgd_track")


def apgd_linf_step_graph(x, x_adv, x_old, grad, step_b, eps, iter_dev):
    """K1 in place with the loop index read from device memory (HIP-graph mode): x_old <- x_adv, x_adv <- new.
    ``eps``: a Python float, or ONE float32 in device memory (the radius as device state too)."""
    _dev(x, x_adv, x_old, grad, step_b, iter_dev)
    for t in (x, x_adv, x_old, grad):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise SeaNativeError("apgd_linf_step_graph: contiguous float32 buffers expected")
    if torch.is_tensor(eps):
        if eps.dtype != torch.float32 or eps.numel() != 1 or eps.device != x.device:
            raise SeaNativeError("apgd_linf_step_graph: a tensor eps must be one float32 on x's device")
        _check(lib().sea_apgd_linf_step_graph_dev(_p(x), _p(x_adv), _p(x_old), _p(grad), _p(_f32c(step_b)), _p(eps), _p(iter_dev),
                                                  x.shape[0], x[0].numel(), _stream()), "sea_apgd_linf_step_graph_dev")
        return
    _check(lib().sea_apgd_linf_step_graph(_p(x), _p(x_adv), _p(x_old), _p(grad), _p(_f32c(step_b)), eps, _p(iter_dev),
                                          x.shape[0], x[0].numel(), _stream()), "sea_apgd_linf_step_graph")


def apgd_track_graph(stats, n_ignored, HW: int, iter_dev, check_table, n_iter, early_stop: bool, st):
    """K7 with the loop index and the checkpoint schedule in device memory; advances ``iter_dev``.  ``n_iter``: an int, or ONE
    int32 in device memory (``check_table`` and ``st.loss_steps`` then sized for the longest run replayed)."""
    if torch.is_tensor(n_iter):
        if n_iter.dtype != torch.int32 or n_iter.numel() != 1:
            raise SeaNativeError("apgd_track_graph: a tensor n_iter must be one int32")
        _check(lib().sea_apgd_track_graph_dev(_p(stats["loss_sum"]), _p(stats["track_sum"]), _p(stats["n_correct"]),
                                              _p(n_ignored), st.B, HW, _p(iter_dev), _p(check_table), _p(n_iter), int(early_stop),
                                              _p(st.acc_cnt), _p(st.acc), _p(st.loss_best), _p(st.loss_best_last),
                                              _p(st.reduced_last), _p(st.step), _p(st.loss_steps), _p(st.flags), _p(st.done),
                                              _p(stats.get("workspace")), _stream()), "sea_apgd_track_graph_dev")
        return
    _check(lib().sea_apgd_track_graph(_p(stats["loss_sum"]), _p(stats["track_sum"]), _p(stats["n_correct"]),
                                      _p(n_ignored), st.B, HW, _p(iter_dev), _p(check_table), n_iter, int(early_stop),
                                      _p(st.acc_cnt), _p(st.acc), _p(st.loss_best), _p(st.loss_best_last),
                                      _p(st.reduced_last), _p(st.step), _p(st.loss_steps), _p(st.flags), _p(st.done),
                                      _p(stats.get("workspace")), _stream()), "sea_apgd_track_graph")


def select_copy(flags, x_adv, grad, x_best, grad_best, x_best_adv, pred=None, pred_best=None):
    B = x_adv.shape[0]
    _check(lib().sea_select_copy(_p(flags), _p(_f32c(x_adv)), _p(_f32c(grad)), _p(_f32c(x_best)),
                                 _p(_f32c(grad_best)), _p(_f32c(x_best_adv)), _p(pred), _p(pred_best),
                                 0 if pred is None else int_bytes(pred), B, x_adv[0].numel(),
                                 0 if pred is None else pred[0].numel(), _stream()), "sea_select_copy")


# ------------------------------------------------------------------------------------------------ M1
def dwconv7x7(x, weight, bias=None, flip: bool = False):
    """Depthwise 7x7 / pad 3 / stride 1 convolution (flip=True: backward-data of the same layer)."""
    _dev(x, weight, bias)
    B, Cc, H, W = x.shape
    x = _f32c(x)
    y = torch.empty_like(x)
    _check(lib().sea_dwconv7x7(_p(x), _p(_f32c(weight)), _p(bias), _p(y), B, Cc, H, W, int(flip), _stream()),
           "sea_dwconv7x7")
    return y


# ------------------------------------------------------------------------------------------------ M2
def upsample_bilinear(x, size):
    """F.interpolate(x, size, mode="bilinear", align_corners=False) for fp32 NCHW device tensors."""
    _dev(x)
    x = _f32c(x)
    B, Cc, h, w = x.shape
    H, W = int(size[0]), int(size[1])
    y = torch.empty(B, Cc, H, W, dtype=torch.float32, device=x.device)
    _check(lib().sea_upsample_bilinear_fwd(_p(x), _p(y), B * Cc, h, w, H, W, _stream()), "sea_upsample_bilinear_fwd")
    return y


def upsample_bilinear_backward(gy, in_size):
    _dev(gy)
    gy = _f32c(gy)
    B, Cc, H, W = gy.shape
    h, w = int(in_size[0]), int(in_size[1])
    gx = torch.empty(B, Cc, h, w, dtype=torch.float32, device=gy.device)
    _check(lib().sea_upsample_bilinear_bwd(_p(gy), _p(gx), B * Cc, h, w, H, W, _stream()), "sea_upsample_bilinear_bwd")
    return gx


def _is_cl(t) -> bool:
    """dense channels_last (and not simultaneously NCHW-contiguous) with C % 4 == 0"""
    return (t.dim() == 4 and t.shape[1] % 4 == 0 and not t.is_contiguous()
            and t.is_contiguous(memory_format=torch.channels_last))


def cl_pixel_stride(t):
    """Floats between consecutive pixels if `t` (B,C,H,W) is an NHWC tensor or a channel slice of one
    (strides (H*W*S, 1, W*S, S) with S >= C, S % 4 == 0, C % 4 == 0, 16-byte aligned), else None."""
    if t.dim() != 4 or t.dtype != torch.float32:
        return None
    B, Cc, H, W = t.shape
    sb, sc, sh, sw = t.stride()
    S = sw if W > 1 else (sh if H > 1 else (sb if B > 1 else Cc))
    ok = (Cc % 4 == 0 and S % 4 == 0 and S >= Cc and (sc == 1 or Cc == 1) and (W == 1 or sw == S)
          and (H == 1 or sh == W * S) and (B == 1 or sb == H * W * S) and t.data_ptr() % 16 == 0)
    return S if ok else None


def upsample_bilinear_cl(x, size, out=None, residual=None):
    """Same op on a channels_last (B,C,h,w) tensor; returns a channels_last (B,C,H,W) tensor.  `out` may be a
    channel slice of a wider channels_last tensor (written in place); `residual` (dense channels_last,
    (B,C,H,W)) is added to the result."""
    _dev(x, out, residual)
    if x.dtype != torch.float32 or cl_pixel_stride(x) != x.shape[1]:
        raise SeaNativeError("expected a dense channels_last float32 tensor with C % 4 == 0")
    B, Cc, h, w = x.shape
    H, W = int(size[0]), int(size[1])
    if out is None:
        out = torch.empty(B, Cc, H, W, dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    S = cl_pixel_stride(out)
    if tuple(out.shape) != (B, Cc, H, W) or S is None:
        raise SeaNativeError("out must be a (B,C,H,W) channels_last tensor or a channel slice of one")
    if residual is not None and (tuple(residual.shape) != (B, Cc, H, W) or cl_pixel_stride(residual) != Cc):
        raise SeaNativeError("residual must be a dense channels_last (B,C,H,W) float32 tensor")
    _check(lib().sea_upsample_bilinear_nhwc_fwd(_p(x), _p(residual), _p(out), B, Cc, h, w, H, W, S, _stream()),
           "sea_upsample_bilinear_nhwc_fwd")
    return out


def upsample_bilinear_backward_cl(gy, in_size):
    """Gradient w.r.t. the low-resolution input; gy is channels_last or a channel slice of such a tensor."""
    _dev(gy)
    S = cl_pixel_stride(gy)
    if S is None:
        raise SeaNativeError("expected a channels_last float32 tensor (or channel slice) with C % 4 == 0")
    B, Cc, H, W = gy.shape
    h, w = int(in_size[0]), int(in_size[1])
    gx = torch.empty(B, Cc, h, w, dtype=torch.float32, device=gy.device, memory_format=torch.channels_last)
    _check(lib().sea_upsample_bilinear_nhwc_bwd(_p(gy), _p(gx), B, Cc, h, w, H, W, S, _stream()),
           "sea_upsample_bilinear_nhwc_bwd")
    return gx


# ------------------------------------------------------------------------------------------------ M6
def tap_gather(G, size, extra=None):
    """G (B,h,w,9,C) coarse per-tap maps -> (B,C,H,W) channels_last: sum over the 3x3 taps of the shifted
    bilinear up-samplings (added to ``extra`` in place when given)."""
    _dev(G, extra)
    B, h, w, nine, Cc = G.shape
    H, W = int(size[0]), int(size[1])
    if nine != 9:
        raise SeaNativeError("tap_gather: G must be (B,h,w,9,C)")
    acc = extra is not None
    if extra is None:
        extra = torch.empty(B, Cc, H, W, dtype=torch.float32, device=G.device, memory_format=torch.channels_last)
    elif tuple(extra.shape) != (B, Cc, H, W) or cl_pixel_stride(extra) != Cc:
        raise SeaNativeError("tap_gather: extra must be a dense channels_last (B,C,H,W) float32 tensor")
    _check(lib().sea_tap_gather_fwd(_p(_f32c(G)), _p(extra), int(acc), B, Cc, h, w, H, W, _stream()), "sea_tap_gather_fwd")
    return extra


def tap_gather_backward(gz, coarse_size):
    """gz dense channels_last (B,C,H,W) -> dG (B,h,w,9,C): the adjoint of tap_gather."""
    _dev(gz)
    B, Cc, H, W = gz.shape
    if cl_pixel_stride(gz) != Cc:
        raise SeaNativeError("tap_gather_backward: dense channels_last float32 gradient expected")
    h, w = int(coarse_size[0]), int(coarse_size[1])
    dG = torch.empty(B, h, w, 9, Cc, dtype=torch.float32, device=gz.device)
    _check(lib().sea_tap_gather_bwd(_p(gz), _p(dG), B, Cc, h, w, H, W, _stream()), "sea_tap_gather_bwd")
    return dG


def gate_scale(g, gate, scale):
    """gate > 0 ? g * scale[c] : 0 for dense channels_last (B,C,H,W) tensors."""
    _dev(g, gate, scale)
    B, Cc, H, W = g.shape
    if cl_pixel_stride(g) != Cc or cl_pixel_stride(gate) != Cc or gate.shape != g.shape or scale.numel() != Cc:
        raise SeaNativeError("gate_scale: dense channels_last float32 tensors of one shape expected")
    out = torch.empty_like(g)
    _check(lib().sea_gate_scale(_p(g), _p(gate), _p(_f32c(scale)), _p(out), B * H * W, Cc, _stream()), "sea_gate_scale")
    return out


# ------------------------------------------------------------------------------------------------ M5
def layernorm(x, weight, bias, eps: float):
    """LayerNorm over the last dim of a contiguous fp32 tensor; returns (y, mean, rstd)."""
    _dev(x, weight, bias)
    x = _f32c(x)
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    _check(lib().sea_layernorm_fwd(_p(x), _p(_f32c(weight)), _p(_f32c(bias)), _p(y), _p(mean), _p(rstd), rows, Cc,
                                   float(eps), _stream()), "sea_layernorm_fwd")
    return y, mean, rstd


def layernorm_backward(g, x, weight, mean, rstd):
    _dev(g, x, weight, mean, rstd)
    Cc = x.shape[-1]
    dx = torch.empty_like(x)
    _check(lib().sea_layernorm_bwd(_p(_f32c(g)), _p(_f32c(x)), _p(_f32c(weight)), _p(mean), _p(rstd), _p(dx),
                                   x.numel() // Cc, Cc, _stream()), "sea_layernorm_bwd")
    return dx


# ------------------------------------------------------------------------------------------------ M2''
def adaptive_avg_pool_nhwc(x, oh: int, ow: int):
    """adaptive average pooling (ATen's bins) of a dense channels_last (B,C,H,W) fp32 tensor -> channels_last (B,C,oh,ow)"""
    _dev(x)
    if x.dtype != torch.float32 or cl_pixel_stride(x) != x.shape[1]:
        raise SeaNativeError("adaptive_avg_pool_nhwc: dense channels_last float32 (B,C,H,W) with C % 4 == 0 expected")
    B, Cc, H, W = x.shape
    out = torch.empty((B, oh, ow, Cc), dtype=torch.float32, device=x.device).permute(0, 3, 1, 2)
    _check(lib().sea_adaptive_avg_pool_nhwc_fwd(_p(x), _p(out), B, Cc, H, W, oh, ow, _stream()), "sea_adaptive_avg_pool_nhwc_fwd")
    return out


def adaptive_avg_pool_nhwc_backward(g, H: int, W: int):
    """input gradient (channels_last (B,C,H,W)) of ``adaptive_avg_pool_nhwc`` given g (B,C,oh,ow)"""
    _dev(g)
    B, Cc, oh, ow = g.shape
    gn = g.permute(0, 2, 3, 1)
    if g.dtype != torch.float32:
        raise SeaNativeError("adaptive_avg_pool_nhwc_backward: float32 expected")
    if not gn.is_contiguous():
        gn = gn.contiguous()
    dx = torch.empty((B, H, W, Cc), dtype=torch.float32, device=g.device)
    _check(lib().sea_adaptive_avg_pool_nhwc_bwd(_p(gn), _p(dx), B, Cc, H, W, oh, ow, _stream()), "sea_adaptive_avg_pool_nhwc_bwd")
    return dx.permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------------------------ M9
STEM_CONV1_CHANNELS = (48,)
LN_GELU_CL_CHANNELS = (48, 96)


def _nhwc_dense(t):
    """(B,C,H,W) tensor whose memory is dense NHWC (a 1 x 1 map, which is NCHW-contiguous as well, included)"""
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)


def _empty_cl(B, Cc, H, W, device):
    return torch.empty((B, H, W, Cc), dtype=torch.float32, device=device).permute(0, 3, 1, 2)


def stem_conv1_ln_gelu(x, weight, bias, gamma=None, beta=None, eps: float = 1e-6):
    """x (B,3,H,W) NCHW -> (y, a): y = conv2d(x, weight (48,3,3,3), bias, stride 2, padding 1) and a = GELU(LayerNorm over the
    channels of y) -- a is None when gamma is None (the convolution alone).  y and a are (B,48,Ho,Wo) tensors in
    channels_last memory."""
    _dev(x, weight, bias, gamma, beta)
    if x.dim() != 4 or x.shape[1] != 3 or tuple(weight.shape[1:]) != (3, 3, 3) or weight.shape[0] not in STEM_CONV1_CHANNELS:
        raise SeaNativeError("stem_conv1_ln_gelu: x (B,3,H,W) and weight (48,3,3,3) expected")
    x = _f32c(x)
    B, _, H, W = x.shape
    CO = weight.shape[0]
    y = _empty_cl(B, CO, (H - 1) // 2 + 1, (W - 1) // 2 + 1, x.device)
    a = torch.empty_like(y) if gamma is not None else None     # (preserves the channels_last strides)
    _check(lib().sea_stem_conv1_ln_gelu(_p(x), _p(_f32c(weight)), _p(_f32c(bias)) if bias is not None else None,
                                        _p(_f32c(gamma)) if gamma is not None else None,
                                        _p(_f32c(beta)) if gamma is not None else None, _p(y),
                                        _p(a) if a is not None else None, B, CO, H, W, float(eps), _stream()),
           "sea_stem_conv1_ln_gelu")
    return y, a


def stem_conv1_backward(dy, weight, H: int, W: int):
    """input gradient (B,3,H,W) NCHW of the stride-2 3x3 stem convolution given dy (B,48,Ho,Wo) in channels_last memory"""
    _dev(dy, weight)
    B, CO, Ho, Wo = dy.shape
    if (Ho, Wo) != ((H - 1) // 2 + 1, (W - 1) // 2 + 1) or CO not in STEM_CONV1_CHANNELS:
        raise SeaNativeError("stem_conv1_backward: dy does not belong to an input of this size")
    if dy.dtype != torch.float32 or not _nhwc_dense(dy):
        raise SeaNativeError("stem_conv1_backward: dy must be float32 in channels_last memory")
    dx = torch.empty((B, 3, H, W), dtype=torch.float32, device=dy.device)
    _check(lib().sea_stem_conv1_bwd(_p(dy), _p(_f32c(weight)), _p(dx), B, CO, H, W, _stream()), "sea_stem_conv1_bwd")
    return dx


def ln_gelu_cl(y, gamma, beta, eps: float = 1e-6, out_nchw: bool = False):
    """GELU(LayerNorm over dim 1) of a (B,C,H,W) fp32 tensor in channels_last memory, C in LN_GELU_CL_CHANNELS; the result is
    NCHW-contiguous when ``out_nchw`` else channels_last"""
    _dev(y, gamma, beta)
    if y.dtype != torch.float32 or not _nhwc_dense(y) or y.shape[1] not in LN_GELU_CL_CHANNELS:
        raise SeaNativeError("ln_gelu_cl: float32 (B,C,H,W) in channels_last memory with C in (48, 96) expected")
    B, Cc, H, W = y.shape
    a = torch.empty((B, Cc, H, W), dtype=torch.float32, device=y.device) if out_nchw else torch.empty_like(y)
    _check(lib().sea_ln_gelu_cl_fwd(_p(y), _p(_f32c(gamma)), _p(_f32c(beta)), _p(a), int(out_nchw), B, Cc, H * W, float(eps),
                                    _stream()), "sea_ln_gelu_cl_fwd")
    return a


def ln_gelu_cl_backward(da, y, gamma, beta, eps: float = 1e-6):
    """d loss / d y (channels_last) of ``ln_gelu_cl`` (gamma, beta frozen); da is NCHW-contiguous or channels_last; the
    statistics are recomputed from y"""
    _dev(da, y, gamma, beta)
    if da.dtype != torch.float32 or da.shape != y.shape or not _nhwc_dense(y):
        raise SeaNativeError("ln_gelu_cl_backward: da float32 of y's shape, y in channels_last memory")
    B, Cc, H, W = y.shape
    # (a tensor with H = W = 1 or C = 1 is both: either reading is the same memory)
    nchw = da.is_contiguous()
    if not nchw and not _nhwc_dense(da):
        raise SeaNativeError("ln_gelu_cl_backward: da must be NCHW-contiguous or channels_last")
    dy = torch.empty_like(y)
    _check(lib().sea_ln_gelu_cl_bwd(_p(da), int(nchw), _p(y), _p(_f32c(gamma)), _p(_f32c(beta)), _p(dy), B, Cc, H * W,
                                    float(eps), _stream()), "sea_ln_gelu_cl_bwd")
    return dy


# ------------------------------------------------------------------------------------------------ M4
def wino_filter(weight, m: int, flip: bool):
    """(Cout,Cin,3,3) -> Winograd-domain filters: (A*A, Cin, Cout), or (A*A, Cout, Cin) rotated when flip."""
    _dev(weight)
    Cout, Cin, kh, kw = weight.shape
    if (kh, kw) != (3, 3):
        raise SeaNativeError("Winograd path is for 3x3 filters")
    A = m + 2
    U = torch.empty((A * A, Cout, Cin) if flip else (A * A, Cin, Cout), dtype=torch.float32, device=weight.device)
    _check(lib().sea_wino_filter_transform(_p(_f32c(weight)), _p(U), Cout, Cin, m, int(flip), _stream()),
           "sea_wino_filter_transform")
    return U


# fewest Winograd tiles PER IMAGE (rows of each of the 36 / 16 Winograd-domain products, divided by the batch) that go through
# M8; below: hipBLASLt's batched fp32 GEMM.  Per image, not in total: which arithmetic an image gets must not depend on how
# many partners its batch has (a sharded evaluation equals the unsharded one bit for bit).  16 = a 16 x 16 map at F(4,3):
# the PSP bottleneck (2816 -> 512) of a 512 x 512 input, 145 + 164 us on the library against ~85 + ~85 us here.
WINO_SPLIT_MIN_TILES = int(os.environ.get("SEA_WINO_SPLIT_MIN_TILES", "16"))


def wino_conv3x3_cl(x, U, m: int, bias=None, scale=None, relu: bool = False, gate=None, gate_scale=None,
                    addend=None, gemm_terms: int = 0):
    """3x3 / stride 1 / pad 1 convolution of a channels_last (B,Cin,H,W) tensor (or channel slice) -- or of the
    channel concatenation of a list of such tensors, which is never materialised -- with Winograd-domain filters
    U (A*A, Cin, Cout); returns channels_last (B,Cout,H,W) = act(scale[c] * (conv + addend) + bias[c]).
    ``gate`` (same shape as x) / ``gate_scale``: the input is read as gate > 0 ? x * gate_scale[c] : 0.
    ``addend`` (dense channels_last (B,Cout,H,W)) is added to the convolution before scale / bias / act."""
    xs = list(x) if isinstance(x, (list, tuple)) else [x]
    _dev(*xs, U, bias, scale, gate, gate_scale, addend)
    B, _, H, W = xs[0].shape
    Cin = sum(t.shape[1] for t in xs)
    A2, Ci, Cout = U.shape
    strides = [cl_pixel_stride(t) for t in xs]  # channel slices of wider channels_last tensors are read in place
    if (any(p is None for p in strides) or any(tuple(t.shape[2:]) != (H, W) or t.shape[0] != B for t in xs) or Ci != Cin
            or A2 != (m + 2) ** 2 or Cout % 4):
        raise SeaNativeError("wino_conv3x3_cl: channels_last float32 inputs (or channel slices) and matching filters expected")
    if gate is not None and (len(xs) != 1 or gate.shape != xs[0].shape or cl_pixel_stride(gate) != Cin):
        raise SeaNativeError("wino_conv3x3_cl: gate must match the (single) input's shape and layout")
    for v, n in ((bias, Cout), (scale, Cout), (gate_scale, Cin)):
        if v is not None and (v.dtype != torch.float32 or v.numel() != n or not v.is_contiguous()):
            raise SeaNativeError("wino_conv3x3_cl: per-channel vectors must be contiguous float32 of the channel count")
    L = lib()
    T = L.sea_wino_tiles(B, H, W, m)
    V = torch.empty(A2, T, Cin, dtype=torch.float32, device=xs[0].device)
    off = 0
    use_split = gemm_terms in (1, 2, 3, 22) and Cin % 32 == 0 and T // B >= WINO_SPLIT_MIN_TILES
    # fp16 x 2: one scale word per tile (= per row of the Winograd-domain GEMMs), filled by the transform itself
    v_amax = (torch.zeros(T, dtype=torch.int32, device=xs[0].device)
              if (use_split and gemm_terms == 22 and AMAX_FROM_PRODUCERS) else None)
    for t, xps in zip(xs, strides):
        if v_amax is not None:
            _check(L.sea_wino_input_transform_amax(_p(t), xps, _p(gate), _p(gate_scale), V.data_ptr() + 4 * off, Cin, B,
                                                   t.shape[1], H, W, m, _p(v_amax), _stream()), "sea_wino_input_transform_amax")
        else:
            _check(L.sea_wino_input_transform(_p(t), xps, _p(gate), _p(gate_scale), V.data_ptr() + 4 * off, Cin, B, t.shape[1],
                                              H, W, m, _stream()), "sea_wino_input_transform")
        off += t.shape[1]
    if use_split:
        # M8: the (A*A) Winograd-domain products on the bf16 matrix cores (operands split into bf16 terms, fp32 accumulate)
        # U is the cached, frozen filter image: packed once per term count and per CONTENT (a caller that refreshes U in place
        # -- fixed addresses for captured graphs -- bumps its version: the packed image is then refreshed in place too)
        packed = U.__dict__.setdefault("_sea_packed", {})
        Up, ver = packed.get(gemm_terms, (None, None))
        if Up is None or ver != U._version:
            fresh = gemm_split_pack(U, trans=True, terms=gemm_terms)
            if Up is None or not Up.refresh_from(fresh):
                Up = fresh
                CACHE_EPOCH[0] += 1
            packed[gemm_terms] = (Up, U._version)
        Mx = gemm_split(V, Up, amax=v_amax, amax_rows=1 if v_amax is not None else 0, groups=B)
    else:
        with torch.autocast("cuda", enabled=False):
            Mx = _f32c(torch.bmm(V, U))  # (A*A) independent fp32 GEMMs: hipBLASLt strided-batched
    del V
    y = torch.empty(B, Cout, H, W, dtype=torch.float32, device=xs[0].device, memory_format=torch.channels_last)
    if addend is not None and (tuple(addend.shape) != (B, Cout, H, W) or cl_pixel_stride(addend) != Cout):
        raise SeaNativeError("wino_conv3x3_cl: addend must be a dense channels_last (B,Cout,H,W) float32 tensor")
    _check(L.sea_wino_output_transform(_p(Mx), _p(addend), _p(scale), _p(bias), int(relu), _p(y), B, Cout, H, W, m,
                                       _stream()), "sea_wino_output_transform")
    return y


def dwconv7x7_nhwc(x, wt, bias=None, flip: bool = False, addend=None):
    """Depthwise 7x7 on a (B,H,W,C) contiguous tensor; wt is the (49,C) taps-major filter bank; ``addend`` (shape of
    the result) is added after the taps (the skip gradient in the backward of a residual block)."""
    _dev(x, wt, bias, addend)
    B, H, W, Cc = x.shape
    if addend is not None and (addend.shape != x.shape or addend.dtype != torch.float32 or not addend.is_contiguous()):
        raise SeaNativeError("dwconv7x7_nhwc: addend must be a contiguous float32 tensor of the input's shape")
    y = torch.empty_like(x)
    _check(lib().sea_dwconv7x7_nhwc_add(_p(_f32c(x)), _p(_f32c(wt)), _p(bias), _p(addend), _p(y), B, Cc, H, W, int(flip),
                                        _stream()), "sea_dwconv7x7_nhwc_add")
    return y


def dwconv7x7_nhwc_weight_grad(x, gy, want_bias: bool = True):
    """(gw (C,1,7,7), gb (C) or None): weight / bias gradient of the depthwise 7x7 on (B,H,W,C) contiguous fp32 tensors"""
    _dev(x, gy)
    B, H, W, Cc = x.shape
    if gy.shape != x.shape or Cc % 4:
        raise SeaNativeError("dwconv7x7_nhwc_weight_grad: x and gy (B,H,W,C) of one shape, C % 4 == 0")
    L = lib()
    ws = torch.empty(int(L.sea_dwconv7x7_nhwc_wgrad_workspace(B, Cc, H)), dtype=torch.float32, device=x.device)
    gw = torch.empty((Cc, 1, 7, 7), dtype=torch.float32, device=x.device)
    gb = torch.empty(Cc, dtype=torch.float32, device=x.device) if want_bias else None
    _check(L.sea_dwconv7x7_nhwc_wgrad(_p(_f32c(x)), _p(_f32c(gy)), _p(gw), _p(gb), _p(ws), B, Cc, H, W, _stream()),
           "sea_dwconv7x7_nhwc_wgrad")
    return gw, gb


# ------------------------------------------------------------------------------------------------ M3
def nchw_to_nhwc(x, scale=None):
    """(B,C,H,W) contiguous -> (B,H,W,C) contiguous, optionally times scale[c]."""
    _dev(x, scale)
    B, Cc, H, W = x.shape
    out = torch.empty(B, H, W, Cc, dtype=torch.float32, device=x.device)
    _check(lib().sea_nchw_to_nhwc(_p(_f32c(x)), _p(scale), _p(out), B, Cc, H * W, _stream()), "sea_nchw_to_nhwc")
    return out


def nhwc_to_nchw(y, scale=None, residual=None):
    """(B,H,W,C) contiguous -> (B,C,H,W) contiguous: residual + scale[c] * y^T."""
    _dev(y, scale, residual)
    B, H, W, Cc = y.shape
    out = torch.empty(B, Cc, H, W, dtype=torch.float32, device=y.device)
    _check(lib().sea_nhwc_to_nchw(_p(_f32c(y)), _p(scale), _p(None if residual is None else _f32c(residual)), _p(out),
                                  B, Cc, H * W, _stream()), "sea_nhwc_to_nchw")
    return out


def patch2x2(x_nhwc):
    """(B,H,W,C) contiguous fp32 -> (B*H/2*W/2, 4*C) rows of 2x2 patches in (di, dj, c) order"""
    _dev(x_nhwc)
    x_nhwc = _f32c(x_nhwc)
    B, H, W, Cc = x_nhwc.shape
    out = torch.empty(B * (H // 2) * (W // 2), 4 * Cc, dtype=torch.float32, device=x_nhwc.device)
    _check(lib().sea_patch2x2(_p(x_nhwc), _p(out), B, H, W, Cc, 0, _stream()), "sea_patch2x2")
    return out


def unpatch2x2(rows, B, H, W):
    """inverse of patch2x2: (B*H/2*W/2, 4*C) -> (B,H,W,C)"""
    _dev(rows)
    rows = _f32c(rows)
    Cc = rows.shape[1] // 4
    out = torch.empty(B, H, W, Cc, dtype=torch.float32, device=rows.device)
    _check(lib().sea_patch2x2(_p(rows), _p(out), B, H, W, Cc, 1, _stream()), "sea_patch2x2")
    return out


# ------------------------------------------------------------------------------------------------ M7
def attention_qkv(qkv, scale: float, terms=None):
    """softmax(q k^T * scale) v for a packed (B,T,3,H,64) fp32 qkv tensor; returns (out (B,T,H*64), lse (B,H,T)).
    ``terms``: arithmetic of the products (22 = fp16 x 2, 3 / 2 = bf16 terms, 0 = fp32 MFMA); None = ``attn_terms_fwd()``."""
    _dev(qkv)
    qkv = _f32c(qkv)
    B, T, three, H, D = qkv.shape
    if three != 3 or D != 64:
        raise SeaNativeError("attention_qkv expects a (B,T,3,H,64) tensor")
    out = torch.empty(B, T, H * D, dtype=torch.float32, device=qkv.device)
    lse = torch.empty(B, H, T, dtype=torch.float32, device=qkv.device)
    p = qkv.data_ptr()
    terms = attn_terms_fwd() if terms is None else int(terms)
    if terms == 22:
        ws = torch.empty(4 * B * H, dtype=torch.int32, device=qkv.device)
        _check(lib().sea_attention_fwd_f16(p, p + 4 * H * D, p + 8 * H * D, T * 3 * H * D, D, 3 * H * D, B, H, T, D, float(scale),
                                           _p(ws), _p(out), _p(lse), _stream()), "sea_attention_fwd_f16")
        return out, lse
    _check(lib().sea_attention_fwd_terms(p, p + 4 * H * D, p + 8 * H * D, T * 3 * H * D, D, 3 * H * D, B, H, T, D, float(scale),
                                         _p(out), _p(lse), terms, _stream()), "sea_attention_fwd_terms")
    return out, lse


ATTN_TERMS_BWD_DEFAULT = 22


def attn_terms_bwd() -> int:
    """Arithmetic of the attention backward products, the ONE place Python callers read it: env SEA_ATTN_TERMS_BWD (looked up per
    call: tests switch it in-process) or 22.  22 = fp16 x 2 (22 significant bits per operand, three MFMA products per pair);
    3 / 2 = bf16 terms per operand (six / three products); 0 = the fp32 MFMA kernels."""
    t = int(os.environ.get("SEA_ATTN_TERMS_BWD", ATTN_TERMS_BWD_DEFAULT))
    return t if t in (0, 2, 3, 22) else ATTN_TERMS_BWD_DEFAULT


ATTN_TERMS_FWD_DEFAULT = 22


def attn_terms_fwd() -> int:
    """forward products: env SEA_ATTN_TERMS (per call) or 22 = fp16 x 2; 3 / 2 = bf16 terms per operand; 0 = fp32 MFMA"""
    t = int(os.environ.get("SEA_ATTN_TERMS", ATTN_TERMS_FWD_DEFAULT))
    return t if t in (0, 2, 3, 22) else ATTN_TERMS_FWD_DEFAULT


def attention_qkv_backward(qkv, out, lse, grad_out, scale: float, terms=None):
    """gradient w.r.t. the packed qkv tensor.  ``terms``: arithmetic of the backward products (22 = fp16 x 2, 3 / 2 = bf16 terms,
    0 = fp32 MFMA); None = ``attn_terms_bwd()``."""
    _dev(qkv, out, lse, grad_out)
    B, T, _, H, D = qkv.shape
    grad_out = _f32c(grad_out.contiguous())
    dqkv = torch.empty_like(qkv)
    delta = torch.empty_like(lse)
    p, g = qkv.data_ptr(), dqkv.data_ptr()
    sb, sh, st = T * 3 * H * D, D, 3 * H * D
    terms = attn_terms_bwd() if terms is None else int(terms)
    head = (p, p + 4 * H * D, p + 8 * H * D, sb, sh, st, B, H, T, D, float(scale), _p(_f32c(out)), _p(grad_out),
            _p(_f32c(lse)), _p(delta))
    tail = (g, g + 4 * H * D, g + 8 * H * D, sb, sh, st)
    if terms == 22:
        ws = torch.empty(4 * B * H, dtype=torch.int32, device=qkv.device)     # max |q|, |k|, |v|, |dO| per (image, head)
        _check(lib().sea_attention_bwd_f16(*head, _p(ws), *tail, _stream()), "sea_attention_bwd_f16")
    else:
        _check(lib().sea_attention_bwd_terms(*head, *tail, terms, _stream()), "sea_attention_bwd_terms")
    return dqkv


# ------------------------------------------------------------------------------------------------ K9 (host)
def worst_miou_greedy(ints: torch.Tensor, unions: torch.Tensor, mt_state, n_rounds: int = 1000):
    """ints/unions: CPU float32 (A,N,C).  mt_state: 625 ints (random.getstate()[1]).
    Returns (miou, selected list, rounds, new mt_state tuple)."""
    import numpy as np
    A, N, Cc = ints.shape
    ti = np.ascontiguousarray(ints.detach().cpu().numpy(), dtype=np.float32)
    tu = np.ascontiguousarray(unions.detach().cpu().numpy(), dtype=np.float32)
    mt = np.array(mt_state, dtype=np.uint32)
    assert mt.shape == (625,)
    sel = np.zeros(N, dtype=np.int32)
    miou = C.c_double(0.0)
    rounds = C.c_int32(0)
    rc = lib().sea_worst_miou_greedy(ti.ctypes.data, tu.ctypes.data, A, N, Cc, mt.ctypes.data, n_rounds,
                                     C.addressof(miou), sel.ctypes.data, C.addressof(rounds))
    _check(rc, "sea_worst_miou_greedy")
    return miou.value, sel.tolist(), rounds.value, tuple(int(v) for v in mt)


# ------------------------------------------------------------------------------------------------ M8
# Bumped whenever a weight-derived cache tensor is (re)created at a NEW device address (first use, or a shape change): a
# caller that captured a HIP graph over the model compares it with the value it saw at capture time.
CACHE_EPOCH = [0]


class PackedWeight:
    """Frozen weights W (N x K) pre-split into `terms` bf16 images in the tile order of sea_gemm_split
    (optionally a batch of them, e.g. the (m+2)^2 Winograd-domain filters)."""

    def __init__(self, data, N, K, terms, batch, src=None):
        self.data, self.N, self.K, self.terms, self.batch = data, N, K, terms, batch
        self.stride = data.numel() // batch
        self.src = src          # (W, trans) of a single matrix: K slices are packed from it on demand (split-K)
        self.slices = {}

    def _slices_of(self, S):
        W, trans = self.src
        Ws = W.view(S, self.K // S, self.N) if trans else W.view(self.N, S, self.K // S).permute(1, 0, 2)
        return gemm_split_pack(Ws, trans=trans, terms=self.terms)

    def k_slices(self, S):
        """this weight as a batch of S packed (N x K/S) matrices, one per K slice"""
        if S not in self.slices:
            self.slices[S] = self._slices_of(S)
            CACHE_EPOCH[0] += 1          # a new device address that a captured graph cannot know about
        return self.slices[S]

    def refresh_from(self, new) -> bool:
        """take the contents of ``new`` (the same weight, packed again after it changed) INTO this object's buffers: the
        packed image -- and every K-slice image derived from it -- keeps its device address, which is what lets a captured
        HIP graph outlive a weight update (PIR-AT: the weights change every outer step).  False if the shapes differ."""
        if (self.data.shape != new.data.shape or (self.N, self.K, self.terms, self.batch) != (new.N, new.K, new.terms, new.batch)
                or (self.src is None) != (new.src is None)):
            return False
        self.data.copy_(new.data)
        self.src = new.src
        for S, sl in self.slices.items():
            sl.data.copy_(self._slices_of(S).data)
        return True


def gemm_split_pack(W, trans: bool = False, terms: int = 3) -> PackedWeight:
    """W: (N, K) fp32 (or (K, N) with ``trans``), or a batch (G, N, K) / (G, K, N) of them; last dim contiguous."""
    _dev(W)
    if W.dtype != torch.float32 or W.dim() not in (2, 3) or W.stride(-1) != 1:
        raise SeaNativeError("gemm_split_pack: float32 (N,K) / (G,N,K) weights with a contiguous last dim expected")
    Wb = W if W.dim() == 3 else W.unsqueeze(0)
    G = Wb.shape[0]
    N, K = (Wb.shape[2], Wb.shape[1]) if trans else (Wb.shape[1], Wb.shape[2])
    L = lib()
    nbytes = L.sea_gemm_split_packed_bytes(N, K, terms)
    if nbytes < 0:
        raise SeaNativeError(f"gemm_split_pack: unsupported shape N={N} K={K} terms={terms} (K % 32 == 0, terms 1, 2, 3 or 22)")
    out = torch.empty(G, nbytes, dtype=torch.uint8, device=W.device)
    for g in range(G):
        _check(L.sea_gemm_split_pack(_p(Wb[g]), Wb.stride(1), int(trans), N, K, terms, _p(out[g]), _stream()),
               "sea_gemm_split_pack")
    return PackedWeight(out, N, K, terms, G, src=(W.detach(), trans) if (W.dim() == 2 and W.is_contiguous()) else None)


# ------------------------------------------------------------------------------------------------ M8f
USE_MLP_FUSED = os.environ.get("SEA_MLP_FUSED", "1") != "0"
USE_CLASSIFIER = os.environ.get("SEA_CLASSIFIER", "1") != "0"


def classifier_ok(P: int, K: int, cls: int) -> bool:
    """the head's 1 x 1 classifier has its own kernels for this shape (M10: at most 32 classes)"""
    return bool(USE_CLASSIFIER and lib().sea_classifier_supported(int(P), int(K), int(cls)))


def classifier_forward(y_rows, w2d, bias, B: int, P: int):
    """logits (B, cls, P) NCHW = W (cls, K) . y^T for y_rows (B P, K) fp32 NHWC rows (sea_classifier_fwd)"""
    _dev(y_rows, w2d, bias)
    cls, K = w2d.shape
    if (y_rows.dtype != torch.float32 or y_rows.shape != (B * P, K) or not y_rows.is_contiguous() or y_rows.data_ptr() % 16
            or w2d.dtype != torch.float32 or not w2d.is_contiguous() or w2d.data_ptr() % 16
            or (bias is not None and (bias.dtype != torch.float32 or bias.numel() != cls or not bias.is_contiguous()))):
        raise SeaNativeError("classifier_forward: y (B P, K) / W (cls, K) must be contiguous, 16-byte aligned float32")
    out = torch.empty((B, cls, P), dtype=torch.float32, device=y_rows.device)
    _check(lib().sea_classifier_fwd(_p(y_rows), _p(w2d), _p(bias), _p(out), B, P, K, cls, _stream()), "sea_classifier_fwd")
    return out


def classifier_backward(g, w2d, gate=None, gate_scale=None):
    """gy (B P, K) NHWC rows = g^T (B, P, cls) . W (cls, K) for the NCHW logit gradient g (B, cls, P) (sea_classifier_bwd).
    ``gate`` (B P, K) with ``gate_scale`` (K): the result is (gate > 0 ? gy * gate_scale : 0) -- ``gate_scale`` of the
    gradient, applied on the way out."""
    _dev(g, w2d, gate, gate_scale)
    cls, K = w2d.shape
    if (gate is None) != (gate_scale is None):
        raise SeaNativeError("classifier_backward: gate and gate_scale come together")
    if gate is not None and (gate.dtype != torch.float32 or gate.dim() != 2 or gate.shape[1] != K or not gate.is_contiguous()
                             or gate_scale.dtype != torch.float32 or gate_scale.numel() != K or not gate_scale.is_contiguous()
                             or g.dim() != 3 or gate.shape[0] != g.shape[0] * g.shape[2]):
        raise SeaNativeError("classifier_backward: gate must be contiguous float32 (B P, K) rows, gate_scale (K)")
    if (g.dtype != torch.float32 or g.dim() != 3 or g.shape[1] != cls or not g.is_contiguous()
            or w2d.dtype != torch.float32 or not w2d.is_contiguous() or w2d.data_ptr() % 16):
        raise SeaNativeError("classifier_backward: g (B, cls, P) / W (cls, K) must be contiguous float32")
    B, _, P = g.shape
    gy = torch.empty((B * P, K), dtype=torch.float32, device=g.device)
    _check(lib().sea_classifier_bwd(_p(g), _p(w2d), _p(gy), B, P, K, cls, _p(gate), _p(gate_scale), _stream()), "sea_classifier_bwd")
    return gy


def mlp_fused_ok(C: int, H: int) -> bool:
    """the fused-MLP kernels exist for this block width (ConvNeXt stages of 96 / 192 channels, hidden = 4 C)"""
    return bool(USE_MLP_FUSED and lib().sea_mlp_fused_supported(int(C), int(H)))


def _mlp_rows(t, C, what):
    if t.dtype != torch.float32 or t.dim() != 2 or t.shape[1] != C or t.stride(1) != 1 or t.stride(0) % 4 or t.data_ptr() % 16:
        raise SeaNativeError(f"mlp_fused: {what} must be float32 (M, {C}) rows, 16-byte aligned, row stride % 4 == 0")


def _ln_args(ln, Cc):
    if ln is None:
        return None
    w, b, eps = ln
    if (w.dtype != torch.float32 or b.dtype != torch.float32 or w.numel() != Cc or b.numel() != Cc or not w.is_contiguous()
            or not b.is_contiguous() or w.data_ptr() % 16 or b.data_ptr() % 16):
        raise SeaNativeError("mlp_fused: LayerNorm weight / bias must be contiguous, 16-byte aligned float32 of C entries")
    return w, b, float(eps)


def mlp_fused_forward(x2, W1p: PackedWeight, b1, W2p: PackedWeight, b2, res, amax_x, amax_h, out=None, ln=None):
    """out = res + W2 GELU(W1 x2 + b1) + b2 in ONE kernel (sea_mlp_fused_fwd): the hidden tensor never leaves the CU.  W1p / W2p:
    the fp16 x 2 packs (terms 22) of w1 (H x C) and w2 (C x H); amax_x / amax_h: one int32 device word each, float bits of a
    bound of max|x2| and of max|GELU(W1 x2 + b1)|.  Bit for bit the two gemm_split launches (a_gelu prologue) plus the add.
    ``ln`` = (weight, bias, eps): x2 is the INPUT of a LayerNorm over its channels whose output feeds the MLP (amax_x then bounds
    that output); the normalisation runs in the kernel's prologue with ``layernorm``'s arithmetic and summation order."""
    _dev(x2, b1, b2, res, out, amax_x, amax_h, *(ln[:2] if ln is not None else ()))
    M, Cc = x2.shape
    H = W1p.N
    if (W1p.terms, W2p.terms, W1p.K, W2p.N, W2p.K, W1p.batch, W2p.batch) != (22, 22, Cc, Cc, H, 1, 1):
        raise SeaNativeError("mlp_fused_forward: fp16 x 2 packs of w1 (H x C) and w2 (C x H) expected")
    _mlp_rows(x2, Cc, "x")
    if res is not None:
        _mlp_rows(res, Cc, "res")
    if out is None:
        out = torch.empty(M, Cc, dtype=torch.float32, device=x2.device)
    _mlp_rows(out, Cc, "out")
    la = _ln_args(ln, Cc)
    if la is not None:
        _check(lib().sea_ln_mlp_fused_fwd(_p(x2), x2.stride(0), _p(la[0]), _p(la[1]), la[2], _p(W1p.data), _p(b1), _p(W2p.data),
                                          _p(b2), _p(res), res.stride(0) if res is not None else 0, _p(out), out.stride(0), M, Cc,
                                          H, _p(amax_x), _p(amax_h), _stream()), "sea_ln_mlp_fused_fwd")
        return out
    _check(lib().sea_mlp_fused_fwd(_p(x2), x2.stride(0), _p(W1p.data), _p(b1), _p(W2p.data), _p(b2), _p(res),
                                   res.stride(0) if res is not None else 0, _p(out), out.stride(0), M, Cc, H, _p(amax_x),
                                   _p(amax_h), _stream()), "sea_mlp_fused_fwd")
    return out


def mlp_fused_backward(g2, x2, W1p: PackedWeight, b1, W2tp: PackedWeight, W1tp: PackedWeight, amax_x, amax_mul, out=None,
                       ln=None):
    """input gradient of ``mlp_fused_forward`` w.r.t. x2 (sea_mlp_fused_bwd): t = W1 x2 + b1 is recomputed, u = g2 W2 and
    dx = (u GELU'(t)) W1 follow in the same kernel.  W2tp / W1tp: the packs of w2 and w1 with trans=True (the operands of the
    two input-gradient products); amax_mul: ONE float32 on the device, rowmax|g2[r]| * amax_mul bounds row r of u GELU'(t).
    Bit for bit the rowmax pass + the two gemm_split launches (a_gelu_grad_of prologue) of the unfused backward."""
    _dev(g2, x2, b1, out, amax_x, amax_mul)
    M, Cc = x2.shape
    H = W1p.N
    if ((W1p.terms, W2tp.terms, W1tp.terms) != (22, 22, 22) or (W1p.K, W2tp.N, W2tp.K, W1tp.N, W1tp.K) != (Cc, H, Cc, Cc, H)
            or amax_mul.dtype != torch.float32 or amax_mul.numel() != 1):
        raise SeaNativeError("mlp_fused_backward: fp16 x 2 packs of w1, w2^T, w1^T and one float32 amax_mul expected")
    _mlp_rows(x2, Cc, "x")
    _mlp_rows(g2, Cc, "g")
    if out is None:
        out = torch.empty(M, Cc, dtype=torch.float32, device=x2.device)
    _mlp_rows(out, Cc, "out")
    la = _ln_args(ln, Cc)
    if la is not None:       # x2 is the LayerNorm's input; the result is the gradient w.r.t. THAT (frozen affine parameters)
        _dev(*la[:2])
        _check(lib().sea_ln_mlp_fused_bwd(_p(g2), g2.stride(0), _p(x2), x2.stride(0), _p(la[0]), _p(la[1]), la[2], _p(W1p.data),
                                          _p(b1), _p(W2tp.data), _p(W1tp.data), _p(out), out.stride(0), M, Cc, H, _p(amax_x),
                                          _p(amax_mul), _stream()), "sea_ln_mlp_fused_bwd")
        return out
    _check(lib().sea_mlp_fused_bwd(_p(g2), g2.stride(0), _p(x2), x2.stride(0), _p(W1p.data), _p(b1), _p(W2tp.data),
                                   _p(W1tp.data), _p(out), out.stride(0), M, Cc, H, _p(amax_x), _p(amax_mul), _stream()),
           "sea_mlp_fused_bwd")
    return out


class AmaxPool:
    """Zero-initialised 4-byte device words for the ``out_amax`` output of the fp16 x 2 GEMMs (atomicMax needs a zeroed
    word): a helper for callers of that ABI feature and its tests.  The shipped models do not use it (their scales come
    from analytic bounds, the Winograd transform and per-row maxima, none of which needs a zeroed word), so no model
    forward resets it."""
    _pools = {}
    SIZE = 1024

    def __init__(self, device):
        self.buf = torch.zeros(self.SIZE, dtype=torch.int32, device=device)
        self.next = 0

    @classmethod
    def get(cls, device):
        device = torch.device(device)
        key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
        if key not in cls._pools:
            cls._pools[key] = AmaxPool(device)
        return cls._pools[key]

    def reset(self):
        self.buf.zero_()
        self.next = 0

    def word(self):
        if self.next >= self.SIZE:      # more GEMMs than words since the last reset: a FRESH zeroed buffer (words that
            self.buf = torch.zeros(self.SIZE, dtype=torch.int32, device=self.buf.device)   # were handed out stay valid)
            self.next = 0
        w = self.buf[self.next:self.next + 1]
        self.next += 1
        return w


# A/B switch (env SEA_AMAX_FUSE=0): every fp16 x 2 GEMM computes max|A| with its own pass instead of taking it from the producer
AMAX_FROM_PRODUCERS = os.environ.get("SEA_AMAX_FUSE", "1") != "0"


def amax_word(device):
    return AmaxPool.get(device).word()


# split-K (env SEA_GEMM_KSPLIT=0 disables): GEMMs with fewer than KSPLIT_BELOW 128 x 128 tiles and a long K run as a batch
# of K slices + sea_gemm_splitk_reduce
KSPLIT = os.environ.get("SEA_GEMM_KSPLIT", "1") != "0"
KSPLIT_BELOW, KSPLIT_TARGET, KSPLIT_MIN_KB = (int(v) for v in os.environ.get("SEA_GEMM_KSPLIT_PARAMS", "400,768,8").split(","))
_ksplit_ws = {}


def _ksplit(M, N, K):
    blocks = -(-M // 128) * -(-N // 128)
    nkb = K // 32
    if not KSPLIT or blocks >= KSPLIT_BELOW or N % 4:
        return 1
    S = min(-(-KSPLIT_TARGET // blocks), nkb // KSPLIT_MIN_KB)
    while S > 1 and nkb % S:
        S -= 1
    return max(S, 1)


def _ksplit_workspace(device, numel):
    """(splits, M, N) partial products: ONE buffer per device, grown on demand.  A product and its reduce pass run back to
    back on one stream, and every consumer of this module orders its streams (ApgdRun: wait_stream both ways around the
    iteration it runs on its capture stream), so one buffer serves them all; keyed by stream it leaked one ~50 MB buffer per
    pooled stream handle.  A captured HIP graph bakes the buffer's address in: ApgdRun pins the tensor for the lifetime
    of its graphs (``ksplit_workspace_pin``), so a later, larger product that replaces the entry cannot free it."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    ws = _ksplit_ws.get(key)
    if ws is None or ws.numel() < numel:
        ws = _ksplit_ws[key] = torch.empty(numel, dtype=torch.float32, device=device)
    return ws[:numel]


def ksplit_workspace_pin(device):
    """the current split-K workspace tensor of ``device`` (or None): hold the reference while a captured graph may replay"""
    return _ksplit_ws.get(device.index if device.index is not None else torch.cuda.current_device())


class _GemmEpilogue(C.Structure):       # SeaGemmEpilogue of include/sea_hip.h
    _fields_ = [("addend", C.c_void_p), ("ld_addend", C.c_int64), ("stride_addend", C.c_int64),
                ("gelu_out", C.c_void_p), ("gelu_grad_of", C.c_void_p), ("a_gelu_grad_of", C.c_void_p), ("a_gate", C.c_int), ("a_gelu", C.c_int),
                ("a_amax_mul", C.c_float), ("a_amax_mul_dev", C.c_void_p)]


def _amax_words(A3, M, K, G, sA, groups, per_row=False):
    """exact per-group maxima of |A| (float bits), one word per M / groups rows -- or one per ROW (``per_row``: gradient
    operands, whose rows span many orders of magnitude)"""
    if per_row and G != 1:
        raise SeaNativeError("gemm_split: row_amax (one scale word per row) is defined for a single matrix, not a batch of them")
    if per_row:
        words = torch.empty(M, dtype=torch.int32, device=A3.device)
        _check(lib().sea_absmax_bits(_p(A3), A3.stride(1), M, K, 1, 0, 1, _p(words), _stream()), "sea_absmax_bits")
        return words, 1
    rpw = M // groups if (groups > 1 and M % groups == 0) else 0
    words = torch.empty(groups if rpw else 1, dtype=torch.int32, device=A3.device)
    _check(lib().sea_absmax_bits(_p(A3), A3.stride(1), M, K, G, sA, rpw, _p(words), _stream()), "sea_absmax_bits")
    return words, rpw


def gemm_split(A, Wp: PackedWeight, bias=None, relu: bool = False, out=None, amax=None, out_amax=None, addend=None,
               gelu_out=None, gelu_grad_of=None, amax_rows: int = 0, groups: int = 1, a_gelu_grad_of=None,
               a_gelu: bool = False, a_relu_gate=None, row_amax: bool = False, amax_mul: float = 1.0):
    """out (.., N) = A (.., K) @ W^T [+ bias] [ReLU] with W pre-split (``gemm_split_pack``).  A: fp32, last dim
    contiguous; 2-D (M, K) with any 4-aligned row stride, or (G, M, K) against a batch of G packed weights.
    fp16 x 2 weights (terms 22): ``amax`` = device words holding the float bits of (upper bounds of) max|A|, one per
    ``amax_rows`` rows (0: one word for the tensor); computed here when None, one word per ``groups``-th of the rows (pass
    the number of images: an image's scale then does not depend on its batch partners); ``out_amax`` = pre-zeroed word that
    receives the bits of max|out|.  ``row_amax``: the words computed here are one per ROW (sea_absmax_bits with
    rows_per_word = 1).  ``amax_mul``: the supplied words bound max|A| only after multiplication by this constant (a per-row
    bound of the producer's input carried through the producer: see SeaGemmEpilogue.a_amax_mul).
    Fused epilogue (sea_gemm_split_fused): ``addend`` (shape of out, last dim contiguous) is added before the activation;
    ``gelu_out`` (layout of out) receives GELU(out) while out keeps the pre-activation; the result is multiplied by
    GELU'(``gelu_grad_of``) (layout of out).  Prologue instead (exclusive): A is read as A * GELU'(``a_gelu_grad_of``)
    (same shape and strides as A; terms 2 or 22), as (``a_relu_gate`` > 0 ? A : 0) (same layout), or as GELU(A)
    (``a_gelu``).
    Split-K products (chosen here for small tile grids) go through ONE partial-product workspace per device: a caller that
    runs these GEMMs on several streams at once must order them itself, and a caller that captures them into a HIP graph
    must hold ``ksplit_workspace_pin(device)`` for as long as the graph may replay (``ApgdRun`` does both)."""
    gate = a_relu_gate is not None
    if gate:
        if a_gelu_grad_of is not None:
            raise SeaNativeError("gemm_split: a_relu_gate and a_gelu_grad_of are exclusive")
        a_gelu_grad_of = a_relu_gate
    _dev(A, bias, out, addend, gelu_out, gelu_grad_of, a_gelu_grad_of)
    batched = Wp.batch > 1 or A.dim() == 3
    A3 = A if A.dim() == 3 else A.unsqueeze(0)
    if (A.dtype != torch.float32 or A3.dim() != 3 or A3.shape[0] != Wp.batch or A3.shape[2] != Wp.K
            or A3.stride(2) != 1):
        raise SeaNativeError("gemm_split: A must be float32 (M,K) / (G,M,K) with K matching the packed weights")
    G, M, K = A3.shape
    if out is None:
        out = torch.empty((G, M, Wp.N) if batched else (M, Wp.N), dtype=torch.float32, device=A.device)
    O3 = out if out.dim() == 3 else out.unsqueeze(0)
    if O3.shape != (G, M, Wp.N) or O3.stride(2) != 1 or out.dtype != torch.float32:
        raise SeaNativeError("gemm_split: output must be float32 (.., M, N) with a contiguous last dim")
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != Wp.N or not bias.is_contiguous()):
        raise SeaNativeError("gemm_split: bias must be contiguous float32 of N entries")
    sA, sC = (A3.stride(0) if G > 1 else 0), (O3.stride(0) if G > 1 else 0)
    if amax is not None and (amax.dtype != torch.int32 or not amax.is_contiguous()
                             or amax.numel() < (-(-M // amax_rows) if amax_rows > 0 else 1)):
        raise SeaNativeError("gemm_split: amax must hold one int32 word per amax_rows rows")
    # a Python float, or ONE float32 in device memory (derived on the device: no host round trip)
    mul_dev = amax_mul if (torch.is_tensor(amax_mul) and Wp.terms == 22) else None
    if mul_dev is not None and (mul_dev.dtype != torch.float32 or mul_dev.numel() != 1 or mul_dev.device != A.device):
        raise SeaNativeError("gemm_split: a tensor amax_mul must be one float32 on A's device")
    amax_mul = float(amax_mul) if (Wp.terms == 22 and mul_dev is None) else 1.0
    fused = (addend is not None or gelu_out is not None or gelu_grad_of is not None or a_gelu_grad_of is not None
             or a_gelu or amax_mul != 1.0 or mul_dev is not None)
    if a_gelu_grad_of is not None and (a_gelu_grad_of.shape != A.shape or a_gelu_grad_of.stride() != A.stride()
                                       or a_gelu_grad_of.dtype != torch.float32 or Wp.terms not in (1, 2, 22)):
        raise SeaNativeError("gemm_split: a_gelu_grad_of must be float32 with A's shape and strides (terms 1, 2 or 22)")
    # split-K takes the prologues (applied per slice) and the addend (added by the reduce pass)
    only_pro = gelu_out is None and gelu_grad_of is None
    add_ok = addend is None or (addend.dim() == 2 and addend.shape == O3.shape[1:] and addend.stride(1) == 1
                                and addend.stride(0) % 4 == 0 and addend.data_ptr() % 16 == 0 and addend.dtype == torch.float32)
    if G == 1 and (not fused or (only_pro and add_ok)) and Wp.src is not None and O3.stride(1) % 4 == 0 and O3.data_ptr() % 16 == 0:
        S = _ksplit(M, Wp.N, K)
        if S > 1:
            if Wp.terms == 22 and amax is None:
                amax, amax_rows = _amax_words(A3, M, K, 1, 0, groups, row_amax)
            part = _ksplit_workspace(A.device, S * M * Wp.N).view(S, M, Wp.N)
            shape, strides = (S, M, K // S), (K // S, A3.stride(1), 1)
            gemm_split(A3[0].as_strided(shape, strides), Wp.k_slices(S), out=part, amax=amax, amax_rows=amax_rows,
                       a_gelu_grad_of=None if (a_gelu_grad_of is None or gate) else a_gelu_grad_of.as_strided(shape, strides),
                       a_gelu=a_gelu, amax_mul=mul_dev if mul_dev is not None else amax_mul,
                       **({"a_relu_gate": a_gelu_grad_of.as_strided(shape, strides)} if gate else {}))
            _check(lib().sea_gemm_splitk_reduce(_p(part), S, M, Wp.N, _p(bias), _p(addend),
                                                addend.stride(0) if addend is not None else 0, int(relu), _p(O3), O3.stride(1),
                                                _p(out_amax) if Wp.terms == 22 else None, _stream()), "sea_gemm_splitk_reduce")
            return out
    if addend is not None and (a_gelu or a_gelu_grad_of is not None):
        # a prologue and an addend only travel together through split-K (the reduce pass adds); when this product does not
        # split (few K blocks, unaligned addend, batched weights) the C ABI rejects the pair: add separately instead
        out = gemm_split(A, Wp, bias=bias, relu=False, out=out, amax=amax, out_amax=None, amax_rows=amax_rows, groups=groups,
                         a_gelu_grad_of=None if gate else a_gelu_grad_of, a_gelu=a_gelu,
                         a_relu_gate=a_gelu_grad_of if gate else None, row_amax=row_amax,
                         amax_mul=mul_dev if mul_dev is not None else amax_mul)
        out += addend
        if relu:
            torch.relu_(out)
        if out_amax is not None and Wp.terms == 22:
            # the contract "out_amax receives the bits of max|out|" holds on this path too (a consumer GEMM that read a
            # zero word would scale by 2^126)
            O2 = out if out.dim() == 3 else out.unsqueeze(0)
            _check(lib().sea_absmax_bits(_p(O2), O2.stride(1), M, Wp.N, G, sC, 0, _p(out_amax), _stream()), "sea_absmax_bits")
        return out
    if fused:
        epi = _GemmEpilogue()
        if addend is not None:
            D3 = addend if addend.dim() == 3 else addend.unsqueeze(0)
            if D3.shape != O3.shape or D3.stride(2) != 1 or addend.dtype != torch.float32:
                raise SeaNativeError("gemm_split: addend must be float32 of the output's shape with a contiguous last dim")
            epi.addend, epi.ld_addend, epi.stride_addend = D3.data_ptr(), D3.stride(1), (D3.stride(0) if G > 1 else 0)
        for name, t in (("gelu_out", gelu_out), ("gelu_grad_of", gelu_grad_of)):
            if t is not None:
                T3 = t if t.dim() == 3 else t.unsqueeze(0)
                if T3.shape != O3.shape or T3.stride() != O3.stride() or t.dtype != torch.float32:
                    raise SeaNativeError(f"gemm_split: {name} must be float32 with the output's shape and strides")
                setattr(epi, name, T3.data_ptr())
        if a_gelu_grad_of is not None:
            epi.a_gelu_grad_of = a_gelu_grad_of.data_ptr()
            epi.a_gate = int(gate)
        epi.a_gelu = int(bool(a_gelu))
        epi.a_amax_mul = amax_mul if amax_mul != 1.0 else 0.0
        epi.a_amax_mul_dev = mul_dev.data_ptr() if mul_dev is not None else None
        if Wp.terms == 22 and amax is None:
            amax, amax_rows = _amax_words(A3, M, K, G, sA, groups, row_amax)
        _check(lib().sea_gemm_split_fused(_p(A3), A3.stride(1), _p(Wp.data), _p(O3), O3.stride(1), _p(bias), int(relu), M, Wp.N,
                                          K, Wp.terms, G, sA, Wp.stride, sC, _p(amax) if Wp.terms == 22 else None, amax_rows,
                                          _p(out_amax) if Wp.terms == 22 else None, C.addressof(epi), _stream()),
               "sea_gemm_split_fused")
        return out
    if Wp.terms == 22:
        # fp16 x 2: the activation scale comes from max|A|, computed on the device (no host round trip)
        if amax is None:
            amax, amax_rows = _amax_words(A3, M, K, G, sA, groups, row_amax)
        _check(lib().sea_gemm_split_f16(_p(A3), A3.stride(1), _p(Wp.data), _p(O3), O3.stride(1), _p(bias), int(relu), M, Wp.N,
                                        K, G, sA, Wp.stride, sC, _p(amax), amax_rows, _p(out_amax), _stream()),
               "sea_gemm_split_f16")
        return out
    _check(lib().sea_gemm_split(_p(A3), A3.stride(1), _p(Wp.data), _p(O3), O3.stride(1), _p(bias), int(relu), M, Wp.N, K,
                                Wp.terms, G, sA, Wp.stride, sC, _stream()), "sea_gemm_split")
    return out
