"""CPU-side checks that run everywhere (`-m "not gpu"`): the C-ABI library loads and exports every
symbol include/sea_hip.h declares, the host C++ greedy (K9) reproduces the reference bit for bit, the
data-independent schedules match the oracle, and device-only entry points refuse CPU tensors."""
import os
import random
import re
import subprocess

import numpy as np
import pytest
import torch

from conftest import PKG, ROOT, load_golden
from oracle import sea_oracle as O


@pytest.fixture(scope="module")
def native():
    from semseg import _native
    if not os.path.exists(_native.LIB_PATH):
        import sys
        sys.path.insert(0, PKG)
        import build_native
        build_native.build(verbose=False)
    return _native


def test_library_exports_every_declared_symbol(native):
    header = open(os.path.join(ROOT, "include", "sea_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|int64_t|size_t|const char\*)\s+(sea_\w+)\s*\(", header, flags=re.M))
    assert declared, "no declarations parsed"
    out = subprocess.run(["nm", "-D", "--defined-only", native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (sea_\w+)", out))
    assert declared <= exported, declared - exported
    assert declared == set(native.EXPORTS)
    lib = native.lib()  # binds every prototype; AttributeError on ABI drift
    assert lib.sea_abi_version() == 1
    assert b"gfx950" in lib.sea_build_info()
    assert lib.sea_loss_workspace_bytes(8, 512 * 512) == (8 * 2048 + 1) * 16  # header + one record per 128-pixel tile (the smallest tile any K2 variant uses)


def test_fastdiv_magic_divides_exactly(native):
    """the multiplier / shift pairs the kernels use instead of integer division (csrc/sea_common.h: FastDiv)"""
    import ctypes as C
    lib = native.lib()
    rng = np.random.default_rng(0)
    ds = np.unique(np.concatenate([np.arange(1, 300), 2 ** np.arange(0, 31), 2 ** np.arange(1, 31) - 1,
                                   2 ** np.arange(1, 30) + 1, rng.integers(1, 2 ** 31 - 1, 300),
                                   [24, 96, 151, 171, 1025, 262144, 2 ** 31 - 1]]))
    for d in ds:
        m, s = C.c_uint32(), C.c_uint32()
        assert lib.sea_fastdiv_magic(int(d), C.byref(m), C.byref(s)) == 0
        n = np.unique(np.concatenate([rng.integers(0, 2 ** 31, 2000), np.arange(0, 64), [2 ** 31 - 1, 2 ** 31 - 2],
                                      np.arange(1, 40) * d - 1, np.arange(1, 40) * d])).astype(np.uint64)
        n = n[n < 2 ** 31]
        q = (((n * np.uint64(m.value)) >> np.uint64(32)) + n) >> np.uint64(s.value)
        assert np.all(q < 2 ** 32)  # the 32-bit sum in the kernel cannot wrap
        np.testing.assert_array_equal(q, n // np.uint64(d))
    assert lib.sea_fastdiv_magic(0, C.byref(m), C.byref(s)) == 1


def test_header_is_plain_c():
    subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "sea_hip.h")], check=True)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_host_greedy_matches_reference_bit_for_bit(native, tag):
    g = load_golden(f"g7_evalsea_{tag}")
    random.seed(225)
    st = random.getstate()
    miou, sel, rounds, new_state = native.worst_miou_greedy(g["ints"], g["unions"], st[1])
    assert miou == g["final_miou"]
    ref, sel_ref, rounds_ref = O.worst_case_miou(g["ints"], g["unions"], rng=random.Random(225))
    assert miou == ref and sel == sel_ref and rounds == rounds_ref
    # the Mersenne-Twister state advanced exactly like `rounds` calls of random.shuffle
    rr = random.Random(225)
    for _ in range(rounds):
        lst = list(range(g["ints"].shape[1]))
        rr.shuffle(lst)
    assert new_state == rr.getstate()[1]


def test_host_greedy_random_tables(native):
    gen = torch.Generator().manual_seed(5)
    for N_, C_ in ((7, 4), (23, 9), (40, 21)):
        tgt = torch.randint(0, C_, (N_, 10, 10), generator=gen)
        preds = torch.stack([torch.where(torch.rand(tgt.shape, generator=gen) < 0.3 + 0.1 * a,
                                         torch.randint(0, C_, tgt.shape, generator=gen), tgt) for a in range(3)])
        ints, unis = O.per_image_tables(preds, tgt, C_)
        seed = 1000 + N_
        ref, sel_ref, r_ref = O.worst_case_miou(ints, unis, rng=random.Random(seed))
        miou, sel, r, _ = native.worst_miou_greedy(ints, unis, random.Random(seed).getstate()[1])
        assert miou == ref and sel == sel_ref and r == r_ref


def test_schedules_match_oracle():
    from semseg import attacker as A
    for n in (1, 2, 3, 5, 10, 25, 90, 120, 300):
        assert A.apgd_checkpoints(n) == O.checkpoints(n)
        assert A.largereps_schedule(n, 4 / 255) == O.largereps_schedule(n, 4 / 255)
    assert A.largereps_schedule(300, 1.0)[0] == [90, 90, 120]
    assert list(A.apgd_checkpoints(90).items())[:3] == [(18, 19), (35, 17), (50, 15)]


def test_device_entry_points_refuse_cpu_tensors(native):
    x = torch.rand(1, 3, 8, 8)
    with pytest.raises(native.SeaNativeError):
        native.apgd_linf_step(x, x, x, x, torch.ones(1), 0.1, 1.0)
    with pytest.raises(native.SeaNativeError):
        native.loss_fwd_bwd(torch.rand(1, 5, 4, 4), torch.zeros(1, 4, 4, dtype=torch.int64), None, 0, 3, 1.0)


def test_api_surface_matches_reference_signatures():
    """names, positional order and defaults of the reference call surface"""
    import inspect
    from semseg import attacker as A, val as V, metrics as M, losses as L
    from semseg.utils import utils as U
    sig = inspect.signature(A.apgd_train)
    assert list(sig.parameters)[:18] == ["model", "x", "y", "norm", "eps", "n_iter", "use_rs", "loss", "verbose",
                                         "is_train", "early_stop", "track_loss", "logger", "y_target", "ignore_index",
                                         "x_init", "num_classes", "weights"]
    assert sig.parameters["n_iter"].default == 10 and sig.parameters["loss"].default == "ce"
    sig = inspect.signature(A.apgd_largereps)
    assert list(sig.parameters)[:17] == ["model", "x", "y", "weights", "norm", "eps", "n_iter", "loss", "verbose",
                                         "n_restarts", "log_path", "early_stop", "eot_iter", "track_loss", "use_rs",
                                         "ignore_index", "num_classes"]
    assert sig.parameters["eps"].default == 8.0 / 255.0
    assert set(A.criterion_dict) == {"ce", "ce-avg", "mask-ce-avg", "mask-ce-bal", "js-avg"}
    assert list(inspect.signature(V.Pgd_Attack_1.__init__).parameters)[1:] == ["epsilon", "alpha", "num_iter", "los"]
    assert list(inspect.signature(V.Pgd_Attack.__init__).parameters)[1:5] == ["eps", "alpha", "num_iter", "los"]
    assert list(inspect.signature(M.Metrics.__init__).parameters)[1:] == ["num_classes", "ignore_label", "device"]
    assert L.get_loss("CrossEntropy", -1, None).criterion.ignore_index == -1
    assert len(U.ADE_WTS) == 151 and len(U.VOC_WTS) == 21
    assert U.getModelName("UperNetForSemanticSegmentation", "ConvNeXt-T_CVST") == "UperNet_ConvNeXt-T_CVST"


# ---------------------------------------------------------------------------- model-side host logic (no GPU needed)
def test_cl_pixel_stride_recognises_nhwc_tensors_and_channel_slices(native):
    cl = torch.channels_last
    wide = torch.zeros(2, 24, 5, 7).contiguous(memory_format=cl)
    assert native.cl_pixel_stride(wide) == 24
    assert native.cl_pixel_stride(wide[:, 4:12]) == 24          # channel slice: pixel stride of the parent
    assert native.cl_pixel_stride(wide[:, 4:10]) is None         # 6 channels: not a multiple of 4
    assert native.cl_pixel_stride(wide[:, 2:10]) is None         # 8-byte aligned only
    assert native.cl_pixel_stride(torch.zeros(2, 24, 5, 7)) is None          # NCHW
    assert native.cl_pixel_stride(wide[:, :, 1:]) is None                     # not dense over pixels
    assert native.cl_pixel_stride(torch.zeros(3, 8, 1, 1)) == 8              # 1x1 maps are NHWC and NCHW at once
    assert native.cl_pixel_stride(wide.double()) is None
    assert native._is_cl(wide) and not native._is_cl(torch.zeros(3, 8, 1, 1)) and not native._is_cl(wide[:, 4:12])


def test_folded_batchnorm_and_frozen_parameter_scope():
    from semseg.attacker import _FrozenParameters
    from semseg.models import convnext_upernet as M
    torch.manual_seed(0)
    mod = M.ConvModule(8, 12, 1).eval()
    mod.batch_norm.running_mean.normal_()
    mod.batch_norm.running_var.uniform_(0.5, 2.0)
    with torch.no_grad():
        mod.batch_norm.weight.uniform_(0.5, 1.5)
        mod.batch_norm.bias.normal_()
    cache = {}
    scale, shift = M._folded_bn(mod.batch_norm, None, cache)
    z = torch.randn(3, 12, 4, 5)
    torch.testing.assert_close(z * scale[None, :, None, None] + shift[None, :, None, None], mod.batch_norm(z),
                               rtol=1e-5, atol=1e-6)
    assert M._folded_bn(mod.batch_norm, None, cache)[0] is scale           # cached
    before, ptr = scale.clone(), scale.data_ptr()
    mod.batch_norm.running_var.mul_(2.0)                                    # in-place update -> a new fold ...
    scale2 = M._folded_bn(mod.batch_norm, None, cache)[0]
    # ... refreshed IN PLACE since round 5 (same tensor, same address, new contents: what lets a captured HIP graph over the
    # model outlive a weight update; convnext_upernet._stable)
    assert scale2 is scale and scale2.data_ptr() == ptr and not torch.allclose(scale2, before)
    torch.testing.assert_close(z * scale2[None, :, None, None] + M._folded_bn(mod.batch_norm, None, cache)[1][None, :, None, None],
                               mod.batch_norm(z), rtol=1e-5, atol=1e-6)
    # fast paths are for HIP tensors with frozen parameters only: CPU tensors take the reference composition
    x = torch.randn(2, 8, 6, 6)
    assert not M._pointwise_ok(mod, x) and not M._wino_ok(mod.conv, x)
    ref = torch.relu(mod.batch_norm(mod.conv(x)))
    torch.testing.assert_close(mod(x), ref)
    # the attack's parameter scope restores every flag, also when the forward raises
    flags = [p.requires_grad for p in mod.parameters()]
    mod.conv.weight.requires_grad_(False)
    flags[0] = False
    with _FrozenParameters(mod):
        assert not any(p.requires_grad for p in mod.parameters())
    assert [p.requires_grad for p in mod.parameters()] == flags
    with pytest.raises(RuntimeError):
        with _FrozenParameters(mod):
            raise RuntimeError("forward failed")
    assert [p.requires_grad for p in mod.parameters()] == flags


def test_split_k_policy(native):
    """which GEMMs run as a batch of K slices (semseg/_native.py:_ksplit): few 128 x 128 tiles, long K, slices of whole
    32-deep K steps, at least 8 of them per slice; everything else single pass"""
    N = native
    for M, Nn, K, want in ((2048, 768, 3072, 8), (8192, 384, 1536, 4), (2048, 512, 4608, 12), (8192, 512, 4608, 3),
                           (2048, 3072, 768, 2), (131072, 384, 96, 1), (8192, 1536, 384, 1), (2048, 768, 128, 1),
                           (2048, 770, 3072, 1), (32768, 192, 768, 1)):
        S = N._ksplit(M, Nn, K)
        assert S == want, (M, Nn, K, S)
        assert S == 1 or ((K // 32) % S == 0 and K // 32 // S >= N.KSPLIT_MIN_KB)
    keep, N.KSPLIT = N.KSPLIT, False
    try:
        assert N._ksplit(2048, 768, 3072) == 1
    finally:
        N.KSPLIT = keep


def test_analytic_activation_bounds_hold():
    """the fp16 x 2 activation scales that need no pass over the activations (convnext_upernet._ln_bound_word /
    _linear_bound_word): float bits of an UPPER bound of |LayerNorm(x)|, and of |x W^T + b| for any x within the input
    bound (hence of GELU / ReLU of it and of convex combinations of its rows)"""
    from semseg.models import convnext_upernet as M
    g = torch.Generator().manual_seed(3)
    for C in (96, 384):
        ln = torch.nn.LayerNorm(C, eps=1e-6)
        with torch.no_grad():
            ln.weight.copy_(torch.randn(C, generator=g) * 2)
            ln.bias.copy_(torch.randn(C, generator=g))
        lin = torch.nn.Linear(C, 4 * C)
        with torch.no_grad():
            lin.weight.copy_(torch.randn(4 * C, C, generator=g) / C ** 0.5)
            lin.bias.copy_(torch.randn(4 * C, generator=g))
        cache = {}
        w_ln = M._ln_bound_word(ln, cache)
        w_lin = M._linear_bound_word(w_ln, lin.weight, lin.bias, cache)
        assert w_ln.dtype == torch.int32 and w_ln.numel() == 1 and M._ln_bound_word(ln, cache) is w_ln   # cached
        b_ln, b_lin = w_ln.view(torch.float32).item(), w_lin.view(torch.float32).item()
        worst_ln = worst_lin = 0.0
        for scale, spike in ((1.0, 0.0), (1e-3, 0.0), (50.0, 0.0), (1.0, 1e4)):        # incl. one dominant channel: the
            x = torch.randn(512, C, generator=g) * scale                                # sqrt(C) extreme of a LayerNorm
            x[:, 7] += spike
            with torch.no_grad():
                y = ln(x)
                t = lin(y)
            worst_ln, worst_lin = max(worst_ln, y.abs().max().item()), max(worst_lin, t.abs().max().item())
            assert torch.nn.functional.gelu(t).abs().max().item() <= t.abs().max().item() + 1e-6
        assert worst_ln <= b_ln and worst_lin <= b_lin
        assert b_ln <= 40 * worst_ln and b_lin <= 2 ** 10 * worst_lin                  # loose, but inside the free range
        with torch.no_grad():
            ln.weight.mul_(3.0)                                                         # in-place change: re-derived
        w_ln2 = M._ln_bound_word(ln, cache)
        assert w_ln2.view(torch.float32).item() > b_ln
        assert M._linear_bound_word(w_ln2, lin.weight, lin.bias, cache).view(torch.float32).item() > b_lin   # follows its input


def test_gemm_epilogue_struct_matches_the_header(native):
    """the ctypes mirror of SeaGemmEpilogue (semseg/_native.py) lists the fields of include/sea_hip.h in the same order
    with the same C types: the struct crosses the ABI by pointer"""
    import ctypes
    src = open(os.path.join(ROOT, "include", "sea_hip.h")).read()
    body = re.search(r"typedef struct SeaGemmEpilogue \{(.*?)\} SeaGemmEpilogue;", src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ctype, names = decl.rsplit(" ", 1)[0], decl
        m = re.match(r"(const float\*|float\*|int64_t|int|float)\s+(.*)", decl)
        assert m, decl
        kind = {"const float*": ctypes.c_void_p, "float*": ctypes.c_void_p, "int64_t": ctypes.c_int64, "int": ctypes.c_int,
                "float": ctypes.c_float}[m.group(1)]
        fields += [(n.strip(), kind) for n in m.group(2).split(",")]
    assert [(n, t) for n, t in native._GemmEpilogue._fields_] == fields
