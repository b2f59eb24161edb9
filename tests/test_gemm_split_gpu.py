"""M8: fp32 GEMM with frozen weights on the bf16 matrix cores by operand splitting (csrc/gemm_split.hip), `-m gpu`.
Reference: float64 matmul of the same fp32 operands; the yardstick is the error of the fp32 GEMM it replaces
(torch.mm on hipBLASLt).  Layers it serves: reference semseg/models/uperforseg.py:119-146, 200-215, 255-262."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def N():
    from semseg import _native
    _native.lib()
    return _native


def _ref(A, W, bias=None, relu=False):
    y = A.double() @ W.double().transpose(-1, -2)
    if bias is not None:
        y = y + bias.double()
    return y.clamp_min(0) if relu else y


def test_exact_on_integer_data(N):
    """small integers are exact in every bf16 term and every partial sum: the result must be exact"""
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.randint(-8, 9, (300, 96), generator=g, device="cuda").float()
    W = torch.randint(-8, 9, (200, 96), generator=g, device="cuda").float()
    for terms in (2, 3):
        out = N.gemm_split(A, N.gemm_split_pack(W, terms=terms))
        assert torch.equal(out, A @ W.t())
    # values that need all three terms: 24-bit integers times powers of two
    A = (torch.randint(-2 ** 23, 2 ** 23, (128, 32), generator=g, device="cuda").float())
    W = torch.zeros(128, 32, device="cuda")
    W[torch.arange(128), torch.arange(128) % 32] = 0.5
    out = N.gemm_split(A, N.gemm_split_pack(W, terms=3))
    assert torch.equal(out, (A.double() @ W.double().t()).float())


@pytest.mark.parametrize("M,K,Nn", [(4096, 512, 512), (1000, 96, 384), (131, 384, 96), (17, 2816, 512), (2048, 768, 21),
                                    (256, 32, 130)])
def test_accuracy_matches_fp32_gemm(N, M, K, Nn):
    g = torch.Generator(device="cuda").manual_seed(M + K)
    A = torch.randn(M, K, generator=g, device="cuda") * torch.rand(M, 1, generator=g, device="cuda") * 3
    W = torch.randn(Nn, K, generator=g, device="cuda") / K ** 0.5
    ref = _ref(A, W)
    scale = ref.abs().max().item()
    e_lib = ((A @ W.t()).double() - ref).abs().max().item() / scale
    e3 = (N.gemm_split(A, N.gemm_split_pack(W, terms=3)).double() - ref).abs().max().item() / scale
    e2 = (N.gemm_split(A, N.gemm_split_pack(W, terms=2)).double() - ref).abs().max().item() / scale
    e22 = (N.gemm_split(A, N.gemm_split_pack(W, terms=22)).double() - ref).abs().max().item() / scale
    print(f"M={M} K={K} N={Nn}: max err / max|C|  hipBLASLt fp32 {e_lib:.2e}   bf16x3 {e3:.2e}   fp16x2 {e22:.2e}   bf16x2 {e2:.2e}")
    assert e22 <= max(6.0 * e_lib, 2e-6), (e22, e_lib)        # 22 significant bits per operand
    assert e3 <= max(4.0 * e_lib, 1e-6) and e3 <= 3e-6, (e3, e_lib)   # fp32-level (measured: 0.8-2.4x hipBLASLt's fp32 error)
    assert e2 <= 2e-5, e2                                      # 16 significant bits per operand


def test_bias_relu_batch_strides_and_determinism(N):
    g = torch.Generator(device="cuda").manual_seed(7)
    G, M, K, Nn = 5, 300, 64, 200
    A = torch.randn(G, M, K + 8, generator=g, device="cuda")[:, :, :K]          # row stride K + 8
    W = torch.randn(G, Nn, K, generator=g, device="cuda")
    Wp = N.gemm_split_pack(W, terms=3)
    out = N.gemm_split(A, Wp)
    ref = _ref(A, W)
    assert (out.double() - ref).abs().max().item() <= 3e-6 * ref.abs().max().item()
    assert torch.equal(out, N.gemm_split(A, Wp))                                # bitwise reproducible
    # transposed weights (K, N): what the Winograd filter transform produces
    Wt = W.transpose(1, 2).contiguous()
    assert torch.equal(out, N.gemm_split(A, N.gemm_split_pack(Wt, trans=True, terms=3)))
    bias = torch.randn(Nn, generator=g, device="cuda")
    o2 = N.gemm_split(A[0], N.gemm_split_pack(W[0], terms=3), bias=bias, relu=True)
    r2 = _ref(A[0], W[0], bias, True)
    assert (o2.double() - r2).abs().max().item() <= 3e-6 * r2.abs().max().item() and o2.min().item() == 0.0
    # output into a wider buffer (ldc > N)
    buf = torch.zeros(M, Nn + 56, device="cuda")
    N.gemm_split(A[0], N.gemm_split_pack(W[0], terms=3), out=buf[:, :Nn])
    assert torch.equal(buf[:, :Nn], out[0]) and float(buf[:, Nn:].abs().max()) == 0.0


def test_rejects_bad_arguments(N):
    A = torch.randn(8, 48, device="cuda")
    with pytest.raises(N.SeaNativeError):
        N.gemm_split_pack(torch.randn(16, 48, device="cuda"))          # K % 32 != 0
    Wp = N.gemm_split_pack(torch.randn(16, 64, device="cuda"))
    with pytest.raises(N.SeaNativeError):
        N.gemm_split(A, Wp)                                            # K mismatch


def test_fp16x2_handles_extreme_scales(N):
    """fp16 has 5 exponent bits: the per-tensor / per-row power-of-two scaling must keep tiny and huge operands exact
    enough (relative to the result's scale) and must never overflow"""
    g = torch.Generator(device="cuda").manual_seed(3)
    for a_scale, w_scale in ((1e-6, 1.0), (3e4, 1.0), (1.0, 1e-5), (2e3, 5e2), (1e-20, 1e10)):
        A = torch.randn(512, 128, generator=g, device="cuda") * a_scale
        W = torch.randn(192, 128, generator=g, device="cuda") * w_scale
        W[5] *= 1e-4                                          # a weight row far below the others: its own scale
        ref = _ref(A, W)
        out = N.gemm_split(A, N.gemm_split_pack(W, terms=22))
        assert torch.isfinite(out).all()
        col = ref.abs().amax(0).clamp_min(1e-300)
        assert ((out.double() - ref).abs().amax(0) / col).max().item() <= 5e-6, (a_scale, w_scale)
    # batch of packed weights (the Winograd-domain products): one activation scale for the batch, one weight scale per row and g
    A = torch.randn(4, 300, 64, generator=g, device="cuda")
    W = torch.randn(4, 200, 64, generator=g, device="cuda") * torch.tensor([1.0, 1e-3, 50.0, 1e2], device="cuda").view(4, 1, 1)
    ref = _ref(A, W)
    out = N.gemm_split(A, N.gemm_split_pack(W, terms=22))
    for i in range(4):
        assert (out[i].double() - ref[i]).abs().max().item() <= 3e-6 * ref[i].abs().max().item()
