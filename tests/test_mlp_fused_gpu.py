"""M8f: the MLP of a ConvNeXt block as one kernel per direction (csrc/mlp_fused.hip), `-m gpu`.

Reference layer: semseg/models/backbones/convnext_orig.py:77-79 (pwconv1 -> GELU -> pwconv2, layer scale folded) and its
autograd input gradient.  The yardstick is the pair of M8 launches the kernel replaces -- `gemm_split` with the GELU / GELU'
prologues, the residual add and the per-row maxima pass: the fused kernels must give THE SAME BITS (same operand split, same
scales, same MFMA products in the same order along K), so every golden / teacher-forced / controller test of the models keeps
its meaning.  Accuracy against float64 is checked on top."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def N():
    """the native module with split-K OFF: a product that `gemm_split` cuts into K slices (few tiles, K >= 512: the second
    projection at C = 192 on small inputs) sums in another order than the single chain the fused kernel reproduces"""
    from semseg import _native
    _native.lib()
    old, _native.KSPLIT = _native.KSPLIT, False
    yield _native
    _native.KSPLIT = old


def _case(C, M, seed, row_spread=0.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    H = 4 * C
    x = torch.randn(M, C, generator=g, device="cuda")                      # a LayerNorm output
    w1 = torch.randn(H, C, generator=g, device="cuda") * 0.05
    b1 = torch.randn(H, generator=g, device="cuda") * 0.1
    w2 = torch.randn(C, H, generator=g, device="cuda") * 0.03
    b2 = torch.randn(C, generator=g, device="cuda") * 0.1
    res = torch.randn(M, C, generator=g, device="cuda")
    gy = torch.randn(M, C, generator=g, device="cuda") * 1e-3
    if row_spread:                                                          # gradient rows over many orders of magnitude
        gy = gy * torch.exp2(torch.rand(M, 1, generator=g, device="cuda") * -row_spread)
    return x, w1, b1, w2, b2, res, gy


def _word(v):
    return torch.tensor([float(v)], dtype=torch.float32, device="cuda").view(torch.int32)


def _unfused_forward(N, x, w1, b1, w2, b2, res, a1, a2):
    t = N.gemm_split(x, N.gemm_split_pack(w1, terms=22), bias=b1, amax=a1)
    y = N.gemm_split(t, N.gemm_split_pack(w2, terms=22), bias=b2, amax=a2, a_gelu=True)
    if res is not None:
        y += res
    return t, y


def _unfused_backward(N, gy, t, w1, w2, mul):
    M, C = gy.shape
    words, _ = N._amax_words(gy.unsqueeze(0), M, C, 1, 0, 1, per_row=True)
    u = N.gemm_split(gy, N.gemm_split_pack(w2, trans=True, terms=22), amax=words, amax_rows=1)
    return N.gemm_split(u, N.gemm_split_pack(w1, trans=True, terms=22), a_gelu_grad_of=t, amax=words, amax_rows=1, amax_mul=mul)


def _bounds(x, w1, b1, w2):
    a1 = _word(x.abs().max().item() * 1.7)                                  # any upper bound will do (the models use analytic ones)
    a2 = _word((x.abs().max() * w1.abs().sum(1) + b1.abs()).max().item())
    mul = (w2.abs().sum(0).max() * (1.13 * (1.0 + 1e-6))).float().reshape(1)
    return a1, a2, mul


@pytest.mark.parametrize("C,M", [(96, 2048), (96, 1000), (96, 257), (96, 31), (192, 1024), (192, 777), (192, 130)])
def test_fused_mlp_gives_the_bits_of_the_two_gemm_form(N, C, M):
    x, w1, b1, w2, b2, res, gy = _case(C, M, 11 * C + M)
    a1, a2, mul = _bounds(x, w1, b1, w2)
    t, y_ref = _unfused_forward(N, x, w1, b1, w2, b2, res, a1, a2)
    P = lambda w, tr=False: N.gemm_split_pack(w, trans=tr, terms=22)       # noqa: E731
    y = N.mlp_fused_forward(x, P(w1), b1, P(w2), b2, res, a1, a2)
    assert torch.equal(y, y_ref), (y - y_ref).abs().max().item()
    # without residual / bias of the second projection
    _, y0_ref = _unfused_forward(N, x, w1, b1, w2, None, None, a1, a2)
    assert torch.equal(N.mlp_fused_forward(x, P(w1), b1, P(w2), None, None, a1, a2), y0_ref)
    gx_ref = _unfused_backward(N, gy, t, w1, w2, mul)
    gx = N.mlp_fused_backward(gy, x, P(w1), b1, P(w2, True), P(w1, True), a1, mul)
    assert torch.equal(gx, gx_ref), (gx - gx_ref).abs().max().item()
    # bitwise reproducible, and rows do not depend on their neighbours (a shard of the rows gives the same rows)
    assert torch.equal(N.mlp_fused_forward(x, P(w1), b1, P(w2), b2, res, a1, a2), y)
    k = min(M, 64)
    assert torch.equal(N.mlp_fused_forward(x[:k].contiguous(), P(w1), b1, P(w2), b2, res[:k].contiguous(), a1, a2), y[:k])
    assert torch.equal(N.mlp_fused_backward(gy[:k].contiguous(), x[:k].contiguous(), P(w1), b1, P(w2, True), P(w1, True), a1, mul),
                       gx[:k])


@pytest.mark.parametrize("C,M", [(96, 2048), (96, 333), (192, 1024), (192, 97)])
def test_layernorm_inside_the_fused_mlp_gives_the_bits_of_the_layernorm_kernel(N, C, M):
    """the block's LayerNorm in the kernels' prologue (and its input gradient in the backward's epilogue): the channel sums are
    taken in the order of sea_layernorm_fwd / _bwd, so the result equals LayerNorm kernel + fused MLP bit for bit"""
    x, w1, b1, w2, b2, res, gy = _case(C, M, 7 * C + M)
    g = torch.Generator(device="cuda").manual_seed(C)
    x = x * (torch.rand(M, 1, generator=g, device="cuda") * 4 + 0.1) + torch.randn(M, 1, generator=g, device="cuda")   # rows with their own mean / scale
    lw = torch.rand(C, generator=g, device="cuda") + 0.5
    lb = torch.randn(C, generator=g, device="cuda") * 0.1
    eps = 1e-6
    yn, mean, rstd = N.layernorm(x, lw, lb, eps)
    a1, a2, mul = _bounds(yn, w1, b1, w2)
    P = lambda w, tr=False: N.gemm_split_pack(w, trans=tr, terms=22)       # noqa: E731
    y_ref = N.mlp_fused_forward(yn, P(w1), b1, P(w2), b2, res, a1, a2)
    y = N.mlp_fused_forward(x, P(w1), b1, P(w2), b2, res, a1, a2, ln=(lw, lb, eps))
    assert torch.equal(y, y_ref), (y - y_ref).abs().max().item()
    gyn = N.mlp_fused_backward(gy, yn, P(w1), b1, P(w2, True), P(w1, True), a1, mul)
    gx_ref = N.layernorm_backward(gyn, x, lw, mean, rstd)
    gx = N.mlp_fused_backward(gy, x, P(w1), b1, P(w2, True), P(w1, True), a1, mul, ln=(lw, lb, eps))
    assert torch.equal(gx, gx_ref), (gx - gx_ref).abs().max().item()


@pytest.mark.parametrize("C", [96, 192])
def test_fused_mlp_gradient_rows_over_many_orders_of_magnitude(N, C):
    """the per-row scales of the gradient operand come from the registers that hold the row: rows spread over 2^60 keep their
    relative accuracy, and the bits are still those of sea_absmax_bits(rows_per_word = 1) + the two launches"""
    M = 1536
    x, w1, b1, w2, b2, res, gy = _case(C, M, 5 + C, row_spread=60.0)
    a1, a2, mul = _bounds(x, w1, b1, w2)
    t, _ = _unfused_forward(N, x, w1, b1, w2, b2, res, a1, a2)
    P = lambda w, tr=False: N.gemm_split_pack(w, trans=tr, terms=22)       # noqa: E731
    gx = N.mlp_fused_backward(gy, x, P(w1), b1, P(w2, True), P(w1, True), a1, mul)
    assert torch.equal(gx, _unfused_backward(N, gy, t, w1, w2, mul))
    td = x.double() @ w1.double().t() + b1.double()
    cdf = 0.5 * (1 + torch.erf(td / 2 ** 0.5))
    pdf = torch.exp(-0.5 * td * td) / (2 * torch.pi) ** 0.5
    ref = ((gy.double() @ w2.double()) * (cdf + td * pdf)) @ w1.double()
    err = ((gx.double() - ref).abs().amax(1) / ref.abs().amax(1).clamp_min(1e-300)).max().item()
    print(f"C={C}: max row-relative error of the fused input gradient against float64: {err:.2e}")
    assert err <= 5e-6, err


@pytest.mark.parametrize("C", [96, 192])
def test_fused_mlp_accuracy_against_float64(N, C):
    M = 4096
    x, w1, b1, w2, b2, res, gy = _case(C, M, 3 * C)
    a1, a2, mul = _bounds(x, w1, b1, w2)
    P = lambda w, tr=False: N.gemm_split_pack(w, trans=tr, terms=22)       # noqa: E731
    y = N.mlp_fused_forward(x, P(w1), b1, P(w2), b2, res, a1, a2)
    td = x.double() @ w1.double().t() + b1.double()
    ref = res.double() + (0.5 * td * (1 + torch.erf(td / 2 ** 0.5))) @ w2.double().t() + b2.double()
    lib = res + torch.nn.functional.gelu(x @ w1.t() + b1) @ w2.t() + b2
    e, e_lib = (y.double() - ref).abs().max().item(), (lib.double() - ref).abs().max().item()
    print(f"C={C}: max |y - float64|  fused fp16x2 {e:.2e}   PyTorch-ROCm fp32 composition {e_lib:.2e}")
    assert e <= max(2.0 * e_lib, 2e-6)


@pytest.mark.parametrize("C,hw", [(96, 32), (192, 24), (192, 116)])
def test_block_takes_the_fused_kernels_and_keeps_its_bits(N, C, hw):
    """through the model's own Block (frozen weights, channels_last trunk): SEA_MLP_FUSED on / off give the same output and the
    same input gradient bit for bit, and the fused path keeps x instead of the 4C-wide t for the backward"""
    from semseg.models import convnext_upernet as M
    torch.manual_seed(C)
    blk = M.Block(C).cuda().eval()
    with torch.no_grad():
        blk.gamma.mul_(0.7)
        blk.pwconv1.bias.normal_(0, 0.1)
    for p in blk.parameters():
        p.requires_grad_(False)
    x = torch.randn(2, C, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
    gy = torch.randn(2, C, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
    outs = []
    calls = []
    real = N.mlp_fused_forward
    N.mlp_fused_forward = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    old = (N.USE_MLP_FUSED, M.FUSE_LN_INTO_MLP)
    try:
        # the two-GEMM form; the fused MLP behind the LayerNorm kernel; LayerNorm + MLP in one kernel (the default)
        for fused, ln_in in ((False, False), (True, False), (True, True)):
            N.USE_MLP_FUSED, M.FUSE_LN_INTO_MLP = fused, ln_in
            xi = x.clone().requires_grad_(True)
            y = blk(xi)
            (gx,) = torch.autograd.grad(y, xi, gy)
            outs.append((y.detach().clone(), gx.clone()))
    finally:
        (N.USE_MLP_FUSED, M.FUSE_LN_INTO_MLP), N.mlp_fused_forward = old, real
    assert len(calls) == 2, "the Block did not reach the fused kernel"
    for o in outs[1:]:
        assert torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1])


def test_branch_free_gelu_is_the_library_gelu_for_every_float(N):
    """the fused kernels evaluate erf without the device library's per-element branch (both sides + select): the same
    operations on the same constants, compared here bit for bit against gelu_f / gelu_grad_f over all 2^32 inputs"""
    out = torch.zeros(2, dtype=torch.int64, device="cuda")
    N._check(N.lib().sea_probe_gelu_mismatches(out.data_ptr(), N._stream()), "sea_probe_gelu_mismatches")
    assert out.tolist() == [0, 0], out.tolist()
