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
    assert e22 <= max(2.0 * e_lib, 2e-6), (e22, e_lib)        # 22 significant bits per operand: measured 0.3-1.5 x the fp32 GEMM's own error
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


def test_fp16x2_activation_scale_from_the_producer(N):
    """``amax`` = any upper bound of max|A| held in a device word (float bits) replaces the pass over A; ``out_amax``
    receives the bits of max|out| for the next GEMM.  A loose bound must not cost accuracy (fp16 is floating point: only
    the sub-normal range moves), a chained pair of GEMMs with GELU in between must match the explicit-pass result."""
    g = torch.Generator(device="cuda").manual_seed(11)
    A = torch.randn(3000, 96, generator=g, device="cuda") * 2.5
    W1 = torch.randn(384, 96, generator=g, device="cuda") / 96 ** 0.5
    W2 = torch.randn(96, 384, generator=g, device="cuda") / 384 ** 0.5
    b1 = torch.randn(384, generator=g, device="cuda")
    P1, P2 = N.gemm_split_pack(W1, terms=22), N.gemm_split_pack(W2, terms=22)
    ref = _ref(A, W1, b1)
    bits = lambda v: torch.tensor([v], dtype=torch.float32, device="cuda").view(torch.int32)
    base = N.gemm_split(A, P1, bias=b1)
    for slack in (1.0, 1.7, 37.0, 1000.0):
        word = N.amax_word(A.device)
        out = N.gemm_split(A, P1, bias=b1, amax=bits(A.abs().max().item() * slack), out_amax=word)
        err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
        assert err <= 2e-6, (slack, err)
        assert word.view(torch.float32).item() == out.abs().max().item()          # exact: a max of the stored values
        if slack == 1.0:
            assert torch.equal(out, base)
    h = torch.nn.functional.gelu(out)
    assert h.abs().max().item() <= word.view(torch.float32).item()
    y_chain = N.gemm_split(h, P2, amax=word)
    y_pass = N.gemm_split(h, P2)
    ref2 = _ref(h, W2)
    for y in (y_chain, y_pass):
        assert (y.double() - ref2).abs().max().item() <= 2e-6 * ref2.abs().max().item()
    # the pool hands out zeroed words, and a reset zeroes the ones in use again
    pool = N.AmaxPool.get(A.device)
    pool.reset()
    w = N.amax_word(A.device)
    assert w.item() == 0
    N.gemm_split(A, P1, amax=bits(20.0), out_amax=w)
    assert w.item() != 0
    pool.reset()
    assert w.item() == 0


@pytest.mark.parametrize("terms", [22, 3, 2])
def test_fused_epilogue_addend_gelu_and_gelu_grad(N, terms):
    """sea_gemm_split_fused: residual addend, second output GELU(C), factor GELU'(t) -- against the composed ATen ops on the
    plain sea_gemm_split result (same products, same order: the GEMM part is bitwise the same)"""
    import torch.nn.functional as F
    g = torch.Generator(device="cuda").manual_seed(21)
    M, K, Nn = 1000, 96, 384                                              # ragged M, N = 3 column blocks
    A = torch.randn(M, K, generator=g, device="cuda") * 2
    W = torch.randn(Nn, K, generator=g, device="cuda") / K ** 0.5
    b = torch.randn(Nn, generator=g, device="cuda")
    res = torch.randn(M, Nn, generator=g, device="cuda")
    t = torch.randn(M, Nn, generator=g, device="cuda") * 2
    P = N.gemm_split_pack(W, terms=terms)
    base = N.gemm_split(A, P, bias=b)
    # addend
    out = N.gemm_split(A, P, bias=b, addend=res)
    torch.testing.assert_close(out, base + res, rtol=0, atol=2e-6 * base.abs().max().item())
    # addend read through a row stride (a column slice of a wider tensor)
    wide = torch.randn(M, Nn + 32, generator=g, device="cuda")
    out = N.gemm_split(A, P, bias=b, addend=wide[:, 16:16 + Nn])
    torch.testing.assert_close(out, base + wide[:, 16:16 + Nn], rtol=0, atol=2e-6 * base.abs().max().item())
    # dual output: C = pre-activation (bitwise the plain GEMM), gelu_out = GELU(C)
    h = torch.empty_like(base)
    word = N.amax_word(A.device) if terms == 22 else None
    out = N.gemm_split(A, P, bias=b, gelu_out=h, out_amax=word)
    assert torch.equal(out, base)
    torch.testing.assert_close(h, F.gelu(base), rtol=1e-5, atol=1e-6)
    if word is not None:
        assert word.view(torch.float32).item() == base.abs().max().item()
    # GELU' factor: the backward of y = GELU(t) applied to the GEMM result
    out = N.gemm_split(A, P, gelu_grad_of=t)
    tt = t.clone().requires_grad_(True)
    (want,) = torch.autograd.grad(F.gelu(tt), tt, N.gemm_split(A, P))
    torch.testing.assert_close(out, want, rtol=1e-5, atol=1e-6 * want.abs().max().item())
    # batch of packed weights with all extras' batch strides
    A3 = torch.randn(3, 260, 64, generator=g, device="cuda")
    W3 = torch.randn(3, 130, 64, generator=g, device="cuda") / 8
    r3 = torch.randn(3, 260, 130, generator=g, device="cuda")
    P3 = N.gemm_split_pack(W3, terms=terms)
    h3 = torch.empty(3, 260, 130, device="cuda")
    o3 = N.gemm_split(A3, P3, addend=r3, gelu_out=h3)
    b3 = N.gemm_split(A3, P3)
    torch.testing.assert_close(o3, b3 + r3, rtol=0, atol=2e-6 * b3.abs().max().item())
    torch.testing.assert_close(h3, F.gelu(o3), rtol=1e-5, atol=1e-6)
    # prologue: A read as GELU(A) while it is staged
    got = N.gemm_split(A, P, bias=b, a_gelu=True)
    plain = N.gemm_split(F.gelu(A), P, bias=b)
    want = _ref(F.gelu(A.double()).float(), W, b)
    e_got, e_plain = ((got.double() - want).abs().max().item(), (plain.double() - want).abs().max().item())
    assert e_got <= 1.5 * e_plain + 2e-6 * want.abs().max().item(), (e_got, e_plain)
    # prologue: A read as A * GELU'(s) while it is staged (terms 2 / 22)
    if terms in (2, 22):
        sA = torch.randn(M, K, generator=g, device="cuda") * 2
        ss = sA.clone().requires_grad_(True)
        (Ag,) = torch.autograd.grad(F.gelu(ss), ss, A)
        want = _ref(Ag, W)
        got = N.gemm_split(A, P, a_gelu_grad_of=sA)
        plain = N.gemm_split(Ag, P)
        e_got, e_plain = ((got.double() - want).abs().max().item(), (plain.double() - want).abs().max().item())
        assert e_got <= 1.5 * e_plain + 2e-6 * want.abs().max().item(), (e_got, e_plain)
        wideA, wideS = torch.zeros(M, K + 32, device="cuda"), torch.zeros(M, K + 32, device="cuda")
        wideA[:, 16:16 + K], wideS[:, 16:16 + K] = A, sA
        assert torch.equal(N.gemm_split(wideA[:, 16:16 + K], P, a_gelu_grad_of=wideS[:, 16:16 + K]), got)
        # prologue: ReLU gate (the backward of a fused GEMM + ReLU): bitwise the GEMM on the masked copy
        gate = torch.randn(M, K, generator=g, device="cuda")
        assert torch.equal(N.gemm_split(A, P, a_relu_gate=gate), N.gemm_split(torch.where(gate > 0, A, torch.zeros_like(A)), P))
    else:
        with pytest.raises(N.SeaNativeError):
            N.gemm_split(A, P, a_gelu_grad_of=A)
    with pytest.raises(N.SeaNativeError):
        N.gemm_split(A, P, gelu_out=h, gelu_grad_of=t)                    # one or the other
    with pytest.raises(N.SeaNativeError):
        N.gemm_split(A, P, addend=res[:, :-1])


@pytest.mark.parametrize("terms", [22, 3, 2])
def test_split_k_for_small_tile_grids(N, terms):
    """few 128 x 128 tiles and a long K: the product runs as a batch of K slices + sea_gemm_splitk_reduce.  Same accuracy
    as the single-pass product, bitwise reproducible, bias / ReLU / max|C| word in the reduce pass."""
    g = torch.Generator(device="cuda").manual_seed(31)
    for M, K, Nn, trans in ((2048, 3072, 768, False), (8192, 1536, 384, True), (2048, 4608, 512, True), (1300, 768, 132, False)):
        A = torch.randn(M, K, generator=g, device="cuda")
        W = torch.randn(Nn, K, generator=g, device="cuda") / K ** 0.5
        b = torch.randn(Nn, generator=g, device="cuda")
        S = N._ksplit(M, Nn, K)
        assert S > 1 and K % (32 * S) == 0, (M, K, Nn, S)
        P = N.gemm_split_pack(W.t().contiguous() if trans else W, trans=trans, terms=terms)
        ref = _ref(A, W, b, relu=True)
        word = N.amax_word(A.device) if terms == 22 else None
        out = N.gemm_split(A, P, bias=b, relu=True, out_amax=word)
        assert S in P.slices                                                   # the split path ran
        N.KSPLIT = False
        try:
            one = N.gemm_split(A, P, bias=b, relu=True)
        finally:
            N.KSPLIT = True
        scale = ref.abs().max().item()
        e_split, e_one = (out.double() - ref).abs().max().item() / scale, (one.double() - ref).abs().max().item() / scale
        print(f"M={M} K={K} N={Nn} terms {terms}: {S} K slices, max err / max|C| split {e_split:.2e}  single pass {e_one:.2e}")
        assert e_split <= 1.5 * e_one + 2e-7
        assert torch.equal(out, N.gemm_split(A, P, bias=b, relu=True))         # fixed summation order
        if word is not None:
            assert word.view(torch.float32).item() == out.abs().max().item()
        # the reduce pass adds a residual; prologues apply per K slice
        res = torch.randn(M, Nn, generator=g, device="cuda")
        torch.testing.assert_close(N.gemm_split(A, P, bias=b, addend=res), N.gemm_split(A, P, bias=b) + res, rtol=0,
                                   atol=1e-6 * scale)
        ga, gb_ = N.gemm_split(A, P, a_gelu=True), N.gemm_split(torch.nn.functional.gelu(A), P)
        torch.testing.assert_close(ga, gb_, rtol=0, atol=3e-6 * gb_.abs().max().item() if terms != 2 else 2e-5 * gb_.abs().max().item())
        # a row-strided output (column slice of a wider tensor)
        wide = torch.zeros(M, Nn + 64, device="cuda")
        N.gemm_split(A, P, bias=b, relu=True, out=wide[:, 32:32 + Nn])
        assert torch.equal(wide[:, 32:32 + Nn], out) and not wide[:, :32].any() and not wide[:, 32 + Nn:].any()
    assert N._ksplit(131072, 384, 96) == 1 and N._ksplit(2048, 768, 128) == 1 and N._ksplit(2048, 770, 3072) == 1


def test_fp16x2_per_image_scales_make_rows_independent_of_their_batch(N):
    """the activation scale is a power of two PER ROW, from the word of the row's group (image / Winograd tile): an
    image's rows give the same bits whatever shares the batch with them.  (A per-tensor scale moves the sub-normal cut-off
    of the low fp16 term: measured to break the sharded evaluation's bitwise 1-rank == 2-rank property.)"""
    g = torch.Generator(device="cuda").manual_seed(17)
    K, Nn, rows = 256, 192, 1536
    W = torch.randn(Nn, K, generator=g, device="cuda") / K ** 0.5
    P = N.gemm_split_pack(W, terms=22)
    a0 = torch.randn(rows, K, generator=g, device="cuda") * torch.rand(rows, 1, generator=g, device="cuda") ** 8   # many tiny rows
    alone = N.gemm_split(a0, P)
    differ = 0
    for partner_scale in (1.0, 37.0, 4096.0, 1e-3):
        a1 = torch.randn(rows, K, generator=g, device="cuda") * partner_scale
        pair = torch.cat([a0, a1])
        out = N.gemm_split(pair, P, groups=2)                        # one exact word per image (sea_absmax_bits, rows_per_word)
        assert torch.equal(out[:rows], alone), partner_scale
        assert torch.equal(out[rows:], N.gemm_split(a1, P))
        differ += int(not torch.equal(N.gemm_split(pair, P, groups=1)[:rows], alone))
        # explicit words: per image, and per row (amax_rows = 1: what the Winograd transform supplies per tile)
        w_img = torch.stack([a0.abs().max(), a1.abs().max()]).view(torch.int32)
        assert torch.equal(N.gemm_split(pair, P, amax=w_img, amax_rows=rows), out)
        w_row = pair.abs().amax(1).contiguous().view(torch.int32)
        per_row = N.gemm_split(pair, P, amax=w_row, amax_rows=1)
        assert torch.equal(per_row[:rows], N.gemm_split(a0, P, amax=w_row[:rows].contiguous(), amax_rows=1))
        ref = _ref(pair, W)
        assert (per_row.double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
    print(f"per-tensor scale changed image 0's bits for {differ} of 4 partners")
    # the grouped maxima themselves, dense and row-strided, batch of matrices
    A3 = torch.randn(3, 1000, 64, generator=g, device="cuda")
    for rpw in (250, 128, 1):
        words = torch.empty(-(-1000 // rpw), dtype=torch.int32, device="cuda")
        N._check(N.lib().sea_absmax_bits(A3.data_ptr(), 64, 1000, 64, 3, 1000 * 64, rpw, words.data_ptr(), N._stream()), "absmax")
        want = torch.stack([A3[:, i:i + rpw].abs().max() for i in range(0, 1000, rpw)])
        assert torch.equal(words.view(torch.float32), want)
    wide = torch.randn(1000, 96, generator=g, device="cuda")
    words = torch.empty(4, dtype=torch.int32, device="cuda")
    N._check(N.lib().sea_absmax_bits(wide[:, 16:].data_ptr(), 96, 1000, 64, 1, 0, 250, words.data_ptr(), N._stream()), "absmax")
    assert torch.equal(words.view(torch.float32), torch.stack([wide[i:i + 250, 16:80].abs().max() for i in range(0, 1000, 250)]))
    with pytest.raises(N.SeaNativeError):
        N.gemm_split(a0, P, amax=w_img, amax_rows=1)                 # too few words


@pytest.mark.parametrize("M,K,lda", [(131072, 96, 96), (5000, 16, 16), (777, 192, 200), (4096, 384, 384), (1031, 512, 512),
                                     (2048, 4608, 4608), (300, 768, 1024)])
def test_row_maxima_words_are_exact(N, M, K, lda):
    """sea_absmax_bits(rows_per_word = 1): one word per row = the float bits of max|row|, exactly (incl. -0, denormals, a row
    of zeros, a strided operand); every word has one writer, so an un-initialised output buffer is fine"""
    g = torch.Generator(device="cuda").manual_seed(M + K)
    buf = torch.randn(M, lda, generator=g, device="cuda") * torch.exp(8 * torch.randn(M, 1, generator=g, device="cuda"))
    buf[3] = 0
    buf[5, :K] = 1e-41
    A = buf[:, :K]
    words, rows = N._amax_words(A.unsqueeze(0), M, K, 1, 0, 1, per_row=True)
    assert rows == 1 and words.shape == (M,)
    assert torch.equal(words, A.abs().amax(1).view(torch.int32))


def test_fp16x2_per_row_scales_keep_every_gradient_row_at_fp32_level(N):
    """The input-gradient products (round 4: fp16 x 2 by default).  A gradient's rows (pixels) span many orders of magnitude:
    with ONE scale per image the small rows fall into fp16's sub-normal range and lose their bits; with one scale per row
    (row_amax) every row is as accurate, relative to its OWN magnitude, as the fp32 GEMM"""
    g = torch.Generator(device="cuda").manual_seed(11)
    M, K, Nn = 8192, 384, 1536
    mag = torch.exp2(torch.randint(-60, 10, (M, 1), generator=g, device="cuda").float())      # 2^-60 .. 2^10 per row
    A = torch.randn(M, K, generator=g, device="cuda") * mag
    W = torch.randn(Nn, K, generator=g, device="cuda") / K ** 0.5
    ref = _ref(A, W)
    row_scale = ref.abs().amax(1, keepdim=True)

    def row_err(out):
        return ((out.double() - ref).abs() / row_scale).amax().item()

    Wp = N.gemm_split_pack(W, terms=22)
    e_lib = row_err(A @ W.t())
    e_row = row_err(N.gemm_split(A, Wp, row_amax=True))
    e_img = row_err(N.gemm_split(A, Wp, groups=8))
    e_b2 = row_err(N.gemm_split(A, N.gemm_split_pack(W, terms=2)))
    print(f"max row-relative error: hipBLASLt fp32 {e_lib:.2e}   fp16x2 per-row scales {e_row:.2e}   "
          f"fp16x2 per-image scales {e_img:.2e}   bf16x2 {e_b2:.2e}")
    assert e_row <= max(4.0 * e_lib, 2e-6), (e_row, e_lib)
    assert e_img > 1e-3            # what per-row scales are for
    # the result of a row does not depend on the other rows of the batch
    sub = A[1000:1256].contiguous()
    assert torch.equal(N.gemm_split(sub, Wp, row_amax=True), N.gemm_split(A, Wp, row_amax=True)[1000:1256])


def test_fp16x2_row_bound_carried_through_a_gemm(N):
    """amax_mul: the second input-gradient product of an MLP takes its row scales from the FIRST product's operand,
    |(g W2)[r]| <= rowmax(g[r]) * max_k sum_n |W2[n][k]| -- no pass over the 4C-wide intermediate.  The bound is loose by
    the ratio of the l1 norm to the realised maximum (3-30x), which fp16 x 2 does not notice"""
    from semseg.models import convnext_upernet as M
    g = torch.Generator(device="cuda").manual_seed(12)
    R, C = 4096, 192
    W2 = torch.randn(C, 4 * C, generator=g, device="cuda") * 0.05          # Linear(4C -> C).weight
    W1 = torch.randn(4 * C, C, generator=g, device="cuda") * 0.05          # Linear(C -> 4C).weight
    gy = torch.randn(R, C, generator=g, device="cuda") * torch.exp2(torch.randint(-40, 4, (R, 1), generator=g, device="cuda").float())
    t = torch.randn(R, 4 * C, generator=g, device="cuda")
    words, _ = N._amax_words(gy.unsqueeze(0), R, C, 1, 0, 1, per_row=True)
    mul_dev = M._l1_bound(W2, {}, dim=0, factor=1.13)        # one float32 on the device (what the model hands over)
    mul = float(mul_dev.item())
    u = N.gemm_split(gy, N.gemm_split_pack(W2, trans=True, terms=22), amax=words, amax_rows=1)
    loose = (words.view(torch.float32) * mul) / (u * torch.ops.aten.gelu_backward(torch.ones_like(u), t)).abs().amax(1)
    assert loose.min() >= 1.0, loose.min()                # it IS a bound
    print(f"row bound / realised row maximum: median {loose.median():.1f}, max {loose.max():.1f}")
    got = N.gemm_split(u, N.gemm_split_pack(W1, trans=True, terms=22), a_gelu_grad_of=t, amax=words, amax_rows=1, amax_mul=mul)
    got_dev = N.gemm_split(u, N.gemm_split_pack(W1, trans=True, terms=22), a_gelu_grad_of=t, amax=words, amax_rows=1,
                           amax_mul=mul_dev)
    assert torch.equal(got, got_dev)                          # host float and device float: the same scales
    ref = (u.double() * torch.ops.aten.gelu_backward(torch.ones_like(u), t).double()) @ W1.double()
    lib = (u * torch.ops.aten.gelu_backward(torch.ones_like(u), t)) @ W1
    rs = ref.abs().amax(1, keepdim=True)
    e_got, e_lib = ((got.double() - ref).abs() / rs).amax().item(), ((lib.double() - ref).abs() / rs).amax().item()
    print(f"max row-relative error: hipBLASLt fp32 {e_lib:.2e}   fp16x2 with carried row bounds {e_got:.2e}")
    assert e_got <= max(4.0 * e_lib, 2e-6), (e_got, e_lib)


def test_mfma_shape_16_variants_match_shape_32(N):
    """sea_gemm_split_mfma_shape(16): every kernel variant (three modes, bias / ReLU, the three prologues, the fused epilogue,
    out_amax) on v_mfma_f32_16x16x32_* fragments gives the 32x32x16 result up to summation order, and the float64 reference
    to the same accuracy"""
    g = torch.Generator(device="cuda").manual_seed(21)
    M, K, Nn = 1000, 192, 200
    A = torch.randn(M, K, generator=g, device="cuda") * torch.exp2(torch.randint(-8, 3, (M, 1), generator=g, device="cuda").float())
    W = torch.randn(Nn, K, generator=g, device="cuda") / K ** 0.5
    bias = torch.randn(Nn, generator=g, device="cuda")
    t = torch.randn(M, K, generator=g, device="cuda")
    add = torch.randn(M, Nn, generator=g, device="cuda")

    def all_variants():
        outs = {}
        for terms in (22, 3, 2):
            Wp = N.gemm_split_pack(W, terms=terms)
            kw = dict(row_amax=True) if terms == 22 else {}
            outs[terms, "plain"] = N.gemm_split(A, Wp, bias=bias, relu=True, **kw)
            outs[terms, "gelu"] = N.gemm_split(A, Wp, a_gelu=True, **kw)
            outs[terms, "addend"] = N.gemm_split(A, Wp, bias=bias, addend=add, **kw)
            if terms != 3:
                outs[terms, "gelu_grad"] = N.gemm_split(A, Wp, a_gelu_grad_of=t, **kw)
                outs[terms, "gate"] = N.gemm_split(A, Wp, a_relu_gate=t, **kw)
        word = N.amax_word(A.device)
        outs[22, "out_amax"] = N.gemm_split(A, N.gemm_split_pack(W, terms=22), out_amax=word).clone()
        outs[22, "out_amax_word"] = word.clone().view(torch.float32)
        return outs

    L = N.lib()
    assert L.sea_gemm_split_mfma_shape(-1) in (16, 32)
    prev = L.sea_gemm_split_mfma_shape(32)
    try:
        o32 = all_variants()
        assert L.sea_gemm_split_mfma_shape(16) == 32
        o16 = all_variants()
    finally:
        L.sea_gemm_split_mfma_shape(prev)
    ref = {"plain": _ref(A, W, bias, True), "gelu": _ref(torch.nn.functional.gelu(A.double()), W),
           "addend": _ref(A, W, bias) + add.double(),
           "gelu_grad": _ref(A.double() * torch.ops.aten.gelu_backward(torch.ones_like(t), t).double(), W),
           "gate": _ref(torch.where(t > 0, A, torch.zeros_like(A)), W)}
    for (terms, name), a in o32.items():
        b = o16[terms, name]
        if name == "out_amax_word":
            assert torch.equal(a, b) or abs(a.item() - b.item()) <= 1e-5 * a.item()
            continue
        r = ref.get(name, _ref(A, W))
        scale = r.abs().max().item()
        tol = 3e-5 if terms == 2 else 3e-6
        assert (a - b).abs().max().item() <= tol * scale, (terms, name, (a - b).abs().max().item() / scale)
        assert (b.double() - r).abs().max().item() <= tol * scale, (terms, name)


def test_one_bf16_term_is_the_bf16_autocast_product_with_fp32_io(N):
    """terms = 1 (BASELINE configs[3]: the attack under bf16 autocast): exactly the product of the bf16-ROUNDED operands,
    accumulated in fp32 -- what autocast's GEMM computes, without its 16-bit output rounding"""
    g = torch.Generator(device="cuda").manual_seed(31)
    M, K, Nn = 2048, 384, 200
    A = torch.randn(M, K, generator=g, device="cuda")
    W = torch.randn(Nn, K, generator=g, device="cuda") / K ** 0.5
    bias = torch.randn(Nn, generator=g, device="cuda")
    t = torch.randn(M, K, generator=g, device="cuda")
    Wp = N.gemm_split_pack(W, terms=1)
    rb = lambda x: x.bfloat16().double()
    ref = rb(A) @ rb(W).t() + bias.double()
    scale = ref.abs().max().item()
    out = N.gemm_split(A, Wp, bias=bias)
    assert (out.double() - ref).abs().max().item() <= 2e-6 * scale
    assert (out.double() - _ref(A, W, bias)).abs().max().item() <= 2e-2 * scale      # bf16 level against the exact product
    # prologues: applied in fp32 BEFORE the rounding to bf16 (the kernel's erf / exp differ from ATen's in the last bits, which
    # moves a few bf16 roundings by one step: bf16-level tolerance for the two GELU forms, exact for the gate)
    ref_g = rb(torch.nn.functional.gelu(A)) @ rb(W).t()
    assert (N.gemm_split(A, Wp, a_gelu=True).double() - ref_g).abs().max().item() <= 2e-3 * ref_g.abs().max().item()
    ref_d = rb(A * torch.ops.aten.gelu_backward(torch.ones_like(t), t)) @ rb(W).t()
    assert (N.gemm_split(A, Wp, a_gelu_grad_of=t).double() - ref_d).abs().max().item() <= 2e-3 * ref_d.abs().max().item()
    ref_r = rb(torch.where(t > 0, A, torch.zeros_like(A))) @ rb(W).t()
    assert (N.gemm_split(A, Wp, a_relu_gate=t).double() - ref_r).abs().max().item() <= 2e-6 * ref_r.abs().max().item()
    # batched (the Winograd-domain products) and bitwise reproducible
    A3 = torch.randn(5, 300, 64, generator=g, device="cuda")
    W3 = torch.randn(5, 130, 64, generator=g, device="cuda")
    o3 = N.gemm_split(A3, N.gemm_split_pack(W3, terms=1))
    r3 = torch.einsum("gmk,gnk->gmn", rb(A3), rb(W3))
    assert (o3.double() - r3).abs().max().item() <= 2e-6 * r3.abs().max().item()
    assert torch.equal(o3, N.gemm_split(A3, N.gemm_split_pack(W3, terms=1)))


@pytest.mark.parametrize("terms", [22, 2])
def test_prologue_with_addend_on_a_product_that_does_not_split(N, terms):
    """GELU-prologue + residual addend travel together only through split-K (the reduce pass adds).  On a product whose
    tile grid is large enough NOT to split, the pair used to reach sea_gemm_split_fused, which rejects it (ADVICE, round 3):
    gemm_split now adds the residual separately; same numbers as the two steps done by hand"""
    g = torch.Generator(device="cuda").manual_seed(41)
    M, K, Nn = 16384, 128, 512           # 128 x 4 tiles >= KSPLIT_BELOW: no split-K
    assert N._ksplit(M, Nn, K) == 1
    A = torch.randn(M, K, generator=g, device="cuda")
    W = torch.randn(Nn, K, generator=g, device="cuda") / K ** 0.5
    res = torch.randn(M, Nn, generator=g, device="cuda")
    Wp = N.gemm_split_pack(W, terms=terms)
    got = N.gemm_split(A, Wp, a_gelu=True, addend=res, groups=4)
    want = N.gemm_split(A, Wp, a_gelu=True, groups=4) + res
    assert torch.equal(got, want)


@pytest.mark.parametrize("M,K,Nn", [(1000, 192, 200), (300, 96, 130), (129, 32, 21), (2048, 512, 512), (517, 1024, 384)])
def test_pingpong_pipeline_gives_the_bits_of_the_single_stage_loop(N, M, K, Nn):
    """sea_gemm_split_pipeline(1) (two LDS stages, one barrier per K step, staging in the MFMA shadow: csrc/gemm_split_pp.hip)
    against pipeline 0 (csrc/gemm_split.hip): same operand split, same MFMA order per accumulator -> bit-identical outputs
    for every mode it serves (fp16 x 2, bf16 x 2, one bf16 term), every prologue, the fused epilogue, out_amax, batches,
    odd / even / single K-step counts and ragged M, N"""
    g = torch.Generator(device="cuda").manual_seed(M + K + Nn)
    A = torch.randn(M, K, generator=g, device="cuda") * torch.exp2(torch.randint(-8, 3, (M, 1), generator=g, device="cuda").float())
    W = torch.randn(Nn, K, generator=g, device="cuda") / K ** 0.5
    bias = torch.randn(Nn, generator=g, device="cuda")
    t = torch.randn(M, K, generator=g, device="cuda")
    add = torch.randn(M, Nn, generator=g, device="cuda")
    pre = torch.randn(M, Nn, generator=g, device="cuda")
    A3 = torch.randn(3, M, K, generator=g, device="cuda")
    W3 = torch.randn(3, Nn, K, generator=g, device="cuda") / K ** 0.5

    def all_variants():
        outs = {}
        for terms in (22, 2, 1):
            Wp = N.gemm_split_pack(W, terms=terms)
            kw = dict(row_amax=True) if terms == 22 else {}
            outs[terms, "plain"] = N.gemm_split(A, Wp, bias=bias, relu=True, **kw)
            outs[terms, "gelu"] = N.gemm_split(A, Wp, a_gelu=True, **kw)
            outs[terms, "addend"] = N.gemm_split(A, Wp, bias=bias, addend=add, **kw)
            outs[terms, "gelu_grad"] = N.gemm_split(A, Wp, a_gelu_grad_of=t, **kw)
            outs[terms, "gate"] = N.gemm_split(A, Wp, a_relu_gate=t, **kw)
            go = torch.empty(M, Nn, device="cuda")
            outs[terms, "gelu_out_pre"] = N.gemm_split(A, Wp, bias=bias, gelu_out=go, **kw)
            outs[terms, "gelu_out"] = go
            outs[terms, "gelu_grad_of"] = N.gemm_split(A, Wp, gelu_grad_of=pre, **kw)
            outs[terms, "batch"] = N.gemm_split(A3, N.gemm_split_pack(W3, terms=terms), groups=1)
        word = N.amax_word(A.device)
        outs[22, "out_amax"] = N.gemm_split(A, N.gemm_split_pack(W, terms=22), out_amax=word).clone()
        outs[22, "out_amax_word"] = word.clone()
        return outs

    L = N.lib()
    assert L.sea_gemm_split_pipeline(-1) in (0, 1, 2, 3)
    prev = L.sea_gemm_split_pipeline(0)
    try:
        o0 = all_variants()
        assert L.sea_gemm_split_pipeline(1) == 0
        o1 = all_variants()
    finally:
        L.sea_gemm_split_pipeline(prev)
    for key, a in o0.items():
        assert torch.equal(a, o1[key]), (key, (a.float() - o1[key].float()).abs().max().item())
    ref = _ref(A, W, bias, True)
    assert (o1[22, "plain"].double() - ref).abs().max().item() <= 3e-6 * ref.abs().max().item()


@pytest.mark.parametrize("G,M,K,Nn", [(3, 1024, 512, 512), (2, 700, 96, 256), (1, 300, 32, 768), (2, 2048, 160, 256), (2, 1000, 384, 384),
                                      (1, 515, 192, 1152)])
def test_one_block_per_cu_kernels_give_the_bits_of_the_128x128_kernels(N, G, M, K, Nn):
    """sea_gemm_split_pipeline(3) (csrc/gemm_split_big.hip: 8 waves, one block per CU; 256 x 256 tiles for the Winograd-domain
    products, 128 x 384 tiles -- with the GELU / GELU' / gate prologues -- where N is a multiple of 384) against pipeline 0: same
    split, same MFMA order per accumulator -> the same bits, for fp16 x 2 and bf16 x 2, bias / ReLU / out_amax, ragged M (rows
    past M load zeros through the buffer descriptor) and odd / single K-step counts"""
    g = torch.Generator(device="cuda").manual_seed(M + K + Nn)
    A = torch.randn(G, M, K, generator=g, device="cuda") * torch.exp2(torch.randint(-6, 3, (G, M, 1), generator=g, device="cuda").float())
    W = torch.randn(G, Nn, K, generator=g, device="cuda") / K ** 0.5
    bias = torch.randn(Nn, generator=g, device="cuda")
    tt = torch.randn(G, M, K, generator=g, device="cuda")

    def variants():
        outs = {}
        for terms in (22, 2):
            Wp = N.gemm_split_pack(W, terms=terms)
            outs[terms, "plain"] = N.gemm_split(A, Wp, groups=1)
            outs[terms, "bias_relu"] = N.gemm_split(A, Wp, bias=bias, relu=True, groups=1)
            if Nn % 384 == 0:          # the prologues (128 x 384 tiles only)
                outs[terms, "gelu"] = N.gemm_split(A, Wp, a_gelu=True, groups=1)
                outs[terms, "gelu_grad"] = N.gemm_split(A, Wp, a_gelu_grad_of=tt, groups=1)
                outs[terms, "gate"] = N.gemm_split(A, Wp, a_relu_gate=tt, groups=1)
        word = N.amax_word(A.device)
        outs[22, "out_amax"] = N.gemm_split(A, N.gemm_split_pack(W, terms=22), out_amax=word, groups=1).clone()
        outs[22, "out_amax_word"] = word.clone()
        return outs

    L = N.lib()
    prev = L.sea_gemm_split_pipeline(0)
    try:
        o0 = variants()
        assert L.sea_gemm_split_pipeline(3) == 0
        o3 = variants()
    finally:
        L.sea_gemm_split_pipeline(prev)
    for key, a in o0.items():
        assert torch.equal(a, o3[key]), (key, (a.float() - o3[key].float()).abs().max().item())
    ref = torch.einsum("gmk,gnk->gmn", A.double(), W.double())
    assert (o3[22, "plain"].double() - ref).abs().max().item() <= 3e-6 * ref.abs().max().item()
