"""M9: the convolutional stem of the robust ConvNeXt backbones through libsea_hip (csrc/stem_kernels.hip) against PyTorch in
float64 -- Conv2d(3,48,3,s2,p1) -> LayerNorm(channels_first) -> GELU -> Conv2d(48,96,3,s2,p1) -> LayerNorm -> GELU
(reference semseg/models/backbones/convnext_orig.py:17-38), forward and input gradient.  `-m gpu`.

Bar: error against float64 no larger than 2 x the error of PyTorch-ROCm's own fp32 ops on the same data (the kernels are plain
fp32 FMA chains; the library convolution they replace is a Winograd kernel), bitwise reproducible, ragged sizes."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def N():
    from semseg import _native
    _native.lib()
    return _native


def _ln_cf(y, g, b, eps):
    u = y.mean(1, keepdim=True)
    s = (y - u).pow(2).mean(1, keepdim=True)
    return (y - u) / torch.sqrt(s + eps) * g[None, :, None, None] + b[None, :, None, None]


def _params(C, seed, cin=3):
    gen = torch.Generator().manual_seed(seed)
    w = torch.randn(C, cin, 3, 3, generator=gen) * 0.3
    b = torch.randn(C, generator=gen) * 0.1
    g = 1 + 0.2 * torch.randn(C, generator=gen)
    be = 0.1 * torch.randn(C, generator=gen)
    return w, b, g, be


def _err(a, ref):
    return ((a.double().cpu() - ref).abs().max() / ref.abs().max()).item()


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (1, 37, 53), (3, 16, 130), (1, 1, 1), (2, 5, 2)])
def test_conv1_ln_gelu_forward_and_input_gradient(N, B, H, W):
    w, b, g, be = _params(48, 1)
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(2))
    xd = x.double().requires_grad_(True)
    yd = F.conv2d(xd, w.double(), b.double(), stride=2, padding=1)
    ad = F.gelu(_ln_cf(yd, g.double(), be.double(), 1e-6))
    da = torch.randn(ad.shape, generator=torch.Generator().manual_seed(3))
    (gxd,) = torch.autograd.grad(ad, xd, da.double())
    c = lambda t: t.cuda()  # noqa: E731
    y, a = N.stem_conv1_ln_gelu(c(x), c(w), c(b), c(g), c(be), 1e-6)
    y2, none = N.stem_conv1_ln_gelu(c(x), c(w), c(b))
    assert none is None and torch.equal(y, y2)
    # PyTorch-ROCm fp32 on the same data: the yardstick
    xs = c(x).requires_grad_(True)
    ys = F.conv2d(xs, c(w), c(b), stride=2, padding=1)
    a_s = F.gelu(_ln_cf(ys, c(g), c(be), 1e-6))
    (gxs,) = torch.autograd.grad(a_s, xs, c(da))
    e_y, e_a = _err(y, yd.detach()), _err(a, ad.detach())
    assert e_y <= max(2 * _err(ys.detach(), yd.detach()), 2e-7), (e_y, _err(ys.detach(), yd.detach()))
    assert e_a <= max(2 * _err(a_s.detach(), ad.detach()), 5e-7), (e_a, _err(a_s.detach(), ad.detach()))
    assert y.is_contiguous(memory_format=torch.channels_last) and a.is_contiguous(memory_format=torch.channels_last)
    dy = N.ln_gelu_cl_backward(c(da), y, c(g), c(be), 1e-6)                                   # da NCHW
    dy_cl = N.ln_gelu_cl_backward(c(da).contiguous(memory_format=torch.channels_last), y, c(g), c(be), 1e-6)
    assert torch.equal(dy, dy_cl)                                                             # the same bits from either layout
    gx = N.stem_conv1_backward(dy, c(w), H, W)
    e_g = _err(gx, gxd)
    assert e_g <= max(2 * _err(gxs, gxd), 2e-6), (e_g, _err(gxs, gxd))
    # reproducible to the bit
    y3, a3 = N.stem_conv1_ln_gelu(c(x), c(w), c(b), c(g), c(be), 1e-6)
    assert torch.equal(y, y3) and torch.equal(a, a3)
    assert torch.equal(gx, N.stem_conv1_backward(N.ln_gelu_cl_backward(c(da), y, c(g), c(be), 1e-6), c(w), H, W))


@pytest.mark.parametrize("C", [48, 96])
@pytest.mark.parametrize("B,H,W", [(2, 32, 32), (1, 7, 19), (3, 1, 129), (1, 1, 1)])
def test_ln_gelu_over_channels(N, C, B, H, W):
    _, _, g, be = _params(C, 5)
    gen = torch.Generator().manual_seed(6)
    y = torch.randn(B, C, H, W, generator=gen) * 2 + 0.3
    da = torch.randn(B, C, H, W, generator=gen)
    yd = y.double().requires_grad_(True)
    ad = F.gelu(_ln_cf(yd, g.double(), be.double(), 1e-6))
    (dyd,) = torch.autograd.grad(ad, yd, da.double())
    c = lambda t: t.cuda()  # noqa: E731
    ycl = c(y).contiguous(memory_format=torch.channels_last)
    a = N.ln_gelu_cl(ycl, c(g), c(be), 1e-6, out_nchw=True)
    a_cl = N.ln_gelu_cl(ycl, c(g), c(be), 1e-6)
    assert a.is_contiguous() and a_cl.is_contiguous(memory_format=torch.channels_last) and torch.equal(a, a_cl)
    dy = N.ln_gelu_cl_backward(c(da), ycl, c(g), c(be), 1e-6)
    assert torch.equal(dy, N.ln_gelu_cl_backward(c(da).contiguous(memory_format=torch.channels_last), ycl, c(g), c(be), 1e-6))
    ys = c(y).requires_grad_(True)
    a_s = F.gelu(_ln_cf(ys, c(g), c(be), 1e-6))
    (dys,) = torch.autograd.grad(a_s, ys, c(da))
    assert _err(a, ad.detach()) <= max(2 * _err(a_s.detach(), ad.detach()), 3e-7)
    assert _err(dy, dyd) <= max(2 * _err(dys, dyd), 1e-6), (_err(dy, dyd), _err(dys, dyd))
    assert torch.equal(a, N.ln_gelu_cl(ycl, c(g), c(be), 1e-6, out_nchw=True))
    assert torch.equal(dy, N.ln_gelu_cl_backward(c(da), ycl, c(g), c(be), 1e-6))


def test_unsupported_widths_are_errors(N):
    x = torch.rand(1, 3, 8, 8).cuda()
    with pytest.raises(N.SeaNativeError):
        N.stem_conv1_ln_gelu(x, torch.randn(32, 3, 3, 3).cuda(), None)
    with pytest.raises(N.SeaNativeError):
        N.ln_gelu_cl(torch.randn(1, 64, 4, 4).cuda().contiguous(memory_format=torch.channels_last), torch.ones(64).cuda(),
                     torch.zeros(64).cuda())
    with pytest.raises(N.SeaNativeError):      # NCHW memory where channels_last is required
        N.ln_gelu_cl(torch.randn(1, 48, 4, 4).cuda(), torch.ones(48).cuda(), torch.zeros(48).cuda())


def test_model_stem_takes_the_fused_path_only_when_frozen():
    """ConvStem: fused kernels for frozen parameters, the library path (with parameter gradients) otherwise; both agree"""
    from semseg.models import convnext_upernet as M
    torch.manual_seed(0)
    stem = M.ConvStem().cuda().eval()
    with torch.no_grad():
        for p in stem.parameters():
            p.add_(0.05 * torch.randn_like(p))
    x = torch.rand(2, 3, 96, 80, device="cuda")
    da = torch.randn(2, 96, 24, 20, device="cuda")

    def run(x):
        xg = x.clone().requires_grad_(True)
        out = stem(xg)
        (g,) = torch.autograd.grad(out, xg, da)
        return out.detach(), g

    assert not stem._fused_ok(x)                      # trainable parameters: nothing may skip their gradients
    out_lib, g_lib = run(x)
    for p in stem.parameters():
        p.requires_grad_(False)
    assert stem._fused_ok(x)
    calls = []
    orig = M._StemConv1LnGelu.apply
    M._StemConv1LnGelu.apply = lambda *a: (calls.append(1), orig(*a))[1]
    try:
        out, g = run(x)
    finally:
        M._StemConv1LnGelu.apply = orig
    assert calls, "the fused stem did not run"
    assert (out - out_lib).abs().max() <= 2e-5 * out_lib.abs().max()
    assert (g - g_lib).abs().max() <= 2e-5 * g_lib.abs().max()
    out2, g2 = run(x)
    assert torch.equal(out, out2) and torch.equal(g, g2)
    old, M.USE_FUSED_STEM = M.USE_FUSED_STEM, False
    try:
        assert not stem._fused_ok(x)
    finally:
        M.USE_FUSED_STEM = old
