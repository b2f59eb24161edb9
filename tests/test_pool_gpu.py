"""Adaptive average pooling of NHWC maps through libsea_hip (csrc/upsample_kernels.hip: the pyramid pooling of the head,
reference semseg/models/uperforseg.py:150-177) against ATen in float64, forward and input gradient.  `-m gpu`."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def N():
    from semseg import _native
    _native.lib()
    return _native


@pytest.mark.parametrize("B,C,H,W,oh,ow", [(8, 768, 16, 16, 1, 1), (8, 768, 16, 16, 2, 2), (8, 768, 16, 16, 3, 3),
                                           (8, 768, 16, 16, 6, 6), (2, 72, 7, 5, 3, 2), (1, 4, 5, 5, 5, 5), (3, 20, 9, 11, 4, 7),
                                           (1, 8, 1, 1, 1, 1), (2, 12, 4, 4, 6, 6)])
def test_adaptive_avg_pool_nhwc(N, B, C, H, W, oh, ow):
    g = torch.Generator().manual_seed(H * 31 + oh)
    x = torch.randn(B, C, H, W, generator=g)
    gy = torch.randn(B, C, oh, ow, generator=g)
    xc = x.cuda().contiguous(memory_format=torch.channels_last)
    if oh > H or ow > W:
        with pytest.raises(N.SeaNativeError):
            N.adaptive_avg_pool_nhwc(xc, oh, ow)
        return
    xd = x.double().requires_grad_(True)
    ref = F.adaptive_avg_pool2d(xd, (oh, ow))
    (gx_ref,) = torch.autograd.grad(ref, xd, gy.double())
    y = N.adaptive_avg_pool_nhwc(xc, oh, ow)
    assert y.shape == (B, C, oh, ow) and y.permute(0, 2, 3, 1).is_contiguous()
    torch.testing.assert_close(y.cpu().double(), ref.detach(), rtol=1e-5, atol=2e-6)
    gx = N.adaptive_avg_pool_nhwc_backward(gy.cuda().contiguous(memory_format=torch.channels_last), H, W)
    assert gx.shape == (B, C, H, W) and gx.permute(0, 2, 3, 1).is_contiguous()
    torch.testing.assert_close(gx.cpu().double(), gx_ref, rtol=1e-5, atol=2e-6)
    # an NCHW-contiguous gradient is accepted as well; reproducible to the bit
    assert torch.equal(gx, N.adaptive_avg_pool_nhwc_backward(gy.cuda(), H, W))
    assert torch.equal(y, N.adaptive_avg_pool_nhwc(xc, oh, ow))
    if W > 1 and C > 1:
        with pytest.raises(N.SeaNativeError):
            N.adaptive_avg_pool_nhwc(x.cuda(), oh, ow)          # NCHW memory refused


def test_pyramid_pooling_uses_the_kernels_and_matches_aten():
    from semseg.models import convnext_upernet as M
    torch.manual_seed(0)
    ppm = M.PyramidPooling((1, 2, 3, 6), 64, 32).cuda().eval()
    x = torch.randn(2, 64, 16, 16, device="cuda").contiguous(memory_format=torch.channels_last)

    def run(flag):
        old, M.USE_HIP_ADAPTIVE_POOL = M.USE_HIP_ADAPTIVE_POOL, flag
        try:
            xg = x.clone().requires_grad_(True)
            outs = ppm(xg)
            (g,) = torch.autograd.grad(sum(o.square().sum() for o in outs), xg)
            return [o.detach() for o in outs], g
        finally:
            M.USE_HIP_ADAPTIVE_POOL = old

    calls = []
    orig = M._AdaptivePoolCL.apply
    M._AdaptivePoolCL.apply = lambda *a: (calls.append(1), orig(*a))[1]
    try:
        outs, g = run(True)
    finally:
        M._AdaptivePoolCL.apply = orig
    assert len(calls) == 4
    outs_ref, g_ref = run(False)
    for o, r in zip(outs, outs_ref):
        torch.testing.assert_close(o, r, rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(g, g_ref, rtol=2e-4, atol=2e-4 * g_ref.abs().max().item())
