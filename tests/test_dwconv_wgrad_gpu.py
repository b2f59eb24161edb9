"""M1w: weight and bias gradient of the NHWC depthwise 7x7 (csrc/dwconv_kernels.hip; PIR-AT's outer backward through
reference convnext_orig.py:55-57) against PyTorch in float64.  `-m gpu`."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def N():
    from semseg import _native
    _native.lib()
    return _native


@pytest.mark.parametrize("B,H,W,C", [(2, 16, 16, 96), (1, 5, 7, 12), (2, 9, 4, 8), (3, 32, 32, 192), (2, 16, 16, 384),
                                     (2, 8, 8, 768), (1, 1, 1, 4), (2, 3, 50, 260), (8, 64, 64, 96)])
def test_depthwise_weight_gradient(N, B, H, W, C):
    g = torch.Generator().manual_seed(B * 100 + C)
    x = torch.randn(B, H, W, C, generator=g)
    gy = torch.randn(B, H, W, C, generator=g)
    ref = torch.nn.grad.conv2d_weight(x.double().permute(0, 3, 1, 2), (C, 1, 7, 7), gy.double().permute(0, 3, 1, 2),
                                      padding=3, groups=C)
    refb = gy.double().sum((0, 1, 2))
    gw, gb = N.dwconv7x7_nhwc_weight_grad(x.cuda(), gy.cuda())
    assert gw.shape == (C, 1, 7, 7) and gb.shape == (C,)
    scale = ref.abs().max().item()
    assert (gw.cpu().double() - ref).abs().max().item() <= 2e-6 * scale * max(1.0, (B * H * W) ** 0.5 / 16)
    assert (gb.cpu().double() - refb).abs().max().item() <= 2e-6 * refb.abs().max().item() * max(1.0, (B * H * W) ** 0.5 / 16) + 1e-6
    gw2, none = N.dwconv7x7_nhwc_weight_grad(x.cuda(), gy.cuda(), want_bias=False)
    assert none is None and torch.equal(gw, gw2)          # deterministic (fixed-order two-pass sum)


def test_block_backward_uses_it_and_matches_the_library():
    from semseg.models import convnext_upernet as M
    torch.manual_seed(0)
    blk = M.Block(96).cuda().train()
    x = torch.randn(2, 96, 16, 16, device="cuda").contiguous(memory_format=torch.channels_last)

    def grads(flag):
        old, M.USE_HIP_DW_WGRAD = M.USE_HIP_DW_WGRAD, flag
        try:
            blk.zero_grad(set_to_none=True)
            blk(x.clone().requires_grad_(True)).square().sum().backward()
            return blk.dwconv.weight.grad.clone(), blk.dwconv.bias.grad.clone()
        finally:
            M.USE_HIP_DW_WGRAD = old

    calls = []
    from semseg import _native as Nn
    real = Nn.dwconv7x7_nhwc_weight_grad
    Nn.dwconv7x7_nhwc_weight_grad = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        gw, gb = grads(True)
    finally:
        Nn.dwconv7x7_nhwc_weight_grad = real
    assert calls, "the block's backward did not go through the kernel"
    gw_ref, gb_ref = grads(False)
    torch.testing.assert_close(gw, gw_ref, rtol=1e-4, atol=1e-5 * gw_ref.abs().max().item())
    torch.testing.assert_close(gb, gb_ref, rtol=1e-4, atol=1e-5 * gb_ref.abs().max().item())
