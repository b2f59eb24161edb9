"""M10: the decode head's 1 x 1 classifier for at most 32 classes (csrc/classifier.hip), `-m gpu`.

Reference layer: semseg/models/uperforseg.py:262 (`cls_seg`: conv_seg, a 1 x 1 convolution onto the classes) and its autograd
input gradient.  The kernels use f32 MFMA operands (exact products), so the yardstick is float64: the error must not exceed
that of the library's fp32 matmul on the same data; results are bitwise reproducible; the `_classify` wrapper of the model takes
the kernels for frozen weights and keeps `nn.Conv2d`'s values."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def N():
    from semseg import _native
    _native.lib()
    return _native


CASES = [(2, 64 * 64, 512, 21), (1, 32 * 32, 256, 5), (1, 1024, 64, 32), (3, 96, 128, 1), (1, 128 * 128, 512, 21), (2, 160, 1024, 22)]


@pytest.mark.parametrize("B,P,K,cls", CASES)
def test_classifier_forward_and_input_gradient_against_float64(N, B, P, K, cls):
    g = torch.Generator(device="cuda").manual_seed(K + cls)
    y = torch.randn(B * P, K, generator=g, device="cuda").relu_()                      # (the head's input is a ReLU output)
    w = torch.randn(cls, K, generator=g, device="cuda") * 0.05
    b = torch.randn(cls, generator=g, device="cuda")
    go = torch.randn(B, cls, P, generator=g, device="cuda") * torch.exp2(torch.randint(-20, 0, (B, 1, 1), generator=g, device="cuda").float())
    assert N.classifier_ok(P, K, cls)
    out = N.classifier_forward(y, w, b, B, P)
    ref = (y.double().view(B, P, K) @ w.double().t() + b.double()).transpose(1, 2)    # (B, cls, P)
    lib = (y.view(B, P, K) @ w.t() + b).transpose(1, 2)
    scale = ref.abs().max()
    err, err_lib = (out.double() - ref).abs().max() / scale, (lib.double() - ref).abs().max() / scale
    assert err <= max(4 * err_lib, 1e-6), (float(err), float(err_lib))    # (one accumulator chain along K: a few ulp)
    assert torch.equal(out, N.classifier_forward(y, w, b, B, P))                        # run to run
    assert torch.equal(N.classifier_forward(y, w, None, B, P) + b.view(1, -1, 1), out)  # (one rounding either way)
    gy = N.classifier_backward(go, w)
    gref = go.double().transpose(1, 2).reshape(B * P, cls) @ w.double()
    glib = go.transpose(1, 2).reshape(B * P, cls) @ w
    # per image: the gradient rows of different images span many orders of magnitude
    for i in range(B):
        sl = slice(i * P, (i + 1) * P)
        s = gref[sl].abs().max()
        e, el = (gy[sl].double() - gref[sl]).abs().max() / s, (glib[sl].double() - gref[sl]).abs().max() / s
        assert e <= max(4 * el, 1e-6), (i, float(e), float(el))
    assert torch.equal(gy, N.classifier_backward(go, w))
    # an image's logits / gradient do not depend on its batch partners
    if B > 1:
        assert torch.equal(N.classifier_forward(y[:P], w, b, 1, P)[0], out[0])
        assert torch.equal(N.classifier_backward(go[:1].contiguous(), w), gy[:P])


def test_classifier_rejects_what_it_does_not_take(N):
    assert not N.classifier_ok(64 * 64, 512, 151)       # ADE20K: stays on the library
    assert not N.classifier_ok(100, 512, 21) and not N.classifier_ok(64, 96, 21) and not N.classifier_ok(64, 2048, 21)
    y, w = torch.randn(100, 512, device="cuda"), torch.randn(21, 512, device="cuda")
    with pytest.raises(N.SeaNativeError):
        N.classifier_forward(y, w, None, 1, 100)
    with pytest.raises(N.SeaNativeError):
        N.classifier_backward(torch.randn(1, 21, 100, device="cuda"), w)


@pytest.mark.parametrize("cls", [21, 151])
def test_head_classifier_takes_the_kernels_and_keeps_conv2d_values(N, cls, monkeypatch):
    from semseg.models import convnext_upernet as M
    torch.manual_seed(cls)
    conv = torch.nn.Conv2d(512, cls, 1).cuda().requires_grad_(False)
    y = torch.randn(2, 512, 32, 32, device="cuda").relu_().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    calls = []
    real_f, real_b = N.classifier_forward, N.classifier_backward
    monkeypatch.setattr(N, "classifier_forward", lambda *a, **k: (calls.append("f"), real_f(*a, **k))[1])
    monkeypatch.setattr(N, "classifier_backward", lambda *a, **k: (calls.append("b"), real_b(*a, **k))[1])
    out = M._classify(conv, y)
    go = torch.randn_like(out)
    (gy,) = torch.autograd.grad(out, y, go)
    assert calls == (["f", "b"] if cls <= 32 else [])
    yr = y.detach().clone().requires_grad_(True)
    ref = conv(yr)
    (gref,) = torch.autograd.grad(ref, yr, go)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gy, gref, rtol=1e-5, atol=1e-5)
    assert gy.is_contiguous(memory_format=torch.channels_last)
    # SEA_CLASSIFIER=0 / USE_CLASSIFIER: the library path, same values within the same tolerance
    monkeypatch.setattr(N, "USE_CLASSIFIER", False)
    calls.clear()
    out2 = M._classify(conv, y)
    assert not calls
    torch.testing.assert_close(out2, ref, rtol=1e-5, atol=1e-5)


def test_bottleneck_and_classifier_as_one_node_keep_the_bits_of_two(N, monkeypatch):
    """_FpnBottleneckClassify (the classifier's backward kernel applies the bottleneck's ReLU gate and BatchNorm scale) against
    _FpnBottleneck + _ClassifierGemm + the separate gate pass: logits and all four input gradients bit for bit"""
    from semseg.models import convnext_upernet as M
    torch.manual_seed(3)
    chans = [96, 192, 384, 768]
    head = M.UperNetHead(chans, 21).cuda().eval()
    with torch.no_grad():
        for mod in head.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.normal_(0, 0.1)
                mod.running_var.uniform_(0.5, 2.0)
                mod.weight.normal_(1, 0.3)
                mod.bias.normal_(0, 0.1)
    for p in head.parameters():
        p.requires_grad_(False)
    # (the real pyramid of a 512 x 512 input: below 16 x 16 the PSP bottleneck falls back to the library's convolution, whose
    # results differ from run to run)
    feats = [torch.randn(1, c, s, s, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
             for c, s in zip(chans, (128, 64, 32, 16))]
    go = None
    res = {}
    for fuse in (True, False):
        monkeypatch.setattr(M, "FUSE_CLASSIFIER_GATE", fuse)
        out = head(tuple(feats))
        assert ("Classify" in type(out.grad_fn).__name__) == fuse, type(out.grad_fn).__name__
        if go is None:
            go = torch.randn_like(out) * 1e-3
        res[fuse] = (out.detach(), torch.autograd.grad(out, feats, go))
    assert torch.equal(res[True][0], res[False][0])
    for a, b in zip(res[True][1], res[False][1]):
        assert torch.equal(a, b)
    # and the gate option of the kernel alone against sea_gate_scale on the stored gradient
    g = torch.Generator(device="cuda").manual_seed(5)
    B, P, K, cls = 2, 1024, 512, 21
    gl = torch.randn(B, cls, P, generator=g, device="cuda")
    w = torch.randn(cls, K, generator=g, device="cuda") * 0.05
    y = torch.randn(B * P, K, generator=g, device="cuda").relu_()
    sc = torch.rand(K, generator=g, device="cuda") + 0.5
    plain = N.classifier_backward(gl, w)
    gated = N.classifier_backward(gl, w, gate=y, gate_scale=sc)
    assert torch.equal(gated, torch.where(y > 0, plain * sc, torch.zeros_like(plain)))
    with pytest.raises(N.SeaNativeError):
        N.classifier_backward(gl, w, gate=y)
