"""BASELINE configs on the REAL models, device path vs goldens produced by the real reference on CPU (`-m gpu`):

  upernet_t  configs[0]/[1]  UperNet-ConvNeXt-T, C=21   (step pins; the 5-step run lives in test_config1_parity.py)
  segmenter  configs[2]      Segmenter ViT-S/16, C=151  (K2u: loss fused with the x16 upsample, auto-selected)
  upernet_s  configs[3]/[4]  UperNet-ConvNeXt-S, C=151  (+ the PIR-AT inner PGD, fp32 and bf16 autocast)

What is pinned: sampled logits; at two fixed points (the clean image and a fixed random start) the per-image loss
of every SEA loss, the tracking loss, the number of correct pixels and sampled input-gradient values; then a
5-iteration apgd_largereps run per loss.  MIOpen / hipBLASLt / Winograd convolutions differ from the CPU's in the
last bits, so float quantities carry the north_star tolerance (1e-4 relative on losses) and integer ones a few
pixels of 262144 (pixels whose two best logits are closer than that rounding noise).
"""
import pytest
import torch

from real_models import CASES, EPS, LOSSES, setup, stage_noises

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    cache = {}

    def get(case):
        if case not in cache:
            cache.clear()  # one real model on the device at a time
            torch.cuda.empty_cache()
            g, model, x, x1, y, w, C = setup(case)
            cache[case] = (g, model.cuda(), x.cuda(), x1.cuda(), y.cuda(), w.cuda(), C)
        return cache[case]
    return get


@pytest.mark.parametrize("case", sorted(CASES))
def test_logits_match_reference_samples(ctx, case):
    g, model, x, x1, y, w, C = ctx(case)
    with torch.no_grad():
        logits = model(x)
    got = logits.flatten()[g["logit_idx"].cuda()].cpu()
    scale = float(g["logit_absmax"])
    assert (got - g["logit_samples"]).abs().max() <= 5e-5 * scale, ((got - g["logit_samples"]).abs().max(), scale)


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("case", sorted(CASES))
def test_step_pins_loss_accuracy_gradient(ctx, case, fused):
    """step 0 / step 1 of the attack on the real model: loss_indiv, tracking loss, n_correct, input gradient."""
    from semseg import _native as N, attacker as A
    g, model, x, x1, y, w, C = ctx(case)
    HW = x.shape[-2] * x.shape[-1]
    yc = A.compact_labels(y, C)
    for tag, xp in (("p0", x), ("p1", x1)):
        for loss in LOSSES:
            key = f"{tag}_{loss.replace('-', '_')}"
            x_in, logits = A._forward_logits(model, xp, True, lowres=fused)
            if fused:
                r = N.loss_fwd_bwd_upsampled(logits.detach().contiguous(), yc, w, N.MODE_BY_NAME[loss], 3, 1.0 / HW,
                                             want_grad=True)
            else:
                r = N.loss_fwd_bwd(logits.detach(), yc, w, N.MODE_BY_NAME[loss], 3, 1.0 / HW, want_grad=True)
            grad = A._input_grad(logits, x_in, r["dlogits"])
            torch.testing.assert_close((r["loss_sum"] / HW).cpu(), g[key + "_img"], rtol=1e-4, atol=1e-6)
            torch.testing.assert_close((r["track_sum"] / HW).cpu(), g[tag + "_ce_img"], rtol=1e-4, atol=1e-6)
            assert (r["n_correct"].cpu().long() - g[tag + "_n_correct"]).abs().max() <= 16
            got = grad.flatten()[g["grad_idx"].cuda()].cpu()
            ref = g[key + "_grad"]
            rel_l2 = ((got - ref).norm() / ref.norm()).item()
            big = ref.abs() > 1e-2 * float(g[key + "_gradmax"])
            sign_ok = (torch.sign(got[big]) == torch.sign(ref[big])).float().mean().item()
            # K1 only uses sign(grad): the sign must agree wherever the gradient is not at rounding level
            assert rel_l2 <= 1e-2 and sign_ok >= 0.995, (key, rel_l2, sign_ok)


@pytest.mark.parametrize("case", ["segmenter", "upernet_s"])
def test_five_step_largereps_matches_reference(ctx, case):
    from semseg import attacker as A
    g, model, x, x1, y, w, C = ctx(case)
    noises = stage_noises(x.cpu())
    for loss in LOSSES:
        key = loss.replace("-", "_")
        xa, _, acc, pred = A.apgd_largereps(model, x, y, w, norm="Linf", eps=EPS, n_iter=5, n_restarts=1, use_rs=True,
                                            loss=loss, verbose=False, track_loss="ce-avg", log_path=None,
                                            num_classes=C, early_stop=True, noises=noises, return_pred=True)
        assert (xa - x).abs().max() <= EPS + 1e-6 and xa.min() >= 0 and xa.max() <= 1
        assert (acc.cpu() - g[key + "_acc"]).abs().max() <= 5e-3, (loss, acc, g[key + "_acc"])
        got = xa.flatten()[g["idx"].cuda()].cpu()
        frac = ((got - g[key + "_x_adv_samples"]).abs() > 1e-6).float().mean().item()
        assert frac <= 0.10, (loss, frac)
        # the argmax map the attack hands out IS the prediction of the returned iterate (no re-forward needed)
        with torch.no_grad():
            again = model(xa).max(1)[1]
        assert (again != pred.long()).float().mean().item() <= 1e-4
        m_acc, a_acc, m_iou = A.compute_iou_acc(pred.long(), y, C)
        assert abs(a_acc.item() - float(g[key + "_adv_aacc"])) <= 5e-3
        assert abs(m_iou.item() - float(g[key + "_adv_miou"])) <= 1e-2


def test_pirat_inner_pgd_convnext_s_fp32_and_bf16(ctx):
    """BASELINE configs[3]: Pgd_Attack_1 (CE, 5 steps, alpha 1e-2, eps 4/255) on UperNet-ConvNeXt-S, C=151.
    fp32 against the reference's run; bf16 autocast (what the config asks for) must stay a valid attack of the same
    strength: same eps-ball, sign-step lattice, and a final cross-entropy within 5 % of the fp32 reference's."""
    from semseg import val as V
    g, model, x, x1, y, w, C = ctx("upernet_s")
    torch.manual_seed(int(g["pgd_seed"]))
    delta0 = torch.zeros(2, 3, 512, 512).uniform_(-EPS, EPS).cuda()      # what the reference drew on the CPU
    atk = V.Pgd_Attack_1(epsilon=EPS, alpha=1e-2, num_iter=int(g["pgd_steps"]), los="pgd")
    xa, logits, _ = atk.adv_attack(model, x, y, delta0=delta0)
    got = xa.flatten()[g["idx"].cuda()].cpu()
    assert ((got - g["pgd_x_adv_samples"]).abs() > 1e-6).float().mean().item() <= 0.05
    assert (xa - x).abs().max() <= EPS + 1e-6
    with torch.no_grad():
        ce = torch.nn.functional.cross_entropy(model(xa), y).item()
    # the random-init model is near-uniform over 151 classes (CE ~ ln 151), so compare the GAIN of the attack
    ref_gain = float(g["pgd_ce_adv"]) - float(g["pgd_ce_clean"])
    with torch.no_grad():
        ce0 = torch.nn.functional.cross_entropy(model(x), y).item()
    assert ce0 == pytest.approx(float(g["pgd_ce_clean"]), rel=1e-4)
    assert ce - ce0 == pytest.approx(ref_gain, rel=0.05), (ce - ce0, ref_gain)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        xb, _, _ = atk.adv_attack(model, x, y, delta0=delta0)
    assert (xb - x).abs().max() <= EPS + 1e-6 and xb.min() >= 0 and xb.max() <= 1
    with torch.no_grad():
        ce_b = torch.nn.functional.cross_entropy(model(xb), y).item()
    print("pgd gain fp32 / bf16 / reference:", ce - ce0, ce_b - ce0, ref_gain)
    assert ce_b - ce0 == pytest.approx(ref_gain, rel=0.05), (ce_b - ce0, ref_gain)   # measured: 0.033689 vs 0.033708
    gotb = xb.flatten()[g["idx"].cuda()].cpu()
    assert ((gotb - g["pgd_x_adv_samples"]).abs() > 1e-6).float().mean().item() <= 0.35
