"""Remaining API surface of the hot path on the device, against goldens from the real reference (`-m gpu`):
val.losses callables incl. l2-loss (semseg/val.py:121-127), general js_div_fn arguments (attacker.py:187-226),
apgd_restarts (574-659), eval_performance (tools/infer.py:56-133), the reference-format argmax logs and their
round trip through evalSEA, the single-attack run of tools.infer, and NaN logits (torch.max semantics)."""
import json
import os
import random

import pytest
import torch

from conftest import PKG, load_golden
from oracle.tiny_models import TinyConvNet

pytestmark = pytest.mark.gpu


def test_val_losses_are_callables_with_reference_values_and_gradients():
    from semseg import val as V
    g = load_golden("g12_val_losses")
    logits, y = g["logits"].cuda(), g["y"].cuda()
    assert set(V.losses) == {"pgd", "mask-ce-avg", "js-avg", "l2-loss"}
    for name, other in (("pgd", y), ("mask-ce-avg", y), ("js-avg", y), ("l2-loss", g["l2_other"].cuda())):
        key = name.replace("-", "_")
        z = logits.clone().requires_grad_(True)
        val = V.losses[name](z, other)
        torch.testing.assert_close(val.detach().cpu(), torch.as_tensor(g[key]), rtol=1e-4, atol=1e-6)
        (gr,) = torch.autograd.grad(val.sum(), [z])
        torch.testing.assert_close(gr.cpu(), g[key + "_grad"], rtol=1e-4, atol=1e-7)


def test_js_div_fn_general_arguments():
    from semseg import attacker as A
    g = load_golden("g12_js_div_general")
    logits, y = g["logits"].cuda(), g["y"].cuda()
    cmp = lambda a, k, **kw: torch.testing.assert_close(a.cpu(), g[k], rtol=1e-4, atol=1e-6, **kw)  # noqa: E731
    cmp(A.js_div_fn(logits, y), "full")
    cmp(A.js_div_fn(logits, y, red_dim=(1)), "sum_c")                      # the fused-kernel configuration
    cmp(A.js_div_fn(logits, y, red_dim=(1, 2, 3)), "sum_chw")
    cmp(A.js_div_fn(torch.softmax(logits, 1), y, softmax_output=True, red_dim=(1)), "from_probs")
    z = logits.clone().requires_grad_(True)
    (gr,) = torch.autograd.grad(A.js_div_fn(z, y).sum(), [z])
    cmp(gr, "full_grad")
    with pytest.raises(ValueError):
        A.js_div_fn(logits, y, reduction="sum")
    out = A.js_div_fn(logits, torch.full_like(y, -1), reduction="sum")
    assert out.shape == g["allign_sum"].shape and torch.equal(out.cpu(), g["allign_sum"])


def test_apgd_restarts_golden():
    from semseg import attacker as A
    g = load_golden("g12_apgd_restarts")
    net = TinyConvNet(5, seed=12, gain=3.0).cuda()
    noises = [g[f"noise_{i}"].cuda() for i in range(int(g["n_noise"]))]
    xa, acc_last, acc = A.apgd_restarts(net, g["x"].cuda(), g["y"].cuda(), norm="Linf", eps=g["eps"],
                                        n_iter=int(g["n_iter"]), loss="mask-ce-avg", n_restarts=int(g["n_restarts"]),
                                        early_stop=True, track_loss="ce-avg", use_rs=True, noises=noises)
    # 12x12 images: one pixel = 1/144; MIOpen vs CPU convolutions may flip a pixel or two
    assert (acc.cpu() - g["acc"]).abs().max() <= 2.0 / 144 and acc[0].item() == 0.0
    assert acc_last.shape == g["acc_last"].shape          # the fully broken image dropped out of the later restarts
    assert (acc_last.cpu() - g["acc_last"]).abs().max() <= 2.0 / 144
    assert ((xa.cpu() - g["x_adv"]).abs() > 1e-6).float().mean().item() <= 0.03
    assert (xa.cpu() - g["x"]).abs().max() <= g["eps"] + 1e-7


@pytest.mark.parametrize("tag", ["a", "b"])
def test_eval_performance_contract_and_values(tag):
    from tools import infer
    g = load_golden(f"g12_eval_performance_{tag}")
    C = int(g["n_cls"])
    net = TinyConvNet(C, seed=int(g["seed_net"])).cuda()
    loader, s = [], 0
    for n in g["sizes"].tolist():
        loader.append((g["images"][s:s + n], g["targets"][s:s + n], "name"))
        s += n
    stats, l_out = infer.eval_performance(net, loader, n_batches=int(g["n_batches"]), n_cls=C, ignore_index=-1)
    assert set(stats) == {"mAcc", "aAcc", "mIoU"} and l_out.dtype == torch.int64 and not l_out.is_cuda
    # 16x16 maps from a conv net: allow the odd pixel whose two best logits tie to rounding
    assert (l_out != g["l_output"]).float().mean().item() <= 2e-3
    for k in ("mAcc", "aAcc", "mIoU"):
        assert stats[k] == pytest.approx(g[k], abs=3e-3)
    # a loader that carries the prediction as third tensor element is evaluated without a forward
    class Boom(torch.nn.Module):
        def forward(self, x):
            raise AssertionError("re-forwarded")
    pre = [(x, t, l_out[i:i + x.shape[0]].clamp_min(0).to(torch.uint8)) for i, (x, t, _) in
           zip([0] + torch.cumsum(g["sizes"], 0).tolist(), loader)]
    boom = Boom().cuda()
    boom.register_buffer("_d", torch.zeros(1, device="cuda"))
    stats2, l_out2 = infer.eval_performance(boom, pre[: len(l_out)], n_batches=int(g["n_batches"]), n_cls=C, device="cuda")
    assert stats2 == stats and torch.equal(l_out2, l_out)


def _cfg(tmp_path):
    import yaml
    cfg = yaml.safe_load(open(os.path.join(PKG, "configs", "pascalvoc_convnext.yaml")))
    cfg["SAVE_DIR"] = str(tmp_path) + "/"
    p = str(tmp_path / "cfg.yaml")
    yaml.safe_dump(cfg, open(p, "w"))
    return p


def test_infer_writes_reference_format_logs_that_evalsea_reads_back(tmp_path):
    """tools/infer.py:366-370 saves ONE (N,H,W) int64 tensor per loss; evalSEA (worse_only.py:191-199, 354-362) loads
    them by name.  The numbers evalSEA recomputes from the files equal the ones the run printed."""
    from tools import infer
    from tools.worse_only import evalSEA
    out = str(tmp_path / "s.json")
    g = torch.Generator().manual_seed(3)
    images = torch.rand(5, 3, 64, 64, generator=g)
    labels = torch.randint(0, 21, (5, 64, 64), generator=g)
    labels[torch.rand(5, 64, 64, generator=g) < 0.05] = -1
    data = str(tmp_path / "data.pt")
    torch.save({"images": images, "labels": labels}, data)
    s = infer.main(["--cfg", _cfg(tmp_path), "--eps", "8", "--n_iter", "10", "--data", data, "--random_init",
                    "--batch_size", "2", "--json", out, "--cleanup", "0", "--save_argmax"])
    name = "UperNet_ConvNeXt-T_CVST"
    logs = [torch.load(os.path.join(str(tmp_path), "argmax-logs", f"{name}_{l}_8.0.pt")) for l in infer.LOSSES]
    for lg in logs:
        assert lg.shape == (5, 64, 64) and lg.dtype == torch.int64
        assert torch.equal(lg == -1, labels == -1)          # masked in place like eval_performance does (infer.py:88-90)
    sd = {"seed": 225, "worst_Acc": 0, "worst_Acc_indiv": 0, "final_miou": 0, "loss-wise_miou": []}
    ev = evalSEA(labels, [], 8.0, 21, "SEA_" + name, str(tmp_path), sd, name)    # empty l_outs: load from disk
    ev.worse_case_eval(bs=2, n_batches=-1)
    random.seed(225)
    ev.worst_case_miou()
    assert ev.saveDict["worst_Acc"] == pytest.approx(s["worst_Acc"], rel=1e-6)
    assert ev.saveDict["final_miou"] == s["final_miou"]
    torch.testing.assert_close(ev.saveDict["worst_Acc_indiv"], torch.tensor(s["worst_Acc_indiv"]), rtol=1e-6, atol=0)


def test_infer_single_attack_sizes_its_tables_for_one_attack(tmp_path):
    """`--attack X`: every table is sized for the one attack (a 3-slot table with two empty slots made the
    worst case min() over zeros)."""
    from tools import infer
    s = infer.main(["--cfg", _cfg(tmp_path), "--eps", "8", "--n_iter", "10", "--synthetic", "4", "--image_size", "64",
                    "--batch_size", "2", "--attack", "mask-ce-avg", "--cleanup", "1"])
    assert s["attacks"] == ["mask-ce-avg"] and len(s["worst_Acc_indiv"]) == 1 and len(s["loss-wise_miou"]) == 1
    assert s["worst_Acc"] == pytest.approx(s["worst_Acc_indiv"][0], rel=1e-6)   # min over ONE attack
    assert 0.0 < s["worst_Acc"] < 1.0 and 0.0 < s["final_miou"] <= s["loss-wise_miou"][0] + 1e-6


@pytest.mark.parametrize("C,grad", [(21, True), (21, False), (151, True), (151, False), (200, True)])
def test_nan_and_inf_logits_follow_torch_max(C, grad):
    """torch.max returns the index of the FIRST NaN if there is one (else the first maximum); +-inf are ordinary
    values.  The reference's argmax (attacker.py:145, 370, 485) inherits that; so does K2."""
    from semseg import _native as N
    g = torch.Generator().manual_seed(C)
    logits = torch.randn(2, C, 8, 16, generator=g) * 3
    y = torch.randint(0, C, (2, 8, 16), generator=g)
    logits[0, 3, 0, 0] = float("nan")                       # one NaN
    logits[0, 5, 0, 1] = float("nan")
    logits[0, 2, 0, 1] = float("nan")                       # two NaNs: the first one wins
    logits[0, 7, 0, 2] = float("inf")                       # +inf is the maximum
    logits[0, 1, 0, 3] = float("inf")
    logits[0, 4, 0, 3] = float("inf")                       # two +inf: first index
    logits[0, :, 0, 4] = float("-inf")                      # all -inf: index 0
    logits[0, 0, 0, 5] = float("-inf")                      # a -inf among finite values is never the maximum
    logits[1, C - 1, 3, 3] = float("nan")                   # NaN in the last class
    ref = logits.max(1)[1]
    for lay in ("nchw", "nhwc"):
        lg = logits.cuda()
        if lay == "nhwc":
            lg = lg.contiguous(memory_format=torch.channels_last)
        pred = torch.empty(2, 8, 16, dtype=torch.int64, device="cuda")
        r = N.loss_fwd_bwd(lg, y.cuda(), None, 0, 3, 1.0, want_grad=grad, pred=pred)
        assert torch.equal(pred.cpu(), ref), lay
        # image 0 holds NaN / inf pixels -> its loss is NaN like the reference's; image 1 too (one NaN logit)
        assert torch.isnan(r["track_sum"]).all()
