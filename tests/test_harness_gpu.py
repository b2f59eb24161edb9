"""SEA harness on the GPU: evalSEA against the reference's goldens and a tiny end-to-end tools.infer
run on synthetic data (`-m gpu`)."""
import json
import os
import random

import pytest
import torch

from conftest import PKG, load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_evalsea_golden(tmp_path, tag):
    from tools.worse_only import evalSEA
    g = load_golden(f"g7_evalsea_{tag}")
    C, bs = int(g["n_cls"]), int(g["bs"])
    sd = {"seed": 225, "worst_Acc": 0, "worst_Acc_indiv": 0, "final_miou": 0, "loss-wise_miou": []}
    ev = evalSEA(g["targets"], [p for p in g["preds"]], 4.0, C, "SEA_test", str(tmp_path), sd, "m")
    ev.worse_case_eval(bs=bs, n_batches=-1, compat_slicing=True)  # reproduce the reference's batch slicing (D9)
    assert ev.saveDict["worst_Acc"] == pytest.approx(g["worst_Acc"], rel=1e-6)
    torch.testing.assert_close(ev.saveDict["worst_Acc_indiv"], g["worst_Acc_indiv"], rtol=1e-6, atol=0)
    random.seed(225)
    ev.worst_case_miou()
    assert ev.saveDict["final_miou"] == g["final_miou"]
    st = torch.load(os.path.join(str(tmp_path), "test_results", "stats_SEA_test_4.0.pt"))
    assert torch.equal(st["run_int_imwise"], g["ints"]) and torch.equal(st["run_union_imwise"], g["unions"])


def test_infer_synthetic_end_to_end(tmp_path):
    import yaml
    from tools import infer
    cfg = yaml.safe_load(open(os.path.join(PKG, "configs", "pascalvoc_convnext.yaml")))
    cfg["SAVE_DIR"] = str(tmp_path) + "/"
    cfg_path = str(tmp_path / "cfg.yaml")
    yaml.safe_dump(cfg, open(cfg_path, "w"))
    out = str(tmp_path / "summary.json")
    infer.main(["--cfg", cfg_path, "--eps", "8", "--n_iter", "10", "--synthetic", "4", "--image_size", "64",
                "--batch_size", "2", "--json", out, "--cleanup", "0"])
    s = json.load(open(out))
    assert s["n_images"] == 4 and s["clean"]["aAcc"] == 1.0      # labels are the clean prediction
    assert 0.0 <= s["worst_Acc"] <= min(s["worst_Acc_indiv"]) + 1e-6
    assert s["final_miou"] <= s["loss-wise_miou"][0] + 1e-6      # the greedy starts from attack 0 and only improves
    saved = torch.load(os.path.join(str(tmp_path), f"worse_SEA_UperNet_ConvNeXt-T_CVST_pascalvoc_8.0.pt"))
    assert set(saved) == {"seed", "worst_Acc", "worst_Acc_indiv", "final_miou", "loss-wise_miou"}
    assert os.path.exists(os.path.join(str(tmp_path), "sea-stats", "loss_wise_UperNet_ConvNeXt-T_CVST_js-avg_N_8.0.txt"))


def test_pirat_training_step_runs_and_keeps_param_grads_clean(tmp_path):
    """inner PGD must not leave anything in .grad (the reference's loss.backward() does) and the
    outer step must train"""
    import yaml
    from tools import train_rob_seg
    from semseg.models import UperNetForSemanticSegmentation
    from semseg.val import Pgd_Attack_1
    model = UperNetForSemanticSegmentation("ConvNeXt-T_CVST", 21, None).cuda().eval()
    x = torch.rand(2, 3, 64, 64, device="cuda")
    y = torch.randint(0, 21, (2, 64, 64), device="cuda")
    x_adv, logits, _ = Pgd_Attack_1(epsilon=4 / 255, alpha=1e-2, num_iter=3, los="pgd").adv_attack(model, x, y)
    assert all(p.grad is None for p in model.parameters())
    assert (x_adv - x).abs().max() <= 4 / 255 + 1e-6 and x_adv.min() >= 0 and x_adv.max() <= 1
    cfg = yaml.safe_load(open(os.path.join(PKG, "configs", "pascalvoc_convnext.yaml")))
    cfg["TRAIN"]["IMAGE_SIZE"] = [64, 64]
    cfg["TRAIN"]["N_ITERS"] = 2
    cfg_path = str(tmp_path / "cfg.yaml")
    yaml.safe_dump(cfg, open(cfg_path, "w"))
    out = str(tmp_path / "train.json")
    train_rob_seg.main(["--cfg", cfg_path, "--synthetic", "8", "--steps", "3", "--warmup", "1", "--batch_size", "2",
                        "--json", out])
    s = json.load(open(out))
    assert s["inner_pgd_steps"] == 2 and s["samples_per_s"] > 0 and s["last_loss"] == s["last_loss"]
