"""Constants and small helpers of the SEA harness (counterpart of semseg/utils/utils.py).

Only what the attack path and its harness touch: the class-balance weights used by ``mask-ce-bal``
(utils.py:16-192; data, reproduced value for value and checked against tests/golden/g0_weights.npz),
output-directory helpers, the per-loss report writer, model naming, the Segmenter config loader and
the two loggers.
"""
from __future__ import annotations

import os
import random
import shutil
from pathlib import Path

import numpy as np
import torch
import yaml

# inverse-frequency style class weights, ADE20K (151 entries incl. background) and PASCAL-VOC (21)
ADE_WTS = [
    2.5511e-05, 3.6983e-05, 4.757e-05, 6.6522e-05, 0.00011128, 0.00010635,
    0.00016683, 0.00017398, 0.00022595, 0.00026104, 0.00034525, 0.0003268,
    0.00046724, 0.00027333, 0.00038664, 0.00051101, 0.00046147, 0.00028707,
    0.00048777, 0.00056734, 0.00052263, 0.00057571, 0.00079284, 0.00071656,
    0.00098619, 0.00073393, 0.00060752, 0.00060696, 0.0010648, 0.0015916,
    0.00074704, 0.0013956, 0.0010427, 0.0016245, 0.0013812, 0.0012803,
    0.0015659, 0.0023384, 0.0026498, 0.0021948, 0.0019984, 0.0021434,
    0.0022654, 0.0023339, 0.0026016, 0.0029368, 0.0024439, 0.0025844,
    0.0023346, 0.001017, 0.0027078, 0.0037222, 0.0030739, 0.0030697,
    0.0050181, 0.0047774, 0.0020477, 0.0031477, 0.0028421, 0.0037206,
    0.0025296, 0.0021699, 0.0028066, 0.002808, 0.0055795, 0.0040186,
    0.0048758, 0.0035471, 0.0031513, 0.0030316, 0.0039002, 0.0050847,
    0.0048401, 0.0059311, 0.0053158, 0.0050188, 0.0040362, 0.0044585,
    0.0052076, 0.0044833, 0.0055491, 0.0057523, 0.0055545, 0.0087588,
    0.0050301, 0.0054497, 0.0076726, 0.0051451, 0.0079943, 0.0044696,
    0.0074416, 0.0067389, 0.007875, 0.0055496, 0.012515, 0.0051635,
    0.0081806, 0.0099495, 0.010522, 0.0060337, 0.011848, 0.010531,
    0.0060837, 0.0080876, 0.01175, 0.0082409, 0.0068528, 0.0081382,
    0.0087929, 0.0076437, 0.0057786, 0.013009, 0.018844, 0.010949,
    0.0042059, 0.0057906, 0.012998, 0.014171, 0.0070287, 0.0090963,
    0.010115, 0.01051, 0.013813, 0.012319, 0.014154, 0.015693,
    0.015035, 0.01112, 0.016888, 0.0073436, 0.014521, 0.0093029,
    0.014782, 0.011918, 0.017509, 0.020762, 0.014547, 0.020312,
    0.010543, 0.018876, 0.036659, 0.020046, 0.022035, 0.014011,
    0.015645, 0.011985, 0.010001, 0.027073, 0.021668, 0.018419,
    0.021877,
]

VOC_WTS = [
    0.0007, 0.0531, 0.1394, 0.05, 0.0814, 0.0575,
    0.0256, 0.0312, 0.0198, 0.0626, 0.0382, 0.0457,
    0.0212, 0.0404, 0.0421, 0.0089, 0.0915, 0.0585,
    0.0366, 0.0279, 0.0677,
]

_CONFIG_DIR = Path(__file__).resolve().parents[2] / "configs"


def make_attack_dirs(saveloc):
    for sub in ("test_results", "sea-stats", "argmax-logs"):
        (Path(saveloc) / sub).mkdir(parents=True, exist_ok=True)


def remove_dirs(saveloc):
    try:
        shutil.rmtree(Path(saveloc) / "test_results")
        shutil.rmtree(Path(saveloc) / "argmax-logs")
    except OSError:
        print("Couldn't delete intermediate files")


def writeIndivloss(saveloc, modelName, clean_stats, test_eps, loss_, adv_stats):
    """Append the per-loss report (same file name and line layout as utils.py:236-245)."""
    path = os.path.join(str(saveloc), "sea-stats", f"loss_wise_{modelName}_{loss_}_N_{test_eps}.txt")
    with open(path, "a+") as f:
        f.write(f"{modelName} \n")
        f.write(f"Clean stats: {clean_stats}\n")
        f.write(f"----- Linf radius: {test_eps} ------")
        f.write(f"Attack: {loss_} \n")
        f.write(f"Adversarial results: {adv_stats}\n")


def getModelName(mname, backname):
    if mname == "SegMenter":
        return "SegMent_" + backname
    if mname == "UperNetForSemanticSegmentation":
        return "UperNet_" + backname
    return "PSPNet_RN50"


def load_config_segmenter(backbone, n_cls, config_path=None):
    """ViT variant + mask-transformer decoder config (utils.py:258-278).  The reference opens
    ./configs/segmenter.yml relative to the CWD; here the packaged copy is the default."""
    path = config_path or (Path("./configs/segmenter.yml") if Path("./configs/segmenter.yml").exists()
                           else _CONFIG_DIR / "segmenter.yml")
    with open(path) as f:
        cfg = yaml.load(f, Loader=yaml.FullLoader)
    model_cfg = cfg["model"][f"{backbone}"]
    dataset_cfg = cfg["dataset"]["ade20k"]
    decoder_cfg = cfg["decoder"]["mask_transformer"]
    crop = dataset_cfg.get("crop_size", 512)
    model_cfg["image_size"] = (crop, crop)
    model_cfg["backbone"] = backbone
    model_cfg["dropout"] = 0.0
    model_cfg["drop_path_rate"] = 0.1
    decoder_cfg["name"] = "mask_transformer"
    model_cfg["decoder"] = decoder_cfg
    model_cfg["n_cls"] = n_cls
    return model_cfg, dataset_cfg


class Logger:
    """print + append to ``<log_path>.txt`` (utils.py:311-320)."""

    def __init__(self, log_path):
        self.log_path = log_path + ".txt"

    def log(self, str_to_log):
        print(str_to_log)
        if self.log_path is not None:
            with open(self.log_path, "a") as f:
                f.write(str_to_log + "\n")
                f.flush()


def fix_seeds(seed: int = 3407) -> None:
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)


def makedir(path):
    os.makedirs(path, exist_ok=True)
