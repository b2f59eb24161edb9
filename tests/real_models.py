"""Shared set-up of the real-model parity tests (BASELINE configs): the build's model with the seeded weights the
golden generator copied into the reference model (oracle/gen_goldens.py --config1-pins / --config3 / --config4),
the two synthetic 512x512 images, the labels (= the reference's clean prediction) and the per-stage random starts
the reference drew from the CPU generator."""
import sys

import torch

from conftest import PKG, load_golden

sys.path.insert(0, PKG)

EPS = 4.0 / 255
CASES = {
    # golden file, model family, backbone, classes
    "upernet_t": ("g11_config1_pins_upernet_t", "upernet", "ConvNeXt-T_CVST", 21),
    "segmenter": ("g9_config3_segmenter_vits", "segmenter", "vit_small_patch16_224", 151),
    "upernet_s": ("g10_config4_upernet_s", "upernet", "ConvNeXt-S_CVST", 151),
}
LOSSES = ("mask-ce-bal", "mask-ce-avg", "js-avg")


def build_model(kind, backbone, n_cls):
    torch.manual_seed(0)
    if kind == "upernet":
        from semseg.models import UperNetForSemanticSegmentation
        return UperNetForSemanticSegmentation(backbone, n_cls, None).eval()
    from semseg.models import create_segmenter
    from semseg.utils.utils import load_config_segmenter
    cfg, _ = load_config_segmenter(backbone, n_cls)
    return create_segmenter(cfg, None, backbone).eval()


def setup(case):
    from semseg.utils.utils import ADE_WTS, VOC_WTS
    name, kind, backbone, C = CASES[case]
    g = load_golden(name)
    model = build_model(kind, backbone, C)
    x = torch.rand(2, 3, 512, 512, generator=torch.Generator().manual_seed(1234))
    u = torch.rand(x.shape, generator=torch.Generator().manual_seed(77))
    x1 = (x + 2 * EPS * (2 * u - 1)).clamp(0.0, 1.0)
    w = torch.tensor(VOC_WTS if C == 21 else ADE_WTS)
    return g, model, x, x1, g["y"].long(), w, C


def stage_noises(x, seed=4321):
    """the reference drew torch.rand_like(x) once per apgd_largereps stage from the CPU generator"""
    torch.manual_seed(seed)
    return [torch.rand_like(x) for _ in range(3)]


class Bounds:
    """Every tolerance of the end-to-end real-model checks goes through here: the MEASURED value is printed next to
    its bound (pytest -s; profiles/r3_real_models_bounds.log holds two leases) and all violations are raised together
    after the table, so a red run shows how far off every quantity was."""

    def __init__(self, title):
        self.title, self.rows = title, []

    def check(self, name, value, bound):
        self.rows.append((name, float(value), float(bound)))

    def report(self):
        print(f"\n[{self.title}]")
        for name, v, b in self.rows:
            print(f"  {name:58s} measured {v:11.4e}   bound {b:9.2e}   margin x{(b / v if v > 0 else float('inf')):.1f}"
                  + ("   <-- VIOLATED" if v > b else ""))
        bad = [(n, v, b) for n, v, b in self.rows if v > b]
        assert not bad, bad
