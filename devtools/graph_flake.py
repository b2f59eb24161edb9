#!/usr/bin/env python3
"""the bitwise eager-vs-graph test of tests/test_attack_gpu.py in a loop (debug aid): failure rate under env switches"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd")]
import torch  # noqa: E402
from semseg import attacker as A  # noqa: E402
from semseg.models import UperNetForSemanticSegmentation  # noqa: E402
from semseg.utils.utils import VOC_WTS  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 4
poll = int(os.environ.get("POLL", "8"))
torch.manual_seed(0)
model = UperNetForSemanticSegmentation("ConvNeXt-T_CVST", 21, None).eval().cuda()
x = torch.rand(2, 3, 512, 512, generator=torch.Generator().manual_seed(5)).cuda()
with torch.no_grad():
    y = model(x).max(1)[1]
y[0, :4] = -1
w = torch.tensor(VOC_WTS).cuda()
noise = torch.rand(x.shape, generator=torch.Generator().manual_seed(6)).cuda()


def run(graph):
    A.USE_HIP_GRAPH = graph
    return A.apgd_train(model, x, y, "Linf", 8.0 / 255, n_iter=30, use_rs=True, loss="mask-ce-bal", early_stop=True,
                        track_loss="ce-avg", num_classes=21, weights=w, noise=noise, return_pred=True, poll_every=poll)


ref = run(False)
bad_g = bad_e = 0
for t in range(trials):
    g = run(True)
    e = run(False)
    bad_g += not all(torch.equal(a, b) for a, b in zip(ref, g))
    bad_e += not all(torch.equal(a, b) for a, b in zip(ref, e))
print(f"{trials} trials: graph differs {bad_g}, eager differs {bad_e}   env " +
      " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("SEA_") or k == "POLL"))
