#!/usr/bin/env python3
"""eager ApgdRun and HIP-graph ApgdRun in lockstep on the same inputs: first iteration / quantity that differs (debug aid)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd")]
import torch  # noqa: E402
from semseg import attacker as A  # noqa: E402
from semseg.models import UperNetForSemanticSegmentation  # noqa: E402
from semseg.utils.utils import VOC_WTS  # noqa: E402

torch.manual_seed(0)
model = UperNetForSemanticSegmentation("ConvNeXt-T_CVST", 21, None).eval().cuda()
x = torch.rand(2, 3, 512, 512, generator=torch.Generator().manual_seed(5)).cuda()
with torch.no_grad():
    y = model(x).max(1)[1]
w = torch.tensor(VOC_WTS).cuda()
noise = torch.rand(x.shape, generator=torch.Generator().manual_seed(6)).cuda()
n_iter = 30
y[0, :4] = -1
loss = sys.argv[1] if len(sys.argv) > 1 else "mask-ce-bal"
recs = []
for graph in (False, True, False):
    A.USE_HIP_GRAPH = graph
    x0 = (x + 8 / 255 * (2 * noise - 1)).clamp(0, 1)
    run = A.ApgdRun(model, x, y, 8 / 255, n_iter, loss, "ce-avg", True, 21, w, x0)
    run.start()
    rec = []
    for i in range(n_iter):
        run.step(i)
        rec.append((run.x_adv.clone(), run.grad.clone(), run.pred.clone(), run.st.loss_best.clone()))
    run.release_graphs()
    torch.cuda.synchronize()
    recs.append(rec)
names = ("x_adv", "grad", "pred", "loss_best")
for tag, other in (("graph", recs[1]), ("eager again", recs[2])):
    first = None
    for i in range(n_iter):
        d = [int((a != b).sum()) for a, b in zip(recs[0][i], other[i])]
        if any(d):
            first = (i, dict(zip(names, d)))
            break
    print(f"{loss}: eager vs {tag}: first difference {first}")
    if first:
        i = first[0]
        gd = (recs[0][i][1] - other[i][1]).abs()
        nz = (gd > 0).nonzero()
        print("   grad diff max", gd.max().item(), "of max|g|", recs[0][i][1].abs().max().item(), "count", len(nz),
              "rows", sorted(set(nz[:, 2].tolist()))[:12] if len(nz) else [], "cols", sorted(set(nz[:, 3].tolist()))[:12] if len(nz) else [])
