"""Where does the input-gradient difference between two correct arithmetics come from?  upernet_s (configs[3]/[4] model), the
clean point p0 of tests/test_real_models_gpu.py: gradient with the M9 stem kernels vs the library stem, against the reference's
sampled values and against each other -- is the difference spread (rounding) or local (a pixel changing sides)?
   gpurun -- python devtools/grad_flip_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "robust-segmentation_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from real_models import setup  # noqa: E402
from semseg import _native as N, attacker as A  # noqa: E402
from semseg.models import convnext_upernet as M  # noqa: E402

g, model, x, x1, y, w, C = setup("upernet_s")
model, x, y, w = model.cuda(), x.cuda(), y.cuda(), w.cuda()
HW = x.shape[-2] * x.shape[-1]
yc = A.compact_labels(y, C)
loss = "mask-ce-avg"
key = "p0_mask_ce_avg"
out = {}
for fused_stem in (True, False):
    old, M.USE_FUSED_STEM = M.USE_FUSED_STEM, fused_stem
    try:
        x_in, logits = A._forward_logits(model, x, True, lowres=False)
        r = N.loss_fwd_bwd(logits.detach(), yc, w, N.MODE_BY_NAME[loss], 3, 1.0 / HW, want_grad=True)
        grad = A._input_grad(logits, x_in, r["dlogits"])
        out[fused_stem] = (grad.detach().clone(), logits.detach().max(1)[1].clone(), r["n_correct"].clone(), logits.detach().clone())
    finally:
        M.USE_FUSED_STEM = old
ref = g[key + "_grad"]
idx = g["grad_idx"].cuda()
for k, name in ((True, "M9 stem"), (False, "library stem")):
    got = out[k][0].flatten()[idx].cpu()
    e = got - ref
    top = e.abs().topk(20).values
    print(f"{name:13s}: rel L2 error vs reference {e.norm() / ref.norm():.3e}; share of the squared error in its 20 / 200 largest of "
          f"{e.numel()} samples: {(top ** 2).sum() / (e ** 2).sum():.2f} / {(e.abs().topk(200).values ** 2).sum() / (e ** 2).sum():.2f};  n_correct {out[k][2].tolist()}")
d = out[True][0] - out[False][0]
print(f"M9 vs library stem: rel L2 difference of the full gradients {d.norm() / out[False][0].norm():.3e}")
e2 = (d ** 2).sum(1)                        # (B, H, W)
for b in range(d.shape[0]):
    m = e2[b]
    iy, ix = divmod(int(m.argmax()), m.shape[1])
    y0, y1, x0, x1_ = max(iy - 48, 0), iy + 48, max(ix - 48, 0), ix + 48
    print(f"  image {b}: {m[y0:y1, x0:x1_].sum() / m.sum():.2f} of the squared difference lies within 48 pixels of ({iy}, {ix}); "
          f"predictions differ at {(out[True][1][b] != out[False][1][b]).sum().item()} pixels, "
          f"correctness differs at {((out[True][1][b] == y[b]) != (out[False][1][b] == y[b])).sum().item()}")
    pm = ((out[True][1][b] == y[b]) != (out[False][1][b] == y[b])).nonzero()
    for q in pm[:4].tolist():
        lt, lf = out[True][3][b, :, q[0], q[1]], out[False][3][b, :, q[0], q[1]]
        t2 = lt.topk(2).values
        print(f"    pixel {q}: top-2 logit gap with the M9 stem {float(t2[0] - t2[1]):.2e}, max |logit difference| between the stems {float((lt - lf).abs().max()):.2e}")
