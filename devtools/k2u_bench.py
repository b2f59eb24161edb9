#!/usr/bin/env python3
"""K2u (loss fused with the model's final bilinear up-sampling) against up-sample + K2 + up-sample-backward, B = 8, 512^2:
    python devtools/k2u_bench.py            (SEA_K2U_POW2=0 in the environment selects the general gather kernel)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd")]
from semseg import _native as N  # noqa: E402
from semseg.models import convnext_upernet as M  # noqa: E402

N.lib()
B, H, W = 8, 512, 512
HW = H * W


def timeit(fns, rounds=7, reps=5):
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    best = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                f()
            b.record()
            torch.cuda.synchronize()
            best[k].append(a.elapsed_time(b) * 1e3 / reps)
    return {k: (min(v), sorted(v)[len(v) // 2]) for k, v in best.items()}


print("K2u kernel:", "general gather (SEA_K2U_POW2=0)" if os.environ.get("SEA_K2U_POW2") == "0" else "power-of-two, lanes = classes")
for C, hl, lab in ((21, 128, "UperNet x4"), (151, 128, "UperNet x4"), (151, 32, "Segmenter x16")):
    g = torch.Generator(device="cuda").manual_seed(C + hl)
    low = torch.randn(B, C, hl, hl, generator=g, device="cuda") * 3
    up0 = torch.nn.functional.interpolate(low, size=(H, W), mode="bilinear")
    y8 = up0.max(1)[1]
    flip = torch.rand(B, H, W, generator=g, device="cuda") < 0.5          # half the pixels misclassified: the masked losses skip them
    y8[flip] = torch.randint(0, C, (int(flip.sum()),), generator=g, device="cuda")
    y8 = y8.to(torch.uint8)
    del up0
    w = torch.rand(C, device="cuda")
    dlow = torch.empty_like(low)
    pred8 = torch.empty(B, H, W, dtype=torch.uint8, device="cuda")
    dl = torch.empty(B, C, H, W, device="cuda")
    ws = N.loss_workspace(B, HW, "cuda")
    wsu = torch.empty(N.lib().sea_loss_upsampled_workspace_bytes(B, C, hl, hl, H, W), dtype=torch.uint8, device="cuda")

    def unfused(mode=1):
        up = M._UpsampleBilinear.apply(low.detach().requires_grad_(True), (H, W)) if hasattr(M, "_UpsampleBilinear") else None
        r = N.loss_fwd_bwd(up.detach(), y8, w, mode, 3, 1.0 / HW, True, pred=pred8, workspace=ws, dlogits=dl)
        return torch.autograd.grad(up, up.grad_fn.next_functions[0][0].variable, r["dlogits"])

    t = timeit({
        f"K2u C={C} {lab} mask-ce-bal +grad": lambda: N.loss_fwd_bwd_upsampled(low, y8, w, 1, 3, 1.0 / HW, True, pred=pred8, dlow=dlow, workspace=wsu),
        f"K2u C={C} {lab} js +grad": lambda: N.loss_fwd_bwd_upsampled(low, y8, w, 2, 3, 1.0 / HW, True, pred=pred8, dlow=dlow, workspace=wsu),
        f"K2u C={C} {lab} no-grad": lambda: N.loss_fwd_bwd_upsampled(low, y8, w, 1, 3, 1.0 / HW, False, pred=pred8, workspace=wsu),
        f"M2 up-sample + K2 + M2 backward C={C} {lab}": unfused,
    })
    for k, (mn, med) in t.items():
        print(f"   {k:55s} min {mn:8.1f} us   median {med:8.1f} us")
    del low, dl, dlow
