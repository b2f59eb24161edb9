#!/usr/bin/env python3
"""Why did the 5-step PGD on UperNet-ConvNeXt-S give 204 mismatching samples on one MI355X lease and 205 on another?

    python devtools/box_determinism.py            (GPU box)

Hypothesis: the only library kernels on the model path whose CHOICE is not a pure function of the shapes are MIOpen's
(the two stride-2 3x3 stem convolutions): `torch.backends.cudnn.benchmark = True` (set by tools/infer.py and
tools/train_rob_seg.py without --deterministic, and leaked into the rest of a pytest process that calls their main())
makes MIOpen time candidates (Find) and write the winner to the user find-db, which immediate mode then consults.

Phases, each hashing the PGD result and counting mismatches against the reference's samples:
  A  fresh process, benchmark False        (twice: run-to-run)
  B  benchmark True (Find runs)
  C  benchmark False again                 (does the Find result stick?)
"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd"), os.path.join(ROOT, "tests")]

import torch  # noqa: E402


def main():
    from real_models import EPS, setup
    from semseg import val as V
    if "--history" in sys.argv:
        # a different allocation / library history before the measured model is built (what a pytest process that ran
        # other models first looks like): UperNet-T 5-step SEA at B=2 and a Segmenter forward+backward
        from semseg import attacker as A
        g0, m0, x0, _, y0, w0, C0 = setup("upernet_t")
        A.apgd_largereps(m0.cuda(), x0.cuda(), y0.cuda(), w0.cuda(), norm="Linf", eps=EPS, n_iter=5, use_rs=True,
                         loss="mask-ce-avg", track_loss="ce-avg", num_classes=C0, early_stop=True)
        g1, m1, x1_, _, y1, w1, C1 = setup("segmenter")
        xin = x1_.cuda().requires_grad_(True)
        m1.cuda()(xin).square().mean().backward()
        del m0, m1, xin
        keep = [torch.empty(n, device="cuda") for n in (12345, 777, 3 * 1000 * 1000 + 5)]   # shifts later addresses
        print("history: ran UperNet-T SEA + Segmenter fwd/bwd first;", torch.cuda.memory_allocated() >> 20, "MiB held", flush=True)
    g, model, x, x1, y, w, C = setup("upernet_s")
    model, x, y = model.cuda(), x.cuda(), y.cuda()
    torch.manual_seed(int(g["pgd_seed"]))
    delta0 = torch.zeros(2, 3, 512, 512).uniform_(-EPS, EPS).cuda()
    atk = V.Pgd_Attack_1(epsilon=EPS, alpha=1e-2, num_iter=int(g["pgd_steps"]), los="pgd")

    def run(tag):
        xa, logits, _ = atk.adv_attack(model, x, y, delta0=delta0)
        got = xa.flatten()[g["idx"].cuda()].cpu()
        n_bad = int(((got - g["pgd_x_adv_samples"]).abs() > 1e-6).sum())
        h = hashlib.sha1(xa.cpu().numpy().tobytes()).hexdigest()[:12]
        hl = hashlib.sha1(logits.float().cpu().numpy().tobytes()).hexdigest()[:12]
        print(f"{tag:34s} x_adv sha1 {h}  logits sha1 {hl}  mismatching samples {n_bad}/4096", flush=True)
        return h

    import platform
    print("host:", platform.processor() or platform.machine(), "| cpu capability:", torch.backends.cpu.get_cpu_capability(),
          "| GPU:", torch.cuda.get_device_properties(0).name, torch.cuda.get_device_properties(0).multi_processor_count, "CUs")
    print("delta0 sha1", hashlib.sha1(delta0.cpu().numpy().tobytes()).hexdigest()[:12], "| weights sha1",
          hashlib.sha1(b"".join(v.detach().cpu().numpy().tobytes() for v in model.state_dict().values())).hexdigest()[:12])
    db = os.path.expanduser("~/.config/miopen")
    print("user find-db before:", sorted(os.listdir(db)) if os.path.isdir(db) else None)
    torch.backends.cudnn.benchmark = False
    a1 = run("A  benchmark=False (fresh)")
    a2 = run("A' benchmark=False (again)")
    torch.backends.cudnn.benchmark = True
    b = run("B  benchmark=True (Find)")
    b2 = run("B' benchmark=True (again)")
    torch.backends.cudnn.benchmark = False
    c = run("C  benchmark=False after Find")
    print("user find-db after:", sorted(os.listdir(db)) if os.path.isdir(db) else None)
    print("run-to-run identical:", a1 == a2, "| Find changes the bits:", a1 != b, "| Find result sticks:", c != a1)


if __name__ == "__main__":
    main()
