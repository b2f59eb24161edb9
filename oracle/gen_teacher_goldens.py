#!/usr/bin/env python3
"""Per-step ("teacher-forced") goldens of the REAL reference on the three real models (build container only).

    python oracle/gen_teacher_goldens.py upernet_t | segmenter | upernet_s | all

The reference's attacks are run UNMODIFIED (semseg.attacker.apgd_largereps: attacker.py:385-569, 662-728;
semseg.val.Pgd_Attack_1: val.py:181-218) on ONE synthetic 512x512 image with the build's seeded weights loaded
strict=True into the reference's model.  A recording wrapper around the model sees every iterate the attack feeds to
the network and, through a tensor hook, every input gradient it gets back.  From that record the fixture stores, per
model evaluation e:

  * a RECIPE that rebuilds the reference's iterate exactly from earlier ones with the L-inf arithmetic of the oracle
    (random start / stage re-projection of evaluation b / APGD step from the pair (x_b, g_b) with x_old = x_o, step
    eps*2/2^m, momentum a) - found by search and verified bit for bit here, so the fixture needs no full images;
  * 1024 sampled values of the iterate (the test's replay is checked against them);
  * the reference's per-image attack loss, tracking loss (ce-avg), number of correct pixels, and the number of pixels
    whose two largest logits are closer than 2e-4 * max|logit| (pixels whose arg-max rounding may legitimately flip);
  * where the attack took a gradient: sign(g) of EVERY element as a packed bit plane (+ the indices of exact zeros)
    and |g| quantised to 2 bits against 1e-4 / 1e-3 / 1e-2 of max|g| (which elements are above rounding level).

With these a GPU test can feed the reference's iterate k to the device path and compare loss, counts, gradient sign
and the next iterate against the reference at every step, instead of comparing chaotic end states.
Only data is written (tests/golden/t1_*.npz); the reference is not modified and none of its text is stored.
"""
import contextlib
import io
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("SEA_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "shims"), REF, ROOT]

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import sea_oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
EPS = 4.0 / 255
N_SAMPLES = 1024
NEAR_TIE = 2e-4

# case -> (model family, backbone, classes, [(loss, n_iter)], PGD steps)
CASES = {
    "upernet_t": ("upernet", "ConvNeXt-T_CVST", 21, [("mask-ce-bal", 10), ("mask-ce-avg", 5), ("js-avg", 5)], 0),
    "segmenter": ("segmenter", "vit_small_patch16_224", 151, [("mask-ce-avg", 5), ("js-avg", 5)], 0),
    "upernet_s": ("upernet", "ConvNeXt-S_CVST", 151, [("mask-ce-bal", 5)], 5),
}


class Recorder(torch.nn.Module):
    """Transparent wrapper: records the input of every forward and the gradient that reaches it."""

    def __init__(self, ref):
        super().__init__()
        self.ref = ref
        self.evals = []
        self.on_logits = None

    def forward(self, x):
        e = {"x": x.detach().clone(), "g": None}
        self.evals.append(e)
        if x.requires_grad:
            x.register_hook(lambda g, e=e: e.__setitem__("g", g.detach().clone()))
        out = self.ref(x)
        if self.on_logits is not None:
            e.update(self.on_logits(out.detach()))
        return out


def pgd_start(shape, eps, seed):
    """U(-eps, eps) start of Pgd_Attack_1 (val.py:192-193), host-independent (restated in tests/teacher.py)"""
    return eps * (2 * torch.rand(shape, generator=torch.Generator().manual_seed(seed)) - 1)


def pack_gradient(g):
    """sign plane + zero list + 2-bit magnitude level of every element"""
    g = g.flatten()
    gmax = g.abs().max()
    neg = np.packbits((g < 0).numpy())
    zeros = (g == 0).nonzero().flatten().to(torch.int32).numpy()
    lvl = ((g.abs() > 1e-4 * gmax).to(torch.uint8) + (g.abs() > 1e-3 * gmax).to(torch.uint8)
           + (g.abs() > 1e-2 * gmax).to(torch.uint8))
    return dict(neg=neg, zeros=zeros, lvl_lo=np.packbits((lvl & 1).bool().numpy()),
                lvl_hi=np.packbits((lvl >> 1).bool().numpy()), gmax=np.float32(gmax))


def logits_stats(A, y, w, loss, mask_bg):
    def fn(logits):
        with torch.no_grad():
            li = A.pixel_to_img_loss(A.criterion_dict[loss](logits, y, w), mask_bg)
            ce = A.pixel_to_img_loss(A.criterion_dict["ce-avg"](logits, y), mask_bg)
            top2 = logits.topk(2, dim=1)[0]
            near = (top2[:, 0] - top2[:, 1]) < NEAR_TIE * logits.abs().max()
            pred = logits.max(1)[1]
        return dict(li=li.clone(), ce=ce.clone(), n_correct=(pred == y).sum().to(torch.int64), n_near=near.sum(),
                    pred=pred.to(torch.uint8), near=near, absmax=logits.abs().max())
    return fn


def run_case(case):
    from gen_goldens import _build_state_dict, _reference_model
    kind, backbone, C, runs, pgd_steps = CASES[case]
    torch.set_num_threads(4)
    sd = _build_state_dict(kind, backbone, C)
    ref = _reference_model(kind, backbone, C, sd)
    os.chdir(REF)
    import semseg.attacker as A
    import semseg.val as V
    from semseg.utils.utils import ADE_WTS, VOC_WTS
    torch.Tensor.cuda = lambda self, *a, **k: self
    w = torch.tensor(VOC_WTS if C == 21 else ADE_WTS)
    x = torch.rand(2, 3, 512, 512, generator=torch.Generator().manual_seed(1234))[:1].clone()
    with torch.no_grad():
        y = ref(x).max(1)[1]
    mask_bg = (y != -1).float()
    sidx = torch.randperm(x.numel(), generator=torch.Generator().manual_seed(5))[:N_SAMPLES]
    rec = Recorder(ref).eval()

    for loss, n_iter in runs:
        rec.evals, rec.on_logits = [], logits_stats(A, y, w, loss, mask_bg)
        torch.manual_seed(4321)
        t0 = time.time()
        with contextlib.redirect_stdout(io.StringIO()):
            xa, _, acc = A.apgd_largereps(rec, x.clone(), y, w, norm="Linf", eps=EPS, n_iter=n_iter, n_restarts=1,
                                          use_rs=True, loss=loss, verbose=False, track_loss="ce-avg", log_path=None,
                                          num_classes=C, early_stop=True)
        dt = time.time() - t0
        torch.manual_seed(4321)
        noises = [torch.rand_like(x) for _ in range(3)]
        n_iters, epss = O.largereps_schedule(n_iter, EPS)
        assert len(rec.evals) == sum(n + 1 for n in n_iters), (len(rec.evals), n_iters)
        out = dict(y=y.to(torch.uint8), eps=np.float64(EPS), n_iter=np.int64(n_iter), seed=np.int64(4321),
                   sample_idx=sidx.to(torch.int32), n_evals=np.int64(len(rec.evals)), acc=acc,
                   seconds=np.float64(dt), near_tie=np.float64(NEAR_TIE))
        e0, final = 0, None
        for s, (n_it, eps_s) in enumerate(zip(n_iters, epss)):
            ev = rec.evals[e0:e0 + n_it + 1]
            # ---- evaluation 0 of the stage: random start, or the re-projected best-accuracy iterate of the last stage
            if s == 0:
                start = O.linf_random_start(x, noises[0], eps_s).clamp(0.0, 1.0)
                recipe = [0, -1, -1, 0, 1.0]
            else:
                prev = rec.evals[e0 - n_iters[s - 1] - 1:e0]
                hit = [j for j, p in enumerate(prev) if torch.equal(O.linf_project(p["x"], x, eps_s).clamp(0.0, 1.0), ev[0]["x"])]
                assert hit, "stage start is not the projection of an earlier iterate"
                b = e0 - len(prev) + hit[-1]
                start = O.linf_project(rec.evals[b]["x"], x, eps_s).clamp(0.0, 1.0)
                recipe = [1, b, -1, 0, 1.0]
            assert torch.equal(start, ev[0]["x"])
            recipes = [recipe]
            # ---- evaluations 1..n_it: one APGD step from an earlier (iterate, gradient) pair of this stage
            for e in range(1, n_it + 1):
                found = None
                for b in range(e - 1, -1, -1):
                    if ev[b]["g"] is None:
                        continue
                    for o in range(e - 1, -1, -1):
                        for m in range(0, 6):
                            for a in ((1.0,) if e == 1 else (0.75, 1.0)):
                                step = torch.full((1,), 2.0 * eps_s) / (2.0 ** m)
                                cand = O.apgd_linf_step(x, ev[b]["x"], ev[o]["x"], ev[b]["g"], step, eps_s, a)
                                if torch.equal(cand, ev[e]["x"]):
                                    found = [2, e0 + b, e0 + o, m, a]
                                    break
                            if found:
                                break
                        if found:
                            break
                    if found:
                        break
                assert found, f"no recipe for stage {s} evaluation {e}"
                recipes.append(found)
            for j, (e, r) in enumerate(zip(ev, recipes)):
                k = f"e{e0 + j}_"
                out.update({k + "recipe": np.array(r, dtype=np.float64), k + "stage": np.int64(s),
                            k + "x_samples": e["x"].flatten()[sidx], k + "li": e["li"], k + "ce": e["ce"],
                            k + "n_correct": e["n_correct"], k + "n_near": e["n_near"], k + "absmax": e["absmax"],
                            k + "has_grad": np.int64(e["g"] is not None)})
                if e["g"] is not None:
                    out.update({k + kk: v for kk, v in pack_gradient(e["g"]).items()})
            e0 += n_it + 1
        # the returned image is the best-accuracy iterate of the last stage
        last = rec.evals[e0 - n_iters[-1] - 1:e0]
        hit = [j for j, p in enumerate(last) if torch.equal(p["x"], xa)]
        assert hit
        out["final_eval"] = np.int64(e0 - len(last) + hit[0])
        for j in (0, len(rec.evals) - 1):          # arg-max maps at the first and last evaluation (+ near-tie pixels)
            out[f"e{j}_pred"] = rec.evals[j]["pred"]
            out[f"e{j}_near"] = np.packbits(rec.evals[j]["near"].flatten().numpy())
        name = f"t1_{case}_{loss.replace('-', '_')}"
        np.savez_compressed(os.path.join(OUT, name + ".npz"),
                            **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                               for k, v in out.items()})
        print(name, "evals", len(rec.evals), "recipes", [[int(r) for r in out[f'e{j}_recipe'][:4]] for j in range(len(rec.evals))],
              "acc", acc.tolist(), f"{dt:.0f}s", os.path.getsize(os.path.join(OUT, name + ".npz")) // 1024, "KiB", flush=True)

    if pgd_steps:
        # ---- PIR-AT inner PGD: Pgd_Attack_1, CE, alpha 1e-2, seeded uniform start (val.py:181-218)
        rec.evals = []
        rec.on_logits = lambda lg: dict(ce_mean=torch.nn.functional.cross_entropy(lg, y).detach().clone(),
                                        absmax=lg.abs().max())
        # the random start: Tensor.uniform_ on the CPU rounds differently per ATEN_CPU_CAPABILITY (from + u*(to-from) is
        # contracted to an FMA in the AVX2/AVX512 builds, not in the default one), so the draw the reference asks for
        # is served from torch.rand (exact 24-bit lattice) with separately rounded ops: identical on every host
        real_uniform = torch.Tensor.uniform_

        def served(self, a=0.0, b=1.0, **kw):
            assert a == -b
            return self.copy_(pgd_start(self.shape, b, 99))

        torch.Tensor.uniform_ = served
        try:
            xa, lg, _ = V.Pgd_Attack_1(epsilon=EPS, alpha=1e-2, num_iter=pgd_steps, los="pgd").adv_attack(rec, x, y)
        finally:
            torch.Tensor.uniform_ = real_uniform
        delta = pgd_start(x.shape, EPS, 99)
        out = dict(y=y.to(torch.uint8), eps=np.float64(EPS), alpha=np.float64(1e-2), seed=np.int64(99),
                   n_evals=np.int64(pgd_steps), sample_idx=sidx.to(torch.int32))
        for e, evl in enumerate(rec.evals):
            assert torch.equal(x + delta, evl["x"]), f"PGD replay differs at step {e}"
            k = f"e{e}_"
            out.update({k + "x_samples": evl["x"].flatten()[sidx], k + "ce_mean": evl["ce_mean"],
                        k + "absmax": evl["absmax"]})
            out.update({k + kk: v for kk, v in pack_gradient(evl["g"]).items()})
            delta = O.pgd_linf_step(x, delta, evl["g"], 1e-2, EPS)
        assert torch.equal((x + delta).clamp(0.0, 1.0), xa)
        out["x_adv_samples"] = xa.flatten()[sidx]
        name = f"t1_{case}_pgd"
        np.savez_compressed(os.path.join(OUT, name + ".npz"),
                            **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                               for k, v in out.items()})
        print(name, "steps", pgd_steps, "ce", [float(e["ce_mean"]) for e in rec.evals],
              os.path.getsize(os.path.join(OUT, name + ".npz")) // 1024, "KiB", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["all"]
    for c in (list(CASES) if "all" in which else which):
        run_case(c)     # one case per process is the tested way (the reference is imported afresh)
