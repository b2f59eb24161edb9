#!/usr/bin/env python3
"""Exact controller goldens at SEA's real stage lengths (build container only; test infrastructure).

    python oracle/gen_controller_goldens.py

The reference's attacks are run UNMODIFIED (semseg.attacker.apgd_train, attacker.py:260-571, at n_iter = 90 and 120 -
the stage lengths of the 300-iteration protocol - and apgd_largereps, attacker.py:662-728, at n_iter = 300 = stages
90 / 90 / 120) on the point-wise net of oracle/tiny_models.py, whose logits are bit-identical on CPU and GPU.  A recording
wrapper sees every iterate and every input gradient.  Per model evaluation e the fixture (tests/golden/g13_*.npz) stores

  * a 64-bit checksum per image of the iterate the reference fed to the model (sum of the float32 bit patterns),
  * sign(g) of every input-gradient element as two bit planes (negative / exactly zero) where a gradient was taken,
  * the reference's per-image attack loss, tracking loss and correct-pixel count,

plus the returned tensors.  The GPU test (tests/test_controller_exact_gpu.py) runs the PRODUCT's apgd_train /
apgd_largereps with HIP-graph replay on and a thin wrapper that replaces the gradient by the stored sign plane: the L-inf
step is bit-exact given the signs, so every iterate - hence every step-size halving, every restart from the best point,
every best-adversarial copy over 90- and 120-iteration stages - must reproduce the checksums bit for bit.  What may
legitimately differ is a comparison of two losses closer than the device kernel's own loss error (2e-6 relative): the
generator measures the smallest relative gap of every decisive comparison (`y1 > loss_best`, the oscillation counts) and
refuses a seed whose gap is below 2e-5.  Only data is written.
"""
import contextlib
import io
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("SEA_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "shims"), REF, ROOT]

import numpy as np  # noqa: E402
import torch  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
MIN_GAP = 2e-5


def checksum(x):
    """per-image sum of the float32 bit patterns as int64 (restated in tests/teacher.py)"""
    return x.contiguous().view(torch.int32).to(torch.int64).flatten(1).sum(1)


class Recorder(torch.nn.Module):
    def __init__(self, net, stats):
        super().__init__()
        self.net, self.stats, self.evals = net, stats, []

    def forward(self, x):
        e = {"chk": checksum(x.detach()), "g": None}
        self.evals.append(e)
        if x.requires_grad:
            x.register_hook(lambda g, e=e: e.__setitem__("g", g.detach().clone()))
        out = self.net(x)
        e.update(self.stats(out.detach()))
        return out


def smallest_gaps(track, n_iters):
    """smallest relative gap (exact ties excluded: the device repeats them exactly) of the comparisons the step-size
    controller makes on the tracking loss, per stage: y1 > loss_best (attacker.py:485-493) and the oscillation count
    loss_steps[j] > loss_steps[j-1] (attacker.py:243-248)"""
    gap, e0 = float("inf"), 0
    for n in n_iters:
        L = track[e0:e0 + n + 1]                      # (n + 1, B): start point + n iterations
        best = L[0].clone()
        for j in range(1, n + 1):
            for a, b in ((L[j], best), (L[j], L[j - 1]) if j > 1 else (L[j], best)):
                d = (a - b).abs() / b.abs().clamp_min(1e-30)
                d = d[d > 0]
                if d.numel():
                    gap = min(gap, float(d.min()))
            best = torch.maximum(best, L[j])
        e0 += n + 1
    return gap


def main():
    os.chdir(REF)
    torch.set_num_threads(2)
    import semseg.attacker as A
    from autoattack.other_utils import Logger
    from semseg.utils.utils import VOC_WTS

    from oracle.tiny_models import PointwiseNet, make_labels
    logger = Logger(None)
    C = 21
    w = torch.tensor(VOC_WTS)

    def stats_fn(y, loss, mask_bg):
        def fn(logits):
            with torch.no_grad():
                li = A.pixel_to_img_loss(A.criterion_dict[loss](logits, y, w), mask_bg)
                ce = A.pixel_to_img_loss(A.criterion_dict["ce-avg"](logits, y), mask_bg)
                n_correct = (logits.max(1)[1] == y).flatten(1).sum(1)
            return dict(li=li.clone(), ce=ce.clone(), n_correct=n_correct)
        return fn

    def pack(rec, extra):
        out = dict(extra, n_evals=np.int64(len(rec.evals)))
        out["chk"] = torch.stack([e["chk"] for e in rec.evals])
        out["li"] = torch.stack([e["li"] for e in rec.evals])
        out["ce"] = torch.stack([e["ce"] for e in rec.evals])
        out["n_correct"] = torch.stack([e["n_correct"] for e in rec.evals])
        out["has_grad"] = np.array([e["g"] is not None for e in rec.evals])
        for j, e in enumerate(rec.evals):
            if e["g"] is not None:
                g = e["g"].flatten()
                out[f"e{j}_neg"] = np.packbits((g < 0).numpy())
                out[f"e{j}_zero"] = np.packbits((g == 0).numpy())     # (masked pixels have an exactly zero gradient)
        return out

    def save(name, out):
        np.savez_compressed(os.path.join(OUT, name + ".npz"),
                            **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                               for k, v in out.items()})
        print("wrote", name, os.path.getsize(os.path.join(OUT, name + ".npz")) // 1024, "KiB", flush=True)

    # ---------------------------------------------------------------- apgd_train at the stage lengths 90 and 120
    for loss in ("mask-ce-bal", "mask-ce-avg", "js-avg"):
        for n_iter in (90, 120):
            for seed in range(20):
                net = PointwiseNet(C, seed=C + seed)
                g = torch.Generator().manual_seed(7000 + 31 * seed + n_iter)
                x = torch.rand(3, 3, 16, 16, generator=g)
                y = make_labels(net, x, ignore_frac=0.05, flip_frac=0.1, seed=C + seed)
                eps = 8.0 / 255
                x_init = (x + eps * (2 * torch.rand(x.shape, generator=g) - 1)).clamp(0, 1)
                rec = Recorder(net, stats_fn(y, loss, (y != -1).float())).eval()
                with contextlib.redirect_stdout(io.StringIO()):
                    xb, acc, lb, xba = A.apgd_train(rec, x, y, "Linf", eps, n_iter=n_iter, use_rs=False, loss=loss,
                                                    track_loss="ce-avg", logger=logger, x_init=x_init, num_classes=C,
                                                    weights=w, early_stop=True)
                assert len(rec.evals) == n_iter + 1, "early stop: pick another case"
                gap = smallest_gaps(torch.stack([e["ce"] for e in rec.evals]), [n_iter])
                if gap >= MIN_GAP:
                    break
                print(f"  {loss} {n_iter}: seed {seed} has a loss comparison within {gap:.1e} of a tie, next seed")
            else:
                raise SystemExit("no seed with a safe margin")
            save(f"g13_ctrl_train_{loss}_{n_iter}", pack(rec, dict(
                x=x, y=y, w=w, x_init=x_init, eps=np.float64(eps), n_iter=np.int64(n_iter), net_seed=np.int64(C + seed),
                min_gap=np.float64(gap), x_best=xb, acc=acc, loss_best=lb, x_best_adv=xba)))

    # ---------------------------------------------------------------- apgd_largereps(n_iter = 300): stages 90 / 90 / 120
    for loss in ("mask-ce-bal", "js-avg"):
        for seed in range(20):
            net = PointwiseNet(C, seed=C + 40 + seed)
            g = torch.Generator().manual_seed(9000 + seed)
            x = torch.rand(2, 3, 16, 16, generator=g)
            y = make_labels(net, x, ignore_frac=0.03, flip_frac=0.1, seed=seed + 1)
            rec = Recorder(net, stats_fn(y, loss, (y != -1).float())).eval()
            torch.manual_seed(4321)
            with contextlib.redirect_stdout(io.StringIO()):
                xa, _, acc = A.apgd_largereps(rec, x.clone(), y, w, norm="Linf", eps=4.0 / 255, n_iter=300, n_restarts=1,
                                              use_rs=True, loss=loss, verbose=False, track_loss="ce-avg", log_path=None,
                                              num_classes=C, early_stop=True)
            if len(rec.evals) != 303:
                print(f"  largereps {loss}: seed {seed} stops early ({len(rec.evals)} evaluations), next seed")
                continue
            gap = smallest_gaps(torch.stack([e["ce"] for e in rec.evals]), [90, 90, 120])
            if gap >= MIN_GAP:
                break
            print(f"  largereps {loss}: seed {seed} has a loss comparison within {gap:.1e} of a tie, next seed")
        else:
            raise SystemExit("no seed with a safe margin")
        save(f"g13_ctrl_largereps_{loss}_300", pack(rec, dict(
            x=x, y=y, w=w, eps=np.float64(4.0 / 255), n_iter=np.int64(300), seed=np.int64(4321),
            net_seed=np.int64(C + 40 + seed), min_gap=np.float64(gap), x_adv=xa, acc=acc)))


def near_tie(loss="mask-ce-bal", n_iter=90, max_gap=1e-5, seeds=400):
    """g13_ctrl_neartie_*: the OPPOSITE selection of main(): the first seed whose reference run makes a decisive loss
    comparison within `max_gap` (relative) of a tie -- closer than the device kernels' own loss error budget of 2e-5 that the
    other fixtures keep clear of.  The GPU test asserts what can be asserted there: the product reproduces the reference's
    iterates through the near-tie, or leaves them at an evaluation AFTER it and nowhere before."""
    os.chdir(REF)
    torch.set_num_threads(2)
    import semseg.attacker as A
    from autoattack.other_utils import Logger
    from semseg.utils.utils import VOC_WTS
    from oracle.tiny_models import PointwiseNet, make_labels
    logger, C, w = Logger(None), 21, torch.tensor(VOC_WTS)
    for seed in range(seeds):
        net = PointwiseNet(C, seed=C + 100 + seed)
        g = torch.Generator().manual_seed(17000 + 31 * seed + n_iter)
        x = torch.rand(3, 3, 16, 16, generator=g)
        y = make_labels(net, x, ignore_frac=0.05, flip_frac=0.1, seed=C + seed)
        eps = 8.0 / 255
        x_init = (x + eps * (2 * torch.rand(x.shape, generator=g) - 1)).clamp(0, 1)
        mask_bg = (y != -1).float()

        def stats(logits):
            with torch.no_grad():
                li = A.pixel_to_img_loss(A.criterion_dict[loss](logits, y, w), mask_bg)
                ce = A.pixel_to_img_loss(A.criterion_dict["ce-avg"](logits, y), mask_bg)
                n_correct = (logits.max(1)[1] == y).flatten(1).sum(1)
            return dict(li=li.clone(), ce=ce.clone(), n_correct=n_correct)

        rec = Recorder(net, stats).eval()
        with contextlib.redirect_stdout(io.StringIO()):
            xb, acc, lb, xba = A.apgd_train(rec, x, y, "Linf", eps, n_iter=n_iter, use_rs=False, loss=loss, track_loss="ce-avg",
                                            logger=logger, x_init=x_init, num_classes=C, weights=w, early_stop=True)
        if len(rec.evals) != n_iter + 1:
            continue
        gap = smallest_gaps(torch.stack([e["ce"] for e in rec.evals]), [n_iter])
        if gap < max_gap:
            break
    else:
        raise SystemExit(f"no seed with a comparison within {max_gap:g} of a tie in {seeds} tries")
    out = dict(x=x, y=y, w=w, x_init=x_init, eps=np.float64(eps), n_iter=np.int64(n_iter), net_seed=np.int64(C + 100 + seed),
               min_gap=np.float64(gap), x_best=xb, acc=acc, loss_best=lb, x_best_adv=xba, n_evals=np.int64(len(rec.evals)))
    for k in ("chk", "li", "ce", "n_correct"):
        out[k] = torch.stack([e[k] for e in rec.evals])
    out["has_grad"] = np.array([e["g"] is not None for e in rec.evals])
    for j, e in enumerate(rec.evals):
        if e["g"] is not None:
            gg = e["g"].flatten()
            out[f"e{j}_neg"] = np.packbits((gg < 0).numpy())
            out[f"e{j}_zero"] = np.packbits((gg == 0).numpy())
    name = f"g13_ctrl_neartie_{loss}_{n_iter}"
    np.savez_compressed(os.path.join(OUT, name + ".npz"),
                        **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()})
    print("wrote", name, f"seed {seed}: smallest relative gap of a decisive loss comparison {gap:.2e}", flush=True)


def real_model(n_iter=40, loss="mask-ce-bal", threads=2):
    """g13_ctrl_real_upernet_t_*: the reference's apgd_largereps on the REAL UperNet-ConvNeXt-T at 512 x 512 with n_iter = 40
    (stages 12 / 12 / 16: long enough for the product's HIP-graph replay in every stage), eps 4/255.  Stored: checksums of
    all 43 iterates, sign(g) planes, per-evaluation losses / counts / near-tie counts, the returned image's checksum."""
    from gen_goldens import _build_state_dict, _reference_model
    torch.set_num_threads(threads)
    C = 21
    sd = _build_state_dict("upernet", "ConvNeXt-T_CVST", C)
    ref = _reference_model("upernet", "ConvNeXt-T_CVST", C, sd)
    os.chdir(REF)
    import semseg.attacker as A
    from semseg.utils.utils import VOC_WTS
    w = torch.tensor(VOC_WTS)
    x = torch.rand(2, 3, 512, 512, generator=torch.Generator().manual_seed(1234))[:1].clone()
    with torch.no_grad():
        y = ref(x).max(1)[1]
    mask_bg = (y != -1).float()

    def stats(logits):
        with torch.no_grad():
            li = A.pixel_to_img_loss(A.criterion_dict[loss](logits, y, w), mask_bg)
            ce = A.pixel_to_img_loss(A.criterion_dict["ce-avg"](logits, y), mask_bg)
            top2 = logits.topk(2, dim=1)[0]
            near = ((top2[:, 0] - top2[:, 1]) < 2e-4 * logits.abs().max()).flatten(1).sum(1)
            n_correct = (logits.max(1)[1] == y).flatten(1).sum(1)
        return dict(li=li.clone(), ce=ce.clone(), n_correct=n_correct, n_near=near)

    rec = Recorder(ref, stats).eval()
    torch.manual_seed(4321)
    import time
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        xa, _, acc = A.apgd_largereps(rec, x.clone(), y, w, norm="Linf", eps=4.0 / 255, n_iter=n_iter, n_restarts=1, use_rs=True,
                                      loss=loss, verbose=False, track_loss="ce-avg", log_path=None, num_classes=C, early_stop=True)
    n_iters = [int(0.3 * n_iter), int(0.3 * n_iter)]
    n_iters.append(n_iter - sum(n_iters))
    assert len(rec.evals) == n_iter + 3, len(rec.evals)
    gap = smallest_gaps(torch.stack([e["ce"] for e in rec.evals]), n_iters)
    out = dict(y=y.to(torch.uint8), eps=np.float64(4.0 / 255), n_iter=np.int64(n_iter), seed=np.int64(4321),
               n_evals=np.int64(len(rec.evals)), min_gap=np.float64(gap), acc=acc, x_adv_chk=checksum(xa),
               seconds=np.float64(time.time() - t0))
    out["chk"] = torch.stack([e["chk"] for e in rec.evals])
    for k in ("li", "ce", "n_correct", "n_near"):
        out[k] = torch.stack([e[k] for e in rec.evals])
    out["has_grad"] = np.array([e["g"] is not None for e in rec.evals])
    for j, e in enumerate(rec.evals):
        if e["g"] is not None:
            g = e["g"].flatten()
            out[f"e{j}_neg"] = np.packbits((g < 0).numpy())
            out[f"e{j}_zero"] = np.packbits((g == 0).numpy())
    name = f"g13_ctrl_real_upernet_t_{loss}_{n_iter}"
    np.savez_compressed(os.path.join(OUT, name + ".npz"),
                        **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()})
    print("wrote", name, os.path.getsize(os.path.join(OUT, name + ".npz")) // 1024, "KiB; smallest relative gap of a loss "
          f"comparison {gap:.2e}; acc {acc.tolist()}; {time.time() - t0:.0f}s", flush=True)


if __name__ == "__main__":
    if "--near-tie" in sys.argv:
        # python oracle/gen_controller_goldens.py --near-tie [loss [n_iter]]
        rest = [a for a in sys.argv[1:] if a != "--near-tie"]
        near_tie(loss=rest[0] if rest else "mask-ce-bal", n_iter=int(rest[1]) if len(rest) > 1 else 90)
    elif "--real" in sys.argv:
        # python oracle/gen_controller_goldens.py --real [loss [n_iter]]
        rest = [a for a in sys.argv[1:] if a != "--real"]
        real_model(loss=rest[0] if rest else "mask-ce-bal", n_iter=int(rest[1]) if len(rest) > 1 else 40)
    else:
        main()
