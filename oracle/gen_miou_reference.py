#!/usr/bin/env python3
"""Worst-case SEA tables of the REAL reference for the north_star's second metric (build container only).

    python oracle/gen_miou_reference.py --parts 0 1 2 3 [--eps255 8] [--n-iter 100] [--threads 4] [--tag _t3]

Runs nmndeep/Robust-Segmentation itself (semseg.attacker.apgd_largereps x the three SEA losses, then
tools.worse_only.evalSEA, reference tools/infer.py:332-408 and tools/worse_only.py:181-422) on CPU, on PARTS of 64
synthetic 128x128 images each, with the build's seeded UperNet-ConvNeXt-T weights loaded strict=True into the
reference's model and the classifier bias fitted so that the clean prediction populates all 21 classes
(tools/synth.py).  Written per part to tests/golden/miou_ref/:

    model.npz                 the fitted classifier bias (once; every part uses the same model)
    part_XX_epsE.npz          labels (uint8), per-attack per-image intersection / union tables and correct-pixel
                              counts exactly as evalSEA computed them, the reference's own worst_Acc / final_miou of
                              the part, and the wall time

Only data is written.  The GPU test (tests/test_miou_claim_gpu.py) runs the device path on the same images, labels and
random starts and compares paired per-image statistics; nothing here runs on the GPU box.

Random starts: the reference draws `torch.rand_like(x)` from the global CPU generator once per stage.  Here that call
is served from one CPU stream per (image, attack) - `start_noise()` below, restated in tests/miou_ref.py - so that
the device run starts every stage from the same point and a part can be generated on its own.
"""
import argparse
import contextlib
import io
import os
import random
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("SEA_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "shims"), REF, ROOT]

import numpy as np  # noqa: E402
import torch  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "miou_ref")
PART, SIZE, C, BATCH = 64, 128, 21, 16
LOSSES = ("mask-ce-bal", "mask-ce-avg", "js-avg")


def part_images(part):
    return torch.rand(PART, 3, SIZE, SIZE, generator=torch.Generator().manual_seed(1234 + part))


def start_noise(image, attack, stage):
    """the uniform draw that replaces rand_like(x[j]) for global image index `image`, attack a, stage s"""
    g = torch.Generator().manual_seed(225 * 1000003 + image * 64 + attack * 8 + stage)
    return torch.rand(3, SIZE, SIZE, generator=g)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--parts", type=int, nargs="+", default=[0])
    ap.add_argument("--eps255", type=float, default=8.0)
    ap.add_argument("--n-iter", type=int, default=100)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--limit", type=int, default=PART, help="images of the part to run (timing probes)")
    ap.add_argument("--no-mkldnn", action="store_true",
                    help="run the reference with oneDNN switched off (torch.backends.mkldnn.enabled = False): every convolution "
                         "then goes through PyTorch's native im2col + BLAS path, i.e. ANOTHER valid fp32 arithmetic in every "
                         "layer (logits differ by ~8e-7 relative) -- the re-run that is comparable to what a different "
                         "implementation such as the device path changes; use with --tag _nomkldnn")
    ap.add_argument("--tag", default="", help="file-name suffix of a RE-RUN of an existing part with another thread count "
                    "(reference-vs-reference noise floor, tests/test_miou_claim_gpu.py), e.g. --threads 3 --tag _t3")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(args.threads)
    if args.no_mkldnn:
        torch.backends.mkldnn.enabled = False

    from gen_goldens import _build_state_dict, _reference_model
    sd = _build_state_dict("upernet", "ConvNeXt-T_CVST", C)

    # ---- the fitted classifier bias (part 0's images, the build's model on CPU; stored, never refitted)
    mpath = os.path.join(OUT, "model.npz")
    if not os.path.exists(mpath):
        pkg = os.path.join(ROOT, "robust-segmentation_amd")
        sys.path.insert(0, pkg)
        cwd = os.getcwd()
        os.chdir(pkg)
        try:
            from semseg.models.convnext_upernet import UperNetForSemanticSegmentation as Mine
            from tools.synth import balance_classes
            mine = Mine("ConvNeXt-T_CVST", C, None).eval()
            mine.load_state_dict(sd)
            share = balance_classes(mine, part_images(0), batch=16)
            bias = mine.decode_head.classifier.bias.detach().clone()
        finally:
            os.chdir(cwd)
            sys.path.remove(pkg)
            for k in [k for k in sys.modules if k in ("semseg", "tools") or k.startswith(("semseg.", "tools."))]:
                del sys.modules[k]
        np.savez_compressed(mpath, bias=bias.numpy(), share=share.numpy())
        print("fitted bias; class shares min/max", float(share.min()), float(share.max()), flush=True)
    bias = torch.from_numpy(np.load(mpath)["bias"])
    sd = dict(sd)
    sd["decode_head.classifier.bias"] = bias
    ref = _reference_model("upernet", "ConvNeXt-T_CVST", C, sd)

    os.chdir(REF)
    import semseg.attacker as A
    import tools.worse_only as W
    from semseg.utils.utils import VOC_WTS
    w = torch.tensor(VOC_WTS)
    eps = args.eps255 / 255.0

    class _DS(torch.utils.data.Dataset):
        def __init__(self, targets):
            self.t = targets

        def __len__(self):
            return self.t.shape[0]

        def __getitem__(self, i):
            return torch.zeros(1), self.t[i], str(i)

    real_rand_like = torch.rand_like
    for part in args.parts:
        images = part_images(part)[:args.limit]
        n = images.shape[0]
        with torch.no_grad():
            labels = torch.cat([ref(images[i:i + BATCH]).max(1)[1] for i in range(0, n, BATCH)])
        t0 = time.time()
        preds = []
        for a, loss in enumerate(LOSSES):
            maps = []
            for i in range(0, n, BATCH):
                idx = list(range(i, min(i + BATCH, n)))
                stage = [0]

                def served(t, *aa, _idx=idx, _a=a, **kk):
                    out = torch.stack([start_noise(part * PART + j, _a, stage[0]) for j in _idx])
                    stage[0] += 1
                    assert out.shape == t.shape
                    return out

                torch.rand_like = served
                try:
                    with contextlib.redirect_stdout(io.StringIO()):
                        xa, _, acc = A.apgd_largereps(ref, images[idx].clone(), labels[idx], w, norm="Linf", eps=eps,
                                                      n_iter=args.n_iter, n_restarts=1, use_rs=True, loss=loss,
                                                      verbose=False, track_loss="ce-avg", log_path=None,
                                                      num_classes=C, early_stop=True)
                finally:
                    torch.rand_like = real_rand_like
                assert stage[0] == 3
                with torch.no_grad():
                    pm = ref(xa).max(1)[1]          # tools/infer.py:354-363: eval_performance on the adversarial set
                pm[labels[idx] == -1] = -1
                maps.append(pm)
                print(f"part {part} {loss} images {i}..{idx[-1]} acc {acc.mean():.4f} "
                      f"{time.time() - t0:.0f}s", flush=True)
            preds.append(torch.cat(maps))
        dt = time.time() - t0
        with tempfile.TemporaryDirectory() as td:
            os.makedirs(os.path.join(td, "test_results"))
            sdict = {"seed": 225, "worst_Acc": 0, "worst_Acc_indiv": 0, "final_miou": 0, "loss-wise_miou": []}
            ev = W.evalSEA(_DS(labels), [p.clone() for p in preds], args.eps255, C, "SEA_ref", td, sdict, "m")
            with contextlib.redirect_stdout(io.StringIO()):
                ev.worse_case_eval(bs=BATCH, n_batches=-1)
                random.seed(225)
                ev.worst_case_miou()
            st = torch.load(os.path.join(td, "test_results", f"stats_SEA_ref_{args.eps255}.pt"))
        P = torch.stack(preds)
        correct = (P == labels[None]).flatten(2).sum(-1)                       # (A, n) correct pixels per attack
        name = (f"part_{part:02d}_eps{int(args.eps255)}" + ("" if n == PART else f"_n{n}")
                + ("" if args.n_iter == 100 else f"_it{args.n_iter}") + args.tag)
        np.savez_compressed(
            os.path.join(OUT, name + ".npz"), labels=labels.to(torch.uint8).numpy(), n_iter=np.int64(args.n_iter),
            eps255=np.float64(args.eps255), part=np.int64(part), ints=st["run_int_imwise"].to(torch.int32).numpy(),
            unions=st["run_union_imwise"].to(torch.int32).numpy(), correct=correct.to(torch.int32).numpy(),
            worst_Acc=np.float64(ev.saveDict["worst_Acc"]), final_miou=np.float64(ev.saveDict["final_miou"]),
            worst_Acc_indiv=ev.saveDict["worst_Acc_indiv"].numpy(), seconds=np.float64(dt),
            threads=np.int64(args.threads), mkldnn=np.int64(0 if args.no_mkldnn else 1))
        print(f"wrote {name}: worst aAcc {100 * ev.saveDict['worst_Acc']:.4f} %  worst mIoU "
              f"{100 * ev.saveDict['final_miou']:.4f} %  {dt:.0f}s", flush=True)


if __name__ == "__main__":
    main()
