#!/usr/bin/env python3
"""SEA evaluation driver (counterpart of tools/infer.py), MI355X-native and shardable over GPUs.

    python -m tools.infer --cfg configs/pascalvoc_convnext.yaml --eps 4                    # 1 GPU
    torchrun --nproc-per-node 8 -m tools.infer --cfg configs/ade20k_convnext.yaml --eps 4  # 8 GPUs

Same CLI flags and YAML keys as the reference (`--cfg --eps --n_iter --adversarial --attack
--n_batches --cleanup`), same output files (`sea-stats/loss_wise_*.txt`, `worse_SEA_*.pt` with keys
seed / worst_Acc / worst_Acc_indiv / final_miou / loss-wise_miou, optional `argmax-logs/*.pt`).
What is different from the reference's data path (tools/infer.py:136-155, 356-370):
  * adversarial images never leave the GPU and are not re-forwarded: the argmax of the returned iterate
    is produced by the attack's own fused kernel;
  * all statistics are integer tables accumulated on the device (K3), packed in one buffer and
    all-reduced ONCE at the end (tools/sea_shard.py);
  * datasets are out of scope of this build: `--synthetic N` evaluates N seeded random images whose
    labels are the model's clean prediction, `--data file.pt` takes a dict
    {images (N,3,H,W) float in [0,1], labels (N,H,W) int64}.
"""
from __future__ import annotations

import argparse
import json
import os
import random
import sys
import time
from functools import partial

import numpy as np
import torch
import torch.distributed as dist
import yaml

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

import semseg.attacker as attacker  # noqa: E402
from semseg import _native as N  # noqa: E402
from semseg.models import UperNetForSemanticSegmentation  # noqa: E402
from semseg.utils.utils import ADE_WTS, VOC_WTS, getModelName, make_attack_dirs, remove_dirs, writeIndivloss  # noqa: E402
from tools.sea_shard import SeaStats, shard_indices  # noqa: E402
from tools.worse_only import worst_acc_from_counts, worst_miou_from_tables  # noqa: E402

SEED = 225
LOSSES = ["mask-ce-bal", "mask-ce-avg", "js-avg"]  # standard SEA attacks, in the reference's order


def stats_from_counts(inter, pred_cnt, tgt_cnt):
    """{mAcc, aAcc, mIoU} (tools/infer.py:93-118, 131) from int64 per-class counts."""
    inter, pc, tc = inter.float().cpu(), pred_cnt.float().cpu(), tgt_cnt.float().cpu()
    union = tc + pc - inter
    ind = tc > 0
    m_acc = (inter[ind] / tc[ind]).mean()
    a_acc = inter.sum() / tc.sum()
    ind = union > 0
    return {"mAcc": m_acc.item(), "aAcc": a_acc.item(), "mIoU": (inter[ind] / union[ind]).mean().item()}


def build_model(cfg, random_init: bool, device):
    model_cfg, test_cfg = cfg["MODEL"], cfg["EVAL"]
    if model_cfg["NAME"] == "UperNetForSemanticSegmentation":
        model = UperNetForSemanticSegmentation(test_cfg["BACKBONE"], test_cfg["N_CLS"], None)
    elif model_cfg["NAME"] == "SegMenter":
        from semseg.models import create_segmenter
        from semseg.utils.utils import load_config_segmenter
        mcfg, _ = load_config_segmenter(backbone=model_cfg["BACKBONE"], n_cls=test_cfg["N_CLS"])
        model = create_segmenter(mcfg, None, test_cfg["BACKBONE"])
    else:
        raise ValueError(f"model family {model_cfg['NAME']} is outside this build (SURVEY 2.1)")
    if not random_init:
        model.load_state_dict(torch.load(test_cfg["MODEL_PATH"], map_location="cpu"))
    return model.to(device).eval()


@torch.no_grad()
def predict(model, x, n_cls):
    """argmax map (uint8/int16) of a batch through the fused kernel, no gradient."""
    logits = model(x)
    pred = torch.empty(x.shape[0], x.shape[2], x.shape[3], dtype=torch.uint8 if n_cls <= 255 else torch.int16,
                       device=x.device)
    dummy = torch.zeros(x.shape[0], x.shape[2], x.shape[3], dtype=torch.uint8, device=x.device)
    N.loss_fwd_bwd(logits, dummy, None, 3, 3, 0.0, want_grad=False, pred=pred)
    return pred


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=str, default="configs/pascalvoc_convnext.yaml")
    ap.add_argument("--eps", type=float, default=8.0)
    ap.add_argument("--n_iter", type=int, default=300)
    ap.add_argument("--adversarial", action="store_true", default=True)
    ap.add_argument("--attack", type=str, default=None)
    ap.add_argument("--n_batches", type=int, default=-1)
    ap.add_argument("--cleanup", type=int, default=1)
    # additions of this build
    ap.add_argument("--synthetic", type=int, default=0, help="evaluate N synthetic images (no dataset, random weights)")
    ap.add_argument("--data", type=str, default=None, help=".pt file with {'images','labels'}")
    ap.add_argument("--image_size", type=int, default=None)
    ap.add_argument("--batch_size", type=int, default=None)
    ap.add_argument("--save_argmax", action="store_true")
    ap.add_argument("--json", type=str, default=None)
    args = ap.parse_args(argv)

    with open(args.cfg) as f:
        cfg = yaml.load(f, Loader=yaml.SafeLoader)
    test_cfg = cfg["EVAL"]
    C = int(test_cfg["N_CLS"])
    bs = args.batch_size or int(test_cfg["BATCH_SIZE"])

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    device = torch.device("cuda", local)
    random.seed(SEED)
    np.random.seed(SEED)
    torch.manual_seed(0)  # same random-init weights on every rank
    torch.backends.cudnn.benchmark = True

    model = build_model(cfg, random_init=bool(args.synthetic), device=device)
    for p in model.parameters():
        p.requires_grad_(False)
    modelName = getModelName(cfg["MODEL"]["NAME"], test_cfg["BACKBONE"])

    # ---- data (resident on the host; one batch at a time goes to HBM) ---------------------------------
    if args.synthetic:
        n_img = args.synthetic
        size = args.image_size or int(test_cfg["IMAGE_SIZE"][0])
        g = torch.Generator().manual_seed(1234)
        images = torch.rand(n_img, 3, size, size, generator=g)
        labels = None  # = clean prediction, filled below
    else:
        blob = torch.load(args.data, map_location="cpu")
        images, labels = blob["images"].float(), blob["labels"].long()
        n_img = images.shape[0]
    if args.n_batches > 0:
        n_img = min(n_img, args.n_batches * bs * world)
    mine = shard_indices(n_img, rank, world)
    batches = [mine[i:i + bs] for i in range(0, len(mine), bs)]
    weights = torch.tensor(ADE_WTS if str(test_cfg["NAME"]).lower() == "ade20k" else VOC_WTS, device=device)
    if weights.numel() != C:
        weights = torch.ones(C, device=device)

    stats = SeaStats(len(LOSSES), n_img, C, device)
    if rank == 0:
        make_attack_dirs(cfg["SAVE_DIR"])
    logs = {l: {} for l in LOSSES}

    # ---- clean pass (tools/infer.py:314-322) ------------------------------------------------------
    dev_batches = []
    for idx in batches:
        x = images[idx].to(device, non_blocking=True)
        pred = predict(model, x, C)
        y = pred.long() if labels is None else labels[idx].to(device)
        stats.add_clean(*N.class_counts(pred, y.contiguous(), C, per_image=False, mask_pred=True))
        dev_batches.append((idx, x, y))
    if not args.adversarial:
        return

    # ---- the three attacks (tools/infer.py:332-379) ----------------------------------------------------
    t_attack = time.time()
    for a, loss_ in enumerate(LOSSES if args.attack is None else [args.attack]):
        attack_fn = partial(attacker.apgd_largereps, norm="Linf", eps=args.eps / 255.0, n_iter=args.n_iter,
                            n_restarts=1, use_rs=True, loss=loss_, verbose=False, track_loss="ce-avg",
                            log_path=None, num_classes=C, early_stop=True)
        for idx, x, y in dev_batches:
            x_adv, _, acc = attack_fn(model, x.clone(), y, weights)
            pred = predict(model, x_adv, C)
            yc = y.contiguous()
            im, pm, tc = N.class_counts(pred, yc, C, per_image=True, mask_pred=True)
            ir, pr, _ = N.class_counts(pred, yc, C, per_image=True, mask_pred=False)
            stats.add_attack_batch(a, idx, im, pm, tc, ir, pr)
            if args.save_argmax:
                for j, gi in enumerate(idx):
                    logs[loss_][gi] = pred[j].cpu()
    torch.cuda.synchronize()
    t_attack = time.time() - t_attack

    # ---- the ONE collective, then host-side worst-case bookkeeping on rank 0 ----------------------------
    stats.all_reduce()
    if rank == 0:
        s = stats.cpu()
        clean_stats = stats_from_counts(s.clean[0], s.clean[1], s.clean[2])
        indiv_mious = []
        for a, loss_ in enumerate(LOSSES if args.attack is None else [args.attack]):
            adv_stats = stats_from_counts(s.attack_totals[a, 0], s.attack_totals[a, 1], s.attack_totals[a, 2])
            indiv_mious.append(adv_stats["mIoU"])
            writeIndivloss(cfg["SAVE_DIR"], modelName, clean_stats, args.eps, loss_, adv_stats)
        worst, indiv, _ = worst_acc_from_counts(s.correct, s.valid)
        random.seed(SEED)
        miou, sel, rounds = worst_miou_from_tables(s.inter, s.union)
        save_dict = {"seed": SEED, "worst_Acc": worst, "worst_Acc_indiv": indiv, "final_miou": miou,
                     "loss-wise_miou": indiv_mious}
        addendum = "SEA_" + modelName
        torch.save(save_dict, os.path.join(cfg["SAVE_DIR"], f"worse_{addendum}_{test_cfg['NAME']}_{args.eps}.pt"))
        summary = {"model": modelName, "n_images": n_img, "world": world, "eps": args.eps, "clean": clean_stats,
                   "worst_Acc": worst, "worst_Acc_indiv": indiv.tolist(), "final_miou": miou,
                   "loss-wise_miou": indiv_mious, "attack_seconds": t_attack,
                   "image_iterations_per_s": n_img * 3 * args.n_iter / t_attack}
        print(json.dumps(summary))
        if args.json:
            json.dump(summary, open(args.json, "w"))
    if args.save_argmax:
        for loss_ in logs:
            if logs[loss_]:
                keys = sorted(logs[loss_])
                torch.save({"index": keys, "argmax": torch.stack([logs[loss_][k] for k in keys]).long()},
                           os.path.join(cfg["SAVE_DIR"], "argmax-logs", f"{modelName}_{loss_}_{args.eps}_rank{rank}.pt"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and bool(args.cleanup):
        remove_dirs(cfg["SAVE_DIR"])


if __name__ == "__main__":
    main()
