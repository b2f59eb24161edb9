#!/usr/bin/env python3
"""SEA evaluation driver (counterpart of tools/infer.py), MI355X-native and shardable over GPUs.

    python -m tools.infer --cfg configs/pascalvoc_convnext.yaml --eps 4                    # 1 GPU
    torchrun --nproc-per-node 8 -m tools.infer --cfg configs/ade20k_convnext.yaml --eps 4  # 8 GPUs

Same CLI flags and YAML keys as the reference (`--cfg --eps --n_iter --adversarial --attack
--n_batches --cleanup`), same helper functions (`eval_performance`, `evaluate`, `check_imgs`, `MaskClass`,
`mask_logits`) with the reference's signatures and return contracts, same output files
(`sea-stats/loss_wise_*.txt`, `worse_SEA_*.pt` with keys seed / worst_Acc / worst_Acc_indiv / final_miou /
loss-wise_miou, `argmax-logs/{model}_{loss}_{eps}.pt` = ONE (N,H,W) int64 tensor per loss, readable by
`tools.worse_only.evalSEA`).  What is different from the reference's data path (tools/infer.py:136-155, 356-370):

  * the data set lives on the HOST (pinned); one batch at a time is uploaded, adversarial images never come
    back to the host and are not re-forwarded: the argmax of the returned iterate is produced by the attack's
    own fused kernel (`apgd_largereps(..., return_pred=True)`);
  * all statistics are integer tables accumulated on the device (K3), packed in one buffer and all-reduced
    ONCE at the end (tools/sea_shard.py); rank r owns the images r, r+world, ...;
  * datasets are out of scope of this build: `--synthetic N` evaluates N seeded random images whose
    labels are the model's clean prediction, `--data file.pt` takes a dict
    {images (N,3,H,W) float in [0,1], labels (N,H,W) int64}.

Like the reference, the three standard SEA losses always run; `--attack X` (which the reference parses and then
overwrites, tools/infer.py:332-336) restricts the run to that one loss here and every table is sized for it.
"""
from __future__ import annotations

import argparse
import json
import os
import random
import sys
import time
from collections import OrderedDict
from functools import partial

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn
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


def check_imgs(adv, x, norm, verbose=False):
    """Perturbation-size / NaN / range report string (tools/infer.py:39-53)."""
    delta = (adv - x).reshape(adv.shape[0], -1)
    if norm == "Linf":
        res = delta.abs().max(dim=1)[0]
    elif norm == "L2":
        res = (delta ** 2).sum(dim=1).sqrt()
    elif norm == "L1":
        res = delta.abs().sum(dim=1)
    else:
        raise ValueError(norm)
    s = (f"max {norm} pert: {res.max():.5f}, nan in imgs: {(adv != adv).sum()}, max in imgs: {adv.max():.5f}, "
         f"min in imgs: {adv.min():.5f}")
    if verbose:
        print(s)
    return s


def stats_from_counts(inter, pred_cnt, tgt_cnt):
    """{mAcc, aAcc, mIoU} (tools/infer.py:93-118, 131) from int64 per-class counts."""
    inter, pc, tc = inter.float().cpu(), pred_cnt.float().cpu(), tgt_cnt.float().cpu()
    union = tc + pc - inter
    ind = tc > 0
    m_acc = (inter[ind] / tc[ind]).mean()
    a_acc = inter.sum() / tc.sum()
    ind = union > 0
    return {"mAcc": m_acc.item(), "aAcc": a_acc.item(), "mIoU": (inter[ind] / union[ind]).mean().item()}


def _pred_dtype(n_cls):
    return torch.uint8 if n_cls <= 255 else torch.int16


@torch.no_grad()
def predict(model, x, n_cls):
    """argmax map (uint8/int16) of a batch through the fused kernel, no gradient."""
    with attacker._FrozenParameters(model):
        logits = model(x)
    pred = torch.empty(x.shape[0], x.shape[2], x.shape[3], dtype=_pred_dtype(n_cls), device=x.device)
    dummy = torch.zeros(x.shape[0], x.shape[2], x.shape[3], dtype=torch.uint8, device=x.device)
    N.loss_fwd_bwd(logits, dummy, None, 3, 3, 0.0, want_grad=False, pred=pred)
    return pred


def _masked_long(pred, target, ignore_index=-1):
    """int64 copy of an argmax map with the ignore label written at ignored pixels (tools/infer.py:88-90)."""
    out = pred.long()
    out[target == ignore_index] = ignore_index
    return out


def eval_performance(model, data_loader, n_batches=-1, n_cls=21, return_output=False, ignore_index=-1,
                     return_preds=False, verbose=False, device=None, counts_out=None):
    """Accuracy / mIoU pass with the reference's contract (tools/infer.py:56-133): ``data_loader`` yields
    ``(input, target, ...)``; returns ``({"mAcc","aAcc","mIoU"}, l_output)`` with ``l_output`` the concatenated
    (N,H,W) int64 argmax maps on the host, ignore label written at ignored pixels.

    One forward + one fused argmax kernel + one histogram kernel (K3) per batch instead of 2*n_cls Python
    iterations on the host.  Extensions: a batch may carry its already known prediction as a third TENSOR element
    ``(input, target, pred)`` -- the forward is then skipped (the attack produced the argmax of its own iterate);
    ``counts_out`` (3 int64 (n_cls,) device tensors) receives the accumulated inter / pred / target counts."""
    model.eval()
    if device is None:
        device = next(model.parameters()).device
    tot = counts_out if counts_out is not None else tuple(
        torch.zeros(n_cls, dtype=torch.int64, device=device) for _ in range(3))
    l_output = []
    for i, vals in enumerate(data_loader):
        inp, target = vals[0], vals[1]
        target = target.to(device, non_blocking=True).contiguous()
        if len(vals) > 2 and torch.is_tensor(vals[2]) and vals[2].shape == target.shape:
            pred = vals[2].to(device)
        else:
            pred = predict(model, inp.to(device, non_blocking=True).float(), n_cls)
        N.class_counts(pred.contiguous(), target, n_cls, per_image=False, mask_pred=True, out=tot)
        l_output.append(_masked_long(pred, target, ignore_index).cpu())
        if verbose:
            s = stats_from_counts(*tot)
            print(f"batch={i} running mAcc={s['mAcc']:.2%} running aAcc={s['aAcc']:.2%}", f" running mIoU={s['mIoU']:.2%}")
        if i + 1 == n_batches:
            print("enough batches seen")
            break
    return stats_from_counts(*tot), torch.cat(l_output)


def evaluate(val_loader, model, attack_fn, n_batches=-1, args=None, weights=None, device=None):
    """Run the attack on every batch (tools/infer.py:136-155).  Returns the list of
    ``(x_adv, target, pred)``: unlike the reference the entries stay ON THE DEVICE and carry the argmax map of
    ``x_adv`` (from the attack's own kernel), so ``eval_performance(model, adv_loader)`` does not forward again."""
    model.eval()
    if device is None:
        device = next(model.parameters()).device
    adv_loader = []
    norm = getattr(args, "norm", "Linf")
    for i, vals in enumerate(val_loader):
        inp = vals[0].to(device, non_blocking=True).float()
        target = vals[1].to(device, non_blocking=True)
        out = attack_fn(model, inp.clone(), target, weights, return_pred=True)
        x_adv, pred = out[0], out[3]
        check_imgs(inp, x_adv, norm=norm)
        adv_loader.append((x_adv, target, pred))
        if i + 1 == n_batches:
            break
    return adv_loader


class MaskClass(nn.Module):
    """Drops one class channel from the logits (tools/infer.py:195-216)."""

    def __init__(self, ignore_index: int) -> None:
        super().__init__()
        self.ignore_index = ignore_index

    def forward(self, input):
        if self.ignore_index == 0:
            return input[:, 1:]
        return torch.cat((input[:, :self.ignore_index], input[:, self.ignore_index + 1:]), dim=1)


def mask_logits(model: nn.Module, ignore_index: int) -> nn.Module:
    return nn.Sequential(OrderedDict([("model", model), ("mask", MaskClass(ignore_index))]))


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


def arithmetic_mode():
    """what the attacked model's frozen-weight GEMMs computed in (recorded in every summary): the reference evaluates in
    fp32; the default here is fp32-equivalent (fp16 x 2 operand split = 22 significant bits, forward and input gradient)"""
    from semseg.models import convnext_upernet as M
    names = {22: "fp16x2 (22 significant bits)", 3: "bf16x3 (24 bits)", 2: "bf16x2 (16 bits)", 0: "hipBLASLt fp32"}
    fwd = M.GEMM_TERMS
    from semseg import _native as N
    return {"gemm_forward": names.get(fwd, str(fwd)), "gemm_input_gradient": names.get(M._bwd_terms(fwd) if fwd else 0, "?"),
            "winograd_tile": M.WINOGRAD_TILE, "accumulate": "fp32", "storage": "fp32",
            # (Segmenter only) M7b attention: bf16 terms per operand, 22 = fp16 x 2, 0 = fp32 MFMA
            "attention_forward_terms": N.attn_terms_fwd(), "attention_input_gradient_terms": N.attn_terms_bwd()}


def _pin(t):
    try:
        return t.pin_memory()
    except RuntimeError:
        return t


def _main(argv=None):
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
    ap.add_argument("--random_init", action="store_true", help="seeded random weights instead of EVAL.MODEL_PATH")
    ap.add_argument("--balance_classes", action="store_true",
                    help="with --synthetic: fit the classifier bias so that the clean prediction (= the labels) populates "
                         "every class (tools/synth.py): a well-conditioned mIoU on random weights")
    ap.add_argument("--image_size", type=int, default=None)
    ap.add_argument("--batch_size", type=int, default=None)
    ap.add_argument("--save_argmax", action="store_true",
                    help="write argmax-logs/{model}_{loss}_{eps}.pt (the reference always does, infer.py:366-370)")
    ap.add_argument("--json", type=str, default=None)
    ap.add_argument("--backend", type=str, default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--dump_stats", type=str, default=None, help="save the all-reduced packed statistics buffer (rank 0)")
    ap.add_argument("--deterministic", action="store_true",
                    help="no MIOpen find-mode benchmarking (its timing-based algorithm choice differs between processes)")
    args = ap.parse_args(argv)

    with open(args.cfg) as f:
        cfg = yaml.load(f, Loader=yaml.SafeLoader)
    test_cfg = cfg["EVAL"]
    C = int(test_cfg["N_CLS"])
    bs = args.batch_size or int(test_cfg["BATCH_SIZE"])
    save_dir = cfg["SAVE_DIR"]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    own_group = False
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(args.backend, rank=rank, world_size=world)
        own_group = True
    device = torch.device("cuda", local)
    random.seed(SEED)
    np.random.seed(SEED)
    torch.manual_seed(0)  # same random-init weights on every rank
    torch.backends.cudnn.benchmark = not args.deterministic

    model = build_model(cfg, random_init=bool(args.synthetic) or args.random_init, device=device)
    for p in model.parameters():
        p.requires_grad_(False)
    modelName = getModelName(cfg["MODEL"]["NAME"], test_cfg["BACKBONE"])

    # ---- data: resident on the HOST (pinned); one batch at a time goes to HBM ---------------------------
    if args.synthetic:
        n_img = args.synthetic
        size = args.image_size or int(test_cfg["IMAGE_SIZE"][0])
        g = torch.Generator().manual_seed(1234)
        images = torch.rand(n_img, 3, size, size, generator=g)
        labels = None  # = clean prediction, filled by the clean pass
    else:
        blob = torch.load(args.data, map_location="cpu")
        images, labels = blob["images"].float(), blob["labels"].long()
        n_img = images.shape[0]
    if args.synthetic and args.balance_classes:
        from tools.synth import balance_classes
        frac = balance_classes(model, images)          # every rank fits the same bias on the same full image set
        if rank == 0:
            print(f"[infer] balanced classes: min / max pixel share {float(frac.min()):.4f} / {float(frac.max()):.4f}")
    if args.n_batches > 0:
        n_img = min(n_img, args.n_batches * bs * world)
    mine = shard_indices(n_img, rank, world)
    batches = [mine[i:i + bs] for i in range(0, len(mine), bs)]
    images = _pin(images[mine].contiguous())                      # this rank's shard only
    H, W = images.shape[-2:]
    lbl_dtype = torch.int16                                       # compact host copy (-1 = ignore survives)
    labels_h = _pin(torch.empty(len(mine), H, W, dtype=lbl_dtype)) if labels is None else _pin(labels[mine].to(lbl_dtype))
    pos = {gi: j for j, gi in enumerate(mine)}
    weights = torch.tensor(ADE_WTS if str(test_cfg["NAME"]).lower() == "ade20k" else VOC_WTS, device=device)
    if weights.numel() != C:
        weights = torch.ones(C, device=device)

    attacks = LOSSES if args.attack is None else [args.attack]
    stats = SeaStats(len(attacks), n_img, C, device)
    if rank == 0:
        make_attack_dirs(save_dir)

    def local_rows(idx):
        """a batch is a run of consecutive rows of this rank's shard: a SLICE, i.e. a view of the pinned host tensor
        (a tensor index would gather into a pageable temporary and make the H2D copy synchronous)"""
        j0 = pos[idx[0]]
        assert pos[idx[-1]] == j0 + len(idx) - 1
        return slice(j0, j0 + len(idx))

    def batch_to_device(idx):
        rows = local_rows(idx)
        return images[rows].to(device, non_blocking=True), rows

    def start_noise(idx, a):
        """Uniform draws of the three random starts (one per apgd_largereps stage), one stream PER IMAGE keyed by
        (seed, global image index, attack): the evaluation does not depend on how images are batched or sharded
        (the reference consumes one global stream batch by batch, tools/infer.py:25-30)."""
        out = [torch.empty(len(idx), 3, H, W, device=device) for _ in range(3)]
        gen = torch.Generator(device=device)
        for j, gi in enumerate(idx):
            gen.manual_seed(SEED * 1000003 + gi * 7 + a)
            for st in range(3):
                out[st][j] = torch.rand(3, H, W, generator=gen, device=device)
        return out

    # ---- clean pass (tools/infer.py:314-322) ------------------------------------------------------
    clean_tot = tuple(stats.clean[k] for k in range(3))
    for idx in batches:
        x, rows = batch_to_device(idx)
        pred = predict(model, x, C)
        if labels is None:
            labels_h[rows] = pred.to(lbl_dtype).cpu()
        y = labels_h[rows].to(device, non_blocking=True).long()
        N.class_counts(pred, y.contiguous(), C, per_image=False, mask_pred=True, out=clean_tot)
    if not args.adversarial:
        return

    # ---- the attacks (tools/infer.py:332-379): x_adv and its argmax never leave the device ----------------
    logs = {l: torch.empty(len(mine), H, W, dtype=lbl_dtype) for l in attacks} if args.save_argmax else {}
    t_attack = time.time()
    for a, loss_ in enumerate(attacks):
        attack_fn = partial(attacker.apgd_largereps, norm="Linf", eps=args.eps / 255.0, n_iter=args.n_iter,
                            n_restarts=1, use_rs=True, loss=loss_, verbose=False, track_loss="ce-avg",
                            log_path=None, num_classes=C, early_stop=True)
        for idx in batches:
            x, rows = batch_to_device(idx)
            y = labels_h[rows].to(device, non_blocking=True).long().contiguous()
            x_adv, _, acc, pred = attack_fn(model, x, y, weights, return_pred=True, noises=start_noise(idx, a))
            # predictions masked at ignored pixels, like the logs eval_performance hands to evalSEA (infer.py:88-90)
            im, pm, tc = N.class_counts(pred, y, C, per_image=True, mask_pred=True)
            stats.add_attack_batch(a, idx, im, pm, tc)
            if args.save_argmax:
                logs[loss_][rows] = _masked_long(pred, y).to(lbl_dtype).cpu()
    torch.cuda.synchronize()
    t_attack = time.time() - t_attack
    # the evaluation's ONE captured graph pair (every stage, loss and batch replayed it) and its activation pool
    attacker.release_graph_cache(model)

    # ---- the ONE collective, then host-side worst-case bookkeeping on rank 0 ----------------------------
    stats.all_reduce()
    if args.dump_stats and rank == 0:
        torch.save(stats.buf.cpu(), args.dump_stats)
    summary = None
    if rank == 0:
        s = stats.cpu()
        clean_stats = stats_from_counts(s.clean[0], s.clean[1], s.clean[2])
        indiv_mious = []
        for a, loss_ in enumerate(attacks):
            adv_stats = stats_from_counts(s.attack_totals[a, 0], s.attack_totals[a, 1], s.attack_totals[a, 2])
            indiv_mious.append(adv_stats["mIoU"])
            writeIndivloss(save_dir, modelName, clean_stats, args.eps, loss_, adv_stats)
        worst, indiv, _ = worst_acc_from_counts(s.correct, s.valid)
        random.seed(SEED)
        miou, sel, rounds = worst_miou_from_tables(s.inter, s.union)
        save_dict = {"seed": SEED, "worst_Acc": worst, "worst_Acc_indiv": indiv, "final_miou": miou,
                     "loss-wise_miou": indiv_mious}
        addendum = "SEA_" + modelName
        torch.save(save_dict, os.path.join(save_dir, f"worse_{addendum}_{test_cfg['NAME']}_{args.eps}.pt"))
        summary = {"model": modelName, "n_images": n_img, "world": world, "eps": args.eps, "attacks": attacks,
                   "clean": clean_stats, "worst_Acc": worst, "worst_Acc_indiv": indiv.tolist(), "final_miou": miou,
                   "loss-wise_miou": indiv_mious, "attack_seconds": t_attack, "arithmetic": arithmetic_mode(),
                   "image_iterations_per_s": n_img * len(attacks) * args.n_iter / t_attack}
        print(json.dumps(summary))
        if args.json:
            json.dump(summary, open(args.json, "w"))

    # ---- argmax logs in the reference's on-disk format: one (N,H,W) int64 tensor per loss -----------------
    if args.save_argmax:
        log_dir = os.path.join(save_dir, "argmax-logs")
        os.makedirs(log_dir, exist_ok=True)
        if world > 1:  # every rank drops its shard, rank 0 merges after the barrier (no collective on the maps)
            for loss_ in attacks:
                torch.save({"index": mine, "argmax": logs[loss_]},
                           os.path.join(log_dir, f".{modelName}_{loss_}_{args.eps}.shard{rank}"))
            dist.barrier()
        if rank == 0:
            for loss_ in attacks:
                full = torch.empty(n_img, H, W, dtype=torch.int64)
                if world > 1:
                    for r in range(world):
                        sp = os.path.join(log_dir, f".{modelName}_{loss_}_{args.eps}.shard{r}")
                        sh = torch.load(sp)
                        full[torch.tensor(sh["index"], dtype=torch.long)] = sh["argmax"].long()
                        os.remove(sp)
                else:
                    full[torch.tensor(mine, dtype=torch.long)] = logs[loss_].long()
                torch.save(full, os.path.join(log_dir, f"{modelName}_{loss_}_{args.eps}.pt"))
    if world > 1:
        dist.barrier()
        if own_group:
            dist.destroy_process_group()
    if rank == 0 and bool(args.cleanup):
        remove_dirs(save_dir)
    return summary


def main(argv=None):
    """`_main` with the process-global switch it sets (MIOpen find mode) restored on exit: a caller that runs this
    in-process (tests do) must not inherit `cudnn.benchmark = True`."""
    prev = torch.backends.cudnn.benchmark
    try:
        return _main(argv)
    finally:
        torch.backends.cudnn.benchmark = prev


if __name__ == "__main__":
    main()
