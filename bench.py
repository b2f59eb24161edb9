#!/usr/bin/env python3
"""SEA attack-loop benchmark (BASELINE.json metric: attack image-iterations/s, UperNet-ConvNeXt-T, 512x512).

    python bench.py --gpus N --steps K --warmup W          # N=1 runs in-process
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One "step" = one APGD loop iteration over one batch of B=8 synthetic 512x512 images per GPU:
L-inf step kernel (K1) + model forward (PyTorch-ROCm) + fused loss/gradient kernel (K2) + model
input-gradient backward + device-side bookkeeping kernels (K4/K7).  Inputs are resident in HBM before
the timed region.  fp32 end to end, like the reference's evaluation.  Images shard across ranks with
no collective in the loop (weak scaling).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "robust-segmentation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def k2_algorithmic_bytes(B, C, HW, logit_bytes=4, with_grad=True):
    """B*H*W*(2*C*s + 8 + 8) with gradient, B*H*W*(C*s + 16) without."""
    return B * HW * ((2 if with_grad else 1) * C * logit_bytes + 16)


def make_model(backbone, C):
    """UperNet-ConvNeXt (BASELINE configs 1, 2, 4, 5) or Segmenter ViT (config 3: --backbone vit_small_patch16_224)."""
    if backbone.startswith("vit_"):
        from semseg.models import create_segmenter
        from semseg.utils.utils import load_config_segmenter
        cfg, _ = load_config_segmenter(backbone, C)
        return create_segmenter(cfg, None, backbone)
    from semseg.models import UperNetForSemanticSegmentation
    return UperNetForSemanticSegmentation(backbone, C, None)


def build_case(rank, B, C, backbone, device):
    torch.manual_seed(0)  # identical weights on every rank
    model = make_model(backbone, C).eval().to(device)
    for p in model.parameters():
        p.requires_grad_(False)
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.rand(B, 3, 512, 512, generator=g).to(device)
    with torch.no_grad():
        y = torch.cat([model(x[i:i + 2]).max(1)[1] for i in range(0, B, 2)])  # labels = clean prediction
    return model, x, y


def cpu_baseline(C, backbone, loss, eps, B=2, n_iter=3):
    """The oracle's APGD loop (the restated reference path) on the host cores, bounded sample
    (about 10-30 s): step 0 + n_iter loop iterations on B images."""
    from oracle import sea_oracle as O
    from semseg.utils.utils import ADE_WTS, VOC_WTS
    torch.manual_seed(0)
    model = make_model(backbone, C).eval()
    cores = min(os.cpu_count() or 1, 32)  # more threads than that slow PyTorch-CPU convolutions down
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(B, 3, 512, 512, generator=g)
    with torch.no_grad():
        y = model(x).max(1)[1]
    w = torch.tensor(VOC_WTS if C == 21 else ADE_WTS)
    t0 = time.perf_counter()
    O.apgd_train(model, x, y, eps=eps, n_iter=n_iter, loss=loss, track_loss="ce-avg", weights=w, early_stop=True)
    dt = time.perf_counter() - t0
    # (1 + n_iter) forwards and n_iter backwards ran (the last iteration has no backward, like the
    # reference); every pass is counted as an iteration here, which favours the CPU number
    return {"value": B * (n_iter + 1) / dt, "unit": "image-iterations/s", "cores": cores, "kind": "port",
            "sample": f"oracle apgd_train (PyTorch-CPU restatement of the reference loop), {backbone}, "
                      f"B={B}x512x512, C={C}, step 0 + {n_iter} iteration(s), {loss}, {dt:.1f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--classes", type=int, default=21)
    ap.add_argument("--backbone", default="ConvNeXt-T_CVST")
    ap.add_argument("--loss", default="mask-ce-bal")
    ap.add_argument("--eps", type=float, default=8.0, help="radius in 1/255 (SEA stage-1 radius for eps=4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fuse-upsample", action="store_true",
                    help="force K2u (loss fused with the model's final bilinear upsample); default: the library's "
                         "heuristic (fused for upsample factors >= 8, i.e. Segmenter; unfused for UperNet)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    # SEA_BENCH_BACKEND=gloo lets the multi-rank control flow be exercised on a single-GPU box (ranks then
    # share device 0); the real thing is "nccl" (= RCCL) with one GPU per rank
    backend = os.environ.get("SEA_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend, rank=rank, world_size=world)  # "nccl" is RCCL on ROCm
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    torch.backends.cudnn.benchmark = True  # MIOpen find mode: pick the fastest conv algorithms

    from semseg import _native as N, attacker as A
    from semseg.utils.utils import ADE_WTS, VOC_WTS
    N.lib()
    B, C, K, W = args.batch, args.classes, args.steps, args.warmup
    model, x, y = build_case(rank, B, C, args.backbone, device)
    weights = torch.tensor(VOC_WTS if C == 21 else ADE_WTS, device=device)[:C]
    eps = args.eps / 255.0

    # default: the model's own upsample + the HBM-bound K2 (the kernel SURVEY 8d prices); --fuse-upsample
    # switches to K2u, which is ~0.1 ms/step faster at C=21 and ~5 ms/step on Segmenter (x16, C=151)
    run = A.ApgdRun(model, x, y, eps, W + K + 1, args.loss, "ce-avg", True, C, weights, x.clone(),
                    fuse_upsample=True if args.fuse_upsample else None)
    run.start()
    for i in range(W):
        run.step(i)
    run.k2_events = []

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for i in range(W, W + K):
        run.step(i)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    k2_ms = sum(a.elapsed_time(b) for a, b in run.k2_events) / max(len(run.k2_events), 1)
    # measured device-copy ceiling of this GPU (SURVEY 8d asks for it next to the 8 TB/s spec peak)
    src = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=device)  # 1 GiB
    dst = torch.empty_like(src)
    dst.copy_(src)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a_, b_ in ev:
        a_.record()
        dst.copy_(src)
        b_.record()
    torch.cuda.synchronize()
    copy_gbs = 2 * src.numel() * 4 / (min(a_.elapsed_time(b_) for a_, b_ in ev) * 1e-3) / 1e9
    del src, dst
    algo = k2_algorithmic_bytes(B, C, 512 * 512)
    kname = "loss_upsampled_kernel (K2u)" if run.fused else f"loss_nchw_reg<C={C}> (K2 fused loss fwd+bwd)"
    achieved = algo / (k2_ms * 1e-3) / 1e9
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "k2_traffic.json")
    if os.path.exists(tfile):
        try:
            traffic = json.load(open(tfile)).get(f"B{B}_C{C}")
        except Exception:
            traffic = None

    if rank == 0:
        out = {
            "metric": "SEA attack-iterations/sec (UperNet-CNX-T, 512x512)",
            "value": world * B * K / dt,
            "unit": "image-iterations/s",
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt * 1e3 / K,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{'Segmenter-' if args.backbone.startswith('vit_') else 'UperNet-'}{args.backbone} C={C} "
                            f"({'PASCAL-VOC' if C == 21 else 'ADE20K'}-shaped), {B}x512x512 per GPU, APGD L-inf "
                            f"eps={args.eps:g}/255, loss {args.loss}, track ce-avg (BASELINE configs[1] loop body)",
                "batch_per_gpu": B, "global_batch": world * B, "sharding": f"images x{world}, no in-loop collective",
                "batch_steps_per_s": world * K / dt,
            },
            "roofline": {"kernel": kname, "bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes": algo, "avg_launch_ms": k2_ms,
                         "measured_copy_ceiling_GBps": copy_gbs, "frac_of_copy_ceiling": achieved / copy_gbs},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(C, args.backbone, args.loss, eps)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
