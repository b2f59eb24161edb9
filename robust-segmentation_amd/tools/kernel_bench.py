#!/usr/bin/env python3
"""Micro-benchmark of the attack-side HIP kernels: achieved algorithmic GB/s vs the HBM roofline.

    python robust-segmentation_amd/tools/kernel_bench.py [--quick]

Inputs follow SURVEY 8(d): logits randn*3 with the label logit boosted on 70% of the pixels, B=8,
512x512.  Variants are timed interleaved in one process (rounds x variants) with HIP events on the
launch stream; the median per variant is reported.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd")]

import torch  # noqa: E402

from semseg import _native as N  # noqa: E402

PEAK = 8000.0


def timeit(fns, rounds=15, warm=3):
    """fns: dict name -> callable.  Interleaved rounds; returns dict name -> median ms."""
    for _ in range(warm):
        for f in fns.values():
            f()
    torch.cuda.synchronize()
    ts = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            f()
            b.record()
            torch.cuda.synchronize()
            ts[k].append(a.elapsed_time(b))
    return {k: sorted(v)[len(v) // 2] for k, v in ts.items()}


def loss_case(B, C, H, W, dtype, channels_last=False, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    logits = torch.randn(B, C, H, W, generator=g, device="cuda") * 3
    y = torch.randint(0, C, (B, H, W), generator=g, device="cuda")
    boost = (torch.rand(B, H, W, generator=g, device="cuda") < 0.7).float() * 6
    logits.scatter_add_(1, y.unsqueeze(1), boost.unsqueeze(1))
    logits = logits.to(dtype)
    if channels_last:
        logits = logits.contiguous(memory_format=torch.channels_last)
    w = torch.rand(C, generator=g, device="cuda")
    return logits, y, w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    N.lib()
    B, H, W = 8, 512, 512
    HW = H * W
    res = []

    def report(name, ms, nbytes):
        gbs = nbytes / (ms * 1e-3) / 1e9
        res.append(dict(kernel=name, ms=ms, algorithmic_MB=nbytes / 1e6, GBps=gbs, frac_of_8TBps=gbs / PEAK))
        print(f"{name:58s} {ms:8.4f} ms  {nbytes / 1e6:9.1f} MB  {gbs:8.1f} GB/s  {gbs / PEAK:6.1%}", flush=True)

    # ---------------- K2 --------------------------------------------------------------------------
    for C, dtype in ((21, torch.float32), (151, torch.float32), (21, torch.bfloat16), (151, torch.bfloat16)):
        if args.quick and C == 151 and dtype != torch.float32:
            continue
        logits, y, w = loss_case(B, C, H, W, dtype)
        y8 = y.to(torch.uint8)
        s = logits.element_size()
        dl = torch.empty_like(logits)
        pred8 = torch.empty(B, H, W, dtype=torch.uint8, device="cuda")
        pred64 = torch.empty(B, H, W, dtype=torch.int64, device="cuda")
        ws = N.loss_workspace(B, HW, "cuda")
        fns = {}
        for mode, mname in ((1, "mask-ce-bal"), (2, "js-avg")):
            for vec in ((4, 2, 1) if C <= 32 else (1,)):
                fns[f"K2 C={C} {str(dtype)[6:]} {mname} vec{vec} u8 labels/pred +grad"] = (
                    lambda mode=mode, vec=vec: N.loss_fwd_bwd(logits, y8, w, mode, 3, 1.0 / HW, True, pred=pred8,
                                                              workspace=ws, dlogits=dl, force_vec=vec))
        if dtype == torch.float32:
            v0 = 4 if C <= 32 else 1
            for tune, tname in ((1, "nt-store"), (2, "nt-load"), (3, "nt-load+store"), (4, "4 waves/SIMD"), (5, "4w+nt-store"),
                                (7, "4w+nt-ld+st")):
                if C > 32 and tune > 3:
                    continue
                fns[f"K2 C={C} float32 mask-ce-bal vec{v0} TUNE[{tname}] u8 +grad"] = (
                    lambda tune=tune, v0=v0: N.loss_fwd_bwd(logits, y8, w, 1, 3, 1.0 / HW, True, pred=pred8, workspace=ws,
                                                           dlogits=dl, force_vec=v0 | (tune << 4)))
        fns[f"K2 C={C} {str(dtype)[6:]} mask-ce-bal auto i64 labels/pred +grad"] = (
            lambda: N.loss_fwd_bwd(logits, y, w, 1, 3, 1.0 / HW, True, pred=pred64, workspace=ws, dlogits=dl))
        fns[f"K2 C={C} {str(dtype)[6:]} mask-ce-bal auto u8 no-grad"] = (
            lambda: N.loss_fwd_bwd(logits, y8, w, 1, 3, 1.0 / HW, False, pred=pred8, workspace=ws))
        t = timeit(fns)
        for k, ms in t.items():
            grad = "no-grad" not in k
            report(k, ms, B * HW * ((2 if grad else 1) * C * s + 16))
        if not args.quick:
            lcl = logits.contiguous(memory_format=torch.channels_last)
            dcl = torch.empty_like(lcl)
            t = timeit({f"K2 C={C} {str(dtype)[6:]} mask-ce-bal NHWC(LDS) u8 +grad":
                        lambda: N.loss_fwd_bwd(lcl, y8, w, 1, 3, 1.0 / HW, True, pred=pred8, workspace=ws, dlogits=dcl)})
            for k, ms in t.items():
                report(k, ms, B * HW * (2 * C * s + 16))
        del logits, dl

    # ---------------- K2u (fused bilinear upsample + loss) vs upsample + K2 + upsample-backward --------------
    for C, hl, lab in ((21, 128, "UperNet x4"), (151, 128, "UperNet x4"), (151, 32, "Segmenter x16")):
        low = torch.randn(B, C, hl, hl, device="cuda") * 3
        y8 = torch.nn.functional.interpolate(low, size=(H, W), mode="bilinear").max(1)[1].to(torch.uint8)
        w = torch.rand(C, device="cuda")
        dlow = torch.empty_like(low)
        pred8 = torch.empty(B, H, W, dtype=torch.uint8, device="cuda")
        hi = torch.empty(B, C, H, W, device="cuda")
        dl = torch.empty_like(hi)
        ws = N.loss_workspace(B, HW, "cuda")

        def unfused():
            lo = low.detach().requires_grad_(True)
            up = torch.nn.functional.interpolate(lo, size=(H, W), mode="bilinear", align_corners=False)
            r = N.loss_fwd_bwd(up.detach(), y8, w, 1, 3, 1.0 / HW, True, pred=pred8, workspace=ws, dlogits=dl)
            return torch.autograd.grad(up, lo, r["dlogits"])

        t = timeit({
            f"K2u fused C={C} {lab} +grad": lambda: N.loss_fwd_bwd_upsampled(low, y8, w, 1, 3, 1.0 / HW, True, pred=pred8, dlow=dlow),
            f"K2u fused C={C} {lab} no-grad": lambda: N.loss_fwd_bwd_upsampled(low, y8, w, 1, 3, 1.0 / HW, False, pred=pred8),
            f"ATen upsample + K2 + upsample-bwd C={C} {lab}": unfused,
        }, rounds=7)
        for k, ms in t.items():
            report(k, ms, B * HW * ((1 if "no-grad" in k else 2) * C * 4 + 16))
        del low, hi, dl, dlow

    # ---------------- K1 / K6 / K5 / K4 -------------------------------------------------------------
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(B, 3, H, W, generator=g, device="cuda")
    xa, xo, gr = (torch.rand_like(x) for _ in range(3))
    gr -= 0.5
    out, xin = torch.empty_like(x), torch.empty_like(x)
    step = torch.full((B,), 16 / 255, device="cuda")
    n = x.numel() * 4
    fl_all = torch.ones(3, B, dtype=torch.uint8, device="cuda")
    fl_all[2] = 0
    xb, gb, xba = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    t = timeit({
        "K1 apgd_linf_step (4 in, 1 out)": lambda: N.apgd_linf_step(x, xa, xo, gr, step, 8 / 255, 0.75, out=out),
        "K6 pgd_linf_step (3 in, 2 out)": lambda: N.pgd_linf_step(x, xa, gr, 1e-2, 4 / 255, delta_out=out, x_in_out=xin),
        "K5 linf_project (2 in, 1 out)": lambda: N.linf_project(xa, x, 8 / 255, out=out),
        "K4 select_copy (adv+best flags: 2 in, 3 out)": lambda: N.select_copy(fl_all, xa, gr, xb, gb, xba),
        "ref torch copy_ (1 in, 1 out)": lambda: out.copy_(x),
    })
    mult = {"K1": 5, "K6": 5, "K5": 3, "K4": 5, "ref": 2}
    for k, ms in t.items():
        report(k, ms, mult[k.split()[0]] * n)
    big_src = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device="cuda")
    big_dst = torch.empty_like(big_src)
    t = timeit({"ref device copy 1 GiB -> 1 GiB (measured HBM copy ceiling)": lambda: big_dst.copy_(big_src)}, rounds=7)
    for k, ms in t.items():
        report(k, ms, 2 * big_src.numel() * 4)
    del big_src, big_dst

    # ---------------- M2 / M1 (model side) ---------------------------------------------------------------
    import torch.nn.functional as F
    for Cc, hl, Hh in ((512, 64, 128), (512, 32, 128), (512, 16, 128), (512, 32, 64), (512, 16, 32), (21, 128, 512), (151, 128, 512), (151, 32, 512)):
        xin = torch.randn(B, Cc, hl, hl, device="cuda")
        gy = torch.randn(B, Cc, Hh, Hh, device="cuda")
        t = timeit({
            f"M2 upsample fwd {Cc}ch {hl}->{Hh}": lambda: N.upsample_bilinear(xin, (Hh, Hh)),
            f"M2 upsample bwd {Cc}ch {hl}->{Hh}": lambda: N.upsample_bilinear_backward(gy, (hl, hl)),
            f"ATen upsample fwd {Cc}ch {hl}->{Hh}": lambda: F.interpolate(xin, size=(Hh, Hh), mode="bilinear", align_corners=False),
        }, rounds=7)
        if Cc % 4 == 0:
            xcl = xin.contiguous(memory_format=torch.channels_last)
            gcl = gy.contiguous(memory_format=torch.channels_last)
            t.update(timeit({
                f"M2 upsample NHWC fwd {Cc}ch {hl}->{Hh}": lambda: N.upsample_bilinear_cl(xcl, (Hh, Hh)),
                f"M2 upsample NHWC bwd {Cc}ch {hl}->{Hh}": lambda: N.upsample_bilinear_backward_cl(gcl, (hl, hl)),
            }, rounds=7))
        for k, ms in t.items():
            report(k, ms, 4 * (xin.numel() + gy.numel()))
    for Cc, hw in ((96, 128), (192, 64), (384, 32), (768, 16)):
        xn = torch.randn(B, hw, hw, Cc, device="cuda")
        xc = torch.randn(B, Cc, hw, hw, device="cuda")
        wt = torch.randn(49, Cc, device="cuda") * 0.1
        wc = torch.randn(Cc, 1, 7, 7, device="cuda") * 0.1
        bb = torch.randn(Cc, device="cuda")
        t = timeit({
            f"M1 dwconv7x7 NHWC C={Cc} {hw}x{hw} fwd": lambda: N.dwconv7x7_nhwc(xn, wt, bb),
            f"M1 dwconv7x7 NHWC C={Cc} {hw}x{hw} bwd-data": lambda: N.dwconv7x7_nhwc(xn, wt, None, flip=True),
            f"M1 dwconv7x7 NCHW C={Cc} {hw}x{hw} fwd": lambda: N.dwconv7x7(xc, wc, bb),
        }, rounds=7)
        for k, ms in t.items():
            report(k, ms, 8 * xn.numel())

    # ---------------- M5 LayerNorm / M4 Winograd transforms --------------------------------------------
    for rows, Cc in ((8 * 256 * 256, 48), (8 * 128 * 128, 96), (8 * 64 * 64, 192), (8 * 32 * 32, 384), (8 * 16 * 16, 768)):
        xl = torch.randn(rows, Cc, device="cuda")
        wl, bl = torch.randn(Cc, device="cuda"), torch.randn(Cc, device="cuda")
        _, mean, rstd = N.layernorm(xl, wl, bl, 1e-6)
        t = timeit({
            f"M5 layernorm fwd rows={rows} C={Cc}": lambda: N.layernorm(xl, wl, bl, 1e-6),
            f"ATen layer_norm fwd rows={rows} C={Cc}": lambda: F.layer_norm(xl, (Cc,), wl, bl, 1e-6),
        }, rounds=7)
        for k, ms in t.items():
            report(k, ms, 8 * xl.numel())
        gl = torch.randn(rows, Cc, device="cuda")
        t = timeit({f"M5 layernorm bwd rows={rows} C={Cc}": lambda: N.layernorm_backward(gl, xl, wl, mean, rstd)}, rounds=7)
        for k, ms in t.items():
            report(k, ms, 12 * xl.numel())
    for m in (2, 4):
        for Cin, Cout in ((2048, 512), (512, 512)):
            xw = torch.randn(B, Cin, 128, 128, device="cuda").contiguous(memory_format=torch.channels_last)
            ww = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.01
            U = N.wino_filter(ww, m, False)
            T = N.lib().sea_wino_tiles(B, 128, 128, m)
            A2 = (m + 2) ** 2
            V = torch.empty(A2, T, Cin, device="cuda")
            Mx = torch.empty(A2, T, Cout, device="cuda")
            yw = torch.empty(B, Cout, 128, 128, device="cuda").contiguous(memory_format=torch.channels_last)
            L = N.lib()
            st = torch.cuda.current_stream().cuda_stream
            t = timeit({
                f"M4 winograd F({m}x{m}) input transform {Cin}ch 128x128": lambda: L.sea_wino_input_transform(
                    xw.data_ptr(), Cin, None, None, V.data_ptr(), Cin, B, Cin, 128, 128, m, st),
                f"M4 winograd F({m}x{m}) output transform {Cout}ch 128x128": lambda: L.sea_wino_output_transform(
                    Mx.data_ptr(), None, None, None, 0, yw.data_ptr(), B, Cout, 128, 128, m, st),
            }, rounds=7)
            report(f"M4 winograd F({m}x{m}) input transform {Cin}ch 128x128", t[f"M4 winograd F({m}x{m}) input transform {Cin}ch 128x128"],
                   4 * (xw.numel() + V.numel()))
            report(f"M4 winograd F({m}x{m}) output transform {Cout}ch 128x128", t[f"M4 winograd F({m}x{m}) output transform {Cout}ch 128x128"],
                   4 * (yw.numel() + Mx.numel()))
            t = timeit({
                f"M4 winograd F({m}x{m}) conv3x3 {Cin}->{Cout} 128x128 (transforms + hipBLASLt bmm)": lambda: N.wino_conv3x3_cl(xw, U, m),
                f"MIOpen conv3x3 {Cin}->{Cout} 128x128": lambda: F.conv2d(xw, ww, padding=1),
            }, rounds=5)
            flops = 2.0 * B * 128 * 128 * Cin * Cout * 9
            for k, ms in t.items():
                print(f"{k:84s} {ms:9.4f} ms   {flops / ms / 1e9:8.1f} direct-equivalent TFLOP/s", flush=True)
            del xw, ww, U, V, Mx, yw

    # ---------------- K3 ----------------------------------------------------------------------------
    for C in (21, 151):
        pred = torch.randint(0, C, (B, H, W), device="cuda", dtype=torch.uint8)
        y8 = pred.clone()
        flip = torch.rand(B, H, W, device="cuda") < 0.3
        y8[flip] = torch.randint(0, C, (int(flip.sum()),), device="cuda", dtype=torch.uint8)
        p64, y64 = pred.long(), y8.long()
        outs = tuple(torch.zeros(B, C, dtype=torch.int64, device="cuda") for _ in range(3))
        hist = torch.zeros(C, C, dtype=torch.int64, device="cuda")
        t = timeit({
            f"K3 class_counts C={C} per-image u8": lambda: N.class_counts(pred, y8, C, True, False, out=outs),
            f"K3 class_counts C={C} per-image i64": lambda: N.class_counts(p64, y64, C, True, False, out=outs),
            f"K3 confusion C={C} u8": lambda: N.confusion(pred, y8, C, hist),
        })
        for k, ms in t.items():
            report(k, ms, B * HW * (16 if "i64" in k else 2))
    if args.json:
        json.dump(res, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
