#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/..., scratch) into the small summaries kept under profiles/.

    python robust-segmentation_amd/tools/summarize_profile.py --round r1 \
        --bench gpurun_out/prof_bench_r1 --kernels gpurun_out/prof_k2_r1 \
        --fetch gpurun_out/pmc_fetch_r1 --write gpurun_out/pmc_write_r1 [--warmup 2]
"""
import argparse
import collections
import csv
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one(d, pat):
    fs = glob.glob(os.path.join(d, "*", pat)) + glob.glob(os.path.join(d, pat))
    if not fs:
        raise SystemExit(f"no {pat} under {d}")
    return fs[0]


def short(name, n=110):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return name if len(name) <= n else name[:n] + "..."


def sea_key(n):
    if "loss_nchw_reg" in n:
        m = re.search(r"loss_nchw_reg<(\w+), (\d+), (\d+), (\w+), (\w+)(?:, (\d+))?>", n)
        if not m:
            return "K2 loss_nchw_reg"
        tune = f",tune={m.group(6)}" if m.group(6) else ""
        return f"K2 loss_nchw_reg<{m.group(1)},C={m.group(2)},vec={m.group(3)},grad={m.group(4)}{tune}>"
    m = re.search(r"loss_nchw_split<(\w+), (\d+), (\d+)>", n)
    if m:
        return f"K2 loss_nchw_split<{m.group(1)},C={m.group(2)},waves={m.group(3)}>"
    m = re.search(r"loss_nchw_fwd<(\w+), (\d+), (\d+)>", n)
    if m:
        return f"K2 loss_nchw_fwd<{m.group(1)},chunk={m.group(2)},waves={m.group(3)}>"
    m = re.search(r"loss_upsampled_pow2_kernel<(\d+), (\d+), (\w+)>", n)
    if m:
        return f"K2u loss_upsampled_pow2<x{m.group(1)},slots={m.group(2)},grad={m.group(3)}>"
    m = re.search(r"loss_upsampled_pow2_combine<(\d+)>", n)
    if m:
        return f"K2u loss_upsampled_pow2_combine<x{m.group(1)}>"
    for pat, key in (("loss_nhwc_lds", "K2 loss_nhwc_lds"), ("loss_upsampled_kernel", "K2u loss_upsampled_kernel"),
                     ("stream_copy_kernel", "probe stream_copy"), ("stream_read_kernel", "probe stream_read"), ("loss_nchw_stream", "K2 loss_nchw_stream"),
                     ("loss_finalize", "K2 loss_finalize"), ("apgd_linf_step", "K1 apgd_linf_step"),
                     ("pgd_linf_step", "K6 pgd_linf_step"), ("ew2_v", "K5 random_start/project"),
                     ("apgd_track_kernel", "K7 apgd_track"), ("select_copy_pred", "K4 select_copy_pred"),
                     ("select_copy_v", "K4 select_copy"), ("class_counts_kernel", "K3 class_counts"),
                     ("confusion_kernel", "K3 confusion"), ("count_ignored", "K3 count_ignored")):
        if pat in n:
            return key
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r1")
    ap.add_argument("--bench")
    ap.add_argument("--kernels")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--title", default="B=8, C=21, 512x512, UperNet-ConvNeXt-T, fp32")
    ap.add_argument("--cases", help="stdout of devtools/profile_cases.py (CASE lines) to price --kernels / --fetch / --write rows")
    ap.add_argument("--sq", help="rocprofv3 --pmc SQ_* output directory (per-kernel medians are tabulated)")
    a = ap.parse_args()
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)

    if a.bench:
        rows = list(csv.DictReader(open(one(a.bench, "*_kernel_trace.csv"))))
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        # one K1 launch per APGD step (none in step 0, none in the cold-launch ring bench.py runs after the timed
        # region): the window runs from the first timed step's K1 to the last K4 copy of the last step
        k1 = [i for i, r in enumerate(rows) if "apgd_linf_step" in r["Kernel_Name"]]
        k4 = [i for i, r in enumerate(rows) if "select_copy" in r["Kernel_Name"]]
        lo, hi = k1[a.warmup], max(i for i in k4)
        steps = len(k1) - a.warmup
        win = rows[lo: hi + 1]
        t0, t1 = int(rows[lo]["Start_Timestamp"]), int(rows[hi]["End_Timestamp"])
        agg = collections.defaultdict(lambda: [0, 0])
        for r in win:
            agg[r["Kernel_Name"]][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            agg[r["Kernel_Name"]][1] += 1
        tot = sum(v[0] for v in agg.values())
        with open(os.path.join(out_dir, f"{a.round}_bench_steady_state.md"), "w") as f:
            f.write(f"# bench.py steady state, rocprofv3 --kernel-trace ({a.round})\n\n")
            f.write(f"Window: {steps} timed APGD steps ({a.title}).\n\n")
            f.write(f"- wall per step: {(t1 - t0) / steps / 1e6:.3f} ms; kernel-busy per step: {tot / steps / 1e6:.3f} ms; "
                    f"launches per step: {sum(v[1] for v in agg.values()) / steps:.0f}\n\n")
            f.write("## attack-side HIP kernels (libsea_hip.so)\n\n| kernel | calls/step | avg us | ms/step | % of step |\n|---|---|---|---|---|\n")
            sea = collections.defaultdict(lambda: [0, 0])
            for n, (d, c) in agg.items():
                k = sea_key(n)
                if k:
                    sea[k][0] += d
                    sea[k][1] += c
            for k, (d, c) in sorted(sea.items()):
                f.write(f"| {k} | {c / steps:.1f} | {d / c / 1e3:.2f} | {d / steps / 1e6:.4f} | {100 * d / tot:.3f} |\n")
            f.write(f"| **all attack-side** | | | {sum(v[0] for v in sea.values()) / steps / 1e6:.4f} | "
                    f"{100 * sum(v[0] for v in sea.values()) / tot:.3f} |\n")
            f.write("\n## top kernels (model forward / input-gradient backward: MIOpen, CK, hipBLASLt, ATen)\n\n"
                    "| ms/step | calls/step | % | kernel |\n|---|---|---|---|\n")
            for n, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
                f.write(f"| {d / steps / 1e6:.3f} | {c / steps:.1f} | {100 * d / tot:.1f} | `{short(n)}` |\n")

    if a.kernels:
        rows = list(csv.DictReader(open(one(a.kernels, "*_kernel_stats.csv"))))
        with open(os.path.join(out_dir, f"{a.round}_attack_kernels_stats.csv"), "w") as f:
            f.write("kernel,calls,avg_ns,min_ns,max_ns\n")
            for r in rows:
                if sea_key(r["Name"]):
                    f.write(f"\"{sea_key(r['Name'])}\",{r['Calls']},{float(r['AverageNs']):.0f},{r['MinNs']},{r['MaxNs']}\n")

    if a.fetch and a.write:
        def load(d, counter):
            out = collections.defaultdict(list)
            for r in csv.DictReader(open(one(d, "*_counter_collection.csv"))):
                k = sea_key(r["Kernel_Name"])
                if k and r["Counter_Name"] == counter:
                    out[k].append(float(r["Counter_Value"]))
            return out
        fe, wr = load(a.fetch, "FETCH_SIZE"), load(a.write, "WRITE_SIZE")
        traffic = {}
        with open(os.path.join(out_dir, f"{a.round}_pmc_hbm_traffic.md"), "w") as f:
            f.write(f"# HBM traffic per launch from rocprofv3 --pmc ({a.round}; separate FETCH_SIZE and WRITE_SIZE passes)\n\n"
                    "Counter unit is KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half of the "
                    "bytes of coalesced streaming reads, so reads = 2 x FETCH_SIZE; WRITE_SIZE is exact.  The K5 row is the "
                    "calibration (known traffic: reads 2 x 25.17 MB, writes 25.17 MB).\n\n"
                    "| kernel | FETCH_SIZE KiB (median) | WRITE_SIZE KiB (median) | read MB (2x) | write MB | total MB |\n|---|---|---|---|---|---|\n")
            for k in sorted(fe):
                a_, b_ = sorted(fe[k])[len(fe[k]) // 2], sorted(wr.get(k, [0]))[len(wr.get(k, [0])) // 2]
                rd, wt = 2 * a_ * 1024 / 1e6, b_ * 1024 / 1e6
                f.write(f"| {k} | {a_:.0f} | {b_:.0f} | {rd:.1f} | {wt:.1f} | {rd + wt:.1f} |\n")
                traffic[k] = (rd + wt) * 1e6
        tj = {}
        for k, v in traffic.items():
            m = re.search(r"K2 loss_nchw_(?:reg|split)<float,C=(\d+),", k)
            if m and "grad=false" not in k:
                tj[f"B8_C{m.group(1)}"] = v
        json.dump(tj, open(os.path.join(out_dir, "k2_traffic.json"), "w"), indent=1)


    if a.cases and a.kernels:
        # price the isolated cold launches: algorithmic and moved bytes per launch / rocprofv3 average duration
        stats = list(csv.DictReader(open(one(a.kernels, "*_kernel_stats.csv"))))
        trace = list(csv.DictReader(open(one(a.kernels, "*_kernel_trace.csv"))))
        fe = wr = None
        if a.fetch and a.write:
            def load_raw(d, counter):
                out = collections.defaultdict(list)
                for r in csv.DictReader(open(one(d, "*_counter_collection.csv"))):
                    if r["Counter_Name"] == counter:
                        out[r["Kernel_Name"]].append(float(r["Counter_Value"]))
                return out
            fe, wr = load_raw(a.fetch, "FETCH_SIZE"), load_raw(a.write, "WRITE_SIZE")
        med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
        with open(os.path.join(out_dir, f"{a.round}_cold_kernel_roofline.md"), "w") as f:
            f.write(f"# Attack-side kernels, cold launches, rocprofv3 --kernel-trace ({a.round})\n\n"
                    "`devtools/profile_cases.py`: B=8, 512x512, every launch on buffers that are not in the 256 MiB Infinity Cache "
                    "(ring of independent sets > 1.5 GB).  avg/min us = rocprofv3 kernel durations of the case's launches; "
                    "alg = SURVEY 8(d) bytes (int64 label + argmax), moved = bytes the kernel moves (uint8 label + argmax); "
                    "fractions of the 8 TB/s HBM3E peak.  PMC = 2 x FETCH_SIZE + WRITE_SIZE (separate passes; KiB; gfx950 "
                    "half-FETCH correction, MI355X_MICROARCH.md).\n\n"
                    "| case | kernel | launches | avg us | min us | alg MB | moved MB | PMC MB | alg frac | moved frac |\n|---|---|---|---|---|---|---|---|---|---|\n")
            for line in open(a.cases):
                if not line.startswith("CASE "):
                    continue
                tag, kn, alg, moved, grp, ngrp = [t.strip() for t in line[5:].split("|")]
                alg, moved, grp, ngrp = float(alg), float(moved), int(grp), int(ngrp)
                pat = re.compile(kn)
                hits = sorted((r for r in trace if pat.search(r["Kernel_Name"])), key=lambda r: int(r["Start_Timestamp"]))
                if not hits:
                    f.write(f"| {tag} | (no launch matched `{kn}`) | | | | | | | | |\n")
                    continue
                per = len(hits) // ngrp
                hits = hits[grp * per: (grp + 1) * per]
                name = hits[0]["Kernel_Name"]
                d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in hits]
                d = d[1:] if len(d) > 2 else d          # the first launch of a case pays the code-object load
                avg, mn = sum(d) / len(d) / 1e3, min(d) / 1e3
                pmc = ""
                if fe is not None and ngrp == 1 and name in fe and name in wr:
                    pmc = f"{(2 * med(fe[name]) + med(wr[name])) * 1024 / 1e6:.1f}"
                f.write(f"| {tag} | `{short(sea_key(name) or name, 70)}` | {len(d)} | {avg:.1f} | {mn:.1f} | {alg / 1e6:.1f} | {moved / 1e6:.1f} | "
                        f"{pmc} | {alg / avg / 1e3 / 8000:.1%} | {moved / avg / 1e3 / 8000:.1%} |\n")

    if a.sq:
        rows = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(one(a.sq, "*_counter_collection.csv"))):
            k = sea_key(r["Kernel_Name"])
            if k:
                rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        names = sorted({c for v in rows.values() for c in v})
        with open(os.path.join(out_dir, f"{a.round}_sq_counters.md"), "w") as f:
            f.write(f"# SQ counters per launch (median), rocprofv3 --pmc ({a.round})\n\n| kernel | " + " | ".join(names) + " |\n|---|" + "---|" * len(names) + "\n")
            for k in sorted(rows):
                f.write(f"| {k} | " + " | ".join(f"{sorted(rows[k][c])[len(rows[k][c]) // 2]:.3g}" if rows[k][c] else "" for c in names) + " |\n")


if __name__ == "__main__":
    main()
