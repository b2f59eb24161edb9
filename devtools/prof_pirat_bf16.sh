#!/bin/bash
# kernel-level breakdown of the PIR-AT outer step (configs[3]) in steady state: kernels of the last 0.8 s of the trace
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
rm -rf /tmp/pp; PIRAT_MODE=${1:-bf16} timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -- python3 devtools/pirat_bench.py > /tmp/pp.log 2>&1
f=$(ls /tmp/pp/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
t0, t1 = min(r[0] for r in rows), max(r[1] for r in rows)
cut = t1 - 0.8e9          # the last 0.8 s: ~4 timed outer steps (warm-up and MIOpen's Find are long over)
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in rows:
    if s >= cut:
        agg[n][0] += e - s
        agg[n][1] += 1
tot = sum(v[0] for v in agg.values())
print(f"window {((t1 - cut) / 1e6):.0f} ms wall, kernel time {tot / 1e6:.0f} ms")
for n, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:32]:
    print(f"{d / tot * 100:5.1f} %  {c:6d} calls  avg {d / c / 1e3:8.1f} us  {n[:120]}")
PY
