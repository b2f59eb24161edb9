#!/bin/bash
# HBM traffic of the M9 stem kernels (separate --pmc passes, counters only: gpurun rules), devtools/stem_bench.py as the driver
#   gpurun -- 'bash devtools/stem_pmc.sh gpurun_out/r5_stem_pmc'
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
OUT=${1:-gpurun_out/r5_stem_pmc}
rm -rf /tmp/sp; mkdir -p $OUT /tmp/sp
: > $OUT/r5_stem_pmc.txt
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/sp/$i -- python3 devtools/stem_bench.py > /tmp/sp/log$i.txt 2>&1 || tail -3 /tmp/sp/log$i.txt
  f=$(ls /tmp/sp/$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY' >> $OUT/r5_stem_pmc.txt
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "stem_conv1" in n or "ln_gelu_cl" in n:
        acc[(n.split("(")[0].replace("void sea::", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    m = sorted(v)[len(v) // 2]
    note = ""
    if c == "FETCH_SIZE":
        note = f"  -> reads {2 * m * 1024 / 1e6:8.1f} MB (2 x FETCH_SIZE KiB: gfx950 correction of MI355X_MICROARCH.md)"
    if c == "WRITE_SIZE":
        note = f"  -> writes {m * 1024 / 1e6:8.1f} MB"
    if c == "TCP_TCC_READ_REQ_sum":
        note = f"  -> L2 -> L1 {m * 128 / 1e6:8.1f} MB (128-byte requests)"
    print(f"{k:44s} {c:24s} median per launch {m:.4e} ({len(v)} launches){note}")
PY
done
cat $OUT/r5_stem_pmc.txt
