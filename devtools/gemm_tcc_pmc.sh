#!/bin/bash
# L2 / fabric counters of one split-GEMM shape (separate PMC passes, counters only)
#   bash devtools/gemm_tcc_pmc.sh OUTDIR -- <python script and args>
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
OUT=$1; shift; shift
rm -rf /tmp/gt; mkdir -p $OUT /tmp/gt
echo "== $*" >> $OUT/tcc_summary.txt
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_NC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/gt/$i -- python3 "$@" > /tmp/gt/log$i.txt 2>&1 || tail -3 /tmp/gt/log$i.txt
  f=$(ls /tmp/gt/$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY' >> $OUT/tcc_summary.txt
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm_split" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:36s} per launch {sum(v) / len(v):.4e}   ({len(v)} launches)")
PY
done
cat $OUT/tcc_summary.txt
