#!/bin/bash
# PMC passes over one split-GEMM shape (separate runs per counter group, kernel-trace only: gpurun rules)
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
OUT=gpurun_out/gemm_pmc; rm -rf /tmp/gp; mkdir -p $OUT /tmp/gp
rocprofv3 --list-avail 2>/dev/null | grep -o "\b\(SQ_[A-Z_0-9]*\|TCC_[A-Za-z_0-9]*\|TCP_[A-Za-z_0-9]*\|TA_[A-Z_0-9]*\|GRBM_[A-Z_0-9]*\)\b" | sort -u > $OUT/counters.txt
wc -l $OUT/counters.txt
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d /tmp/gp/$i -- python3 devtools/gemm_split_case.py "$@" > /tmp/gp/log$i.txt 2>&1 || tail -3 /tmp/gp/log$i.txt
  f=$(ls /tmp/gp/$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY' >> $OUT/summary.txt
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm_split_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:36s} per launch {sum(v) / len(v):.4e}   ({len(v)} launches)")
PY
done
cat $OUT/summary.txt
