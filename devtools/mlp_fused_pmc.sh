#!/bin/bash
# PMC passes over one fused-MLP shape (separate runs per counter group, counters + kernel trace only: gpurun rules)
#   bash devtools/mlp_fused_pmc.sh C M fwd|bwd
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
OUT=gpurun_out/mlp_pmc; rm -rf /tmp/gp; mkdir -p $OUT /tmp/gp
echo "== mlp_fused $*" >> $OUT/summary.txt
i=0
for grp in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/gp/$i -- python3 devtools/mlp_fused_case.py "$@" > /tmp/gp/log$i.txt 2>&1 || tail -3 /tmp/gp/log$i.txt
  f=$(ls /tmp/gp/$i/*/*counter_collection.csv 2>/dev/null | head -1)
  k=$(ls /tmp/gp/$i/*/*kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - "$f" "$k" <<'PY' >> $OUT/summary.txt
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "mlp_fused_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(sys.argv[2])) if "mlp_fused_kernel" in r["Kernel_Name"]] if len(sys.argv) > 2 and sys.argv[2] else []
d = sum(dur) / len(dur) if dur else float("nan")
print(f"kernel duration (profiled) {d / 1e3:.1f} us")
for k, v in acc.items():
    m = sum(v) / len(v)
    extra = ""
    if k == "GRBM_GUI_ACTIVE":
        extra = f"  -> clock {m / 8 / d:.2f} GHz"
    if k == "SQ_VALU_MFMA_BUSY_CYCLES":
        extra = f"  -> per SIMD {m / 1024:.0f} cycles"
    print(f"{k:36s} per launch {m:.4e}{extra}")
PY
done
cat $OUT/summary.txt
