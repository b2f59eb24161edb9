#!/bin/bash
# kernel-level breakdown of the PIR-AT outer step under bf16 autocast (configs[3])
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
rm -rf /tmp/pp; PIRAT_MODE=${1:-bf16} timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 devtools/pirat_bench.py > /tmp/pp.log 2>&1
f=$(ls /tmp/pp/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over the whole run (8 outer steps incl. warm-up)")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f} %  {int(r['Calls']):6d} calls  avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:110]}")
PY
