#!/usr/bin/env python3
"""Aggregate a rocprofv3 kernel trace by (kernel, grid size): separates the launches of one kernel by problem size.
   python devtools/trace_by_grid.py <kernel_trace.csv> [substring] [last-seconds window]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
win = float(sys.argv[3]) if len(sys.argv) > 3 else 0.8
end = max(int(r["End_Timestamp"]) for r in rows)
rows = [r for r in rows if int(r["Start_Timestamp"]) >= end - win * 1e9]
span = (end - min(int(r["Start_Timestamp"]) for r in rows)) / 1e6
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / 1e6
print(f"window {span:.1f} ms, kernel-busy {busy:.1f} ms, {len(rows)} launches")
gcol = next((c for c in ("Grid_Size", "Grid_Size_X", "Grid_X") if c in rows[0]), None)
if gcol is None:
    print("columns:", list(rows[0].keys()))
acc = collections.defaultdict(list)
for r in rows:
    if sub in r["Kernel_Name"]:
        acc[(r["Kernel_Name"][:70], r[gcol] if gcol else "?")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tab = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
for (name, grid), v in tab[:60]:
    print(f"{sum(v) / 1e6:9.3f} ms  {len(v):5d} x {sum(v) / len(v) / 1e3:9.1f} us  grid {grid:>10s}  {name}")
