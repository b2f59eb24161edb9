#!/usr/bin/env python3
"""The kernel sequence of ONE attack step from a rocprofv3 kernel trace (the last complete step: between two launches of the
K1 step kernel), one line per launch: start offset, duration, grid, name.  Shows WHICH launch a library kernel belongs to.
   python devtools/trace_step_sequence.py <kernel_trace.csv> [anchor substring = apgd_linf]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
anchor = sys.argv[2] if len(sys.argv) > 2 else "apgd_linf"
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
if len(idx) < 3:
    sys.exit(f"fewer than three launches of '{anchor}'")
a, b = idx[-3], idx[-2]
gcol = next((c for c in ("Grid_Size", "Grid_Size_X", "Grid_X") if c in rows[0]), None)
t0 = int(rows[a]["Start_Timestamp"])
print(f"{b - a} launches, {(int(rows[b]['Start_Timestamp']) - t0) / 1e6:.3f} ms")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  grid {r[gcol]:>9s}  {r['Kernel_Name'][:110]}")
