#!/usr/bin/env python3
"""List the launches of one steady-state APGD step from a rocprofv3 kernel trace whose name matches a pattern:
   python list_kernels.py <trace dir> <regex>   (start offset in the step, duration, grid, name)"""
import csv, glob, os, re, sys
d, pat = sys.argv[1], re.compile(sys.argv[2])
f = (glob.glob(os.path.join(d, "*", "*_kernel_trace.csv")) + glob.glob(os.path.join(d, "*_kernel_trace.csv")))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
k1 = [i for i, r in enumerate(rows) if "apgd_linf_step" in r["Kernel_Name"]]
lo, hi = k1[-3], k1[-2]
t0 = int(rows[lo]["Start_Timestamp"])
prev = ""
for i in range(lo, hi):
    r = rows[i]
    if pat.search(r["Kernel_Name"]):
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us  grid {r.get('Grid_Size', r.get('Grid_Size_X', '?')):>10}  "
              f"{r['Kernel_Name'][:70]}   <- after {prev[:50]}")
    prev = r["Kernel_Name"]
