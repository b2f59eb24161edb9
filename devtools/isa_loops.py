#!/usr/bin/env python3
"""Per kernel of a hipcc -S listing: the loops (backward branches) with their instruction mix -- MFMAs, other vector
instructions, LDS reads, DMA / global loads, scratch (spill) traffic.  `python devtools/isa_loops.py file.s [name filter]`"""
import re
import sys

PATS = [("mfma", r".*v_mfma"), ("other vector", r"\s+v_(?!mfma)"), ("ds_read", r".*ds_read"), ("ds_write", r".*ds_write"),
        ("lds-dma", r".*global_load_lds"), ("global/buffer loads", r"\s+(global|buffer)_load_dword"),
        ("stores", r"\s+(global|buffer)_store"), ("scratch", r".*scratch_"), ("s_waitcnt", r"\s+s_waitcnt"),
        ("barriers", r"\s+s_barrier")]
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
funcs = re.split(r'\n(?=_Z\w+:)', s)
for f in funcs[1:]:
    name = f.split(':')[0]
    if flt not in name:
        continue
    lines = f.split('\n')
    ends = [i for i, l in enumerate(lines) if 's_endpgm' in l]
    end = ends[-1] if ends else len(lines)
    lines = lines[:end]

    def cnt(pat, a=0, b=None):
        return sum(bool(re.match(pat, l)) for l in lines[a:b])

    labels = {l.split(':')[0]: i for i, l in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:', l)}
    br = [(i, l.split()[-1]) for i, l in enumerate(lines) if re.match(r'\s+s_c?branch', l)]
    loops = [(labels[t], i) for i, t in br if t in labels and labels[t] < i]
    print(name)
    print("   whole kernel:", ", ".join(f"{n} {cnt(p)}" for n, p in PATS))
    for a, b in loops:
        print(f"   loop [{a}, {b}]:", ", ".join(f"{n} {cnt(p, a, b)}" for n, p in PATS))
