#!/usr/bin/env python3
"""instruction-class pictures of the stretches between s_barrier instructions of one kernel in a hipcc -S listing
(M mfma, v other vector, d LDS, G LDS-DMA, W s_waitcnt, B barrier, n s_nop, . scalar):
    python devtools/isa_between_barriers.py file.s <kernel name substring> [min length]"""
import re
import sys
import textwrap

s = open(sys.argv[1]).read()
funcs = [f for f in re.split(r'\n(?=_Z\w+:)', s)[1:] if sys.argv[2] in f.split(':')[0]]
lines = funcs[0].split('\n')
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 300
bars = [i for i, l in enumerate(lines) if 's_barrier' in l]
for a, b in zip(bars, bars[1:]):
    if b - a < minlen:
        continue
    seq = ''
    for l in lines[a:b + 1]:
        m = re.match(r'\s+([a-z_0-9]+)', l)
        if not m:
            continue
        op = m.group(1)
        seq += ('M' if op.startswith('v_mfma') else 'v' if op.startswith('v_') else 'G' if op.startswith('global_load_lds') else
                'd' if op.startswith('ds_') else 'W' if op.startswith('s_waitcnt') else 'B' if op.startswith('s_barrier') else
                'n' if op.startswith('s_nop') else '.')
    print(f"lines {a}..{b}: mfma {seq.count('M')}, vector {seq.count('v')}, scalar {seq.count('.')}, lds {seq.count('d')}, waits {seq.count('W')}")
    print('\n'.join(textwrap.wrap(seq, 130)))
    print()
