#!/usr/bin/env python3
"""instruction-class picture of a line range of a kernel in a hipcc -S listing: M mfma, v other vector, d LDS, G LDS-DMA,
W s_waitcnt, B barrier, n s_nop, . scalar.  `python devtools/isa_seq.py file.s <kernel substring> <first> <last>`
(line numbers relative to the kernel, as devtools/isa_loops.py prints them)"""
import re
import sys
import textwrap

s = open(sys.argv[1]).read()
funcs = [f for f in re.split(r'\n(?=_Z\w+:)', s)[1:] if sys.argv[2] in f.split(':')[0]]
lines = funcs[0].split('\n')
a, b = int(sys.argv[3]), int(sys.argv[4])
seq = ''
for l in lines[a:b + 1]:
    m = re.match(r'\s+([a-z_0-9]+)', l)
    if not m:
        continue
    op = m.group(1)
    seq += ('M' if op.startswith('v_mfma') else 'v' if op.startswith('v_') else 'G' if op.startswith('global_load_lds') else
            'd' if op.startswith('ds_') else 'W' if op.startswith('s_waitcnt') else 'B' if op.startswith('s_barrier') else
            'n' if op.startswith('s_nop') else '.')
print('\n'.join(textwrap.wrap(seq, 120)))
