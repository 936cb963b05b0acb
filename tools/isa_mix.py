#!/usr/bin/env python
"""Instruction-mix summary of one kernel in a hipcc -S dump: per basic block, the order of MFMA (M), LDS (D), VALU (v), SALU (s),
waits (W), barriers (B), global (G), jumps (J), scratch (X); runs of four or more are written x{n}.

    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only file.hip -o /tmp/k.s;  python tools/isa_mix.py /tmp/k.s <substring of the kernel symbol>
"""
import re, sys
L = open(sys.argv[1]).read().split('\n')
st = [i for i, l in enumerate(L) if l.endswith(':') is False and l.split(':')[0].startswith('_Z') and sys.argv[2] in l.split(':')[0] and ':' in l][0]
en = [i for i in range(st, len(L)) if 's_endpgm' in L[i]][0]
s = ''
tot = {}
for l in L[st + 1:en]:
    t = l.strip()
    if t.startswith('.LBB'):
        s += '\n' + t.split(':')[0] + ': '
        continue
    if not t or t[0] in ';.':
        continue
    op = t.split()[0]
    k = ('M' if op.startswith('v_mfma') else 'D' if op.startswith('ds_') else 'v' if op.startswith('v_') else 'W' if op.startswith('s_waitcnt')
         else 'B' if op.startswith('s_barrier') else 'G' if op.startswith('global_') or op.startswith('buffer_') else
         'J' if op.startswith('s_cbranch') or op.startswith('s_branch') else 'X' if op.startswith('scratch') else 's')
    s += k
for line in s.split('\n'):
    c = {k: line.count(k) for k in 'MDvsWBGX' if line.count(k)}
    print(re.sub(r'(.)\1{3,}', lambda m: f"{m.group(1)}{{{len(m.group(0))}}}", line)[:600], '  ', c)
