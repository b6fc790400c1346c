#!/usr/bin/env python
"""Static instruction mix of a kernel between s_barrier instructions (the phases of the strip kernel) from `hipcc -S` output.
usage: python tools/isa_phases.py file.s kernel-symbol-prefix [phase-index-for-opcode-histogram]"""
import collections
import sys

lines = open(sys.argv[1]).read().split('\n')
pref = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith(pref))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
segs, cur = [], []
for l in lines[start:end + 1]:
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'):
        continue
    op = t.split()[0]
    cur.append(op)
    if op == 's_barrier':
        segs.append(cur)
        cur = []
segs.append(cur)
for i, s in enumerate(segs):
    c = collections.Counter(s)
    grp = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    print(i, len(s), 'valu', grp('v_'), 'salu', grp('s_') - c.get('s_waitcnt', 0) - c.get('s_barrier', 0), 'wait', c.get('s_waitcnt', 0), 'lds', grp('ds_'),
          'vmem', grp('global_') + grp('buffer_') + grp('scratch_'))
if len(sys.argv) > 3:
    c = collections.Counter(segs[int(sys.argv[3])])
    print(c.most_common(40))
