#!/usr/bin/env python3
"""Where a kernel spills: scratch stores/loads, MFMAs and VALU instructions per basic block of a
hipcc -save-temps .s file.   usage: spill_map.py file.s kernel_symbol_prefix"""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith(sys.argv[2])][0]
end = [i for i in range(start, len(lines)) if lines[i].strip().startswith('.Lfunc_end')][0]
cur, stats, order = 'entry', {'entry': Counter()}, ['entry']
for l in lines[start:end]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        cur = m.group(1); stats[cur] = Counter(); order.append(cur); continue
    t = l.strip().split()
    if not t:
        continue
    op = t[0]
    if op.startswith('scratch_store'): stats[cur]['st'] += 1
    elif op.startswith('scratch_load'): stats[cur]['ld'] += 1
    elif op.startswith('v_mfma'): stats[cur]['mfma'] += 1
    elif op.startswith('v_'): stats[cur]['valu'] += 1
tot = Counter()
for b in order:
    c = stats[b]; tot.update(c)
    if c['st'] + c['ld'] > 8 or c['mfma'] > 100:
        print(b, dict(c))
print('total', dict(tot))
