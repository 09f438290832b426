#!/usr/bin/env python3
"""Dev tool: per-basic-block instruction mix of one kernel in an assembly listing (blocks with >= min_instr instructions).
usage: isa_blocks.py file.s <mangled-name-substring> [min_instr]"""
import re, sys
from collections import Counter
t = open(sys.argv[1]).read().splitlines()
key = sys.argv[2]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 100
start = [k for k, l in enumerate(t) if l.startswith('_ZN') and key in l and ':' in l.split(';')[0]][0]
end = [k for k in range(start, len(t)) if '.amdhsa_kernel' in t[k]][0]
blocks, cur, name = [], [], 'entry'
for l in t[start + 1:end]:
    s = l.strip()
    if not s or s.startswith(';'): continue
    if re.match(r'^\.LBB\d+_\d+:', s):
        blocks.append((name, cur)); cur = []; name = s.split(':')[0]; continue
    if s.startswith('.'): continue
    cur.append(s)
blocks.append((name, cur))
print('total instructions', sum(len(b) for _, b in blocks), 'blocks', len(blocks))
for name, b in blocks:
    if len(b) < mn: continue
    c = Counter(x.split()[0] for x in b)
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    print(f"{name}: {len(b)} instr, {valu} VALU | " + ' '.join(f"{k}:{v}" for k, v in c.most_common(16)) + f" | ends: {b[-1][:44]}")
