#!/usr/bin/env python3
"""Dev tool: one character per instruction of a kernel in an assembly listing (M mfma, r/w LDS read/write, v VALU,
W s_waitcnt, n s_nop, G/S global load/store, s SALU, B branch, | barrier): shows how loads, waits and MFMAs interleave."""
import sys
t = open(sys.argv[1]).read()
i = t.index(sys.argv[2])
body = t[t.index('\n', i):t.index('.end_amdhsa_kernel', i)]
lines = [l.strip() for l in body.splitlines() if l.strip() and not l.strip().startswith((';', '.'))]
def cat(l):
    op = l.split()[0]
    if op.startswith('v_mfma'): return 'M'
    if op.startswith('ds_read') or op.startswith('ds_bperm'): return 'r'
    if op.startswith('ds_'): return 'w'
    if op.startswith('v_'): return 'v'
    if op.startswith('s_waitcnt'): return 'W'
    if op.startswith('s_nop'): return 'n'
    if op.startswith(('s_cbranch', 's_branch')): return 'B'
    if op.startswith('s_barrier'): return '|'
    if op.startswith('s_'): return 's'
    if op.startswith('global_load'): return 'G'
    if op.startswith('global_store'): return 'S'
    return '?'
s = ''.join(cat(l) for l in lines)
for k in range(0, len(s), 160): print(s[k:k + 160])
