#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a hipcc -S dump: tools/isa_blocks.py file.s <kernel name substring> [min instrs]"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 60
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*:', l) and pat in l)
end = next(i for i in range(start, len(lines)) if '.end_amdhsa_kernel' in lines[i] or (i > start and re.match(r'^_Z\w*:', lines[i])))
name = 'entry'; cur = collections.Counter(); blocks = []
for l in lines[start:end]:
    t = l.strip()
    m = re.match(r'^(\.LBB\d+_\d+):', t)
    if m:
        blocks.append((name, cur)); cur = collections.Counter(); name = m.group(1); continue
    if not t or t.startswith(('.', ';', '//')) or t.endswith(':'):
        continue
    op = t.split()[0]
    if op.startswith('scratch_'): cur['scratch'] += 1
    elif op.startswith('global_load'): cur['gl'] += 1
    elif op.startswith('global_store'): cur['gs'] += 1
    elif op.startswith('ds_'): cur['ds'] += 1
    elif op.startswith('s_load'): cur['s_load'] += 1
    elif op.startswith('s_waitcnt'): cur['wait'] += 1
    elif op.startswith('v_accvgpr'): cur['acc'] += 1
    elif op.startswith(('v_readlane', 'v_writelane')): cur['lane'] += 1
    elif re.match(r'v_(fma|mul|add|max|min)_f64', op): cur['fp64'] += 1
    elif op == 's_barrier': cur['barrier'] += 1
    elif op.startswith('s_cbranch'):
        cur['br'] += 1
        tgt = t.split()[-1]
        cur['loop->' + tgt] += 0
        if tgt == name: cur['SELF_LOOP'] += 1
    cur['n'] += 1
blocks.append((name, cur))
tot = collections.Counter()
for n, c in blocks:
    tot.update({k: v for k, v in c.items() if not k.startswith('loop->')})
    if c['n'] >= minn or c['barrier'] or c['SELF_LOOP']:
        print(n, {k: v for k, v in c.items() if not k.startswith('loop->')})
print('total', dict(tot))
