#!/usr/bin/env python3
"""Instruction mix of a kernel between consecutive s_memtime stamps of a -DMPC_STAMPS build (hipcc -save-temps .s file)."""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*:', l) and pat in l)
end = next(i for i in range(start, len(lines)) if '.end_amdhsa_kernel' in lines[i])
body = [l.strip() for l in lines[start:end]]
body = [l for l in body if l and not l.startswith((';', '.'))]


def cls(op):
    if re.match(r"v_(fma|mul|add|max|min|fmac)_f64", op): return "fp64"
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_rcp"): return "rcp"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("global_"): return "global"
    if op.startswith("ds_"): return op.split()[0][:12]
    if op.startswith("v_accvgpr"): return "accvgpr"
    if op.startswith(("v_readlane", "v_writelane")): return "lane_spill"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_load"): return "s_load"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("v_cndmask"): return "cndmask"
    if op.startswith("v_cmp"): return "v_cmp"
    if op.startswith("v_mov"): return "v_mov"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_"): return "valu_other"
    return "other"


seg, cur = [], collections.Counter()
for l in body:
    op = l.split()[0]
    if op == 's_memtime':
        seg.append(cur); cur = collections.Counter()
    cur[cls(op)] += 1
seg.append(cur)
for i, c in enumerate(seg):
    print(i, sum(c.values()), dict(c.most_common(14)))
