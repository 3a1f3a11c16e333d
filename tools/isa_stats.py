#!/usr/bin/env python3
"""Instruction mix per basic-block loop of a kernel in a hipcc -S dump (which loops are hot, what is in them)."""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and pat in l)
end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or (i > start and re.match(r"^_Z\w*:", lines[i])))
body = lines[start:end]


def cls(op):
    if re.match(r"v_(fma|mul|add|max|min)_f64", op): return "fp64"
    if op.startswith("v_rcp"): return "rcp"
    if op.startswith("scratch_"): return op.split("_dword")[0]
    if op.startswith("global_"): return op
    if op.startswith("v_accvgpr"): return "accvgpr"
    if op.startswith(("v_readlane", "v_writelane")): return "lane_spill"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_load"): return "s_load"
    if op.startswith("v_cndmask"): return "cndmask"
    if op.startswith("v_cmp"): return "v_cmp"
    if op.startswith("v_mov"): return "v_mov"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_"): return "valu_other"
    return "other"


# basic blocks
blocks, cur, name = [], collections.Counter(), "entry"
for l in body:
    t = l.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        blocks.append((name, cur)); cur = collections.Counter(); name = m.group(1); continue
    if not t or t.startswith((".", ";", "//")) or t.endswith(":"):
        continue
    cur[cls(t.split()[0])] += 1
blocks.append((name, cur))
tot = collections.Counter()
for n, c in blocks:
    tot.update(c)
print("kernel total", sum(tot.values()), dict(tot.most_common()))
for n, c in sorted(blocks, key=lambda x: -sum(x[1].values()))[:8]:
    print(f"{n:12s} {sum(c.values()):6d}", dict(c.most_common()))
