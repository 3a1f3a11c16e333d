#!/usr/bin/env python3
"""Per-kernel registers / scratch / occupancy / LDS from a hipcc log made with -Rpass-analysis=kernel-resource-usage:  python tools/kernel_resources.py build.log"""
import re
import subprocess
import sys

t = open(sys.argv[1]).read()
for b in re.split(r"remark: Function Name: ", t)[1:]:
    name = b.split()[0]
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0][:58]
    print("%-60s VGPR %4s AGPR %4s scratch %6s B/lane  waves/SIMD %s  LDS %6s B" % (dn, g(r"    VGPRs"), g(r"AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
