#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files for one kernel into a small JSON (committed under profiles/).

    python3 tools/pmc_summary.py <kernel substring> out.json <instance-steps per launch> "<profiled command>" dir1 dir2 ...
HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KiB, collected in
separate passes; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads, so it is doubled (an upper bound for
narrow accesses).  bench.py scales `hbm_bytes_per_instance_step` to the launches of its own run.
"""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the sources a kernel family is compiled from: bench.py recomputes this fingerprint and marks a traffic figure whose kernel has changed since it was measured
KERNEL_SOURCES = {"loop_kernel": ("mpc_amd.hip", "mpc_wave.hpp", "mpc_device.hpp", "mpc_tp.hpp", "mpc_sym.hpp"),
                  "nmpc": ("mpc_nmpc.hip", "mpc_nmpc.hpp", "mpc_wave.hpp", "mpc_device.hpp", "mpc_tp.hpp", "mpc_sym.hpp"),
                  "enmpc": ("mpc_enmpc.hip", "mpc_enmpc.hpp", "mpc_rk4s2.hpp", "mpc_device.hpp", "mpc_tp.hpp", "mpc_sym.hpp")}


def sources_sha16(kernel_name):
    fam = "enmpc" if "enmpc" in kernel_name else ("nmpc" if "nmpc" in kernel_name else "loop_kernel")
    h = hashlib.sha256()
    for f in KERNEL_SOURCES[fam]:
        h.update(open(os.path.join(ROOT, "mpc-code_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]



kern = None
if __name__ == "__main__":      # (bench.py imports the fingerprint only)
    if len(sys.argv) < 6:
        raise SystemExit(__doc__)
    kern, out, inst_steps, command, dirs = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4], sys.argv[5:]
if kern is not None:
    rows = []
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if kern in r["Kernel_Name"]:
                    rows.append((r["Kernel_Name"], int(r["Grid_Size"]), r["Counter_Name"], float(r["Counter_Value"])))
    # the workload's launches are the ones of the most frequent (instantiation, grid) pair: a library's create-time self-test launches the same kernels on a handful of instances
    freq = collections.Counter((n, g) for n, g, _, _ in rows)
    keep = freq.most_common(1)[0][0] if freq else None
    acc = collections.defaultdict(list)
    names = set()
    for n, g, c, v in rows:
        if (n, g) == keep:
            acc[c].append(v)
            names.add(n)
    summ = {k: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for k, v in acc.items()}
    summ["kernel"] = sorted(names)[0] if names else None
    summ["kernel_short"] = kern
    summ["command"] = command
    summ["instance_steps_per_launch"] = inst_steps
    summ["sources_sha16"] = sources_sha16(kern)      # the kernel sources this figure was measured on
    if "FETCH_SIZE" in summ and "WRITE_SIZE" in summ:
        summ["hbm_bytes_per_launch"] = (2.0 * summ["FETCH_SIZE"]["mean_per_launch"] + summ["WRITE_SIZE"]["mean_per_launch"]) * 1024.0
        summ["hbm_bytes_per_instance_step"] = summ["hbm_bytes_per_launch"] / inst_steps
        summ["note"] = "bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB: gfx950 FETCH_SIZE counts 64 B per 128-B request on streaming reads"
    if "SQ_WAIT_ANY" in summ and "SQ_WAVE_CYCLES" in summ:
        summ["wait_fraction"] = summ["SQ_WAIT_ANY"]["mean_per_launch"] / summ["SQ_WAVE_CYCLES"]["mean_per_launch"]
    json.dump(summ, open(out, "w"), indent=1)
    print(json.dumps(summ, indent=1))
