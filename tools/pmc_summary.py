#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files for one kernel into a small JSON (committed under profiles/).

    python3 tools/pmc_summary.py loop_kernel out.json dir1 dir2 ...
HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KiB, collected in
separate passes; on gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) coalesced reads, so it is doubled.
"""
import collections
import csv
import glob
import json
import sys

kern, out, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
acc = collections.defaultdict(list)
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {k: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for k, v in acc.items()}
if "FETCH_SIZE" in summ and "WRITE_SIZE" in summ:
    summ["hbm_bytes_per_launch"] = (2.0 * summ["FETCH_SIZE"]["mean_per_launch"] + summ["WRITE_SIZE"]["mean_per_launch"]) * 1024.0
    summ["note"] = "bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB: gfx950 FETCH_SIZE counts 64 B per 128-B request on 16-B-per-lane streaming reads"
json.dump(summ, open(out, "w"), indent=1)
print(json.dumps(summ, indent=1))
