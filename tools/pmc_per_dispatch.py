#!/usr/bin/env python3
"""Per-dispatch counter values of one kernel from a rocprofv3 --pmc run (csv): pmc_per_dispatch.py <dir> <kernel substring>"""
import collections
import csv
import glob
import os
import sys

d, sub = sys.argv[1], sys.argv[2]
rows = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if sub not in r["Kernel_Name"]:
            continue
        e = rows.setdefault((f, r["Dispatch_Id"]), {"grid": int(r.get("Grid_Size", 0) or 0), "wg": int(r.get("Workgroup_Size", 1) or 1),
                                                     "t0": int(r.get("Start_Timestamp", 0) or 0), "t1": int(r.get("End_Timestamp", 0) or 0)})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
names = sorted({k for e in rows.values() for k in e if k not in ("grid", "wg", "t0", "t1")})
print("# workgroups lanes dur_us " + " ".join(names))
for e in rows.values():
    print("%8d %5d %8.1f " % (e["grid"] // max(1, e["wg"]), e["wg"], (e["t1"] - e["t0"]) / 1e3) + " ".join("%.4g" % e.get(n, 0) for n in names))
