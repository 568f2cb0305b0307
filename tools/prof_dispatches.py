"""Per-dispatch PMC table for one kernel-name substring: rows in dispatch order, one column per counter.
usage: prof_dispatches.py <prof dir> <kernel substring>"""
import csv
import glob
import os
import sys
from collections import defaultdict

d, sub = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    rows = defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    if not rows:
        continue
    names = sorted({c for v in rows.values() for c in v})
    print("==", os.path.relpath(f, d))
    print("  disp " + " ".join(f"{n[-18:]:>18s}" for n in names))
    for k in sorted(rows):
        print(f"  {k:4d} " + " ".join(f"{rows[k].get(n, 0):18.0f}" for n in names))
