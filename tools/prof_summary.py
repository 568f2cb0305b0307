"""Summarise a tools/prof.sh output directory: per-kernel average duration from the kernel
trace and per-kernel PMC counter means (per dispatch) from the counter-collection CSVs."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name.split("(")[0].replace("void tbvh::", "")[:60]


for f in glob.glob(os.path.join(d, "kt", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats", os.path.relpath(f, d))
    for r in csv.DictReader(open(f)):
        print(f"  {short(r['Name']):60s} calls {r['Calls']:>6s} avg_ns {float(r['AverageNs']):14.0f} total% {r['Percentage']}")
for f in sorted(glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    agg = defaultdict(lambda: defaultdict(list))
    meta = {}
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"))
    print("== pmc", os.path.relpath(f, d))
    for k, cs in agg.items():
        if "k_cwbvh" not in k and "k_bvh" not in k and "k_tlas" not in k:
            continue
        print(f"  {k}  vgpr/agpr/sgpr/lds/scratch={meta[k]}")
        for c, v in cs.items():
            print(f"      {c:36s} n={len(v):3d} mean={sum(v)/len(v):16.1f} min={min(v):16.1f} max={max(v):16.1f}")
            if len(v) <= 16:   # per dispatch, in launch order (tools/ab_probe.py: 3 preparation launches, then primary / diffuse passes)
                print("          per dispatch: " + " ".join(f"{x:.4g}" for x in v))
