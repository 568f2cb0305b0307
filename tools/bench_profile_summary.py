"""Summarise tools/bench_profile.sh: kernel-trace stats, per-launch HBM-side traffic of the
dominant kernel (k_cwbvh<false> on the diffuse batch = every second launch of that kernel), and
the FETCH_SIZE calibration on the gather micro-benchmark (known bytes)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]
out = {}


def rows(pattern):
    for f in glob.glob(os.path.join(d, pattern), recursive=True):
        yield from csv.DictReader(open(f))


print("== kernel stats (bench.py --steps 3 --warmup 1)")
for r in rows("kt/**/*kernel_stats.csv"):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void tbvh::", "").split("(")[0]
    print(f"  {n:64s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e6:9.4f} ms  {r['Percentage']} %")
# per-launch durations of the intersect kernel in dispatch order
tr = [r for r in rows("kt/**/*kernel_trace.csv") if "k_cwbvh<false" in r["Kernel_Name"]]
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in tr]
print("  k_cwbvh<false> launches (ms, dispatch order):", [round(x, 3) for x in dur])
# bench order: 3 launches while building the batches, then (primary, diffuse) per step for the 1 warm-up +
# 3 timed steps of tools/bench_profile.sh; whatever follows belongs to the shadow / wavefront extras
N_PAIRS = 4
steps = dur[3:3 + 2 * N_PAIRS]
if len(steps) >= 2:
    out["rocprof_primary_ms"] = sum(steps[0::2]) / len(steps[0::2])
    out["rocprof_diffuse_ms"] = sum(steps[1::2]) / len(steps[1::2])
for name, key in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    v = [r for r in rows(name + "/**/*counter_collection.csv") if "k_cwbvh<false" in r["Kernel_Name"] and r["Counter_Name"] == key]
    v.sort(key=lambda r: int(r["Start_Timestamp"]))
    vals = [float(r["Counter_Value"]) for r in v][3:3 + 2 * N_PAIRS]
    if len(vals) >= 2:
        out[key + "_primary_KB"] = sum(vals[0::2]) / len(vals[0::2])
        out[key + "_diffuse_KB"] = sum(vals[1::2]) / len(vals[1::2])
# calibration: gather_rate runs k<NLOADS,STRIDE>; bytes actually requested = blocks*64*iters*NLOADS*16, distinct lines known
cal = defaultdict(list)
for r in rows("calib/**/*counter_collection.csv"):
    if r["Counter_Name"] == "FETCH_SIZE":
        cal[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), float(r["Counter_Value"]), int(r["Grid_Size"])))
print("== FETCH_SIZE calibration (gather_rate: every launch reads grid*4000*NLOADS*16 bytes; the last launches use the 512 MB table = all misses)")
for k, v in cal.items():
    v.sort()
    ts, val, grid = v[-1]  # last launch of this instantiation: 512 MB table, 24 waves/CU
    import re
    m = re.search(r"k<(\d+), (\d+)>", k)
    if not m:
        continue
    nl, stride = int(m.group(1)), int(m.group(2))
    req = grid * 4000 * nl * 16
    lines = grid * 4000 * (1 if stride == 128 else (stride + 127) // 128 + (0.5 if stride % 128 else 0)) * 128
    print(f"  k<{nl},{stride}>: FETCH_SIZE {val:14.1f} KB = {val*1024/1e9:8.2f} GB ; requested {req/1e9:8.2f} GB ; ~distinct-line bytes {lines/1e9:8.2f} GB ; ratio FETCH/requested {val*1024/req:5.2f}")
    if nl == 8 and stride == 128:
        out["calib_fetch_over_true_128B_records"] = val * 1024 / req
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)
