#!/bin/bash
# A 1280 x 720 path-traced frame kernel by kernel: rocprofv3 --kernel-trace of tools/wavefront_small_frame.py, then per steady-state frame the
# wall time, the time inside kernels, the launches by name and the gaps between them (how DESIGN.md par. 10 found the 20 fill / copy launches).
set -u
O=$PWD/gpurun_out/frame_trace
mkdir -p $O
export TMPDIR=/tmp
HERE=$PWD
cd /tmp
timeout 300 rocprofv3 --output-format csv --kernel-trace --memory-copy-trace -d $O/kt -o kt -- python $HERE/tools/wavefront_small_frame.py > $O/run.txt 2>&1
cd $HERE
tail -3 $O/run.txt
python - <<'PY'
import csv, glob, os
from collections import defaultdict
d = "gpurun_out/frame_trace/kt"
tr = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    tr += list(csv.DictReader(open(f)))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
print(len(tr), "kernel launches")
# take the last 40 % of launches (steady state) and find the frame period by the generate kernel
gen = [i for i, r in enumerate(tr) if "k_wf_generate" in r["Kernel_Name"]]
if len(gen) > 12:
    a, b = gen[-11], gen[-1]
    frames = 10
    seg = tr[a:b]
    wall = (int(tr[b]["Start_Timestamp"]) - int(tr[a]["Start_Timestamp"])) / 1e3 / frames
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e3 / frames
    print(f"steady state: {wall:.1f} us per frame wall, {busy:.1f} us in kernels, {len(seg) / frames:.1f} launches per frame")
    by = defaultdict(lambda: [0, 0.0])
    for r in seg:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void tbvh::", "").replace("tbvh::", "").split("(")[0][:70]
        by[n][0] += 1; by[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for n, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print(f"  {n:70s} {c / frames:5.1f} x {t / c:7.1f} us = {t / frames:7.1f} us per frame")
    gaps = [(int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"])) / 1e3 for i in range(len(seg) - 1)]
    print(f"  gaps between consecutive kernels: mean {sum(gaps) / len(gaps):.2f} us, total {sum(gaps) / frames:.1f} us per frame, max {max(gaps):.1f}")
PY
