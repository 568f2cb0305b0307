#!/bin/bash
# a 4096 x 4096 three-bounce frame of the Bistro stand-in kernel by kernel (rocprofv3 --kernel-trace of tools/wavefront_probe.py: 4 frames)
set -u
O=$PWD/gpurun_out/r04_frame_big
mkdir -p $O
export TMPDIR=/tmp
HERE=$PWD
cd /tmp
timeout 300 rocprofv3 --output-format csv --kernel-trace -d $O/kt -o kt -- python $HERE/tools/wavefront_probe.py bistro 4096 > $O/run.txt 2>&1
cd $HERE
grep "^frame" $O/run.txt
python - <<'PY'
import csv, glob, os
d = "gpurun_out/r04_frame_big/kt"
tr = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    tr += list(csv.DictReader(open(f)))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
gen = [i for i, r in enumerate(tr) if "k_wf_generate" in r["Kernel_Name"]]
a = gen[-1]
seg = tr[a:]
t0 = int(seg[0]["Start_Timestamp"])
prev_end = None
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void tbvh::", "").split("(")[0][:70]
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t0) / 1e6:8.3f} ms  +{(e - s) / 1e6:7.3f} ms  gap {gap:6.1f} us  {n}")
    prev_end = e
print(f"frame: {(int(seg[-1]['End_Timestamp']) - t0) / 1e6:.3f} ms, in kernels {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e6:.3f} ms")
PY
