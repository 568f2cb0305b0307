"""Summary of `rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-pmc --no-configs --no-cpu-baseline --no-strong`
(tools/runs/r03_run6.sh) next to the JSON line the same process printed: per traversal kernel the launches that did work (a probed query is TWO
launches, one per verdict of the coherence probe; the flavor the verdict is not for leaves after ~4 us and is listed apart), whose average must
agree with the HIP-event averages inside bench.py (detail.kernel_ms).
    python tools/bench_profile_summary_r03.py gpurun_out/r03_6/kt gpurun_out/r03_6/kt_bench.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d, jpath = sys.argv[1], sys.argv[2]
try:
    j = json.load(open(jpath))                       # round 6: the run's full record (bench.py --detail-out), indented JSON
except ValueError:
    j = json.loads([l for l in open(jpath).read().strip().split("\n") if l.startswith("{")][-1])   # rounds 3-5: the one-line record
print("bench line under rocprofv3: value %.1f MRays/s, ms_per_step %.3f, HIP-event kernel ms %s" % (j["value"], j["ms_per_step"], {k: round(v, 3) for k, v in j["detail"]["kernel_ms"].items()}))
tr = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    tr += list(csv.DictReader(open(f)))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
by = defaultdict(list)
for r in tr:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void tbvh::", "").split("(")[0]
    by[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("%-64s %6s %10s %10s   %s" % ("kernel", "calls", "avg ms", "total ms", "(launches that left at once: < 0.02 ms)"))
tot = sum(sum(v) for v in by.values())
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    real = [x for x in v if x >= 0.02]
    empty = [x for x in v if x < 0.02]
    if sum(v) / tot < 0.002 and "k_cwbvh" not in n:
        continue
    if "k_cwbvh" in n:
        print("%-64s %6d %10.4f %10.3f   %d empty, avg %.4f ms" % (n[:64], len(real), sum(real) / max(len(real), 1), sum(v), len(empty), sum(empty) / max(len(empty), 1)))
    else:
        print("%-64s %6d %10.4f %10.3f" % (n[:64], len(v), sum(v) / len(v), sum(v)))
# the timed steps: 16.7 M-ray launches of the two flavors, in dispatch order
def launches(prefix):   # (the template argument list grew a trailing TRI2 in round 5: match by prefix)
    return [x for n_, v_ in by.items() if n_.startswith(prefix) for x in v_ if x >= 0.02]


prim = launches("k_cwbvh<false, 8, 16, 8, true, false, 0, 5, 3, 0, 8")
pk = launches("k_cwbvh_packet<false, false>")
if len(pk) > len(prim):   # the scene's tuner settled on one traversal per wave for camera rays (round 5)
    print("camera rays: the tuner settled on k_cwbvh_packet (%d launches; %d of the deferred flavor while it measured)" % (len(pk), len(prim)))
    prim = pk
diff = launches("k_cwbvh<false, 8, 16, 1, false, false, 0, 13, 2, 0, 8")
print("16.7 M-ray launches, coherent flavor   (camera rays; the last %d are the warm-up + timed steps): %s" % (j["steps"] + j["warmup"], [round(x, 3) for x in prim]))
print("16.7 M-ray launches, incoherent flavor (bounce rays):                                            %s" % [round(x, 3) for x in diff])
k = j["steps"]
if len(prim) >= k and len(diff) >= k:
    # the timed steps are followed by the 64 M-ray batch / wavefront frames, so take the k launches of ~the step's size that the bench timed: by value
    ps = sorted(prim, key=lambda x: abs(x - j["detail"]["kernel_ms"]["primary"]))[:k]
    ds = sorted(diff, key=lambda x: abs(x - j["detail"]["kernel_ms"]["diffuse"]))[:k]
    print("rocprofv3 average of the %d timed launches: camera %.4f ms (HIP events %.4f), bounce %.4f ms (HIP events %.4f)" % (k, sum(ps) / k, j["detail"]["kernel_ms"]["primary"], sum(ds) / k, j["detail"]["kernel_ms"]["diffuse"]))
