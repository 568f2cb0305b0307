#!/bin/bash
# Round 4: the GPU suite (per-test timeout) after the boundary work, then the driver-shaped bench line with the counters-based roofline.
set -u
O=$PWD/gpurun_out/r04_run4
mkdir -p $O
export TMPDIR=/tmp
bash tools/r04_suite.sh r04_run4 -x
grep -E "passed|failed|Timeout|rc " $O/pytest.txt | tail -5
( timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ); echo "bench rc $?" >> $O/bench.err; tail -12 $O/bench.err
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04_run4/bench.json").read().split("\n") if l.startswith("{")][-1])
print({k: j[k] for k in ("metric", "value", "unit", "ms_per_step", "n_gpus", "parity_checked", "parity_ok")})
d = j["detail"]
print({k: d[k] for k in ("primary_mrays", "diffuse_mrays", "shadow_mrays", "kernel_ms", "dispatch_gap_ms")})
r = j["roofline"]
print("roofline", {k: r[k] for k in ("bound", "achieved", "peak", "frac", "frac_of_measured_read", "traffic")})
print("valu", r["valu"]); print("latency", r["l2_miss_latency"]); print("primary", json.dumps(r["primary"])[:900])
print("hbm_regime", json.dumps(d.get("hbm_regime"))[:2500])
print("ref_opencl", json.dumps(d.get("ref_opencl_cwbvh")))
print("tlas", json.dumps(d.get("tlas_1000_instances")))
print("cpu", json.dumps(j["cpu_baseline"])[:300])
PY
