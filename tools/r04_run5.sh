#!/bin/bash
# Round 4: full GPU suite, the driver-shaped bench line, rocprofv3 --kernel-trace --stats of the same bench command (short form).
set -u
O=$PWD/gpurun_out/r04_run5
mkdir -p $O
export TMPDIR=/tmp
HERE=$PWD
bash tools/r04_suite.sh r04_run5
( timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ); echo "bench rc $?" >> $O/bench.err; tail -6 $O/bench.err
cd /tmp
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt -o kt -- python $HERE/bench.py --steps 5 --warmup 2 --no-pmc --no-configs --no-cpu-baseline --no-strong --no-hbm-regime > $O/kt_bench.json 2> $O/kt.log
cd $HERE
python tools/bench_profile_summary_r03.py $O/kt $O/kt_bench.json > $O/bench_profile.txt 2>&1; head -30 $O/bench_profile.txt
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04_run5/bench.json").read().split("\n") if l.startswith("{")][-1])
print({k: j[k] for k in ("metric", "value", "unit", "ms_per_step", "n_gpus", "parity_checked", "parity_ok")})
d = j["detail"]
print({k: d[k] for k in ("primary_mrays", "diffuse_mrays", "shadow_mrays", "kernel_ms", "dispatch_gap_ms")})
r = j["roofline"]
print("roofline", {k: r[k] for k in ("bound", "achieved", "peak", "frac", "frac_of_measured_read", "traffic")})
print("hbm_regime", json.dumps(d.get("hbm_regime"))[:3000])
print("ref_opencl", json.dumps(d.get("ref_opencl_cwbvh")))
PY
