#!/bin/bash
set -u
O=$PWD/gpurun_out/r04_run8
mkdir -p $O
export TMPDIR=/tmp
bash tools/r04_suite.sh r04_run8
( timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ); echo "bench rc $?" >> $O/bench.err; tail -6 $O/bench.err
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04_run8/bench.json").read().split("\n") if l.startswith("{")][-1])
print({k: j[k] for k in ("metric", "value", "unit", "ms_per_step", "n_gpus", "parity_checked", "parity_ok")})
d = j["detail"]
print({k: d[k] for k in ("primary_mrays", "diffuse_mrays", "shadow_mrays", "kernel_ms", "dispatch_gap_ms", "coherent_schedule")})
PY
