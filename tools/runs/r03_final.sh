#!/bin/bash
# Round 3, end: the full GPU suite, the driver-shaped bench line, and rocprofv3 --kernel-trace --stats of the same bench command.
set -u
O=$PWD/gpurun_out/r03_final
mkdir -p $O
export TMPDIR=/tmp
HERE=$PWD
( timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.txt 2>&1 ); echo "rc $?" >> $O/pytest.txt; tail -4 $O/pytest.txt
( timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1 ); tail -2 $O/smoke.txt
( timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err ); echo "bench rc $?" >> $O/bench.err; tail -2 $O/bench.err
cd /tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt -o kt -- python $HERE/bench.py --steps 5 --warmup 2 --no-pmc --no-configs --no-cpu-baseline --no-strong > $O/kt_bench.json 2> $O/kt.log
cd $HERE
python tools/bench_profile_summary_r03.py $O/kt $O/kt_bench.json > $O/bench_profile.txt 2>&1; head -30 $O/bench_profile.txt
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r03_final/bench.json").read().split("\n") if l.startswith("{")][-1])
print({k: j[k] for k in ("metric", "value", "unit", "ms_per_step", "n_gpus")})
print("roofline", {k: j["roofline"][k] for k in ("bound", "achieved", "peak", "frac", "traffic") if k in j["roofline"]})
print("cpu_baseline", j["cpu_baseline"])
d = j["detail"]
print({k: d[k] for k in ("primary_mrays", "diffuse_mrays", "shadow_mrays", "parity_sample")})
PY
