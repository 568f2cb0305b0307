#!/bin/bash
# Round 6, call 4: BVH_GPU's 8-wide copy — its tests, the whole GPU suite, the layouts side by side again, the bench line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r06_run4
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_bvh_gpu_wide_copy.py tests/test_examples.py -m gpu -q > $O/pytest_new.log 2>&1; echo "pytest rc $?" >> $O/pytest_new.log ); tail -25 $O/pytest_new.log
( timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log ); tail -12 $O/pytest.log
for sc in sponza bistro; do
  timeout 300 python tools/perf_probe.py --scene $sc --width 1024 --height 1024 --layouts 5 > $O/probe_${sc}_1024.log 2>&1; tail -2 $O/probe_${sc}_1024.log
  timeout 300 python tools/perf_probe.py --scene $sc --width 4096 --height 4096 --layouts 5 > $O/probe_${sc}_4096.log 2>&1; tail -2 $O/probe_${sc}_4096.log
done
SECONDS=0
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail-out $O/bench_detail.json > $O/bench.out 2> $O/bench.err ); echo "bench rc $? in $SECONDS s" | tee -a $O/bench.err
tail -1 $O/bench.out | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','kernel_mrays','config2','config5') if k in d})"
