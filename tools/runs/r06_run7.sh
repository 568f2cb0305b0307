#!/bin/bash
# Round 6, call 7: BVH4_GPU's 8-wide copy — tests, whole suite, layouts side by side, TLAS numbers.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r06_run7
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_bvh_gpu_wide_copy.py tests/test_examples.py -m gpu -q > $O/pytest_new.log 2>&1; echo "pytest rc $?" >> $O/pytest_new.log ); tail -25 $O/pytest_new.log
( timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log ); tail -12 $O/pytest.log
for sc in sponza bistro; do
  for w in 1024 4096; do
    timeout 300 python tools/perf_probe.py --scene $sc --width $w --height $w --layouts 8 > $O/probe_${sc}_$w.log 2>&1; tail -1 $O/probe_${sc}_$w.log
    TBVH_WIDE_COPY_MIN=0 timeout 300 python tools/perf_probe.py --scene $sc --width $w --height $w --layouts 8 > $O/probe_${sc}_${w}_native.log 2>&1; tail -1 $O/probe_${sc}_${w}_native.log
  done
done
