#!/bin/bash
# Round 6, call 3: deep-tree tests (caterpillar), examples incl. the shim, config 4 at its stated size; layouts side by side on the smaller scenes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r06_run3
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_deep_tree.py tests/test_examples.py -m gpu -q -s > $O/pytest_new.log 2>&1; echo "pytest rc $?" >> $O/pytest_new.log ); tail -15 $O/pytest_new.log
( timeout 900 python -m pytest tests/test_full_size.py -m gpu -q -s -k config4 > $O/pytest_c4.log 2>&1; echo "pytest rc $?" >> $O/pytest_c4.log ); tail -8 $O/pytest_c4.log
for sc in sponza dragon; do
  for w in 1024 4096; do
    timeout 300 python tools/perf_probe.py --scene $sc --width $w --height $w > $O/probe_${sc}_$w.log 2>&1; tail -5 $O/probe_${sc}_$w.log
  done
done
