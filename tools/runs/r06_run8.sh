#!/bin/bash
# Round 6, call 8: the two-level wave-packet kernel — parity tests, A/B against the per-lane kernel; the full suite again (BVH4 copy: nested boxes).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r06_run8
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_tlas_packet.py tests/test_tlas.py tests/test_examples.py -m gpu -q -x > $O/pytest_new.log 2>&1; echo "pytest rc $?" >> $O/pytest_new.log ); tail -25 $O/pytest_new.log
for r in 1 2; do
  for p in 1 0; do
    TBVH_TLAS_PACKET=$p timeout 300 python tools/tlas_probe.py --layout 10 2>&1 | tail -1 | tee -a $O/tlas_ab.log
  done
done
timeout 300 python tools/tlas_probe.py --layout 8 2>&1 | tail -1 | tee -a $O/tlas_ab.log
( timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log ); tail -12 $O/pytest.log
