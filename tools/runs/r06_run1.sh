#!/bin/bash
# Round 6, first call: the re-shaped bench line (driver-shaped command, timed by `SECONDS`), then the full GPU suite.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r06_run1
mkdir -p $O
export TMPDIR=/tmp
SECONDS=0
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail-out $O/bench_detail.json > $O/bench.out 2> $O/bench.err ); echo "bench rc $? in $SECONDS s" | tee -a $O/bench.err
tail -1 $O/bench.out | wc -c
tail -1 $O/bench.out
grep "leg " $O/bench.err
( timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log ); tail -3 $O/pytest.log
