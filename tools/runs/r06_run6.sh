#!/bin/bash
# Round 6, call 6: shim bit-compare in one process; the full --detail bench record (docs tables), timed.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r06_run6
mkdir -p $O
export TMPDIR=/tmp
./examples/_build/shim_bits > $O/shim_bits.log 2>&1; tail -12 $O/shim_bits.log
SECONDS=0
( timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 --detail --ceilings --detail-out $O/bench_detail.json > $O/bench.out 2> $O/bench.err ); echo "bench --detail rc $? in $SECONDS s" | tee -a $O/bench.err
tail -1 $O/bench.out | wc -c
grep "leg " $O/bench.err | tr '\n' ';'
