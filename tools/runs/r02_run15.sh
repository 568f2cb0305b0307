#!/bin/bash
# after restricting the coherence probe to scenes <= 384 MB: size sweep, GPU tests, bench
set -u
O=gpurun_out/r02y; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/size_sweep.py --sizes 2.8,30,60 --variants 0 > $O/size_sweep_autopad.log 2>&1; cat $O/size_sweep_autopad.log
( timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -4 $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; tail -c 600 $O/bench.json
