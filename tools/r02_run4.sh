#!/bin/bash
set -u
O=gpurun_out/r02d; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_cwbvh_schedules.py tests/test_sharded.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -3 $O/pytest.log
timeout 900 python tools/ab_probe.py --variants 0,51,52,68,69,70,71,0 --passes 6 > $O/ab_bistro.log 2>&1; cat $O/ab_bistro.log
timeout 300 python tools/ab_probe.py --scene sponza --side 1024 --variants 0,51,52,68,69,71,0 --passes 6 > $O/ab_sponza.log 2>&1; cat $O/ab_sponza.log
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; tail -5 $O/bench.err; cat $O/bench.json
