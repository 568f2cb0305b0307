#!/bin/bash
# GPU run 1 of round 2: parity of the new schedules, lane-count gather microbenchmark, A/B of the CWBVH schedule variants
# on the contract bench's batches, lane statistics, SQ counters of the strict and the deferred schedule.
set -u
O=gpurun_out/r02a; mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ) 
tail -3 $O/pytest.log
timeout 200 tools/ubench/gather_lanes > $O/gather_lanes.txt 2>&1
timeout 900 python tools/ab_probe.py --variants 0,45,51,52,53,54,55,56,57,58,62,63 --stats 59,60,61 --out $O/ab_bistro.json > $O/ab_bistro.log 2>&1
cat $O/ab_bistro.log
timeout 300 python tools/ab_probe.py --scene sponza --side 1024 --variants 0,45,51,53,40 --stats 59,60 > $O/ab_sponza.log 2>&1
cat $O/ab_sponza.log
tools/prof_cmd.sh r02a_v0 python $PWD/tools/ab_probe.py --variants 0 --passes 1 > $O/prof_v0.log 2>&1
tools/prof_cmd.sh r02a_v53 python $PWD/tools/ab_probe.py --variants 53 --passes 1 > $O/prof_v53.log 2>&1
tail -5 $O/prof_v0.log
