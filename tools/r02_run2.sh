#!/bin/bash
set -u
O=gpurun_out/r02b; mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -3 $O/pytest.log
timeout 900 python tools/ab_probe.py --variants 0,51,64,65,66,67 --out $O/ab_bistro.json > $O/ab_bistro.log 2>&1
cat $O/ab_bistro.log
tools/prof_cmd.sh r02b_v64 python $PWD/tools/ab_probe.py --variants 64 --passes 1 > $O/prof_v64.log 2>&1
grep -A3 "TCC_MISS_sum\|FETCH_SIZE\|TCP_TCC_READ_REQ\|TCP_TOTAL_CACHE" gpurun_out/prof_r02b_v64/summary.txt | grep -v "k_cwbvh<true" | head -40
