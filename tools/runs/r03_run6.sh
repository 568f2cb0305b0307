#!/bin/bash
set -u
mkdir -p gpurun_out/r03_6
O=$PWD/gpurun_out/r03_6
export TMPDIR=/tmp
( timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ); echo "bench rc $?" >> $O/bench.err
HERE=$PWD
cd /tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt -o kt -- python $HERE/bench.py --steps 5 --warmup 2 --no-pmc --no-configs --no-cpu-baseline --no-strong > $O/kt_bench.json 2> $O/kt.log
cd $HERE
bash tools/runs/r03_counters.sh > $O/counters.log 2>&1
tail -3 $O/bench.err
