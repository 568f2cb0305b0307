#!/bin/bash
# Round 3: the scene-size sweep beyond the Infinity Cache with S/T counts next to the counter traffic (review item 2).
# Two runs of the same command: timed (+ S/T from the instrumented kernel), then under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE.
set -u
O=$PWD/gpurun_out/r03_sweep
mkdir -p $O
export TMPDIR=/tmp
HERE=$PWD
CMD="python $HERE/tools/size_sweep.py --sizes 12,30,60 --variants 0 --st --order $O/order.json"
( timeout 900 $CMD > $O/times.txt 2>&1 ); cat $O/times.txt
cd /tmp
timeout 900 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o pmc -- $CMD > $O/pmc_fetch.log 2>&1
cd $HERE
python tools/size_sweep_pmc.py $O/pmc_fetch $O/order.json $((2048*2048)) > $O/fetch.txt 2>&1; cat $O/fetch.txt
