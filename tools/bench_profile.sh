#!/bin/bash
# usage: tools/bench_profile.sh <tag>   (GPU box)  — rocprofv3 of the SAME command as the contract bench:
# kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in separate PMC passes, then a calibration of
# FETCH_SIZE on a known byte count in this path's access pattern (divergent 16-byte loads).
set -u
TAG=$1
OUT=$PWD/gpurun_out/bench_prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-strong --no-rotated --no-other-layouts --no-host-rays --no-hbm-regime --no-configs"
[ -x tools/ubench/gather_rate ] || make -C tools/ubench gather_rate > /dev/null
cd /tmp
timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/kt -o kt -- $CMD > $OUT/kt.log 2>&1
timeout 400 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1
timeout 200 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OUT/calib -o pmc -- $OLDPWD/tools/ubench/gather_rate > $OUT/calib.log 2>&1
cd - > /dev/null
python tools/bench_profile_summary.py $OUT
