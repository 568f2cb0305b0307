#!/bin/bash
# usage: tools/prof_mem.sh <tag> [perf_probe args...]  — cache / memory-pipe counters (two PMC passes)
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/tools/perf_probe.py $*"
cd /tmp
for pass in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
            "SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" ; do
  n=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 180 rocprofv3 --output-format csv --pmc $pass --kernel-trace -d $OUT/pmc_$n -o pmc -- $CMD > $OUT/pmc_$n.log 2>&1
done
cd - > /dev/null
