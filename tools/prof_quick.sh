#!/bin/bash
# usage: tools/prof_quick.sh <tag> [perf_probe args...]  — one PMC pass (instruction mix) + kernel trace
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/tools/perf_probe.py $*"
cd /tmp
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU" ; do
  n=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 180 rocprofv3 --output-format csv --pmc $pass --kernel-trace -d $OUT/pmc_$n -o pmc -- $CMD > $OUT/pmc_$n.log 2>&1
done
cd - > /dev/null
python tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
