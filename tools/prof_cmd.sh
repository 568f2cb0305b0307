#!/bin/bash
# usage: tools/prof_cmd.sh <tag> <command ...>   (GPU box, through gpurun)
# rocprofv3 kernel-trace stats of the command, then SQ / TCP / TCC counters in separate --pmc passes (with --kernel-trace
# only, as gpurun requires).  Writes gpurun_out/prof_<tag>/ and its summary.txt (tools/prof_summary.py).
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="$*"
HERE=$PWD
cd /tmp
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/kt -o kt -- $CMD > $OUT/kt.log 2>&1
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
            "FETCH_SIZE GRBM_GUI_ACTIVE" \
            "WRITE_SIZE GRBM_GUI_ACTIVE" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
            "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
            "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE" ; do
  n=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 240 rocprofv3 --output-format csv --pmc $pass --kernel-trace -d $OUT/pmc_$n -o pmc -- $CMD > $OUT/pmc_$n.log 2>&1
done
cd $HERE
python tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
