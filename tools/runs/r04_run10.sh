#!/bin/bash
# counters of the two-lanes-per-ray kernel on the bench's bounce batch (TBVH_PAIR_KERNEL=1), two --pmc passes
set -u
O=$PWD/gpurun_out/r04_run10
mkdir -p $O
export TMPDIR=/tmp
HERE=$PWD
export TBVH_PAIR_KERNEL=1
cd /tmp
for pass in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"; do
  n=$(echo $pass | tr ' ' '_' | cut -c1-30)
  timeout 200 rocprofv3 --output-format csv --pmc $pass --kernel-trace -d $O/pmc_$n -o pmc -- python $HERE/tools/ab_probe.py --scene bistro --side 4096 --layout 10 --variants 0 --passes 2 > $O/pmc_$n.log 2>&1
done
cd $HERE
python tools/prof_summary.py $O > $O/summary.txt 2>&1
grep -A40 "k_cwbvh_pair<false" $O/summary.txt | grep -E "k_cwbvh_pair|SQ_|TCP_|GRBM|per dispatch" | head -60
