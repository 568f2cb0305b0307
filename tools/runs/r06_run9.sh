#!/bin/bash
# Round 6, call 9: full GPU suite (BVH4 copy with magnitude-scaled padding), bench line, rocprofv3 --kernel-trace --stats of the bench command,
# refill-threshold A/B of the incoherent flavor (experiment library).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r06_run9
mkdir -p $O
export TMPDIR=/tmp
HERE=$PWD
( timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log ); tail -6 $O/pytest.log
SECONDS=0
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail-out $O/bench_detail.json > $O/bench.out 2> $O/bench.err ); echo "bench rc $? in $SECONDS s" | tee -a $O/bench.err
tail -1 $O/bench.out > $O/bench_line.json; wc -c $O/bench_line.json
cd /tmp
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt -o kt -- python $HERE/bench.py --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-strong --no-reference-blob --detail-out $O/kt_detail.json > $O/kt_bench.out 2> $O/kt.log
cd $HERE
python tools/bench_profile_summary_r03.py $O/kt $O/kt_detail.json > $O/bench_profile.txt 2>&1; head -12 $O/bench_profile.txt
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
rm -rf $O/kt
TBVH_LIB_OVERRIDE=$PWD/tinybvh_amd/libtinybvh_amd_exp.so timeout 600 python tools/ab_configs.py --side 4096 --rounds 5 --check base=keep:0:0 refill8=keep:131072:0 refill32=keep:262144:0 w24=keep:6144:0 w32=keep:8192:0 > $O/ab_refill.txt 2>&1; cat $O/ab_refill.txt | tail -12
