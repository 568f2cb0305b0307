#!/bin/bash
# Round 6, end: the full GPU suite, smoke, the driver-shaped bench line (timed), the --detail record, the same line through torch.distributed.run (one rank)
# and from one process over two contexts, rocprofv3 --kernel-trace --stats of the bench command.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r06_final
mkdir -p $O
export TMPDIR=/tmp
HERE=$PWD
( timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log ); tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
SECONDS=0
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail-out $O/bench_default_detail.json > $O/bench.out 2> $O/bench.err ); echo "bench rc $? in $SECONDS s" | tee -a $O/bench.err
tail -1 $O/bench.out > $O/bench_line.json; wc -c < $O/bench_line.json
SECONDS=0
( timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 --detail --ceilings --detail-out $O/bench_detail.json > $O/bench_detail.out 2> $O/bench_detail.err ); echo "bench --detail rc $? in $SECONDS s" | tee -a $O/bench_detail.err
( TBVH_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --detail-out $O/dist_detail.json > $O/bench_dist.out 2> $O/bench_dist.err ); echo "dist rc $?"; tail -1 $O/bench_dist.out > $O/bench_dist_line.json
( TBVH_BENCH_DEVICE_MAP=0,0 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --detail-out $O/2ctx_detail.json > $O/bench_2ctx.out 2> $O/bench_2ctx.err ); echo "2ctx rc $?"; tail -1 $O/bench_2ctx.out > $O/bench_2ctx_line.json
cd /tmp
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt -o kt -- python $HERE/bench.py --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-strong --no-reference-blob --detail-out $O/kt_detail.json > $O/kt_bench.out 2> $O/kt.log
cd $HERE
python tools/bench_profile_summary_r03.py $O/kt $O/kt_detail.json > $O/bench_profile.txt 2>&1; tail -4 $O/bench_profile.txt
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
rm -rf $O/kt
python - <<PY
import json
for f in ("bench_line.json", "bench_dist_line.json", "bench_2ctx_line.json"):
    d = json.load(open("$O/" + f))
    print(f, d["value"], d["n_gpus"], d["kernel_mrays"], d.get("reference_blob"), d.get("config2"), d.get("config5"), d["parity"])
PY
