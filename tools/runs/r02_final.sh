#!/bin/bash
# end-of-round measurements with the library AS SHIPPED (default build): GPU tests, the contract bench, its rocprofv3 profile,
# TLAS probes, the HBM-regime points of the size sweep
set -u
O=gpurun_out/r02w; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -4 $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
TBVH_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $O/bench_dist.json 2> $O/bench_dist.err; tail -2 $O/bench_dist.err; python -c "
import json
for f in ('$O/bench.json','$O/bench_dist.json'):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j['value'], j['detail']['kernel_ms'], j['detail']['config4_strong']['mrays'] if j['detail']['config4_strong'] else None)"
bash tools/bench_profile.sh r02w > $O/bench_profile.txt 2>&1; tail -25 $O/bench_profile.txt
for L in 8 10; do timeout 300 python tools/tlas_probe.py --layout $L --random 4194304 --frames 3 > $O/tlas_layout$L.log 2>&1; tail -3 $O/tlas_layout$L.log | cut -c1-300; done
timeout 900 python tools/size_sweep.py --sizes 2.8,30,60 --variants 0 > $O/size_sweep_autopad.log 2>&1; cat $O/size_sweep_autopad.log
