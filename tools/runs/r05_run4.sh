#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 300 python tools/split_halves_probe.py > $O/r05_split_halves.txt 2>&1 ); cat $O/r05_split_halves.txt
( timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-pmc --no-hbm-regime --no-configs --no-cpu-baseline > $O/r05_bench_torchrun_1gpu.json 2> $O/r05_bench_torchrun_1gpu.err; echo "torchrun rc $?" )
tail -2 $O/r05_bench_torchrun_1gpu.err; python -c "
import json; d=json.loads([l for l in open('gpurun_out/r05_bench_torchrun_1gpu.json') if l.startswith('{')][-1]); print(d['value'], d['n_gpus'], d['parity_ok'], d['detail']['per_gpu'])"
for k in 1 2; do timeout 600 python -m pytest tests -m gpu -x -q > $O/r05_7_pytest_$k.log 2>&1; echo "suite run $k rc $?"; tail -1 $O/r05_7_pytest_$k.log; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
