#!/bin/bash
# Round 6, call 12: the N > 1 code paths of the re-shaped bench on one GPU: torch.distributed.run with one rank (RCCL barrier / reduce, blob replication
# through the Save-format file) and one process driving two contexts on device 0.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r06_run12
mkdir -p $O
export TMPDIR=/tmp
( TBVH_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --detail-out $O/dist_detail.json > $O/bench_dist.out 2> $O/bench_dist.err ); echo "dist rc $?" | tee -a $O/bench_dist.err
tail -1 $O/bench_dist.out | cut -c1-600
( TBVH_BENCH_DEVICE_MAP=0,0 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --detail-out $O/2ctx_detail.json > $O/bench_2ctx.out 2> $O/bench_2ctx.err ); echo "2ctx rc $?" | tee -a $O/bench_2ctx.err
tail -1 $O/bench_2ctx.out | cut -c1-600
tail -3 $O/bench_2ctx.err
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
