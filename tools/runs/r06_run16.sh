#!/bin/bash
# Round 6, call 16: small scenes probed for the packet kernel: schedule tests, Sponza / Dragon stand-ins before (TBVH debug flag 64 = no probe) and after,
# config 2 through the bench.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r06_run16
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_cwbvh_schedules.py tests/test_bench_kernels.py tests/test_split_rays.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log ); tail -8 $O/pytest.log
for sc in sponza dragon; do for w in 1024 2048 4096; do for lay in 10 5; do
  timeout 300 python tools/perf_probe.py --scene $sc --layouts $lay --width $w --height $w --passes 14 2>&1 | tail -1 | sed "s/^/$sc $w: /" | tee -a $O/probe.txt
done; done; done
python bench.py --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-reference-blob --no-strong 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d.get('config2'))"
