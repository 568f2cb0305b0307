#!/bin/bash
# Round 6, call 13: the whole GPU suite and the driver-shaped bench line on the tree as committed.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r06_run13
mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log ); tail -6 $O/pytest.log
SECONDS=0
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail-out $O/bench_detail.json > $O/bench.out 2> $O/bench.err ); echo "bench rc $? in $SECONDS s" | tee -a $O/bench.err
tail -1 $O/bench.out > $O/bench_line.json; wc -c $O/bench_line.json; python -c "
import json; d=json.load(open('$O/bench_line.json')); print(d['value'], d['kernel_mrays'], d['reference_blob'], d.get('config2'), d.get('config5'), d['parity'])"
