#!/bin/bash
# Round 6, call 2: new tests (deep tree / stack overflow, config 4 at its stated size, the reference's minimal main unmodified on the shim), bench again.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r06_run2
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_deep_tree.py tests/test_examples.py tests/test_full_size.py -m gpu -q -x -s > $O/pytest_new.log 2>&1; echo "pytest rc $?" >> $O/pytest_new.log ); tail -30 $O/pytest_new.log
SECONDS=0
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail-out $O/bench_detail.json > $O/bench.out 2> $O/bench.err ); echo "bench rc $? in $SECONDS s" | tee -a $O/bench.err
tail -1 $O/bench.out | wc -c
tail -1 $O/bench.out | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','kernel_mrays','reference_blob','config2','config5','legs_s') if k in d})"
