#!/bin/bash
# Round 6: split references reach a pad's width across their cut faces (host_builder.cpp) -> full GPU suite, the two randomised hunts, the bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r06_run19
mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log ); tail -3 $O/pytest.log
( TBVH_RANDOM_LARGE_SEEDS=300 timeout 1500 python -m pytest tests/test_random_large.py -m gpu -q > $O/hunt_large.log 2>&1 ); grep -n "^E   *AssertionError\|passed\|failed" $O/hunt_large.log | cut -c1-400
( TBVH_RANDOM_SEEDS=1500 timeout 900 python -m pytest tests/test_random_configs.py -m gpu -q > $O/hunt_small.log 2>&1 ); tail -1 $O/hunt_small.log
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail-out $O/bench_detail.json > $O/bench.out 2> $O/bench.err ); tail -1 $O/bench.out > $O/bench_line.json
python - <<PY
import json
d = json.load(open("$O/bench_line.json"))
print(d["value"], d["kernel_mrays"], d.get("config2"), d.get("config5"), d["parity"], d["roofline"]["nodes_per_ray"], d["roofline"]["tris_per_ray"])
PY
