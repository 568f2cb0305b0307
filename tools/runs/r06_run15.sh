#!/bin/bash
# Round 6, call 15: examples (shim example factored), the --detail bench record with the native-kernel rows.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r06_run15
mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_examples.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log ); tail -4 $O/pytest.log
SECONDS=0
( timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 --detail --ceilings --detail-out $O/bench_detail.json > $O/bench.out 2> $O/bench.err ); echo "bench --detail rc $? in $SECONDS s" | tee -a $O/bench.err
tail -1 $O/bench.out | wc -c
grep -v "leg " $O/bench.err | tail -8
python - <<PY
import json
d = json.load(open("$O/bench_detail.json"))
for k, r in d["detail"]["other_layouts"].items():
    if "error" in r: print(k, r); continue
    print(k, {kind: (round(r[kind]["mrays"]), round(r[kind].get("ref_opencl_mrays", 0)), {x: round(v, 3) for x, v in (r[kind].get("valu") or {}).items()}, round(r[kind].get("fabric", {}).get("frac", 0), 3)) for kind in ("primary", "diffuse")})
PY
