#!/bin/bash
# round 5, GPU call 3: GPU tests (all, after the split default / launch table / host pipeline), final rotated table, bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q > $O/r05_3_pytest.log 2>&1; echo "pytest rc $?" >> $O/r05_3_pytest.log )
tail -15 $O/r05_3_pytest.log
( TBVH_COHERENT_TUNER=0 timeout 600 python tools/rotated_table.py > $O/r05_rotated.txt 2> $O/r05_rotated.err; echo "rc $?" >> $O/r05_rotated.err )
cat $O/r05_rotated.txt; tail -3 $O/r05_rotated.err
( timeout 200 python tools/hostpath_probe.py > $O/r05_hostpath_after.txt 2>&1 ); cat $O/r05_hostpath_after.txt
( timeout 900 python bench.py --steps 20 --warmup 5 > $O/r05_bench_b.json 2> $O/r05_bench_b.err; echo "bench rc $?" >> $O/r05_bench_b.err )
tail -5 $O/r05_bench_b.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r05_bench_b.json") if l.startswith("{")][-1])
print(d.get("value"), d.get("n_gpus"), d.get("parity_ok"), (d.get("roofline") or {}).get("frac"))
det=d["detail"]
print({k: det[k] for k in ("primary_mrays","diffuse_mrays","shadow_mrays","dispatch_gap_ms")})
print("host_rays", json.dumps(det.get("host_rays"))[:1500])
print("rotated", {k: (det["rotated_scene"][k]["mrays"]) for k in ("primary","diffuse")} if det.get("rotated_scene") and "primary" in det["rotated_scene"] else det.get("rotated_scene"))
PY
