#!/bin/bash
# round 5, GPU call 1: GPU tests, the rotated-scene table, the TRI2 A/B, the bench with its new legs, the one-process 2-context path
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests -m gpu -x -q > $O/r05_1_pytest.log 2>&1; echo "pytest rc $?" >> $O/r05_1_pytest.log ) 
tail -5 $O/r05_1_pytest.log
( TBVH_COHERENT_TUNER=0 timeout 600 python tools/rotated_table.py > $O/r05_rotated.txt 2> $O/r05_rotated.err; echo "rc $?" >> $O/r05_rotated.err )
cat $O/r05_rotated.txt; tail -3 $O/r05_rotated.err
EXP=$PWD/tinybvh_amd/libtinybvh_amd_exp.so
( TBVH_LIB_OVERRIDE=$EXP TBVH_COHERENT_TUNER=0 timeout 400 python tools/ab_configs.py --rounds 7 --check base=keep:0:0 tri2=keep:65536:0 > $O/r05_ab_tri2.txt 2>&1 )
( TBVH_LIB_OVERRIDE=$EXP TBVH_COHERENT_TUNER=0 timeout 400 python tools/ab_configs.py --scene street_rot --rounds 5 base=keep:0:0 tri2=keep:65536:0 >> $O/r05_ab_tri2.txt 2>&1 )
cat $O/r05_ab_tri2.txt
( timeout 900 python bench.py --steps 20 --warmup 5 > $O/r05_bench_a.json 2> $O/r05_bench_a.err; echo "bench rc $?" >> $O/r05_bench_a.err )
tail -3 $O/r05_bench_a.err
( TBVH_BENCH_DEVICE_MAP=0,0 timeout 400 python bench.py --gpus 2 --steps 5 --warmup 2 --no-pmc --no-configs --no-hbm-regime --no-rotated --no-other-layouts --no-host-rays --no-cpu-baseline > $O/r05_bench_2ctx.json 2> $O/r05_bench_2ctx.err; echo "rc $?" >> $O/r05_bench_2ctx.err )
tail -2 $O/r05_bench_2ctx.err
python - <<'PY'
import json
for f in ("gpurun_out/r05_bench_a.json","gpurun_out/r05_bench_2ctx.json"):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, d.get("value"), d.get("n_gpus"), d.get("parity_ok"), (d.get("roofline") or {}).get("frac"))
    except Exception as e: print(f, "unreadable", e)
PY
