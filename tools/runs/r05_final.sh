#!/bin/bash
# Round 5, end: the full GPU suite, smoke, the driver-shaped bench line (timed), the same through torch.distributed.run (the N > 1 code path on one GPU), the
# one-process 2-context path, and rocprofv3 --kernel-trace --stats of the bench command (short form).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r05_final
mkdir -p $O
export TMPDIR=/tmp
HERE=$PWD
( timeout 600 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log ); tail -2 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
SECONDS=0
( timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ); echo "bench rc $? in $SECONDS s" >> $O/bench.err; tail -4 $O/bench.err
( TBVH_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-pmc --no-configs --no-cpu-baseline --no-hbm-regime > $O/bench_dist.json 2> $O/bench_dist.err ); echo "dist rc $?" >> $O/bench_dist.err; tail -2 $O/bench_dist.err
( TBVH_BENCH_DEVICE_MAP=0,0 timeout 400 python bench.py --gpus 2 --steps 5 --warmup 2 --no-pmc --no-configs --no-hbm-regime --no-rotated --no-other-layouts --no-host-rays --no-cpu-baseline > $O/bench_2ctx.json 2> $O/bench_2ctx.err; echo "2ctx rc $?" >> $O/bench_2ctx.err ); tail -2 $O/bench_2ctx.err
cd /tmp
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt -o kt -- python $HERE/bench.py --steps 5 --warmup 2 --no-pmc --no-configs --no-cpu-baseline --no-strong --no-hbm-regime --no-rotated --no-other-layouts --no-host-rays > $O/kt_bench.json 2> $O/kt.log
cd $HERE
python tools/bench_profile_summary_r03.py $O/kt $O/kt_bench.json > $O/bench_profile.txt 2>&1; head -8 $O/bench_profile.txt; tail -4 $O/bench_profile.txt
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
rm -rf $O/kt
python - <<'PY'
import json
for f in ("bench.json", "bench_dist.json", "bench_2ctx.json"):
    try:
        j = json.loads([l for l in open("gpurun_out/r05_final/" + f).read().split("\n") if l.startswith("{")][-1])
    except Exception as e:
        print(f, "no line", e); continue
    print(f, {k: j.get(k) for k in ("metric", "value", "unit", "ms_per_step", "n_gpus", "parity_checked", "parity_ok")})
    d = j["detail"]
    print({k: d.get(k) for k in ("primary_mrays", "diffuse_mrays", "shadow_mrays", "kernel_ms", "dispatch_gap_ms")})
    if j.get("roofline"):
        r = j["roofline"]; print("roofline", {k: r[k] for k in ("bound", "achieved", "peak", "frac", "frac_of_measured_read", "traffic")}, "valu", {k: v for k, v in (r.get("valu") or {}).items() if k != "source"})
PY
