#!/bin/bash
# Round 4, first GPU call: counters available on this box, the GPU suite after the capi split + in-kernel probe + timing ring, the driver-shaped bench line.
set -u
O=$PWD/gpurun_out/r04_run1
mkdir -p $O
export TMPDIR=/tmp
HERE=$PWD
( cd /tmp && timeout 120 rocprofv3 -L > $O/counters_list.txt 2>&1 ); grep -i -c "name" $O/counters_list.txt
grep -i -E "mall|dram|hbm|EA0_RDREQ|EA_RDREQ|TCC_EA|TCC_MISS|TCC_HIT|TCC_REQ" $O/counters_list.txt | cut -c1-200 | sort -u | head -60 > $O/counters_mem.txt; wc -l $O/counters_mem.txt
( timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider -s > $O/pytest.txt 2>&1 ); echo "rc $?" >> $O/pytest.txt; tail -6 $O/pytest.txt
grep -E "differences from the real|config 5 differences" $O/pytest.txt
( timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ); echo "bench rc $?" >> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04_run1/bench.json").read().split("\n") if l.startswith("{")][-1])
print({k: j[k] for k in ("metric", "value", "unit", "ms_per_step", "n_gpus", "parity_checked", "parity_ok")})
d = j["detail"]
print({k: d[k] for k in ("primary_mrays", "diffuse_mrays", "shadow_mrays", "kernel_ms", "dispatch_gap_ms")})
print("parity", json.dumps(d["parity_sample"]))
print("reference_blob", json.dumps(d["reference_blob"]))
PY
