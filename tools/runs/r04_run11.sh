#!/bin/bash
# counters of the incoherent flavor with the first 16 nodes in every wave's LDS (debug flags 0x10000) next to the shipped kernel, bounce batch
set -u
O=$PWD/gpurun_out/r04_run11
mkdir -p $O
export TMPDIR=/tmp
HERE=$PWD
cd /tmp
for fl in 0 65536; do
  for pass in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
    n=f${fl}_$(echo $pass | tr ' ' '_' | cut -c1-24)
    timeout 200 rocprofv3 --output-format csv --pmc $pass --kernel-trace -d $O/pmc_$n -o pmc -- python $HERE/tools/ab_probe.py --scene bistro --side 4096 --layout 10 --variants 0 --passes 2 --flags $fl > $O/pmc_$n.log 2>&1
  done
done
cd $HERE
python - $O <<'PY'
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
for f in sorted(glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    agg = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if ", 13, 2, " not in k: continue
        agg[(k.split("k_cwbvh")[1][:60], r.get("LDS_Block_Size"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", os.path.relpath(f, d))
    for k, cs in agg.items():
        print("  ", k)
        for c, v in cs.items():
            big = [x for x in v if x > 0.2 * max(v)]   # the launches that traced a batch (the other flavor's verdict: a few us)
            print(f"      {c:32s} n={len(big):2d} mean={sum(big) / max(len(big), 1):16.1f}")
PY
