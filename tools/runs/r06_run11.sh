#!/bin/bash
# Round 6, call 11: SQ counters of the two-rays-per-lane kernel next to the shipped incoherent flavor (same batch, same process).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r06_run11
mkdir -p $O
export TMPDIR=/tmp
HERE=$PWD
cd /tmp
for pass in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "FETCH_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  n=$(echo $pass | tr ' ' '_' | cut -c1-24)
  timeout 300 rocprofv3 --output-format csv --pmc $pass --kernel-trace -d $O/pmc_$n -o pmc -- python $HERE/tools/debug/dual_child.py > /dev/null 2>&1
done
cd $HERE
python - <<'PY' | tee gpurun_out/r06_run11/counters.txt
import csv, glob, collections
n = 16777216
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r06_run11/pmc_*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float); name = {}
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        k = "dual" if "k_cwbvh_dual" in kn else "base" if ", 13, 2, 0, 8, 0>" in kn and "k_cwbvh<false" in kn else None
        if not k: continue
        per[(int(r["Dispatch_Id"]), r["Counter_Name"])] += float(r["Counter_Value"]); name[int(r["Dispatch_Id"])] = k
    for (d, cn), v in per.items():
        tot[name[d]][cn].append(v)
for k in ("base", "dual"):
    print(k)
    for cn, vals in sorted(tot[k].items()):
        big = sorted(vals, reverse=True)[:2]          # (the launches that did the work; the first one warms the caches)
        v = sum(big) / len(big)
        print(f"   {cn:32s} {v:16.0f}   per ray {v / n:10.2f}")
    c = {cn: sum(sorted(v, reverse=True)[:2]) / 2 for cn, v in tot[k].items()}
    if "SQ_INSTS_VALU" in c:
        print(f"   lane-slots per ray {c['SQ_INSTS_VALU'] * 64 / n:.0f}   lanes at work {c['SQ_THREAD_CYCLES_VALU'] / (64 * c['SQ_ACTIVE_INST_VALU']):.3f}")
PY
rm -rf $O/pmc_*
