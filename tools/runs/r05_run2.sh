#!/bin/bash
# round 5, GPU call 2: GPU tests on the new defaults (split triangles), link-rate microbenchmark, TRI2 gates + counters, host-thread sweep
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
nproc > $O/r05_2_nproc.txt
( timeout 700 python -m pytest tests -m gpu -x -q > $O/r05_2_pytest.log 2>&1; echo "pytest rc $?" >> $O/r05_2_pytest.log )
tail -8 $O/r05_2_pytest.log
( timeout 300 tools/ubench/link_rate 16 > $O/r05_link_rate.txt 2>&1 ); cat $O/r05_link_rate.txt
EXP=$PWD/tinybvh_amd/libtinybvh_amd_exp.so
# TRI2 gated: flags = 0x10000 | gate << 20
( TBVH_LIB_OVERRIDE=$EXP TBVH_COHERENT_TUNER=0 timeout 400 python tools/ab_configs.py --rounds 5 base=keep:0:0 tri2=keep:65536:0 tri2_g4=keep:4259840:0 tri2_g8=keep:8454144:0 tri2_g12=keep:12648448:0 > $O/r05_ab_tri2_gates.txt 2>&1 ); cat $O/r05_ab_tri2_gates.txt
# counters of the bounce launch: shipped vs TRI2 (exp library, flags from the environment)
for tag in base tri2; do
  fl=0; [ $tag = tri2 ] && fl=65536
  for pass in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
    n=$(echo $pass | tr ' ' '_' | cut -c1-30)
    ( cd /tmp && TBVH_LIB_OVERRIDE=$EXP TBVH_DEBUG_FLAGS=$fl TBVH_COHERENT_TUNER=0 timeout 200 rocprofv3 --output-format csv --pmc $pass --kernel-trace -d $OLDPWD/$O/pmc_${tag}_$n -o pmc -- python $OLDPWD/bench.py --pmc-child --scene bistro --side 4096 --layout 10 > $OLDPWD/$O/pmc_${tag}_$n.log 2>&1 )
  done
done
python - <<'PY' > gpurun_out/r05_tri2_counters.txt 2>&1
import csv, glob, collections
for tag in ("base","tri2"):
    tot=collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_{tag}_*/**/*counter_collection.csv", recursive=True):
        per=collections.defaultdict(float); names={}
        for r in csv.DictReader(open(f)):
            if ", 13, 2, " in r["Kernel_Name"]:
                per[(int(r["Dispatch_Id"]), r["Counter_Name"])]+=float(r["Counter_Value"])
        ids=sorted({k[0] for k in per})
        # the bounce launches are the incoherent-flavor dispatches that did work: take the 3 largest by any counter
        for cn in {k[1] for k in per}:
            vals=sorted((per[(i,cn)] for i in ids), reverse=True)[:3]
            tot[cn]=sum(vals)/max(len(vals),1)
    print(tag, {k: round(v) for k,v in sorted(tot.items())})
    if tot.get("SQ_ACTIVE_INST_VALU"): print("   lane utilisation", tot["SQ_THREAD_CYCLES_VALU"]/(64*tot["SQ_ACTIVE_INST_VALU"]), " VALU wave-instructions per ray", tot["SQ_INSTS_VALU"]/16777216, " L1 accesses per ray", tot.get("TCP_TOTAL_CACHE_ACCESSES_sum",0)/16777216)
PY
cat gpurun_out/r05_tri2_counters.txt
rm -rf gpurun_out/pmc_base_* gpurun_out/pmc_tri2_*
# host-thread sweep of the staged host-ray path
for th in 4 8 12 16; do echo "TBVH_HOST_THREADS=$th"; TBVH_HOST_THREADS=$th timeout 200 python tools/hostpath_probe.py 2>&1 | grep "16777216"; done > $O/r05_hostpath_threads.txt 2>&1; cat $O/r05_hostpath_threads.txt
