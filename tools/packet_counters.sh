export TMPDIR=/tmp
HERE=$PWD
cd /tmp
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU" "SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $pass | tr ' ' '_' | cut -c1-24)
  TBVH_COHERENT_TUNER=3 timeout 200 rocprofv3 --output-format csv --pmc $pass --kernel-trace -d $HERE/gpurun_out/pkc_$n -o pmc -- python $HERE/tools/coherent_modes.py --child bistro 4096 > /dev/null 2>&1
done
cd $HERE
python - > gpurun_out/r06_packet_counters.txt <<'PY'
import csv, glob, collections
tot=collections.defaultdict(list)
for f in glob.glob("gpurun_out/pkc_*/**/*counter_collection.csv", recursive=True):
    per=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "k_cwbvh_packet<false" in r["Kernel_Name"]:
            per[(int(r["Dispatch_Id"]), r["Counter_Name"])]+=float(r["Counter_Value"])
    ids=sorted({k[0] for k in per})
    for cn in {k[1] for k in per}:
        vals=sorted((per[(i,cn)] for i in ids), reverse=True)[:4]
        tot[cn]=sum(vals)/max(len(vals),1)
n=16777216
for k,v in sorted(tot.items()): print(f"{k:28s} {v:16.0f}   per ray {v/n:10.2f}")
PY
rm -rf gpurun_out/pkc_*
