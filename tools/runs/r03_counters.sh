#!/bin/bash
# Round 3 evidence pair for DESIGN.md §5: SQ / TCP / TCC counters of the bench's own 16.7 M-ray batches (tools/ab_probe.py) with the
# incoherent-batch copies off (TBVH_INCOHERENT_COPIES=0: every probed launch is ONE kernel on the packed arrays = round 2's placement) and
# on (the library as shipped: the bounce batch runs the incoherent flavor on the hybrid node copy and the 64-byte triangle records).
set -u
TBVH_INCOHERENT_COPIES=0 bash tools/prof_cmd.sh r03_before python $PWD/tools/ab_probe.py --scene bistro --side 4096 --layout 10 --variants 0 --passes 2 > gpurun_out/r03_counters_before.txt 2>&1
bash tools/prof_cmd.sh r03_after python $PWD/tools/ab_probe.py --scene bistro --side 4096 --layout 10 --variants 0 --passes 2 > gpurun_out/r03_counters_after.txt 2>&1
wc -l gpurun_out/r03_counters_before.txt gpurun_out/r03_counters_after.txt
