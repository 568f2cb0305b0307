#!/bin/bash
set -u
O=gpurun_out/r02f; mkdir -p $O
tools/prof_cmd.sh r02f_tlas python $PWD/tools/tlas_probe.py --layout 8 --random 4194304 --frames 1 > $O/prof_tlas.log 2>&1
grep -v "per dispatch" gpurun_out/prof_r02f_tlas/summary.txt | head -150
