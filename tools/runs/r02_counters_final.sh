#!/bin/bash
# SQ / TCP / TCC counters of the kernels as shipped on the bench's own batches (16.7 M rays: the probed schedule for camera and shadow rays, the strict
# one for bounce rays) and on 1 M-ray batches (the kernels with split rays)
set -u
bash tools/prof_cmd.sh final16m python /root/repo/tools/ab_probe.py --scene bistro --side 4096 --layout 10 --variants 0 --passes 2 > gpurun_out/counters_final16m.txt 2>&1
bash tools/prof_cmd.sh final1m python /root/repo/tools/ab_probe.py --scene bistro --side 1024 --layout 10 --variants 0 --passes 4 > gpurun_out/counters_final1m.txt 2>&1
wc -l gpurun_out/counters_final16m.txt gpurun_out/counters_final1m.txt
