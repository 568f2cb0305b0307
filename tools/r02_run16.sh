#!/bin/bash
# counters of k_bvh2 / k_bvh4 on config 2's batch (Sponza stand-in, 1 M camera rays)
set -u
for L in 5 8; do
bash tools/prof_cmd.sh cfg2_l$L python /root/repo/tools/ab_probe.py --scene sponza --side 1024 --layout $L --variants 0 --passes 8 > gpurun_out/cfg2_l$L.txt 2>&1
tail -60 gpurun_out/cfg2_l$L.txt
done
