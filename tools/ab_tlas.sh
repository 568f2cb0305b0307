#!/bin/bash
# A/B of two builds on config 5 (tools/tlas_probe.py: 1000 instances, 8.3 M camera rays per frame + 4.2 M random rays), BLAS layout BVH4_GPU and BVH8_CWBVH
set -u
for r in 1 2 3; do
  for n in base tree; do
    if [ "$n" = tree ]; then L=""; else L=$PWD/tools/_ab/libbase.so; fi
    for lay in 8 10; do
      TBVH_LIB_OVERRIDE=$L timeout 200 python tools/tlas_probe.py --layout $lay --frames 6 --random 4194304 2>&1 | grep -E "frame +5|incoherent" | sed "s/^/$n L$lay /" | cut -c1-200
    done
  done
done
