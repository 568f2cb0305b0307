#!/bin/bash
# small-batch grid shape sweep (config 2: Sponza stand-in, 1 M camera rays, BVH_GPU and BVH4_GPU)
for rpb in 64 96 128 192 256; do for bpc in 24 32; do
  echo "== rays/block $rpb  blocks/CU $bpc"; TBVH_RAYS_PER_BLOCK=$rpb TBVH_BLOCKS_PER_CU=$bpc timeout 200 python tools/perf_probe.py --scene sponza --layouts 5,8,10 --passes 8 2>&1 | grep "^layout" | cut -c1-200
done; done
echo "== default"; timeout 200 python tools/perf_probe.py --scene sponza --layouts 5,8,10 --passes 8 2>&1 | grep "^layout" | cut -c1-200
