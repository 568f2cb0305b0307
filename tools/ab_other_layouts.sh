#!/bin/bash
# A/B of two builds (tools/_ab/libbase.so against the in-tree library) on the kernels the contract bench does not time: BVH_GPU / BVH4_GPU
# (tools/ab_probe.py, Sponza stand-in 1 M rays and Bistro stand-in 4 M rays) and the two-level kernels of config 5 (tools/tlas_probe.py)
set -u
O=$PWD/gpurun_out/ab_other
rm -rf $O; mkdir -p $O
for r in 1 2; do
  for n in base tree; do
    if [ "$n" = tree ]; then L=""; else L=$PWD/tools/_ab/libbase.so; fi
    for lay in 5 8; do
      TBVH_LIB_OVERRIDE=$L timeout 200 python tools/ab_probe.py --scene sponza --side 1024 --layout $lay --variants 0 --passes 6 2>&1 | grep -E "variant +0" | sed "s/^/$n sponza L$lay /" >> $O/$n.txt
      TBVH_LIB_OVERRIDE=$L timeout 200 python tools/ab_probe.py --scene bistro --side 2048 --layout $lay --variants 0 --passes 6 2>&1 | grep -E "variant +0" | sed "s/^/$n bistro L$lay /" >> $O/$n.txt
    done
    for lay in 8 10; do
      TBVH_LIB_OVERRIDE=$L timeout 200 python tools/tlas_probe.py --layout $lay --frames 5 --random 4194304 2>&1 | tail -2 | sed "s/^/$n tlas L$lay /" >> $O/$n.txt
    done
  done
done
cat $O/base.txt; cat $O/tree.txt
