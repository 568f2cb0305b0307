#!/bin/bash
# A/B of two BUILDS of the library (tools/_ab/libbase.so = the previous commit, the in-tree library = the working tree), interleaved processes
set -u
O=$PWD/gpurun_out/r04_ab_lib
mkdir -p $O
for r in 1 2 3; do
  TBVH_LIB_OVERRIDE=$PWD/tools/_ab/libbase.so timeout 200 python tools/ab_configs.py --side 4096 --rounds 5 base=keep:0:0 > $O/base_$r.txt 2>&1
  timeout 200 python tools/ab_configs.py --side 4096 --rounds 5 new=keep:0:0 > $O/new_$r.txt 2>&1
done
tail -n 4 $O/base_*.txt $O/new_*.txt
