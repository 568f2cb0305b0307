#!/bin/bash
# interleaved A/B of several BUILDS of the library: tools/ab_libs.sh <name> ... (tools/_ab/lib<name>.so; "tree" = the in-tree library)
set -u
O=$PWD/gpurun_out/ab_libs
rm -rf $O; mkdir -p $O
for r in 1 2 3; do
  for n in "$@"; do
    if [ "$n" = tree ]; then L=""; else L=$PWD/tools/_ab/lib$n.so; fi
    TBVH_LIB_OVERRIDE=$L timeout 200 python tools/ab_configs.py --side 4096 --rounds 5 $n=keep:0:0 2>&1 | tail -1 >> $O/$n.txt
  done
done
for n in "$@"; do cat $O/$n.txt; done
