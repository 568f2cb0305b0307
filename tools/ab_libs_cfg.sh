#!/bin/bash
# like tools/ab_libs.sh with one ab_configs configuration string per library: tools/ab_libs_cfg.sh name=cfg ...   (cfg = hybridK:flags:variant)
set -u
O=$PWD/gpurun_out/ab_libs
rm -rf $O; mkdir -p $O
for r in 1 2 3; do
  for nc in "$@"; do
    n=${nc%%=*}; cfg=${nc#*=}
    if [ "$n" = tree ]; then L=""; else L=$PWD/tools/_ab/lib$n.so; fi
    TBVH_LIB_OVERRIDE=$L timeout 200 python tools/ab_configs.py --side 4096 --rounds 5 $n=$cfg 2>&1 | tail -1 >> $O/$n.txt
  done
done
for nc in "$@"; do cat $O/${nc%%=*}.txt; done
