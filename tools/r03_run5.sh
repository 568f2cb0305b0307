#!/bin/bash
set -u
mkdir -p gpurun_out/r03_5
O=gpurun_out/r03_5
export TMPDIR=/tmp
python tools/ab_configs.py --side 4096 --rounds 7 base=keep:0:0 one_kernel=keep:4:0 cohonly=keep:8:0 w20=keep:5120:0 w28=keep:7168:0 w32=keep:8192:0 k4096=4096:0:0 k16384=16384:0:0 k8192=8192:0:0 > $O/ab_16m.txt 2>&1
python tools/ab_configs.py --side 2048 --rounds 9 base=keep:0:0 one_kernel=keep:4:0 w28=keep:7168:0 > $O/ab_4m.txt 2>&1
( timeout 1200 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > $O/pytest.txt 2>&1 )
tail -3 $O/pytest.txt
