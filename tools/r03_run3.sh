#!/bin/bash
set -u
mkdir -p gpurun_out/r03_3
O=gpurun_out/r03_3
export TMPDIR=/tmp
python tools/tie_probe.py > $O/tie_probe.txt 2>&1
python tools/ab_configs.py --side 4096 --rounds 7 base=-1:0:0 hyall=all:0:0 hy8k=8192:0:0 hy32k=32768:0:0 nt=-1:1:0 t64=-1:2:0 hy8k_nt=8192:1:0 hy8k_t64=8192:2:0 hy8k_nt_t64=8192:3:0 strict=-1:0:72 > $O/ab_16m.txt 2>&1
python tools/ab_configs.py --side 2048 --rounds 9 base=-1:0:0 hy8k=8192:0:0 old_spill=-1:0:89 t64=-1:2:0 > $O/ab_4m.txt 2>&1
python tools/ab_configs.py --side 1024 --rounds 15 base=-1:0:0 hy8k=8192:0:0 hy32k=32768:0:0 t64=-1:2:0 > $O/ab_1m.txt 2>&1
python tools/ab_configs.py --scene sponza --side 1024 --rounds 15 base=-1:0:0 hy8k=8192:0:0 t64=-1:2:0 > $O/ab_sponza_1m.txt 2>&1
( timeout 2400 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > $O/pytest.txt 2>&1 )
tail -5 $O/pytest.txt
