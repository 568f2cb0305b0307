#!/bin/bash
set -u
mkdir -p gpurun_out/r03_4
O=gpurun_out/r03_4
export TMPDIR=/tmp
python tools/ab_configs.py --side 4096 --rounds 7 base=-1:0:0 one_kernel=-1:4:0 incoh_forced=-1:0:90 strict=-1:0:72 > $O/ab_16m.txt 2>&1
python tools/ab_configs.py --side 2048 --rounds 9 base=-1:0:0 one_kernel=-1:4:0 > $O/ab_4m.txt 2>&1
( timeout 600 python bench.py --steps 10 --warmup 3 --no-pmc > $O/bench.json 2> $O/bench.err )
( timeout 2400 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > $O/pytest.txt 2>&1 )
tail -5 $O/pytest.txt
