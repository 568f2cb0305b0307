#!/bin/bash
set -u
mkdir -p gpurun_out/r03_2
O=gpurun_out/r03_2
export TMPDIR=/tmp
( timeout 900 python tools/layout_probe.py --hybrid all,8192,32768 --flags 1,2,3 --no-bins > $O/layout_probe_16m.txt 2>&1 )
( timeout 900 python tools/layout_probe.py --side 1024 --hybrid all,8192,32768 --flags 1 --no-bins --passes 6 > $O/layout_probe_1m.txt 2>&1 )
( timeout 900 python tools/layout_probe.py --side 2048 --hybrid all,8192,32768 --no-bins --passes 4 > $O/layout_probe_4m.txt 2>&1 )
( timeout 900 python tools/layout_probe.py --side 2048 --variant 89 --hybrid "" --no-bins --passes 4 > $O/layout_probe_4m_v89.txt 2>&1 )
( timeout 900 python tools/layout_probe.py --side 2048 --variant 88 --hybrid "" --no-bins --passes 4 > $O/layout_probe_4m_v88.txt 2>&1 )
( timeout 2400 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > $O/pytest.txt 2>&1 )
tail -5 $O/pytest.txt
