#!/bin/bash
# round 3, GPU call 1: bench after the tie rule, layout / ray-order probe, micro-benchmarks, the GPU test suite
set -u
mkdir -p gpurun_out/r03_1
O=gpurun_out/r03_1
export TMPDIR=/tmp
( timeout 600 python bench.py --steps 5 --warmup 2 --no-pmc > $O/bench.json 2> $O/bench.err ) 
( timeout 900 python tools/layout_probe.py > $O/layout_probe.txt 2>&1 )
( cd tools/ubench && timeout 300 ./valu_issue > ../../$O/valu_issue.txt 2>&1; timeout 300 ./copy_rate > ../../$O/copy_rate.txt 2>&1; timeout 600 ./gather_coop > ../../$O/gather_coop.txt 2>&1 )
( timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > $O/pytest.txt 2>&1 )
tail -5 $O/pytest.txt
tail -3 $O/layout_probe.txt
