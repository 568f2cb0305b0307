#!/bin/bash
set -u
O=$PWD/gpurun_out/r04_run9
mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_cwbvh_schedules.py -m gpu -q -p no:cacheprovider -x --timeout 60 --timeout-method thread > $O/pytest.txt 2>&1 ); echo "rc $?" >> $O/pytest.txt; tail -15 $O/pytest.txt
if grep -q "rc 0" $O/pytest.txt; then
( timeout 300 python tools/ab_configs.py --side 4096 --rounds 7 base=keep:0:0 pair=keep:128:0 base2=keep:0:0 pair2=keep:128:0 > $O/ab_pair.txt 2>&1 ); cat $O/ab_pair.txt
fi
