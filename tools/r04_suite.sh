#!/bin/bash
# the GPU suite (+ smoke) on the GPU box; usage: tools/r04_suite.sh <tag> [pytest args]
set -u
TAG=$1; shift
O=$PWD/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -s --timeout 150 --timeout-method thread "$@" > $O/pytest.txt 2>&1 ); echo "rc $?" >> $O/pytest.txt; tail -8 $O/pytest.txt
grep -E "differences from the real|config 5 differences" $O/pytest.txt
( timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1 ); tail -2 $O/smoke.txt
