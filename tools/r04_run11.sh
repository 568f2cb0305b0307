#!/bin/bash
# Round 4: the GPU suite with a longer randomised hunt (TBVH_RANDOM_SEEDS=300: 300 single-level + 100 two-level + 150 hybrid-copy configurations)
set -u
O=$PWD/gpurun_out/r04_run11
mkdir -p $O
export TMPDIR=/tmp
export TBVH_RANDOM_SEEDS=300
( timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 150 --timeout-method thread -x > $O/pytest.txt 2>&1 ); echo "rc $?" >> $O/pytest.txt; tail -12 $O/pytest.txt | cut -c1-400
