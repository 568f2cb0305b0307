#!/bin/bash
set -u
mkdir -p gpurun_out/r03_7
O=gpurun_out/r03_7
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_sharded.py tests/test_examples.py tests/test_wavefront.py tests/test_cwbvh_schedules.py tests/test_bench_kernels.py -m gpu -q --maxfail=25 -p no:cacheprovider > $O/pytest.txt 2>&1 )
tail -5 $O/pytest.txt
examples/_build/wavefront_demos > $O/wavefront_demos.txt 2>&1; tail -8 $O/wavefront_demos.txt
examples/_build/speedtest_gpu_section > $O/speedtest.txt 2>&1; tail -14 $O/speedtest.txt
( timeout 900 python bench.py --no-pmc --no-configs --no-cpu-baseline --one-process-devices 2 > $O/bench_one_process.json 2> $O/bench_one_process.err ); tail -2 $O/bench_one_process.err
