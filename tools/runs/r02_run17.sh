#!/bin/bash
# default build after split rays: GPU tests, config 2 against the reference's OpenCL kernels, contract bench
set -u
O=gpurun_out/r02x; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -3 $O/pytest.log
for k in primary bounce; do timeout 600 python tools/vs_reference_opencl.py --scene sponza --side 1024 --kind $k 2>&1 | grep -v "^\[" | tail -4; done > $O/vs_ocl.txt 2>&1; cat $O/vs_ocl.txt
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; tail -c 300 $O/bench.json
