#!/bin/bash
set -u
O=gpurun_out/r02e; mkdir -p $O
( timeout 900 python -m pytest tests/test_wavefront_reference.py tests/test_wavefront.py -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -15 $O/pytest.log
python -c "
import tinybvh_amd as tb
c=tb.Context(0); print('copy GB/s', c.copy_bandwidth_gbps(1<<30,3), c.copy_bandwidth_gbps(1<<31,3))"
