#!/bin/bash
set -u
O=gpurun_out/r02i; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py tests/test_golden.py tests/test_cwbvh_schedules.py tests/test_wavefront.py tests/test_refit_device.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -4 $O/pytest.log
timeout 600 python tools/ab_probe.py --variants 0,72,52,0,72 --passes 6 --stats 59 2>&1 | grep -E "^variant|^stats" | cut -c1-260
timeout 300 python tools/ab_probe.py --scene sponza --side 1024 --variants 0,72,0,72 --passes 8 2>&1 | grep -E "^variant" | cut -c1-110
timeout 300 python tools/ab_probe.py --scene sponza --side 4096 --variants 0,72,0 --passes 4 2>&1 | grep -E "^variant" | cut -c1-110
