#!/bin/bash
set -u
O=gpurun_out/r02j; mkdir -p $O
( timeout 900 python -m pytest tests/test_tlas.py tests/test_tlas_device_build.py tests/test_wavefront.py tests/test_opacity_micromaps.py tests/test_examples.py tests/test_refit_device.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -12 $O/pytest.log
for v in 7 0 26 21 22 23 24 25 29 30 31; do echo "== variant $v"; timeout 300 python tools/tlas_probe.py --layout 10 --random 4194304 --frames 2 --variant $v 2>&1 | grep -E "frame 1: DEVICE|incoherent" | tail -3 | cut -c1-420 | sed 's/host call.*device time/dev/'; done
