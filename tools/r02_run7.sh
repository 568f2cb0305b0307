#!/bin/bash
set -u
O=gpurun_out/r02g; mkdir -p $O
( timeout 900 python -m pytest tests/test_tlas.py tests/test_tlas_device_build.py tests/test_convert_device.py tests/test_wavefront.py tests/test_opacity_micromaps.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -12 $O/pytest.log
for v in 0 26 21 22 23 24 25 7 13; do echo "== variant $v"; timeout 300 python tools/tlas_probe.py --layout 8 --random 4194304 --frames 2 --variant $v 2>&1 | grep -E "DEVICE TLAS|incoherent" | tail -3; done
