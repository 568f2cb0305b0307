#!/bin/bash
python -m pytest tests/test_tlas.py tests/test_tlas_device_build.py tests/test_wavefront.py tests/test_opacity_micromaps.py tests/test_examples.py -m gpu -x -q 2>&1 | tail -4
for L in 8 10; do timeout 300 python tools/tlas_probe.py --layout $L --random 4194304 --frames 3 2>&1 | grep -E "frame 2: DEVICE|incoherent" | tail -2 | cut -c1-330 | sed 's/host call.*device time/dev/'; done
