#!/bin/bash
for v in 0 22 23 24 25 27 28 29 30 31 26; do echo "== variant $v"; timeout 300 python tools/tlas_probe.py --layout 8 --random 4194304 --frames 2 --variant $v 2>&1 | grep -E "frame 1: DEVICE|incoherent" | tail -3 | cut -c1-420; done
