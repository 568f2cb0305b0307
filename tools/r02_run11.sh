#!/bin/bash
for bpc in 24 28 32; do echo "== blocks/CU $bpc"; TBVH_BLOCKS_PER_CU=$bpc timeout 300 python tools/ab_probe.py --variants 0,58,52,0 --passes 5 2>&1 | grep "^variant" | cut -c1-110; done
