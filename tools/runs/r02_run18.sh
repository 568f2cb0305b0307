#!/bin/bash
# grid-shape sweep with split rays: rays per workgroup x workgroups per CU, 1 M and 4 M rays, three layouts (default kernels)
for sc in "sponza 1024" "bistro 1024" "bistro 2048"; do set -- $sc
for L in 5 8 10; do
for bpc in 24 32; do for rpb in 96 128 192 256 384; do
  r=$(TBVH_BLOCKS_PER_CU=$bpc TBVH_RAYS_PER_BLOCK=$rpb timeout 200 python tools/ab_probe.py --scene $1 --side $2 --layout $L --variants 0 --passes 4 2>&1 | tail -1 | cut -c12-78)
  echo "$1 $2 layout $L blocks/CU $bpc rays/block $rpb: $r"
done; done
r=$(timeout 200 python tools/ab_probe.py --scene $1 --side $2 --layout $L --variants 0 --passes 4 2>&1 | tail -1 | cut -c12-78)
echo "$1 $2 layout $L defaults: $r"
done; done
