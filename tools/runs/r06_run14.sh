#!/bin/bash
# Round 6, call 14: v_pk_fma_f32 plane FMAs in the packet kernel's node test (library built with -DTBVH_PACKET_PK_FMA) against the shipped form:
# tools/coherent_modes.py child (16.7 M camera + shadow rays, pinned to the packet schedule), alternating, 4 rounds; CRC of the records.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r06_run14
mkdir -p $O
for r in 1 2 3 4; do
  for lib in base pk; do
    if [ $lib = pk ]; then L=$PWD/tinybvh_amd/libtinybvh_amd_pk.so; else L=""; fi
    for sc in bistro street_rot; do
      echo -n "$lib $sc " >> $O/ab_pk.txt
      TBVH_LIB_OVERRIDE=$L TBVH_COHERENT_TUNER=3 timeout 200 python tools/coherent_modes.py --child $sc 4096 2>/dev/null | tail -1 | cut -c1-200 >> $O/ab_pk.txt
    done
  done
done
cat $O/ab_pk.txt
