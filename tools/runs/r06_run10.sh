#!/bin/bash
# Round 6, call 10: two rays per lane (kernels_cwbvh_dual.hip, debug flag 0x80000) against the shipped incoherent flavor; records byte-compared.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r06_run10
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/ab_configs.py --side 4096 --rounds 5 --check base=keep:0:0 dual=keep:524288:0 gate8=keep:2621440:0 gate16=keep:4718592:0 gate24=keep:6815744:0 gate32=keep:8912896:0 > $O/ab_dual.txt 2>&1; tail -14 $O/ab_dual.txt
