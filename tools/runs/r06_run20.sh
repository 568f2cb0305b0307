#!/bin/bash
# Round 6: node loads ahead of the triangle phase in the incoherent flavor (experiment build: flags 0x80000 / 0x1000000 / 0x2000000 = register budgets of 8 / 7 / 6 waves per SIMD)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r06_run20
mkdir -p $O
TBVH_LIB_OVERRIDE=$PWD/tinybvh_amd/libtinybvh_amd_exp.so timeout 900 python tools/ab_configs.py --side 4096 --rounds 5 --check base=keep:0:0 pref8=keep:524288:0 pref7=keep:16777216:0 pref6=keep:33554432:0 > $O/ab_pref.txt 2>&1; tail -12 $O/ab_pref.txt
