#!/bin/bash
# Round 6: launchQuery split into helpers (same behaviour: the full GPU suite); deferred triangles in the INCOHERENT flavor (experiment build, flags 0x100000.. 0x800000: gate 1 / 4 / 8 / 16 lanes)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r06_run18
mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log ); tail -3 $O/pytest.log
TBVH_LIB_OVERRIDE=$PWD/tinybvh_amd/libtinybvh_amd_exp.so timeout 900 python tools/ab_configs.py --side 4096 --rounds 5 --check base=keep:0:0 spec1=keep:1048576:0 spec4=keep:2097152:0 spec8=keep:4194304:0 spec16=keep:8388608:0 > $O/ab_spec.txt 2>&1; tail -14 $O/ab_spec.txt
