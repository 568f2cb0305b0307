#!/bin/bash
# builds the library of the last commit into tools/_ab/libbase.so (for tools/runs/r04_ab_lib.sh: A/B of two builds)
set -e
rm -rf /tmp/basebuild && mkdir -p /tmp/basebuild
git -C /root/repo archive HEAD tinybvh_amd/csrc include | tar -x -C /tmp/basebuild
make -C /tmp/basebuild/tinybvh_amd/csrc -j8 2>&1 | grep -E "error|warning" || true
mkdir -p /root/repo/tools/_ab && cp /tmp/basebuild/tinybvh_amd/libtinybvh_amd.so /root/repo/tools/_ab/libbase.so
ls -la /root/repo/tools/_ab/libbase.so
