#!/bin/bash
# builds the WORKING TREE's library with extra compiler flags for kernels_cwbvh.hip only into tools/_ab/lib<name>.so
#   usage: tools/build_variant_lib.sh <name> <flags...>
set -e
NAME=$1; shift
D=/tmp/varbuild_$NAME
rm -rf $D && mkdir -p $D/tinybvh_amd && cp -r /root/repo/tinybvh_amd/csrc $D/tinybvh_amd/ && cp -r /root/repo/include $D/
cd $D/tinybvh_amd/csrc && rm -rf build ../libtinybvh_amd.so && mkdir -p build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-pass-failed "$@" -c kernels_cwbvh.hip -o build/kernels_cwbvh.o
make -j8 2>&1 | grep -E "error|warning" || true
mkdir -p /root/repo/tools/_ab && cp ../libtinybvh_amd.so /root/repo/tools/_ab/lib$NAME.so
ls -la /root/repo/tools/_ab/lib$NAME.so
