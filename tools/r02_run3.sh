#!/bin/bash
set -u
O=gpurun_out/r02c; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/ab_probe.py --variants 0,51,8,16 > $O/ab_pk.log 2>&1; cat $O/ab_pk.log
timeout 900 python tools/ref_blobs_probe.py --side 2048 > $O/ref_blobs.log 2>&1; cat $O/ref_blobs.log
timeout 300 python tools/vs_reference_opencl.py --scene sponza --side 1024 --kind primary > $O/vs_ocl_sponza_primary.log 2>&1; cat $O/vs_ocl_sponza_primary.log
timeout 300 python tools/vs_reference_opencl.py --scene sponza --side 1024 --kind bounce > $O/vs_ocl_sponza_bounce.log 2>&1; tail -4 $O/vs_ocl_sponza_bounce.log
timeout 300 python tools/tlas_probe.py --layout 8 --random 4194304 > $O/tlas_bvh4.log 2>&1; tail -4 $O/tlas_bvh4.log
timeout 300 python tools/tlas_probe.py --layout 10 --random 4194304 > $O/tlas_cwbvh.log 2>&1; tail -3 $O/tlas_cwbvh.log
timeout 1500 python tools/size_sweep.py --sizes 2.8,12,30,60 --variants 0,62,64 --host-tree --order $O/sweep_order.json > $O/size_sweep.log 2>&1; cat $O/size_sweep.log
cd /tmp; timeout 1500 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OLDPWD/$O/sweep_pmc -o pmc -- python $OLDPWD/tools/size_sweep.py --sizes 2.8,12,30,60 --variants 0,62,64 --host-tree --passes 1 --order $OLDPWD/$O/sweep_order_pmc.json > $OLDPWD/$O/size_sweep_pmc_run.log 2>&1; cd $OLDPWD
python tools/size_sweep_pmc.py $O/sweep_pmc $O/sweep_order_pmc.json 4194304 > $O/size_sweep_pmc.txt 2>&1; cat $O/size_sweep_pmc.txt
