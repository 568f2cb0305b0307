#!/bin/bash
# Round 6, call 5: host-array path (1 M-ray groups, two-group result buffers, direct path for small batches), the shim test with the hidden rand(), host-ray rates.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r06_run5
mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log ); tail -12 $O/pytest.log
timeout 300 python tools/hostpath_probe.py > $O/hostpath.log 2>&1; tail -5 $O/hostpath.log
python - > $O/host_rays.log 2>&1 <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import tinybvh_amd as tb
from tinybvh_amd import rays as R, scenes
import bench_detail as bd
verts, label = scenes.get("bistro")
ctx = tb.Context(0)
sc = tb.BVH8_CWBVH(ctx).Build(verts)
n = 4096 * 4096
cam = R.camera(*scenes.cameras("bistro")[0], 4096, 4096, 1, 1)
d = ctx.malloc(n * 64); ctx.generate_primary(cam, d, 0, n)
print(json.dumps(bd.host_rays_leg(tb, ctx, sc, d, n), indent=1))
PY
tail -40 $O/host_rays.log
