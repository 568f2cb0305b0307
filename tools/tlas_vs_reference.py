"""BASELINE config 5 next to the reference on the same GPU: 1000 instances of the Dragon stand-in (BVH8_CWBVH BLAS, the configuration of
tiny_bvh_gpu2.cpp) under a TLAS in BVH_GPU format; camera rays and random rays traced by this library's two-level kernel and by the reference's
traverse_tlas (traverse_tlas.cl:13-107) through wavefront2.cl's Extend kernel on ROCm OpenCL, launched as the demo does.  Both sides get the
same TLAS nodes, instance records, BLAS blobs and rays.  The reference derives rD with native_recip on the device and composes the hit's
prim with (instance << 24), so the comparison is on t (relative 1e-4) and on hit / miss."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import tinybvh_amd as tb  # noqa: E402
from tinybvh_amd import rays as R  # noqa: E402
from tinybvh_amd import scenes  # noqa: E402
from oracle_lib import ReferenceOpenCL  # noqa: E402
from tlas_probe import instances  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--side", type=int, default=10)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--random", type=int, default=1 << 20)
a = ap.parse_args()
verts, label = scenes.get("dragon")
ctx = tb.Context(0)
ocl = ReferenceOpenCL()
blas = tb.BVH8_CWBVH(ctx).Build(verts)
inst = instances(a.side, 1.0)
tlas = tb.TLAS(ctx).Build(inst, [blas])
nodes, idx, irec = tlas.Download()
h = blas.host
bn, bt = h.blob(0, np.uint32, 4), h.blob(1, np.uint32, 4)
ext = 2.0 * a.side
print(f"{inst.shape[0]} instances of {label} ({verts.shape[0] // 3} tris, BVH8_CWBVH); TLAS {nodes.shape[0]} BVH_GPU nodes; OpenCL device {ocl.device}")
cam = R.primary(R.camera((-0.6 * ext, 0.8 * ext, -0.9 * ext), (0.62, -0.38, 0.68), a.width, a.height, 1, 1))
rnd = R.random_rays(a.random, (-1.0, -1.0, -1.0), (ext, ext, ext), seed=9)
for tag, rays in (("camera", cam), ("random", rnd)):
    n = (rays.shape[0] // 64) * 64
    rays = np.ascontiguousarray(rays[:n])
    d = ctx.malloc(n * 64); ctx.to_device(d, rays)
    ms = []
    for k in range(4):
        tlas.intersect_device_fresh(d, n, 1e30); t = ctx.time_last_ms()
        if k:
            ms.append(t)
    mine = np.zeros(n, tb.RAY_DTYPE); ctx.from_device(mine, d); ctx.free(d)
    ref, ref_ms = ocl.tlas_extend(nodes, idx, irec, bn, bt, rays, passes=3)
    my_ms = float(np.mean(ms))
    mh, rh = mine["t"] < 1e30, ref[:, 0] < 1e30
    both = mh & rh
    rel = np.abs(mine["t"][both] - ref[both, 0]) / np.maximum(np.abs(ref[both, 0]), 1e-20)
    print(f"  {tag:6s} {n:8d} rays: this library {my_ms:.3f} ms = {n / my_ms / 1e3:7.1f} MRays/s   reference traverse_tlas (OpenCL) {ref_ms:.3f} ms = {n / ref_ms / 1e3:7.1f} MRays/s   "
          f"x{ref_ms / my_ms:.2f}   [hits {int(mh.sum())} / {int(rh.sum())}, hit/miss differs on {int((mh != rh).sum())}, t off by more than 1e-4 on {int((rel > 1e-4).sum())}]", flush=True)
ctx.close()
