"""End-to-end rate of the HOST-ray entry points (tbvh_intersect / tbvh_occluded: rays in caller memory, stride 64 or
128 like tinybvh::Ray[]): upload + kernel + read-back, wall clock."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C  # noqa: E402

import tinybvh_amd as tb  # noqa: E402
from tinybvh_amd import rays as R  # noqa: E402
from tinybvh_amd import scenes  # noqa: E402

verts, label = scenes.get("sponza")
ctx = tb.Context(0)
sc = tb.BVH8_CWBVH(ctx).Build(verts)
for side in (1024, 4096):
    cam = R.camera(*scenes.SPONZA_CAMERAS[0], side, side, 1, 1)
    n = side * side
    d = ctx.malloc(n * 64)
    ctx.generate_primary(cam, d, 0, n)
    r64 = np.zeros(n, tb.RAY_DTYPE); ctx.from_device(r64, d)
    r128 = np.zeros((n, 128), np.uint8); r128[:, :64] = r64.view(np.uint8).reshape(n, 64)
    for name, arr, stride in (("stride 64", r64, 64), ("stride 128 (tinybvh::Ray[])", r128, 128)):
        ts = []
        for p in range(4):
            work = arr.copy()
            t0 = time.perf_counter()
            tb.check(tb.lib.tbvh_intersect(sc._h, C.c_void_p(work.ctypes.data), n, stride), "tbvh_intersect")
            ts.append(time.perf_counter() - t0)
        k = ctx.time_last_ms()
        print(f"{n} host rays, {name}: {np.mean(ts[1:]) * 1e3:.2f} ms wall = {n / np.mean(ts[1:]) / 1e6:.0f} MRays/s end to end (kernel alone {k:.2f} ms)", flush=True)
    ctx.free(d)
ctx.close()
