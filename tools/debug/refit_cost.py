"""device refit of a 100 k-triangle BLAS of each layout: alone, and with the copies a TLAS over it (closest-hit + any-hit queries seen) has made"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import tinybvh_amd as tb
from tinybvh_amd import rays as R, scenes
from test_tlas import grid_instances
ctx = tb.Context(0)
m = scenes.blob(100_000, seed=3)
d_v = ctx.malloc(m.nbytes); ctx.to_device(d_v, m)
rays = R.random_rays(65536, (-2, -2, -2), (21, 21, 21), seed=1)
for lay in (tb.LAYOUT_BVH_GPU, tb.LAYOUT_BVH4_GPU, tb.LAYOUT_CWBVH):
    b = tb.LAYOUT_CLASSES[lay](ctx).Build(m)
    def refit_ms():
        ms = []
        for k in range(5):
            b.Refit((d_v, m.shape[0] // 3), on_device=True); ctx.synchronize(); ms.append(ctx.time_last_ms())
        return float(np.median(ms[1:]))
    alone = refit_ms()
    t = tb.TLAS(ctx).Build(grid_instances(10, 0.5, 3), [b])
    t.Intersect(rays.copy()); t.IsOccluded(rays.copy())
    import time
    t0 = time.perf_counter(); b.Refit((d_v, m.shape[0] // 3), on_device=True); ctx.synchronize(); wall = (time.perf_counter() - t0) * 1e3
    print(f"layout {lay}: refit alone {alone:.3f} ms (last timed op); under a TLAS with its copies: last op {ctx.time_last_ms():.3f} ms, wall of the whole call {wall:.3f} ms, device bytes {b.device_bytes}")
    t.free(); b.free()
