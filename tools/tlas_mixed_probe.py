"""A TLAS over BLASes of three layouts and very different sizes (1000 instances): camera / shadow / random MRays/s.  With TBVH_WIDE_COPY_MIN=32768 in the
environment the small BLASes get no copies (the behaviour before the TLAS-side threshold of 64 entries) and the TLAS falls back to the flat loop."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import tinybvh_amd as tb
from tinybvh_amd import rays as R, scenes
from test_tlas import grid_instances

ctx = tb.Context(0)
def unit(m):
    m = m.copy(); m[:, :3] -= 0.5 * (m[:, :3].min(0) + m[:, :3].max(0)); m[:, :3] *= np.float32(1.6 / float((m[:, :3].max(0) - m[:, :3].min(0)).max())); return np.ascontiguousarray(m)
blas = [tb.BVH_GPU(ctx).Build(unit(scenes.blob(100_000, seed=3))), tb.BVH8_CWBVH(ctx).Build(unit(scenes.blob(20_000, seed=4))), tb.BVH4_GPU(ctx).Build(unit(scenes.blob(5_000, seed=5)))]
inst = grid_instances(10, 0.5, 3, n_blas=3)
tlas = tb.TLAS(ctx).Build(inst, blas)
cam = R.camera((-12.0, 16.0, -18.0), (0.62, -0.38, 0.68), 2560, 1600, 1, 1)
nt = 2560 * 1600
d = ctx.malloc(nt * 64); d_sh = ctx.malloc(nt * 64); d_occ = ctx.malloc(nt)
ctx.generate_primary(cam, d, 0, nt)
def timed(fn, passes=6):
    ms = []
    for p in range(passes):
        fn(); ctx.synchronize()
        if p: ms.append(ctx.time_last_ms())
    return float(np.median(ms))
ms_cam = timed(lambda: tlas.intersect_device_fresh(d, nt, 1e30))
ctx.generate_shadow(d, d_sh, nt, (10.0, 40.0, 10.0), 1e-4)
ms_sh = timed(lambda: tlas.occluded_device(d_sh, nt, d_occ))
rr = R.random_rays(1 << 22, (-1.0, -1.0, -1.0), (20.0, 20.0, 20.0), seed=9)
d_r = ctx.malloc(rr.shape[0] * 64); ctx.to_device(d_r, rr)
ms_r = timed(lambda: tlas.intersect_device_fresh(d_r, rr.shape[0], 1e30))
rec = np.zeros(nt, tb.RAY_DTYPE); ctx.from_device(rec, d)
print(f"TBVH_WIDE_COPY_MIN={os.environ.get('TBVH_WIDE_COPY_MIN', 'unset')}: camera {nt / ms_cam / 1e3:8.1f} MRays/s  shadow {nt / ms_sh / 1e3:8.1f}  random {rr.shape[0] / ms_r / 1e3:8.1f}   hits {int((rec['t'] < 1e30).sum())} checksum {int(rec['prim'][rec['t'] < 1e30].astype(np.uint64).sum())}")
