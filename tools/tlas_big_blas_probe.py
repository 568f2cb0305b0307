"""A TLAS over a few instances of a LARGE BLAS (the Bistro stand-in four times): the static-world case of tiny_bvh_gpu2.cpp, where the two-level
kernel does the work the single-level kernel does for the same geometry.  Usage: tlas_big_blas_probe.py [layout [statistics variant]]"""
import sys, os, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import tinybvh_amd as tb
from tinybvh_amd import rays as R, scenes
verts, label = scenes.get("bistro")
ctx = tb.Context(0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 10
blas = tb.LAYOUT_CLASSES[L](ctx).Build(verts)
ext = float((verts[:, :3].max(0) - verts[:, :3].min(0)).max())
k = 4
T = np.zeros((k, 4, 4), np.float32)
for i in range(k):
    T[i] = np.eye(4); T[i, 0, 3] = (i % 2) * ext * 1.05; T[i, 2, 3] = (i // 2) * ext * 1.05
tlas = tb.TLAS(ctx).Build(tb.make_instances(T, np.zeros(k, np.uint32)), [blas])
n = 2048 * 2048
cam = R.camera(*scenes.STREET_CAMERAS[0], 2048, 2048, 1, 1)
d = ctx.malloc(n * 64); ctx.generate_primary(cam, d, 0, n)
ms = []
for p in range(5):
    tlas.intersect_device_fresh(d, n, 1e30); ms.append(ctx.time_last_ms())
lo = verts[:, :3].min(0); hi = lo + ext * 2.1
rr = R.random_rays(1 << 21, tuple(lo), tuple(hi), seed=3)
d2 = ctx.malloc(rr.shape[0] * 64); ctx.to_device(d2, rr)
m2 = []
for p in range(5):
    tlas.intersect_device_fresh(d2, rr.shape[0], 1e30); m2.append(ctx.time_last_ms())
if len(sys.argv) > 2:
    import ctypes as C
    tlas.set_variant(int(sys.argv[2]))
    st = (C.c_uint64 * 8)()
    tb.lib.tbvh_debug_stats(ctx._h, st, 1)
    tlas.intersect_device_fresh(d, n, 1e30); ctx.synchronize()
    tb.lib.tbvh_debug_stats(ctx._h, st, 1)
    it, act, pa, la, pb, lb, pc, lc = [int(x) for x in st]
    print(f"   camera rays, statistics variant {sys.argv[2]}: wave passes {it} ({it * 64 / n:.1f} per ray), lanes holding a ray {act / max(it, 1):.1f}; node phases {pa / it:.3f} per pass at {la / max(pa, 1):.1f} lanes "
          f"({la / n:.1f} per ray); instance phases {pb / it:.3f} at {lb / max(pb, 1):.1f} ({lb / n:.2f} per ray); triangle phases {pc / it:.3f} at {lc / max(pc, 1):.1f} ({lc / n:.1f} per ray)")
    tlas.set_variant(0)
print(f"layout {L}: {k} instances of {label}: camera {n / np.mean(ms[1:]) / 1e3:.0f} MRays/s, random {rr.shape[0] / np.mean(m2[1:]) / 1e3:.0f} MRays/s  (blocks/CU {os.environ.get('TBVH_BLOCKS_PER_CU', 'default')})")
