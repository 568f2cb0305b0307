"""child for rocprofv3: the bench's diffuse batch through the shipped incoherent flavor (3 launches) and through the two-rays-per-lane kernel (3 launches)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import tinybvh_amd as tb
from tinybvh_amd import rays as R, scenes
from ab_probe import make_batches
verts, label = scenes.get("bistro")
ctx = tb.Context(0)
sc = tb.BVH8_CWBVH(ctx).Build(verts)
side = 4096; n = side * side
cam = R.camera(*scenes.cameras("bistro")[0], side, side, 1, 1)
d_prim, d_diff, d_shad = make_batches(ctx, sc, verts, cam, n)
for flags in (0, 0x80000):
    ctx.set_debug_flags(flags)
    for _ in range(3):
        sc.intersect_device_fresh(d_diff, n, 1e30)
    ctx.synchronize()
ctx.close()
