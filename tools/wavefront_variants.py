"""A 4096 x 4096 path-traced frame of the Bistro stand-in with the traversal kernels picked by the probe (0) or forced: 90 = the incoherent flavor for every
query, 91 = the coherent flavor for every query.  (Shadow rays of later depths point at one light — the probe calls them coherent — but start all over the scene.)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tinybvh_amd as tb
from tinybvh_amd import rays as R, scenes
verts, label = scenes.get("bistro")
ctx = tb.Context(0)
sc = tb.BVH8_CWBVH(ctx).Build(verts)
d_verts = ctx.malloc(verts.nbytes); ctx.to_device(d_verts, verts)
W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cam = R.camera(*scenes.STREET_CAMERAS[0], W, H, 1, 1)
wf = tb.Wavefront(ctx, W, H)
light = (0.0, 0.9 * float(verts[:, 1].max()), 0.0)
for rnd in range(3):
    for v in (0, 90, 91):
        sc.set_variant(v)
        ms = []
        for f in range(4):
            st = wf.render(sc, d_verts, cam, light, (3000.0, 3000.0, 3000.0), max_depth=3, seed=f + 1)
            ms.append(st["frame_ms"])
        print(f"variant {v:2d}: frame {np.median(ms[1:]):.2f} ms  extend {st['extend_rays'][:3]} shadow {st['shadow_rays'][:3]}", flush=True)
