"""Triangle cost of the SAH-optimal collapse (how a triangle test is priced against a node visit when leaves are formed) against the
traced rate, on the bench's batches: the collapse decides S (node visits) and T (triangle tests) per ray."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import tinybvh_amd as tb
from tinybvh_amd import rays as R, scenes
from ab_probe import make_batches

side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
verts, label = scenes.get("bistro")
ctx = tb.Context(0)
n = side * side
cam = R.camera(*scenes.STREET_CAMERAS[0], side, side, 1, 1)
base = tb.BVH8_CWBVH(ctx).Build(verts)
d_prim, d_diff, d_shad = make_batches(ctx, base, verts, cam, n)
d_occ = ctx.malloc(n)
print(label, n, "rays per batch")
for leaf, c_prim in ((0, 0.0), (1, 1.0), (2, 0.3), (2, 1.0), (3, 0.3), (3, 0.6), (3, 1.0), (3, 2.0)):
    sc = base if leaf == 0 else tb.BVH8_CWBVH(ctx).Build(verts, optimal_collapse=True, c_prim=c_prim, max_leaf_tris=leaf)
    row = []
    for kind, d in (("primary", d_prim), ("diffuse", d_diff), ("shadow", d_shad)):
        ms = []
        for p in range(4):
            if kind == "shadow":
                sc.occluded_device(d, n, d_occ)
            else:
                sc.intersect_device_fresh(d, n, 1e30)
            if p:
                ms.append(ctx.time_last_ms())
        row.append(n / np.mean(ms) / 1e3)
    print(f"  max leaf {leaf if leaf else 'default'} triangle cost {c_prim if leaf else 'default':>8}: {sc.device_bytes / 1e6:6.0f} MB   camera {row[0]:7.0f}  bounce {row[1]:7.0f}  shadow {row[2]:7.0f} MRays/s", flush=True)
    if sc is not base:
        sc.free()
