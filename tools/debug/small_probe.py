"""debug aid: camera / shadow rays on a small scene: unprobed (debug flag 64) against the two-kernel path pinned to strict (TBVH_COHERENT_TUNER=2) or packet (3)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tinybvh_amd as tb
from tinybvh_amd import rays as R, scenes
name, side = sys.argv[1], int(sys.argv[2])
verts, label = scenes.get(name)
ctx = tb.Context(0)
sc = tb.BVH8_CWBVH(ctx).Build(verts)
n = side * side
cams = scenes.SPONZA_CAMERAS if name == "sponza" else scenes.STREET_CAMERAS
d, d_s, d_occ = ctx.malloc(n * 64), ctx.malloc(n * 64), ctx.malloc(n)
ctx.generate_primary(R.camera(*cams[0], side, side, 1, 1), d, 0, n)
sc.intersect_device_fresh(d, n, 1e30)
ext = float((verts[:, :3].max(0) - verts[:, :3].min(0)).max())
ctx.generate_shadow(d, d_s, n, (0.0, 0.9 * float(verts[:, 1].max()), 0.0), ext * 5e-7)
out = []
for flags in (64, 0):
    ctx.set_debug_flags(flags)
    for kind, fn in (("camera", lambda: sc.intersect_device_fresh(d, n, 1e30)), ("shadow", lambda: sc.occluded_device(d_s, n, d_occ))):
        ms = []
        for p in range(9):
            fn(); ctx.synchronize()
            if p >= 3: ms.append(ctx.time_last_ms())
        out.append(f"{'unprobed' if flags else 'probed  '} {kind} {n / np.median(ms) / 1e3:8.0f} MRays/s ({np.median(ms) * 1e3:6.1f} us)")
print(name, side, "pin", os.environ.get("TBVH_COHERENT_TUNER"), " | ".join(out), "hits", int((np.frombuffer(b"", np.uint8)).sum()))
