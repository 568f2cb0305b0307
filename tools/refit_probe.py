"""Device BLAS refit timing (tbvh_refit): Bistro / Sponza stand-in, vertices already on the device,
then trace the camera batch through the refitted blob."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tinybvh_amd as tb  # noqa: E402
from tinybvh_amd import rays as R  # noqa: E402
from tinybvh_amd import scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="bistro")
ap.add_argument("--layouts", default="10,5")
ap.add_argument("--side", type=int, default=2048)
a = ap.parse_args()
verts, label = scenes.get(a.scene)
cams = scenes.SPONZA_CAMERAS if a.scene == "sponza" else scenes.STREET_CAMERAS
cam = R.camera(*cams[0], a.side, a.side, 1, 1)
n = a.side * a.side
ctx = tb.Context(0)
d_rays = ctx.malloc(n * 64)
ext = float((verts[:, :3].max(0) - verts[:, :3].min(0)).max())
v2 = verts.copy(); v2[:, 1] += (0.002 * ext * np.sin(verts[:, 0] * (20.0 / ext))).astype(np.float32)   # a gentle wave
d_v = ctx.malloc(v2.nbytes); ctx.to_device(d_v, v2)
for layout in [int(x) for x in a.layouts.split(",")]:
    sc = tb.LAYOUT_CLASSES[layout](ctx).Build(verts)
    ctx.generate_primary(cam, d_rays, 0, n); sc.intersect_device(d_rays, n); t0 = ctx.time_last_ms()
    ms = []
    for it in range(4):
        sc.Refit((d_v, v2.shape[0] // 3), on_device=True); ctx.synchronize(); ms.append(ctx.time_last_ms())
    ctx.generate_primary(cam, d_rays, 0, n); sc.intersect_device(d_rays, n); t1 = ctx.time_last_ms()
    print(f"{label}: layout {layout}: {verts.shape[0] // 3} tris, {sc.device_bytes / 1e6:.0f} MB on device; refit {ms[0]:.3f} ms first call, "
          f"{np.mean(ms[1:]):.3f} ms after = {verts.shape[0] // 3 / np.mean(ms[1:]) / 1e3:.0f} Mtris/s; trace {n} camera rays {t0:.3f} ms before, {t1:.3f} ms after", flush=True)
    sc.free()
ctx.close()
