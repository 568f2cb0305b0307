"""Device build (tbvh_build_device: LBVH + collapse + encode) vs the host SAH build: build time and the
trace rate through either tree."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tinybvh_amd as tb  # noqa: E402
from tinybvh_amd import rays as R  # noqa: E402
from tinybvh_amd import scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "bistro"
side = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
verts, label = scenes.get(name)
ctx = tb.Context(0)
t0 = time.perf_counter(); host = tb.BVH8_CWBVH(ctx).Build(verts); ctx.synchronize(); t_host = time.perf_counter() - t0
for it in range(3):
    t0 = time.perf_counter(); dev = tb.BVH8_CWBVH(ctx).BuildOnDevice(verts); t_call = time.perf_counter() - t0
    ms = ctx.time_last_ms()
    print(f"{label}: device build {ms:.2f} ms on the GPU ({verts.shape[0] // 3 / ms / 1e3:.0f} Mtris/s), {t_call * 1e3:.1f} ms wall incl. uploading {verts.nbytes / 1e6:.0f} MB of vertices; "
          f"{dev.device_bytes / 1e6:.0f} MB  (host SAH build + encode + upload: {t_host * 1e3:.0f} ms, {host.device_bytes / 1e6:.0f} MB)", flush=True)
    if it < 2: dev.free()
n = side * side
cams = scenes.SPONZA_CAMERAS if name == "sponza" else scenes.STREET_CAMERAS
cam = R.camera(*cams[0], side, side, 1, 1)
d = ctx.malloc(n * 64); d_b = ctx.malloc(n * 64); d_verts = ctx.malloc(verts.nbytes); ctx.to_device(d_verts, verts)
for nm, sc in (("host SAH tree", host), ("device LBVH tree", dev)):
    ts, tb_ = [], []
    for p in range(4):
        ctx.generate_primary(cam, d, 0, n); sc.intersect_device(d, n); ts.append(ctx.time_last_ms())
        ctx.generate_bounce(d_verts, d, d_b, n, 1); sc.intersect_device(d_b, n); tb_.append(ctx.time_last_ms())
    print(f"  {nm}: camera rays {n / np.mean(ts[1:]) / 1e3:.0f} MRays/s, bounce rays {n / np.mean(tb_[1:]) / 1e3:.0f} MRays/s")
ctx.close()
