"""Host encode vs device conversion of the same BVH2 (tbvh_convert_bvh2_device), Bistro stand-in."""
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
verts, label = scenes.get(name)
ctx = tb.Context(0)
t0 = time.perf_counter(); host = tb.HostBVH(verts, tb.LAYOUT_CWBVH); t_host = time.perf_counter() - t0
n2, pi = host.bvh2_nodes(), host.bvh2_prim_idx()
t0 = time.perf_counter(); up = tb.BVH8_CWBVH(ctx).Upload(host.blob(0, np.uint32, 4), host.blob(1, np.uint32, 4)); ctx.synchronize(); t_up = time.perf_counter() - t0
for it in range(3):
    t0 = time.perf_counter(); dev = tb.BVH8_CWBVH(ctx).ConvertFromBVH2(n2, pi, verts); t_call = time.perf_counter() - t0
    ms = ctx.time_last_ms()
    print(f"{label}: BVH2 {n2.nbytes // 32} nodes -> CWBVH {dev.device_bytes / 1e6:.0f} MB: device conversion {ms:.2f} ms on the GPU, {t_call * 1e3:.1f} ms wall incl. uploading "
          f"{(n2.nbytes + pi.nbytes + verts.nbytes) / 1e6:.0f} MB of BVH2 + vertices  (host build+collapse+encode {t_host * 1e3:.0f} ms, blob upload {t_up * 1e3:.1f} ms)", flush=True)
    if it < 2: dev.free()
side = 2048; n = side * side
cams = scenes.SPONZA_CAMERAS if name == "sponza" else scenes.STREET_CAMERAS
cam = R.camera(*cams[0], side, side, 1, 1)
d = ctx.malloc(n * 64)
for nm, sc in (("host-encoded", up), ("device-converted", dev)):
    ts = []
    for p in range(4):
        ctx.generate_primary(cam, d, 0, n); sc.intersect_device(d, n); ts.append(ctx.time_last_ms())
    print(f"  trace {n} camera rays through the {nm} blob: {np.mean(ts[1:]):.3f} ms")
ctx.close()
