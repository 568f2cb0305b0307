"""Small batches from TWO contexts on one device: each context has its own stream, ray pool and timing events, so independent batches issued
alternately overlap on the GPU — one launch's tail (a few waves finishing the longest rays) is filled by the other's body.  Same scene data
uploaded to both contexts; K launches each, wall clock over all of them against the same 2 K launches from one context."""
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
side = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
layout = int(sys.argv[3]) if len(sys.argv) > 3 else 10
verts, label = scenes.get(name)
cams = scenes.SPONZA_CAMERAS if name == "sponza" else scenes.STREET_CAMERAS
n = side * side
cam = R.camera(*cams[0], side, side, 1, 1)
host = tb.HostBVH(verts, layout)
ctxs = [tb.Context(0), tb.Context(0)]
scs, bufs = [], []
for c in ctxs:
    sc = tb.LAYOUT_CLASSES[layout](c)
    sc.host = host
    if layout == 10:
        sc.Upload(host.blob(0, np.uint32, 4), host.blob(1, np.uint32, 4))
    elif layout == 8:
        sc.Upload(host.blob(0, np.uint32, 4))
    else:
        sc.Upload(host.blob(0, np.uint32, 16), host.blob(1, np.uint32, 1), host.verts)
    d = c.malloc(n * 64); c.generate_primary(cam, d, 0, n); c.synchronize()
    scs.append(sc); bufs.append(d)
K = 50


def run(order):
    for c in ctxs:
        c.synchronize()
    t0 = time.perf_counter()
    for i in order:
        scs[i].intersect_device_fresh(bufs[i], n, 1e30)
    for c in ctxs:
        c.synchronize()
    return (time.perf_counter() - t0) / len(order)


run([0, 1] * 5)
one = run([0] * (2 * K))
two = run([0, 1] * K)
print(f"{label}, layout {layout}, {n} camera rays per launch: one context {one * 1e3:.3f} ms per launch = {n / one / 1e6:.0f} MRays/s; "
      f"two contexts alternating {two * 1e3:.3f} ms per launch = {n / two / 1e6:.0f} MRays/s (x{one / two:.2f})")
