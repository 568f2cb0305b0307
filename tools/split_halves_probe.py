"""Round 5, review item 9: would ONE small query gain from the library splitting it into two halves on two internal streams?  The A/B without building it:
the same camera batch (BASELINE config 2's 1 M rays on the Sponza stand-in; also 0.26 M and 2 M) traced (a) as one launch on one context and (b) as two
halves on two contexts of the same device, both enqueued before either is waited for; wall clock per query (synchronised after every query: an isolated
query, not a stream of them — that case is tools/two_context_overlap.py), best and median of 30."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tinybvh_amd as tb
from tinybvh_amd import rays as R, scenes

for name in ("sponza", "bistro"):
    verts, label = scenes.get(name)
    cams = scenes.cameras(name)
    for layout in (5, 10):
        host = tb.HostBVH(verts, layout)
        ctxs = [tb.Context(0), tb.Context(0)]
        scs = []
        for c in ctxs:
            sc = tb.LAYOUT_CLASSES[layout](c); sc.host = host
            if layout == 10:
                sc.Upload(host.blob(0, np.uint32, 4), host.blob(1, np.uint32, 4))
            else:
                sc.Upload(host.blob(0, np.uint32, 16), host.blob(1, np.uint32, 1), host.verts)
            scs.append(sc)
        for side in (512, 1024, 1448):
            side = side // 4 * 4
            n = side * side
            cam = R.camera(*cams[0], side, side, 1, 1)
            h = (n // 2) // 64 * 64
            d_all = ctxs[0].malloc(n * 64); ctxs[0].generate_primary(cam, d_all, 0, n)
            d_a = ctxs[0].malloc(h * 64); ctxs[0].generate_primary(cam, d_a, 0, h)
            d_b = ctxs[1].malloc((n - h) * 64); ctxs[1].generate_primary(cam, d_b, h, n - h)
            for c in ctxs:
                c.synchronize()
            one, two = [], []
            for r in range(34):
                t0 = time.perf_counter(); scs[0].intersect_device_fresh(d_all, n, 1e30); ctxs[0].synchronize(); t1 = time.perf_counter()
                scs[0].intersect_device_fresh(d_a, h, 1e30); scs[1].intersect_device_fresh(d_b, n - h, 1e30); ctxs[0].synchronize(); ctxs[1].synchronize(); t2 = time.perf_counter()
                if r >= 4:
                    one.append(t1 - t0); two.append(t2 - t1)
            f = lambda x: f"{np.min(x) * 1e6:7.0f} / {np.median(x) * 1e6:7.0f} us = {n / np.median(x) / 1e6:6.0f} MRays/s"
            print(f"{label[:28]:28s} layout {layout:2d} {n:8d} camera rays: one launch {f(one)}   two halves at once {f(two)}   x{np.median(one) / np.median(two):.2f}", flush=True)
            ctxs[0].free(d_all); ctxs[0].free(d_a); ctxs[1].free(d_b)
        for c in ctxs:
            c.close()
