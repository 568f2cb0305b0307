"""What the per-query HIP-event pair costs a caller that issues many small queries back to back (tbvh_set_timing): wall clock per query of
200 launches, timing on / off, for a few batch sizes on the Sponza stand-in (BVH8_CWBVH)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinybvh_amd as tb
from tinybvh_amd import rays as R, scenes

verts, label = scenes.get("sponza")
ctx = tb.Context(0)
sc = tb.BVH8_CWBVH(ctx).Build(verts)
side = 1024
rays = R.primary(R.camera(*scenes.SPONZA_CAMERAS[0], side, side, 1, 1))
d = ctx.malloc(rays.nbytes); ctx.to_device(d, rays)
print(label)
for n in (16_384, 65_536, 262_144, 1_048_576):
    row = []
    for rep in range(3):
        for on in (True, False):
            ctx.set_timing(on)
            for _ in range(20): sc.intersect_device_fresh(d, n, 1e30)
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(200): sc.intersect_device_fresh(d, n, 1e30)
            ctx.synchronize()
            row.append((on, (time.perf_counter() - t0) / 200 * 1e6))
    on_ = np.median([t for o, t in row if o]); off_ = np.median([t for o, t in row if not o])
    print(f"{n:8d} rays per query: {on_:7.1f} us timed, {off_:7.1f} us untimed ({(off_ / on_ - 1) * 100:+.1f} %)")
ctx.set_timing(True)
