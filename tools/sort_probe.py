"""How much would ordering a bounce batch help?  Trace the Bistro bounce-1 batch as generated, then the
same rays reordered on the host by (origin cell Morton, direction octant) — an upper bound for any
on-the-fly sorting / binning scheme (the reorder itself is not timed here)."""
import os
import sys

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
sc = tb.BVH8_CWBVH(ctx).Build(verts)
n = side * side
cams = scenes.SPONZA_CAMERAS if name == "sponza" else scenes.STREET_CAMERAS
cam = R.camera(*cams[0], side, side, 1, 1)
d = ctx.malloc(n * 64); d_b = ctx.malloc(n * 64); d_verts = ctx.malloc(verts.nbytes); ctx.to_device(d_verts, verts)
ctx.generate_primary(cam, d, 0, n); sc.intersect_device(d, n)
ctx.generate_bounce(d_verts, d, d_b, n, 1); ctx.synchronize()
rays = np.zeros(n, tb.RAY_DTYPE); ctx.from_device(rays, d_b)


def spread(v):
    v = v.astype(np.uint64) & 0x3ff
    v = (v | (v << 16)) & 0x30000ff
    v = (v | (v << 8)) & 0x300f00f
    v = (v | (v << 4)) & 0x30c30c3
    v = (v | (v << 2)) & 0x9249249
    return v


def trace(arr, tag):
    ctx.to_device(d_b, arr)
    ts = []
    for p in range(4):
        ctx.reset_hits(d_b, n, 1e30) if hasattr(ctx, "reset_hits") else ctx.to_device(d_b, arr)
        sc.intersect_device(d_b, n); ts.append(ctx.time_last_ms())
    print(f"  {tag}: {np.mean(ts[1:]):.3f} ms = {n / np.mean(ts[1:]) / 1e3:.0f} MRays/s", flush=True)


print(f"{label}: {n} bounce rays")
trace(rays, "as generated (pixel tile order)")
O = rays["O"]; lo, hi = O.min(0), O.max(0)
for bits in (4, 6, 8):
    q = np.clip(((O - lo) / np.maximum(hi - lo, 1e-20) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    cell = (spread(q[:, 0]) << 2) | (spread(q[:, 1]) << 1) | spread(q[:, 2])
    octant = ((rays["D"][:, 0] < 0).astype(np.uint64) << 2) | ((rays["D"][:, 1] < 0).astype(np.uint64) << 1) | (rays["D"][:, 2] < 0).astype(np.uint64)
    for nm, key in ((f"origin cell {bits} bits/axis, then octant", (cell << 3) | octant), (f"octant, then origin cell {bits} bits/axis", (octant << (3 * bits)) | cell)):
        order = np.argsort(key, kind="stable")
        trace(np.ascontiguousarray(rays[order]), "sorted by " + nm)
rng = np.random.default_rng(1)
trace(np.ascontiguousarray(rays[rng.permutation(n)]), "randomly shuffled (lower bound)")
ctx.close()
