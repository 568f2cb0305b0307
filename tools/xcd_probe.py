"""Would it pay to give every XCD its own spatial region of a bounce batch (one L2 per XCD: eight copies of the same hot
set today)?  The ray pool deals chunk c of the batch to stripe ((c mod 32) - 5 (c div 32)) mod 32, and a stripe's waves
all sit on XCD stripe mod 8.  So a host-side permutation of the 64-ray chunks decides which XCD traces which rays without
touching the kernels: as generated, Morton-sorted by origin, and both again with the chunks placed so that XCD x gets the
x-th contiguous eighth of the order."""
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
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 1
verts, label = scenes.get(name)
ctx = tb.Context(0)
sc = tb.BVH8_CWBVH(ctx).Build(verts)
n = side * side
cams = scenes.SPONZA_CAMERAS if name == "sponza" else scenes.STREET_CAMERAS
cam = R.camera(*cams[0], side, side, 1, 1)
d = ctx.malloc(n * 64); d_b = ctx.malloc(n * 64); d_verts = ctx.malloc(verts.nbytes); ctx.to_device(d_verts, verts)
ctx.generate_primary(cam, d, 0, n); sc.intersect_device(d, n)
for k in range(depth):
    ctx.generate_bounce(d_verts, d, d_b, n, 1 + k); ctx.synchronize()
    if k + 1 < depth:
        sc.intersect_device(d_b, n); d, d_b = d_b, d
rays = np.zeros(n, tb.RAY_DTYPE); ctx.from_device(rays, d_b)


def spread(v):
    v = v.astype(np.uint64) & 0x3ff
    v = (v | (v << 16)) & 0x30000ff
    v = (v | (v << 8)) & 0x300f00f
    v = (v | (v << 4)) & 0x30c30c3
    v = (v | (v << 2)) & 0x9249249
    return v


def per_xcd(arr):
    """chunks of `arr` placed so that XCD x traces the x-th contiguous eighth of arr"""
    nch = n // 64
    c = np.arange(nch)
    xcd = ((c % 32) - 5 * (c // 32)) % 8
    out = np.empty_like(arr)
    per = nch // 8
    a = arr.reshape(nch, 64)
    o = out.reshape(nch, 64)
    for x in range(8):
        slots = c[xcd == x]
        assert len(slots) == per
        o[slots] = a[x * per:(x + 1) * per]
    return out


def trace(arr, tag):
    ctx.to_device(d_b, arr)
    ts = []
    for p in range(5):
        ctx.reset_hits(d_b, n, 1e30)
        sc.intersect_device(d_b, n); ts.append(ctx.time_last_ms())
    print(f"  {tag}: {np.mean(ts[1:]):.3f} ms = {n / np.mean(ts[1:]) / 1e3:.0f} MRays/s", flush=True)


print(f"{label}: {n} bounce rays (depth {depth})")
trace(rays, "as generated (pixel tile order), chunks dealt over the XCDs")
trace(per_xcd(rays), "as generated, one contiguous eighth per XCD")
O = rays["O"]; lo, hi = O.min(0), O.max(0)
q = np.clip(((O - lo) / np.maximum(hi - lo, 1e-20) * 1024).astype(np.int64), 0, 1023)
cell = (spread(q[:, 0]) << 2) | (spread(q[:, 1]) << 1) | spread(q[:, 2])
srt = np.ascontiguousarray(rays[np.argsort(cell, kind="stable")])
trace(srt, "Morton-sorted by origin, chunks dealt over the XCDs")
trace(per_xcd(srt), "Morton-sorted by origin, one contiguous eighth (a region of space) per XCD")
ctx.close()
