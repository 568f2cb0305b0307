"""Stress of tbvh_pinned_malloc: arrays of random size in page-locked memory of the library's are traced (packed: straight from there by DMA) and given back,
with queries and plain copies from fresh pageable arrays in between (the pattern that faulted the GPU when the library still page-locked CALLER memory with
hipHostRegister: DESIGN.md par. 0); every result compared with the pageable path's.  A GPU memory fault aborts the process: the last line printed says how far
it got."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tinybvh_amd as tb
from tinybvh_amd import rays as R, scenes

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
verts = scenes.soup(8192, seed=7)
ctx = tb.Context(0)
sc = tb.BVH8_CWBVH(ctx).Build(verts)
rng = np.random.default_rng(1)
bad = 0
for it in range(iters):
    n = int(rng.choice([1, 63, 64, 777, 4097, 20000, 33333, 200000, 1 << 20]))
    rays = R.random_rays(n, (0, 0, 0), (10, 10, 10), seed=it + 1)
    want = sc.Intersect(rays.copy())
    packed = ctx.pinned_array((n,), tb.RAY_DTYPE)
    packed[:] = rays
    got = sc.Intersect(packed)
    occ = sc.IsOccluded(packed)
    bad += int(not np.array_equal(got.view(np.uint8), want.view(np.uint8)))
    ctx.pinned_free(packed)
    del packed
    for _ in range(3):                                       # fresh pageable arrays right after: host queries and plain copies
        m = int(rng.choice([1, 64, 777, 20000, 300000]))
        r2 = R.random_rays(m, (0, 0, 0), (10, 10, 10), seed=1000 + it)
        a = sc.Intersect(r2.copy()); b = sc.IsOccluded(r2.copy())
        d = ctx.malloc(m * 64); ctx.to_device(d, r2.copy()); sc.intersect_device(d, m); back = np.zeros(m, tb.RAY_DTYPE); ctx.from_device(back, d); ctx.free(d)
    if it % 20 == 0:
        print(f"iteration {it}: ok so far, {bad} mismatches", flush=True)
print(f"done: {iters} iterations, {bad} mismatches")
ctx.close()
