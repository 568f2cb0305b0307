"""Stress of tbvh_pin_host: arrays of random size and alignment are pinned, traced (packed: straight from the caller's memory by DMA; 128-byte stride: packed by
the host threads), unpinned and freed, with unpinned queries on fresh arrays (which may land on formerly pinned addresses) in between; every result compared
with the staged path's.  A GPU memory fault aborts the process: the last line printed says how far it got."""
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
    off = int(rng.integers(0, 64)) * 16                      # 16-byte aligned, mostly not page aligned
    raw = np.zeros(n * 64 + off + 64, np.uint8)
    packed = raw[off:off + n * 64].view(tb.RAY_DTYPE)
    packed[:] = rays
    ctx.pin_host(packed)
    got = sc.Intersect(packed)
    occ = sc.IsOccluded(packed)
    ctx.unpin_host(packed)
    bad += int(not np.array_equal(got.view(np.uint8), want.view(np.uint8)))
    del packed, raw
    for _ in range(3):                                       # fresh, unpinned arrays right after the free
        m = int(rng.choice([1, 64, 777, 20000]))
        r2 = R.random_rays(m, (0, 0, 0), (10, 10, 10), seed=1000 + it)
        a = sc.Intersect(r2.copy()); b = sc.IsOccluded(r2.copy())
    if it % 20 == 0:
        print(f"iteration {it}: ok so far, {bad} mismatches", flush=True)
print(f"done: {iters} iterations, {bad} mismatches")
ctx.close()
