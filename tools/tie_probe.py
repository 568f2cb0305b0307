"""How deterministic are the records on a scene with coplanar, overlapping, axis-aligned faces (the Sponza stand-in)?  Per layout: three runs of
the default kernels (split rays below 12 M rays) against each other and against a context created under TBVH_SPLIT_RAYS=0; differing rays
classified: t bitwise equal / within 16 ulps / farther."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tinybvh_amd as tb  # noqa: E402
from tinybvh_amd import rays as R  # noqa: E402
from tinybvh_amd import scenes  # noqa: E402


def classify(a, b):
    d = np.flatnonzero((a["prim"] != b["prim"]) | (a["t"] != b["t"]))
    if d.size == 0:
        return "identical"
    ta, tb_ = a["t"][d].view(np.int32).astype(np.int64), b["t"][d].view(np.int32).astype(np.int64)
    ulps = np.abs(ta - tb_)
    return f"{d.size} rays differ: t bitwise equal {int((ulps == 0).sum())}, within 16 ulps {int(((ulps > 0) & (ulps <= 16)).sum())}, farther {int((ulps > 16).sum())} (max {int(ulps.max())} ulps)"


verts, label = scenes.get("sponza")
cam = R.primary(R.camera(*scenes.SPONZA_CAMERAS[0], 512, 512, 1, 1))
rnd = R.random_rays(1 << 20, verts[:, :3].min(0), verts[:, :3].max(0), seed=12)
ctx = tb.Context(0)
os.environ["TBVH_SPLIT_RAYS"] = "0"
plain = tb.Context(0)
del os.environ["TBVH_SPLIT_RAYS"]
print(label)
for cls in (tb.BVH_GPU, tb.BVH4_GPU, tb.BVH8_CWBVH):
    a, b = cls(ctx).Build(verts), cls(plain).Build(verts)
    for name, rays in (("camera", cam), ("random", rnd)):
        runs = [a.Intersect(rays.copy()) for _ in range(3)]
        p = b.Intersect(rays.copy())
        print(f"{cls.__name__:11s} {name:7s} {rays.shape[0]} rays, {int((p['t'] < 1e30).sum())} hits: run 2 vs 1: {classify(runs[1], runs[0])}; run 3 vs 1: {classify(runs[2], runs[0])}; "
              f"split vs unsplit kernels: {classify(runs[0], p)}", flush=True)
    a.free(); b.free()
