"""HIP engine vs the reference's own OpenCL kernels on the same GPU, same blobs, same rays
(BASELINE config 2: "Sponza BVH_GPU 1 M primary rays on 1 x MI355X vs reference traverse_bvh2.cl",
plus BVH4_GPU and CWBVH).  Blobs come from the library's host builder (reference formats), so both
sides traverse byte-identical data.  Needs oracle/_ref/libtinybvh_refocl.so (built where the
reference checkout exists; it travels to the GPU box inside the repo snapshot)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import tinybvh_amd as tb  # noqa: E402
from tinybvh_amd import rays as R  # noqa: E402
from tinybvh_amd import scenes  # noqa: E402
from oracle_lib import ReferenceOpenCL, compare_hits  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="sponza")
ap.add_argument("--side", type=int, default=1024)
ap.add_argument("--kind", default="primary")
ap.add_argument("--out", default="")
a = ap.parse_args()
verts, label = scenes.get(a.scene)
cams = scenes.SPONZA_CAMERAS if a.scene == "sponza" else scenes.STREET_CAMERAS
cam = R.camera(*cams[0], a.side, a.side, 1, 1)
n = a.side * a.side
ocl = ReferenceOpenCL()
print(f"scene: {label}; OpenCL device: {ocl.device}; {n} {a.kind} rays", flush=True)
ctx = tb.Context(0)
d = ctx.malloc(n * 64)
d_verts = ctx.malloc(verts.nbytes); ctx.to_device(d_verts, verts)
res = {}
for layout, name in ((5, "BVH_GPU"), (8, "BVH4_GPU"), (10, "BVH8_CWBVH")):
    sc = tb.LAYOUT_CLASSES[layout](ctx).Build(verts)
    h = sc.host
    ctx.generate_primary(cam, d, 0, n)
    if a.kind != "primary":
        sc.intersect_device(d, n)
        ctx.generate_bounce(d_verts, d, d, n, 7)
    rays = np.zeros(n, tb.RAY_DTYPE); ctx.from_device(rays, d)
    ms = []
    for p in range(4):
        sc.intersect_device_fresh(d, n, 1e30)
        t = ctx.time_last_ms()
        if p:
            ms.append(t)
    mine = np.zeros(n, tb.RAY_DTYPE); ctx.from_device(mine, d)
    if layout == 5:
        blobs = [h.blob(0, np.uint32, 16), h.blob(1, np.uint32, 1), verts]
    elif layout == 8:
        blobs = [h.blob(0, np.uint32, 4)]
    else:
        blobs = [h.blob(0, np.uint32, 4), h.blob(1, np.uint32, 4)]
    theirs, ref_ms = ocl.run(layout, blobs, rays, passes=3)
    # the .cl kernels always overwrite `hit` (miss = t 1e30) and use strict comparisons / native_recip:
    # compare loosely (hit/miss and prim; t to 1e-4) just to show both sides trace the same thing
    c = compare_hits(mine[: theirs.shape[0]], theirs, rtol=1e-4)
    hip = n / (np.mean(ms) * 1e-3) / 1e6
    ref = theirs.shape[0] / (ref_ms * 1e-3) / 1e6
    res[name] = {"hip_mrays": hip, "reference_opencl_mrays": ref, "speedup": hip / ref, "hits": c["hits"], "hitmiss": c["hitmiss"], "prim_mismatch": c["prim_mismatch"]}
    print(f"{name:11s} HIP {hip:8.1f} MRays/s   reference OpenCL ({'batch_ailalaine' if layout == 5 else 'batch_gpu4way' if layout == 8 else 'batch_cwbvh'}) {ref:8.1f} MRays/s   x{hip / ref:.2f}   "
          f"[agreement: hits {c['hits']}, hit/miss diff {c['hitmiss']}, prim diff {c['prim_mismatch']}]", flush=True)
    sc.free()
if a.out:
    json.dump({"scene": label, "rays": n, "kind": a.kind, "opencl_device": ocl.device, "results": res}, open(a.out, "w"), indent=1)
