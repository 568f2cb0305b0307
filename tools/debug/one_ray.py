#!/usr/bin/env python
"""One ray of a tests/test_random_large.py seed: what every path says about it.  usage: one_ray.py SEED RAY"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import tinybvh_amd as tb
from tinybvh_amd import rays as R
from oracle_lib import Oracle
import test_random_large as T

seed, ri = int(sys.argv[1]), int(sys.argv[2])
ctx = tb.Context(0)
orc = Oracle(tie_rule=1)
rng = np.random.default_rng(9000 + seed)
name, verts = T.make_scene(rng)
layout = T.LAYOUTS[int(rng.integers(0, 3))]
sc = tb.LAYOUT_CLASSES[layout](ctx).Build(verts)
host = sc.host
lo, hi = verts[:, :3].min(0), verts[:, :3].max(0)
n_min = int(rng.choice([800_000, 1_100_000, 1_600_000, 2_200_000]))
kind = ["camera", "bounce", "shadow", "random"][int(rng.integers(0, 4))]
assert kind == "camera"
cam = T.camera_rays(rng, lo, hi, n_min)
r = cam[ri:ri + 1].copy()
np.set_printoptions(precision=9, floatmode="unique")
print("ray", r["O"][0], r["D"][0], r["rD"][0])
print("oracle bvh2      ", orc.bvh2_intersect(host.bvh2_nodes(), host.bvh2_prim_idx(), verts, r)[["t", "u", "v", "prim"]])
if layout == tb.LAYOUT_CWBVH:
    nodes, tris = host.blob(0, np.float32, 4), host.blob(1, np.float32, 4)
    out, cnt = orc.cwbvh_intersect(nodes, tris, r, counts=True)
    print("oracle cwbvh     ", out[["t", "u", "v", "prim"]], cnt)
for v in (72, 92, 0):
    sc.set_variant(v)
    for nrep in (1, 64, 4096):
        rr = np.repeat(r, nrep)
        got = sc.Intersect(rr.copy())
        print(f"variant {v} x{nrep}", got[["t", "u", "v", "prim"]][0], "all same" if np.all(got.view(np.uint8).reshape(nrep, 64) == got.view(np.uint8).reshape(nrep, 64)[0]) else "VARIES")
# neighbours through the packet kernel, as in the batch: the 64-ray chunk the ray sits in
c0 = ri // 64 * 64
chunk = cam[c0:c0 + 64].copy()
for v in (72, 92):
    sc.set_variant(v)
    got = sc.Intersect(chunk.copy())
    print(f"variant {v} on the ray's chunk of 64:", got[["t", "u", "v", "prim"]][ri - c0])
for p in (143740, 143810):
    print("prim", p, verts.reshape(-1, 3, 4)[p, :, :3].tolist())
