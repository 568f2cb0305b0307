#!/usr/bin/env python
"""Generates tests/golden/*.npz FROM THE REFERENCE ITSELF (needs oracle/_ref, i.e. the
reference checkout at /root/reference at build time; run in the build container, commit the
output).  Each fixture is self-contained: the triangles, the ray records, and the hit
records / occlusion flags produced by tinybvh's own BVH::Intersect / BVH::IsOccluded
(tiny_bvh.h:3222, 3382) on a BVH::Build tree, plus the reference-encoded layout blobs
(BVH_GPU / BVH4_GPU / BVH8_CWBVH, Build and BuildHQ) so the drop-in path — reference blobs in,
hit records out — can be replayed on a machine that has neither the reference nor its meshes.

    python oracle/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import tinybvh_amd as tb  # noqa: E402
from tinybvh_amd import rays as R  # noqa: E402
from tinybvh_amd import scenes  # noqa: E402
from oracle_lib import Reference  # noqa: E402


def fixture(name, verts, ref):
    lo, hi = verts[:, :3].min(0), verts[:, :3].max(0)
    c = (lo + hi) / 2; ext = float((hi - lo).max())
    cam = R.camera(c + np.array([0.15 * ext, 0.25 * ext, 1.3 * ext], np.float32), (-0.08, -0.16, -1.0), 64, 48, 1, 1)
    prim = R.primary(cam)
    rnd = R.random_rays(3072, lo - 0.05 * ext, hi + 0.05 * ext, seed=17)
    short = R.random_rays(1024, lo, hi, seed=18, tmax=np.float32(0.15 * ext))
    rays = np.concatenate([prim, rnd, short])
    out = {"verts": verts, "rays": rays}
    for hq in (0, 1):
        rs = ref.build(verts, hq=bool(hq), threaded=False)
        hits = rs.intersect(1, rays)
        if hq == 0:
            out["hits"] = hits.copy()
            sh = R.shadow(hits, hi * 1.05, ext * 5e-7)
            out["shadow_rays"] = sh
            out["occluded"] = rs.occluded(1, sh)
        else:
            out["hits_hq_tree"] = hits  # same records up to ties; kept to show builder independence
        out[f"bvhgpu_nodes_{hq}"] = rs.blob(5, 0, np.uint32, 16); out[f"bvhgpu_idx_{hq}"] = rs.blob(5, 1, np.uint32, 1)
        out[f"bvh4_{hq}"] = rs.blob(8, 0, np.uint32, 4)
        out[f"cwbvh_nodes_{hq}"] = rs.blob(10, 0, np.uint32, 4); out[f"cwbvh_tris_{hq}"] = rs.blob(10, 1, np.uint32, 4)
        # the BVH2 (leaves of at most 3 triangles) the reference converted that CWBVH from: input of the device-side conversion
        out[f"bvh2s3_nodes_{hq}"] = rs.blob(110, 0, np.uint32, 8); out[f"bvh2s3_idx_{hq}"] = rs.blob(110, 1, np.uint32, 1)
        out[f"mirror4_{hq}"] = rs.intersect(5, rays); out[f"mirror6_{hq}"] = rs.intersect(8, rays); out[f"mirror9_{hq}"] = rs.intersect(10, rays)
    # hit records are stored as their last 16 bytes (t, u, v, prim) to keep the fixtures small
    for k in list(out):
        if k.startswith(("hits", "mirror")):
            out[k] = np.ascontiguousarray(out[k]).view(np.uint32).reshape(-1, 16)[:, 12:16].copy()
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print(name, verts.shape[0] // 3, "tris", rays.shape[0], "rays", int((out['hits'][:, 0].view(np.float32) < 1e30).sum()), "hits", os.path.getsize(path) // 1024, "KiB")


def main():
    ref = Reference()
    print(ref.lib.ref_version().decode())
    fixture("soup_2k", scenes.soup(2048, seed=7), ref)
    fixture("atrium_6k", scenes.atrium(6000, seed=1), ref)
    b = "/root/reference/testdata/suzanne.bin"
    if os.path.exists(b):
        v = scenes.load_bin(b)
        fixture("suzanne_decimated", np.ascontiguousarray(v.reshape(-1, 3, 4)[::4].reshape(-1, 4)), ref)  # every 4th triangle of the reference's suzanne.bin


if __name__ == "__main__":
    main()
