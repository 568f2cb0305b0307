"""Golden vectors generated from the reference itself (oracle/make_golden.py): the oracle must
reproduce BVH::Intersect / IsOccluded bit for bit on the stored scene + rays, both on its own
BVH and by traversing the stored reference-encoded blobs."""
import glob
import os

import numpy as np
import pytest

import tinybvh_amd as tb
from oracle_lib import compare_hits

HIT_DTYPE = np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("prim", "<u4")])


def hitrec(a):
    """(n, 4) uint32 -> structured t,u,v,prim view."""
    return np.ascontiguousarray(a).view(HIT_DTYPE).reshape(-1)


GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def test_fixtures_exist():
    assert len(GOLDEN) >= 2


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_on_reference_split_leaf_bvh2(oracle_ref, path):
    """The BVH2 the reference converts its CWBVH from (leaves of at most 3 triangles; the device conversion's input
    fixture) is a valid BVH2: the restated BVH::Intersect on it returns the reference's hit records."""
    g = np.load(path)
    verts, rays, hits = g["verts"], g["rays"], hitrec(g["hits"])
    for k in (0, 1):
        n2, pi = g[f"bvh2s3_nodes_{k}"], g[f"bvh2s3_idx_{k}"].reshape(-1)
        assert int(n2[:, 7].max()) <= 3
        got = oracle_ref.bvh2_intersect(n2, pi, verts, rays)
        c = compare_hits(got, hits)
        assert c["hitmiss"] == 0 and c["prim_real"] == 0 and c["t_bad"] == 0 and c["tie"] <= 4, c


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_reproduces_reference_hits(oracle_ref, path):
    g = np.load(path)
    verts, rays, hits = g["verts"], g["rays"], hitrec(g["hits"])
    h = tb.HostBVH(verts, tb.LAYOUT_BVH2_WALD)
    got = oracle_ref.bvh2_intersect(h.bvh2_nodes(), h.bvh2_prim_idx(), verts, rays)
    c = compare_hits(got, hits)
    assert c["hitmiss"] == 0 and c["prim_real"] == 0 and c["t_bad"] == 0 and c["uv_bad"] == 0 and c["tie"] <= 1, c
    assert c["bit_identical"] == c["same_prim"], c
    occ = oracle_ref.bvh2_occluded(h.bvh2_nodes(), h.bvh2_prim_idx(), verts, g["shadow_rays"])
    assert np.array_equal(occ, g["occluded"])


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
@pytest.mark.parametrize("hq", [0, 1])
def test_oracle_mirrors_reproduce_reference_mirrors_on_reference_blobs(oracle_ref, path, hq):
    g = np.load(path)
    verts, rays = g["verts"], g["rays"]
    got4 = oracle_ref.bvhgpu_intersect(g[f"bvhgpu_nodes_{hq}"], g[f"bvhgpu_idx_{hq}"], verts, rays)
    got6 = oracle_ref.bvh4_intersect(g[f"bvh4_{hq}"], rays)
    got9 = oracle_ref.cwbvh_intersect(g[f"cwbvh_nodes_{hq}"], g[f"cwbvh_tris_{hq}"], rays)
    for got, key in ((got4, f"mirror4_{hq}"), (got6, f"mirror6_{hq}")):
        for f in ("t", "u", "v", "prim"):
            assert np.array_equal(got[f].view(np.uint32), hitrec(g[key])[f].view(np.uint32)), (key, f)
    # CWBVH mirror: identical wherever a triangle was accepted; on a miss with finite tmax the
    # reference mirror zeroes u,v,prim (tiny_bvh.h:7148-7149) while the contract (and the
    # restatement) leaves the record untouched
    m9 = hitrec(g[f"mirror9_{hq}"])
    changed = got9["t"] != rays["t"]
    for f in ("t", "u", "v", "prim"):
        assert np.array_equal(got9[f][changed].view(np.uint32), m9[f][changed].view(np.uint32)), f
    assert np.array_equal(got9["t"], m9["t"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
@pytest.mark.parametrize("hq", [0, 1])
def test_hip_kernels_on_reference_blobs_reproduce_reference_hits(ctx, path, hq):
    """The drop-in path end to end on the GPU box: reference-encoded blobs in, the reference's
    own BVH::Intersect hit records out."""
    g = np.load(path)
    verts, rays, hits = g["verts"], g["rays"], hitrec(g["hits"])
    scenes_ = [
        tb.BVH_GPU(ctx).Upload(g[f"bvhgpu_nodes_{hq}"], g[f"bvhgpu_idx_{hq}"], verts),
        tb.BVH4_GPU(ctx).Upload(g[f"bvh4_{hq}"]),
        tb.BVH8_CWBVH(ctx).Upload(g[f"cwbvh_nodes_{hq}"], g[f"cwbvh_tris_{hq}"]),
    ]
    for sc in scenes_:
        got = sc.Intersect(rays.copy())
        c = compare_hits(got, hits)
        assert c["hitmiss"] + c["prim_real"] <= 1 and c["t_bad"] == 0 and c["uv_bad"] == 0 and c["tie"] <= 2 and c["onsurf"] <= 2, (sc.layout, c)
        assert c["bit_identical"] == c["same_prim"], (sc.layout, c)
        occ = sc.IsOccluded(g["shadow_rays"])
        assert int((occ != g["occluded"]).sum()) <= 1, sc.layout
