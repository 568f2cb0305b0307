"""Pins the C restatement (oracle/tbvh_oracle.c) against the REAL reference
(oracle/_ref/libtinybvh_ref.so, built from tiny_bvh.h where the checkout exists): on the
reference's own trees and blobs the restated traversals reproduce BVH::Intersect /
IsOccluded and the three layout mirrors bit for bit."""
import os

import numpy as np
import pytest

import tinybvh_amd as tb
from tinybvh_amd import rays as R
from tinybvh_amd import scenes
from oracle_lib import compare_hits


def batches(verts):
    lo, hi = verts[:, :3].min(0), verts[:, :3].max(0)
    c = (lo + hi) / 2; ext = float((hi - lo).max())
    cam = R.camera(c + np.array([0.1 * ext, 0.2 * ext, 1.4 * ext], np.float32), (-0.05, -0.12, -1.0), 96, 96, 1, 1)
    prim = R.primary(cam)
    rnd = R.random_rays(12_000, lo - 0.1 * ext, hi + 0.1 * ext, seed=5)
    return prim, rnd


def exact(got, want):
    for f in ("t", "u", "v", "prim", "inst"):
        assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), f


SCENES = ["soup", "atrium", "bunny"]


def get_scene(name):
    if name == "soup":
        return scenes.soup(8192, seed=7)
    if name == "atrium":
        return scenes.atrium(25_000, seed=1)
    p = "/root/reference/testdata/bunny.bin"
    if not os.path.exists(p):
        pytest.skip("reference test mesh not on this machine")
    return scenes.load_bin(p)


@pytest.mark.parametrize("scene", SCENES)
@pytest.mark.parametrize("hq", [False, True])
def test_restated_traversals_equal_the_reference(oracle, reference, scene, hq):
    verts = get_scene(scene)
    rs = reference.build(verts, hq=hq)
    n2, pi = rs.blob(1, 0, np.uint32, 8), rs.blob(1, 1, np.uint32, 1)
    for rays in batches(verts):
        want = rs.intersect(1, rays)
        assert (want["t"] < 1e30).sum() > 100
        exact(oracle.bvh2_intersect(n2, pi, verts, rays), want)                       # BVH::Intersect
        exact(oracle.bvhgpu_intersect(rs.blob(4, 0, np.uint32, 16), rs.blob(4, 1, np.uint32, 1), verts, rays), rs.intersect(4, rays))
        exact(oracle.bvh4_intersect(rs.blob(6, 0, np.uint32, 4), rays), rs.intersect(6, rays))
        # the CWBVH mirror zeroes u,v,prim of missed finite-tmax rays (tiny_bvh.h:7148); tmax is
        # 1e30 here so records are comparable field by field
        got = oracle.cwbvh_intersect(rs.blob(9, 0, np.uint32, 4), rs.blob(9, 1, np.uint32, 4), rays)
        ref9 = rs.intersect(9, rays)
        for f in ("t", "u", "v", "prim"):
            assert np.array_equal(got[f].view(np.uint32), ref9[f].view(np.uint32)), f
        # shadow rays
        sh = R.shadow(want, verts[:, :3].max(0) * 1.1, 1e-4)
        assert np.array_equal(oracle.bvh2_occluded(n2, pi, verts, sh), rs.occluded(1, sh))


def test_own_builder_gives_the_reference_hits(oracle, reference):
    """Hit records are builder independent (up to ties): the library's own BVH and the
    reference's BVH::Build agree ray by ray."""
    verts = scenes.atrium(25_000, seed=1)
    rs = reference.build(verts)
    h = tb.HostBVH(verts, tb.LAYOUT_BVH2_WALD)
    for rays in batches(verts):
        got = oracle.bvh2_intersect(h.bvh2_nodes(), h.bvh2_prim_idx(), verts, rays)
        c = compare_hits(got, rs.intersect(1, rays))
        assert c["hitmiss"] == 0 and c["prim_real"] == 0 and c["t_bad"] == 0 and c["tie"] <= 2, c
        assert c["bit_identical"] == c["same_prim"], c


def test_reference_counts_match_oracle_counts(oracle, reference):
    verts = scenes.atrium(25_000, seed=1)
    rs = reference.build(verts)
    rays = batches(verts)[0]
    for layout, fn in ((1, lambda: oracle.bvh2_intersect(rs.blob(1, 0, np.uint32, 8), rs.blob(1, 1, np.uint32, 1), verts, rays, counts=True)),
                       (4, lambda: oracle.bvhgpu_intersect(rs.blob(4, 0, np.uint32, 16), rs.blob(4, 1, np.uint32, 1), verts, rays, counts=True)),
                       (6, lambda: oracle.bvh4_intersect(rs.blob(6, 0, np.uint32, 4), rays, counts=True))):
        s, t = rs.counts(layout, rays)
        _, c = fn()
        assert (int(c[0]), int(c[1])) == (s, t), layout
