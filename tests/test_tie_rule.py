"""The library's tie rule (tinybvh_amd/csrc/device_common.h: hit_wins; oracle/tbvh_oracle.c: orc_set_tie_rule 1) against the reference's
(tiny_bvh.h:1656: a candidate at t == hit.t is accepted, so the LATER test wins), on the CPU restatement:
  * under the reference's rule the four traversals of one scene (BVH::Intersect and the three layout mirrors) disagree on the winner of an
    exact tie — the visit-order class of SURVEY.md par. 7;
  * under the library's rule they all report the same triangle: the smaller prim;
  * the two rules differ in nothing but the prim (and u, v) of exact ties."""
import numpy as np

import tinybvh_amd as tb
from tinybvh_amd import rays as R
from tinybvh_amd import scenes


def all_traversals(orc, verts, rays):
    out = {}
    h2 = tb.HostBVH(verts, tb.LAYOUT_BVH2_WALD)
    out["bvh2"] = orc.bvh2_intersect(h2.bvh2_nodes(), h2.bvh2_prim_idx(), verts, rays)
    h = tb.HostBVH(verts, tb.LAYOUT_BVH_GPU)
    out["bvh_gpu"] = orc.bvhgpu_intersect(h.blob(0, np.uint32, 16), h.blob(1, np.uint32, 1), verts, rays)
    h = tb.HostBVH(verts, tb.LAYOUT_BVH4_GPU)
    out["bvh4_gpu"] = orc.bvh4_intersect(h.blob(0, np.uint32, 4), rays)
    h = tb.HostBVH(verts, tb.LAYOUT_CWBVH)
    out["cwbvh"] = orc.cwbvh_intersect(h.blob(0, np.uint32, 4), h.blob(1, np.uint32, 4), rays)
    return out


def test_every_triangle_twice(oracle, oracle_ref):
    base = scenes.soup(3000, seed=17)
    verts = np.ascontiguousarray(np.concatenate([base, base]))
    ntri = base.shape[0] // 3
    rays = R.random_rays(20_000, (0, 0, 0), (10, 10, 10), seed=4)
    lib = all_traversals(oracle, verts, rays)
    ref = all_traversals(oracle_ref, verts, rays)
    hit = lib["bvh2"]["t"] < 1e30
    assert hit.sum() > 3000
    for name, r in lib.items():   # the library's rule: one answer, the smaller copy, whatever the traversal
        assert np.array_equal(r.view(np.uint8), lib["bvh2"].view(np.uint8)), name
        assert np.all(r["prim"][hit] < ntri), name
    differ = 0
    for name, r in ref.items():   # the reference's rule: same t, a copy of the same triangle, WHICH copy depends on the traversal
        assert np.array_equal(r["t"], lib["bvh2"]["t"]), name
        assert np.array_equal(r["prim"][hit] % ntri, lib["bvh2"]["prim"][hit] % ntri), name
        differ += int((r["prim"][hit] != ref["bvh2"]["prim"][hit]).sum())
    assert differ > 100   # the reference's own traversals do disagree on this scene


def test_the_rules_differ_on_exact_ties_only(oracle, oracle_ref):
    verts = scenes.atrium(40_000, seed=2)      # coplanar overlapping faces: some exact ties, mostly none
    h = tb.HostBVH(verts, tb.LAYOUT_BVH2_WALD)
    rays = R.random_rays(40_000, verts[:, :3].min(0), verts[:, :3].max(0), seed=3)
    a = oracle.bvh2_intersect(h.bvh2_nodes(), h.bvh2_prim_idx(), verts, rays)
    b = oracle_ref.bvh2_intersect(h.bvh2_nodes(), h.bvh2_prim_idx(), verts, rays)
    assert np.array_equal(a["t"].view(np.uint32), b["t"].view(np.uint32))
    same = a["prim"] == b["prim"]
    assert np.array_equal(a[same].view(np.uint8), b[same].view(np.uint8))
    assert np.all(a["prim"][~same] < b["prim"][~same])


def test_a_hit_already_in_the_record_is_not_a_tie_partner(oracle):
    """The first hit of a traversal is accepted at t == hit.t whatever prim the record carried in (the reference accepts t <= hit.t)."""
    verts = scenes.soup(2000, seed=5)
    h = tb.HostBVH(verts, tb.LAYOUT_BVH2_WALD)
    rays = R.random_rays(5000, (0, 0, 0), (10, 10, 10), seed=9)
    first = oracle.bvh2_intersect(h.bvh2_nodes(), h.bvh2_prim_idx(), verts, rays)
    again = first.copy()
    again["prim"] = 0; again["u"] = 0.5; again["v"] = 0.25      # tmax = exactly the hit distance, a smaller "prim" in the record
    got = oracle.bvh2_intersect(h.bvh2_nodes(), h.bvh2_prim_idx(), verts, again)
    hit = first["t"] < 1e30
    assert np.array_equal(got[hit].view(np.uint8), first[hit].view(np.uint8))
