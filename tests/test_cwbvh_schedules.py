"""Every schedule of the BVH8_CWBVH kernel returns the oracle's records — EXACTLY, ties included: the library's tie rule (smaller prim at
exactly equal t, tinybvh_amd/csrc/device_common.h: hit_wins) makes the result independent of the order in which a schedule tests
triangles, so the deferred-triangle schedule (which tests them in another order than the CPU mirror), the strict one, and the kernels that
split their last rays must all agree bit for bit with oracle/tbvh_oracle.c under the same rule, and with each other byte for byte.
Variants (tbvh_set_variant; kernels_cwbvh.hip): 72 strict, 52 coherent (deferred + gated), 75 / 88 the same two with split rays whatever
the batch size, 59 / 61 the instrumented kernels.  Reference semantics: BVH::Intersect / IsOccluded, tiny_bvh.h:3222-3304, 3382-3453."""
import numpy as np
import pytest

import tinybvh_amd as tb
from tinybvh_amd import rays as R
from tinybvh_amd import scenes
from oracle_lib import compare_hits


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 52, 72, 75, 88, 59, 61])
def test_schedule_variant_matches_the_oracle(ctx, oracle_ties, variant):
    verts = scenes.soup(20_000, seed=5)
    sc = tb.BVH8_CWBVH(ctx).Build(verts)
    rng = R.random_rays(150_000, (0, 0, 0), (10, 10, 10), seed=3)
    cam = R.primary(R.camera((-3.0, 5.0, -4.0), (0.6, -0.2, 0.75), 256, 256, 1, 1))
    h = sc.host
    for rays in (rng, cam):
        want = oracle_ties.bvh2_intersect(h.bvh2_nodes(), h.bvh2_prim_idx(), verts, rays)
        sc.set_variant(0)
        base = sc.Intersect(rays.copy())
        sc.set_variant(variant)
        got = sc.Intersect(rays.copy())
        c = compare_hits(got, want)
        assert c["hits"] > 1000 and c["hitmiss"] == 0 and c["prim_real"] == 0 and c["t_bad"] == 0 and c["uv_bad"] == 0, (variant, c)
        assert c["tie"] == 0 and c["onsurf"] <= max(4, c["n"] // 5000), (variant, c)
        assert c["bit_identical"] == c["same_prim"], (variant, c)
        assert np.array_equal(got.view(np.uint8), base.view(np.uint8)), variant          # and the default kernel's, byte for byte
        if variant not in (59, 61):                     # (the instrumented kernels exist for Intersect only)
            occ = sc.IsOccluded(rays.copy())
            assert int((occ.astype(bool) != (want["t"] < 1e30)).sum()) <= 2
    sc.free()


@pytest.mark.gpu
def test_unknown_variants_are_refused(ctx):
    verts = scenes.soup(500, seed=1)
    for cls in (tb.BVH_GPU, tb.BVH4_GPU, tb.BVH8_CWBVH):
        sc = cls(ctx).Build(verts)
        with pytest.raises(tb.TbvhError):
            sc.set_variant(47)
        sc.set_variant(0)
        sc.free()


@pytest.mark.gpu
@pytest.mark.parametrize("packed", [0, 64, 1024, 10**9])
def test_node_placement_does_not_change_a_record(ctx, oracle_ties, packed):
    """tbvh_cwbvh_set_hybrid: priority-ordered nodes, the first `packed` at 80 bytes, the others one per 128-byte line, triangle records at 64
    bytes; the incoherent flavor of the kernel (non-temporal ray records).  Same records, byte for byte; a refit keeps the copies current."""
    verts = scenes.atrium(60_000, seed=3)
    sc = tb.BVH8_CWBVH(ctx).Build(verts)
    rays = np.concatenate([R.random_rays(60_000, (-20, 0, -10), (20, 15, 10), seed=8), R.primary(R.camera(*scenes.SPONZA_CAMERAS[0], 256, 128, 1, 1))])
    base = sc.Intersect(rays.copy())
    occ0 = sc.IsOccluded(rays.copy())
    want = oracle_ties.bvh2_intersect(sc.host.bvh2_nodes(), sc.host.bvh2_prim_idx(), verts, rays)
    c = compare_hits(base, want)
    assert c["hits"] > 10_000 and c["hitmiss"] == 0 and c["prim_real"] == 0 and c["t_bad"] == 0 and c["tie"] == 0, c
    sc.set_hybrid(packed)
    sc.set_variant(90)                      # the incoherent flavor on the placed copies (hybrid nodes, 64-byte triangle records), whatever the batch
    got = sc.Intersect(rays.copy())
    assert np.array_equal(got.view(np.uint8), base.view(np.uint8))
    assert np.array_equal(sc.IsOccluded(rays.copy()), occ0)
    moved = verts.copy(); moved[:, 1] += np.float32(0.01) * np.sin(verts[:, 0]).astype(np.float32)
    sc.Refit(moved)
    got2 = sc.Intersect(rays.copy())
    sc.set_variant(0)
    sc.set_hybrid(-1)
    base2 = sc.Intersect(rays.copy())
    assert np.array_equal(got2.view(np.uint8), base2.view(np.uint8))
    assert not np.array_equal(base2["t"], base["t"])
    sc.free()
