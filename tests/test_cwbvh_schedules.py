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
@pytest.mark.parametrize("variant", [0, 72, 75, 88, 91])
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


@pytest.mark.gpu
def test_upload_refuses_a_node_array_that_is_not_a_tree(ctx):
    """In-range indices keep every read in bounds, but only a TREE keeps the traversal finite: a child range shared by two parents can close a
    cycle, and a cyclic blob would be a launch that never ends.  The upload validation walks the tree from the root and refuses it
    (TBVH_E_FORMAT) for all three layouts' node arrays; so the hybrid copy's renumbering (cwbvh_priority_order) only ever sees strict trees."""
    verts = scenes.soup(3000, seed=4)
    h = tb.HostBVH(verts, tb.LAYOUT_CWBVH)
    nodes, tris = h.blob(0, np.uint32, 4).copy(), h.blob(1, np.uint32, 4)
    nd = nodes.reshape(-1, 5, 4)
    inner = np.where((nd[:, 0, 3] >> 24) != 0)[0]
    pc = np.array([bin(int(x)).count("1") for x in (nd[inner, 0, 3] >> 24)])
    a = int(inner[np.argmax(pc)]); b = next(int(i) for i in inner if i != a and i != 0)
    nd[b, 1, 0] = nd[a, 1, 0]               # b's children are now (a prefix of) a's: every index in range, the range shared
    with pytest.raises(tb.TbvhError, match="not a strict tree"):
        tb.BVH8_CWBVH(ctx).Upload(nodes, tris)
    bad = h.blob(0, np.uint32, 4).copy().reshape(-1, 5, 4)
    k = int(inner[1]); slot = next(s for s in range(8) if (int(bad[k, 0, 3]) >> (24 + s)) & 1)
    bad[k, 0, 3] &= np.uint32(~(1 << (24 + slot)) & 0xFFFFFFFF)      # interior mask and meta bytes disagree
    with pytest.raises(tb.TbvhError, match="disagree"):
        tb.BVH8_CWBVH(ctx).Upload(bad.reshape(-1, 4), tris)
    h2 = tb.HostBVH(verts, tb.LAYOUT_BVH_GPU)
    n2 = h2.blob(0, np.uint32, 16).copy()
    # BVH_GPU node: lmin, left, lmax, right, rmin, triCount, rmax, firstTri (tiny_bvh.h:1095-1105): point some interior node's right child at the root
    inner2 = np.where(n2[:, 11] == 0)[0]
    n2[inner2[3], 7] = 0
    with pytest.raises(tb.TbvhError, match="not a tree"):
        tb.BVH_GPU(ctx).Upload(n2, h2.blob(1, np.uint32, 1), h2.verts)


@pytest.mark.gpu
def test_small_scene_batches_are_probed_for_the_packet_kernel(ctx, oracle_ties):
    """Scenes under 48 MB (round 6): a batch of 768 k rays or more is probed; a coherent one is traced by the coherent flavor the scene's tuner is trying or has
    settled on (strict per-lane or one traversal per wave), an incoherent one by the unprobed kernel behind it.  Whatever the tuner does over a dozen
    launches, every launch leaves the bytes of the forced strict schedule (variant 72) — and the oracle's records on a sample."""
    verts = scenes.atrium(60_000, seed=3)
    sc = tb.BVH8_CWBVH(ctx).Build(verts)
    h = sc.host
    cam = R.primary(R.camera(*scenes.SPONZA_CAMERAS[0], 1024, 1024, 1, 1))
    rnd = R.random_rays(1 << 20, (-20, 0, -10), (20, 15, 10), seed=8)
    n = cam.shape[0]
    d = ctx.malloc(n * 64)
    d_occ = ctx.malloc(n)
    for name, rays, verdict in (("camera", cam, 2), ("random", rnd, 1)):
        sc.set_variant(72)
        ctx.to_device(d, rays); sc.intersect_device_fresh(d, n, 1e30)
        want = np.zeros(n, tb.RAY_DTYPE); ctx.from_device(want, d)
        sc.occluded_device(d, n, d_occ)
        want_occ = np.zeros(n, np.uint8); ctx.from_device(want_occ, d_occ)
        sc.set_variant(0)
        seen = set()
        for k in range(12):
            ctx.to_device(d, rays); sc.intersect_device_fresh(d, n, 1e30)
            got = np.zeros(n, tb.RAY_DTYPE); ctx.from_device(got, d)
            assert ctx.last_probe()[2] in (verdict, 0), (name, k, ctx.last_probe())     # (0: the tuner has settled on the per-lane kernel for this class: one unprobed kernel again)
            assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (name, k)
            sc.occluded_device(d, n, d_occ)
            occ = np.zeros(n, np.uint8); ctx.from_device(occ, d_occ)
            assert np.array_equal(occ, want_occ), (name, k)
            seen.add(sc.coherent_schedule(False)[0])
        if name == "camera":
            assert sc.coherent_schedule(False)[0] in (2, 3) and sc.coherent_schedule(True)[0] in (2, 3), (sc.coherent_schedule(False), sc.coherent_schedule(True))     # decided: per-lane or packet, never the deferred schedule
        idx = np.arange(0, n, 16)
        ref = oracle_ties.bvh2_intersect(h.bvh2_nodes(), h.bvh2_prim_idx(), verts, rays[idx])
        c = compare_hits(want[idx], ref)
        assert c["hitmiss"] == 0 and c["prim_real"] == 0 and c["t_bad"] == 0 and c["tie"] == 0 and c["bit_identical"] == c["same_prim"], (name, c)
    # a batch below the threshold is not probed
    ctx.to_device(d, cam[: 1 << 19]); sc.intersect_device_fresh(d, 1 << 19, 1e30)
    assert ctx.last_probe()[2] == 0
    # the decision can be read, and given back, through the schedule hint (its `reserved` bytes carry the class of 768 k .. 1.5 M-ray batches)
    hint = sc.schedule_hint()
    sc.set_schedule_hint(hint)
    ctx.free(d); ctx.free(d_occ); sc.free()
