"""Split rays (tinybvh_amd/csrc/ray_split.h): once the ray pool of a launch is dry, idle lanes take pending subtrees off
the lanes that are still traversing.  Batches of fewer rays than the launch has lanes are all tail: every ray is split
as far as its traversal branches, so these are the cases where a wrong merge of the members' results would show.
The records must be the oracle's (reference: BVH::Intersect / IsOccluded, tiny_bvh.h:3222-3304, 3382-3453) — exactly, and the same
from run to run: the members of a split ray merge their hits under the library's tie rule (smaller prim at exactly equal t), which is
also what one lane alone applies, so which rays get split (timing) cannot show in a record."""
import numpy as np
import pytest

import tinybvh_amd as tb
from tinybvh_amd import rays as R
from tinybvh_amd import scenes
from oracle_lib import compare_hits

LAYOUTS = [tb.BVH_GPU, tb.BVH4_GPU, tb.BVH8_CWBVH]


def _check(got, want, what):
    c = compare_hits(got, want)
    assert c["hits"] > 100 and c["hitmiss"] == 0 and c["prim_real"] == 0 and c["t_bad"] == 0 and c["uv_bad"] == 0, (what, c)
    assert c["tie"] == 0 and c["onsurf"] <= max(4, c["n"] // 5000), (what, c)


@pytest.mark.gpu
@pytest.mark.parametrize("cls", LAYOUTS)
@pytest.mark.parametrize("n", [64, 1000, 40_000])
def test_batches_that_are_all_tail_match_the_oracle(ctx, oracle, cls, n):
    verts = scenes.soup(30_000, seed=11)
    sc = cls(ctx).Build(verts)
    h = sc.host
    # long rays through the whole soup (many nodes each) and short ones from inside it
    rays = R.random_rays(n, (0, 0, 0), (10, 10, 10), seed=n)
    want = oracle.bvh2_intersect(h.bvh2_nodes(), h.bvh2_prim_idx(), verts, rays)
    first = None
    for rep in range(3):   # which lanes help which ray depends on timing: the records must not
        got = sc.Intersect(rays.copy())
        if first is None:
            first = got.copy()
        assert np.array_equal(got.view(np.uint8), first.view(np.uint8)), (cls.__name__, n, rep)   # byte-identical run to run
        if n >= 1000:
            _check(got, want, (cls.__name__, n, rep))
        else:
            c = compare_hits(got, want)
            assert c["hitmiss"] == 0 and c["prim_real"] == 0 and c["t_bad"] == 0 and c["uv_bad"] == 0, c
        occ = sc.IsOccluded(rays.copy())
        assert int((occ.astype(bool) != (want["t"] < 1e30)).sum()) <= 2
    sc.free()


@pytest.mark.gpu
@pytest.mark.parametrize("cls", LAYOUTS)
def test_a_closer_hit_already_in_the_record_survives_a_split_ray(ctx, oracle, cls):
    """Intersect only ever shortens hit.t (tiny_bvh.h:8515-8530): rays that arrive with a hit closer than anything in the
    scene keep their record bit for bit, rays that arrive with a farther one get the scene's — also when the ray was split."""
    verts = scenes.soup(30_000, seed=12)
    sc = cls(ctx).Build(verts)
    h = sc.host
    rays = R.random_rays(3000, (0, 0, 0), (10, 10, 10), seed=5)
    want = oracle.bvh2_intersect(h.bvh2_nodes(), h.bvh2_prim_idx(), verts, rays)
    pre = rays.copy()
    pre["t"][::2] = 1e-4; pre["u"][::2] = 0.25; pre["v"][::2] = 0.5; pre["prim"][::2] = 123456   # closer than any triangle
    pre["t"][1::2] = 1e29                                                                          # farther than all of them
    got = sc.Intersect(pre.copy())
    for f in ("t", "u", "v", "prim"):
        assert np.array_equal(got[f][::2].view(np.uint32), pre[f][::2].view(np.uint32)), f
    odd_want = want[1::2].copy()
    miss = odd_want["t"] >= 1e29
    odd_want["t"][miss] = 1e29
    c = compare_hits(got[1::2][~miss], odd_want[~miss])
    assert c["hitmiss"] == 0 and c["prim_real"] == 0 and c["t_bad"] == 0 and c["uv_bad"] == 0, c
    assert np.all(got["t"][1::2][miss] == np.float32(1e29))
    sc.free()


def same_up_to_face_ulps(a, b, what):
    """Byte-identical, except for the residual every BVH traversal has (the reference's included): a triangle lying exactly IN a face of its
    leaf box (the Sponza stand-in is all axis-aligned walls) has its distance computed twice, by the slab test and by the triangle test, and
    when a ray grazes the wall the two differ by more than the 2^-20 slack of device_common.h: cull_bound — then which of two coplanar
    triangles at (nearly) the same t is still tested depends on the order.  Such rays are a few per million, and their t agrees to 16 ulps."""
    d = np.flatnonzero((a["prim"] != b["prim"]) | (a["t"] != b["t"]))
    assert d.size <= max(2, a.shape[0] // 20_000), (what, d.size)
    ulps = np.abs(a["t"][d].view(np.int32).astype(np.int64) - b["t"][d].view(np.int32).astype(np.int64))
    assert np.all(ulps <= 16), (what, int(ulps.max()))
    keep = np.ones(a.shape[0], bool); keep[d] = False
    assert np.array_equal(a[keep].view(np.uint8), b[keep].view(np.uint8)), what


@pytest.mark.gpu
def test_split_and_unsplit_kernels_agree(ctx, monkeypatch):
    """The kernels WITHOUT split rays (a context created under TBVH_SPLIT_RAYS=0) return the same bytes as the default ones, in every layout:
    byte for byte on camera rays, and on random rays inside the axis-aligned atrium up to the box-face residual."""
    verts, _ = scenes.get("sponza")
    cam = R.primary(R.camera(*scenes.SPONZA_CAMERAS[0], 512, 512, 1, 1))
    rnd = R.random_rays(200_000, verts[:, :3].min(0), verts[:, :3].max(0), seed=12)
    monkeypatch.setenv("TBVH_SPLIT_RAYS", "0")
    plain_ctx = tb.Context(0)
    monkeypatch.delenv("TBVH_SPLIT_RAYS")
    try:
        for cls in LAYOUTS:
            a, b = cls(ctx).Build(verts), cls(plain_ctx).Build(verts)
            ga, gb = a.Intersect(cam.copy()), b.Intersect(cam.copy())
            assert int((ga["t"] < 1e30).sum()) > cam.shape[0] // 2
            assert np.array_equal(ga.view(np.uint8), gb.view(np.uint8)), cls.__name__
            same_up_to_face_ulps(a.Intersect(rnd.copy()), b.Intersect(rnd.copy()), cls.__name__)
            for rays in (cam, rnd):
                assert int((a.IsOccluded(rays.copy()) != b.IsOccluded(rays.copy())).sum()) <= 2, cls.__name__
            a.free(); b.free()
    finally:
        plain_ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cls", LAYOUTS)
def test_large_batches_run_the_unsplit_kernels_and_agree(ctx, cls):
    """Batches of 12 M rays and more run the kernels WITHOUT split rays (the tail is 5 % of such a launch): the same rays traced as part of a
    12.8 M-ray launch and as a 1 M-ray launch of their own give the same records, byte for byte."""
    verts, _ = scenes.get("sponza")
    sc = cls(ctx).Build(verts)
    side = 3584                                   # 12.8 M rays
    n, m = side * side, 1 << 20
    cam = R.camera(*scenes.SPONZA_CAMERAS[0], side, side, 1, 1)
    d = ctx.malloc(n * 64)
    ctx.generate_primary(cam, d, 0, n)
    sc.intersect_device_fresh(d, n, 1e30)
    big = np.zeros(m, tb.RAY_DTYPE); ctx.from_device(big, d)
    sc.intersect_device_fresh(d, m, 1e30)
    small = np.zeros(m, tb.RAY_DTYPE); ctx.from_device(small, d)
    ctx.free(d)
    c = compare_hits(small, big)
    assert c["hits"] > m // 2 and c["hitmiss"] == 0 and c["prim_real"] == 0 and c["t_bad"] == 0 and c["uv_bad"] == 0, c
    assert np.array_equal(small.view(np.uint8), big.view(np.uint8))
    sc.free()


@pytest.mark.gpu
def test_split_rays_can_be_switched_off(oracle, monkeypatch):
    """TBVH_SPLIT_RAYS=0 (read at tbvh_init): small batches run the kernels without split rays; same records."""
    monkeypatch.setenv("TBVH_SPLIT_RAYS", "0")
    c2 = tb.Context(0)
    try:
        verts = scenes.soup(20_000, seed=13)
        for cls in LAYOUTS:
            sc = cls(c2).Build(verts)
            rays = R.random_rays(30_000, (0, 0, 0), (10, 10, 10), seed=2)
            want = oracle.bvh2_intersect(sc.host.bvh2_nodes(), sc.host.bvh2_prim_idx(), verts, rays)
            _check(sc.Intersect(rays.copy()), want, cls.__name__)
            sc.free()
    finally:
        c2.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cls", LAYOUTS + ["mixed"])
def test_large_two_level_batches_agree_with_small_ones(ctx, cls):
    """The two-level kernels: 12.8 M rays (kernels without split rays) against the first 1 M of them as a launch of their own.
    "mixed": BVH8_CWBVH and BVH_GPU BLASes under one TLAS, instance by instance."""
    verts = scenes.blob(6_000, seed=6)
    blases = [tb.BVH8_CWBVH(ctx).Build(verts), tb.BVH_GPU(ctx).Build(verts)] if cls == "mixed" else [cls(ctx).Build(verts)]
    blas = blases[0]
    g = np.stack(np.meshgrid(np.arange(5), np.arange(5), np.arange(5), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    k = g.shape[0]
    ang = (0.2 + np.arange(k) * 0.41).astype(np.float32)
    T = np.zeros((k, 4, 4), np.float32)
    T[:, 0, 0] = np.cos(ang) * 0.7; T[:, 0, 2] = np.sin(ang) * 0.7; T[:, 1, 1] = 0.7; T[:, 2, 0] = -np.sin(ang) * 0.7; T[:, 2, 2] = np.cos(ang) * 0.7; T[:, 3, 3] = 1
    T[:, :3, 3] = g * 2.0
    tlas = tb.TLAS(ctx).Build(tb.make_instances(T, (np.arange(k) % len(blases)).astype(np.uint32)), blases)
    side = 3584
    n, m = side * side, 1 << 20
    cam = R.camera((-6.0, 8.0, -9.0), (0.62, -0.38, 0.68), side, side, 1, 1)
    d = ctx.malloc(n * 64)
    ctx.generate_primary(cam, d, 0, n)
    tlas.intersect_device_fresh(d, n, 1e30)
    big = np.zeros(m, tb.RAY_DTYPE); ctx.from_device(big, d)
    tlas.intersect_device_fresh(d, m, 1e30)
    small = np.zeros(m, tb.RAY_DTYPE); ctx.from_device(small, d)
    ctx.free(d)
    c = compare_hits(small, big)
    assert c["hits"] > 1000 and c["hitmiss"] == 0 and c["prim_real"] == 0 and c["t_bad"] == 0 and c["uv_bad"] == 0, c
    assert np.array_equal(small.view(np.uint8), big.view(np.uint8))
    tlas.free()
    for x in blases:
        x.free()


@pytest.mark.gpu
def test_every_hit_a_tie(ctx, oracle):
    """A scene in which every triangle exists twice: every hit is a tie between two prims, the case in which the members of a split ray — and
    the three layouts, and the schedules — disagree about the winner all the time under "the later test wins".  Under the library's rule the
    record is the same everywhere: t, u, v the oracle's bit for bit, prim the SMALLER copy; three runs of a 1 M-ray batch byte-identical, in
    every layout (1 M rays: the split-ray kernels; small batches are all tail)."""
    base = scenes.soup(8_000, seed=17)
    verts = np.ascontiguousarray(np.concatenate([base, base]))
    ntri = base.shape[0] // 3
    rays = R.random_rays(1 << 20, (0, 0, 0), (10, 10, 10), seed=4)
    small = rays[:20_000]
    records = {}
    for cls in LAYOUTS:
        sc = cls(ctx).Build(verts)
        h = sc.host
        want = oracle.bvh2_intersect(h.bvh2_nodes(), h.bvh2_prim_idx(), verts, small)
        hit = want["t"] < 1e30
        assert hit.sum() > 5000 and np.all(want["prim"][hit] < ntri)          # the oracle under the same rule reports the smaller copy
        for rep in range(3):
            got = sc.Intersect(small.copy())
            assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (cls.__name__, rep)
        runs = [sc.Intersect(rays.copy()) for _ in range(3)]
        assert np.array_equal(runs[0].view(np.uint8), runs[1].view(np.uint8)) and np.array_equal(runs[0].view(np.uint8), runs[2].view(np.uint8)), cls.__name__
        h1 = runs[0]["t"] < 1e30
        assert np.all(runs[0]["prim"][h1] < ntri), cls.__name__
        records[cls.__name__] = runs[0]
        sc.free()
    names = list(records)
    for nm in names[1:]:   # and across layouts
        assert np.array_equal(records[nm].view(np.uint8), records[names[0]].view(np.uint8)), nm


@pytest.mark.gpu
@pytest.mark.parametrize("cls", LAYOUTS)
def test_every_instance_twice(ctx, cls):
    """Two-level: every instance exists twice with the same transform, so every hit is a tie between two instances.  The record must be the
    one of the scene with each instance once — t, u, v, prim bit for bit — with hit.inst naming the FIRST copy (the smaller instance index),
    the same in three runs of a batch large enough for split rays."""
    verts = scenes.blob(5_000, seed=8)
    blas = cls(ctx).Build(verts)
    g = np.stack(np.meshgrid(np.arange(3), np.arange(3), np.arange(3), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    k = g.shape[0]
    ang = (0.5 + np.arange(k) * 0.29).astype(np.float32)
    T = np.zeros((k, 4, 4), np.float32)
    T[:, 0, 0] = np.cos(ang) * 0.8; T[:, 0, 2] = np.sin(ang) * 0.8; T[:, 1, 1] = 0.8; T[:, 2, 0] = -np.sin(ang) * 0.8; T[:, 2, 2] = np.cos(ang) * 0.8; T[:, 3, 3] = 1
    T[:, :3, 3] = g * 2.2
    once = tb.TLAS(ctx).Build(tb.make_instances(T, np.zeros(k, np.uint32)), [blas])
    twice = tb.TLAS(ctx).Build(tb.make_instances(np.concatenate([T, T]), np.zeros(2 * k, np.uint32)), [blas])
    rays = R.random_rays(400_000, (-1.0, -1.0, -1.0), (6.0, 6.0, 6.0), seed=6)
    want = once.Intersect(rays.copy())
    hit = want["t"] < 1e30
    assert hit.sum() > 50_000
    for rep in range(3):
        got = twice.Intersect(rays.copy())
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (cls.__name__, rep)
    once.free(); twice.free(); blas.free()
