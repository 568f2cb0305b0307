"""Randomised parity on the paths only LARGER inputs reach (tests/test_random_configs.py stays under 20 k triangles and 34 k rays): scenes of
30 k .. 400 k triangles in all three layouts — BVH_GPU and BVH4_GPU scenes then trace their 8-wide copy (capi_scene.hip: makeWideCopy) — and batches of
0.8 .. 2.2 M camera, bounce, shadow and random rays, which a scene under 48 MB probes for the wave-packet kernel while its tuner measures (capi_query.hip:
launchCoherentFlavor).  Eight launches in a row — the tuner moves through its schedules meanwhile.

Two traversals of one scene can part where a box test decides at rounding level — the classes the reference's own layouts disagree on with each other
(DESIGN.md par. 4) — whenever one of them TESTS MORE candidates than the other: the 8-wide copy's boxes are re-quantised outward from the uploaded tree's, and
the wave-packet kernel walks the union of its 64 rays' paths (a lane tests the triangles of a leaf another lane's ray entered).  The classes:
  * a hit that grazes an edge of its triangle (min(u, v, 1 - u - v) < 2e-3),
  * a hit at t = +-0: the ray starts IN the plane of an axis-aligned triangle (bounce rays off axis-aligned geometry; MOLLER_TRUMBORE_TEST, tiny_bvh.h:1644-1656,
    accepts t = -0; the flat box of such a triangle is entered or not by the sign of a rounding error),
  * a ray that skims its triangle (|cos| of the angle between direction and normal below 1e-4): the slab distance through the triangle's flat box is the
    quotient of two rounding errors,
  * coplanar triangles hit at the same t, or within a few ulp of it: the closest — at equal t the smaller index (the tie rule) — wins among the candidates a
    traversal TESTS; a flat box 1e-5 (relative to t) beyond the hit is culled by one tree and not by the other.
Every differing record must be in one of these classes, must be a REAL hit (the oracle's triangle test on that one triangle reproduces its bytes), and there
are at most n / 4000 of them for BVH_GPU / BVH4_GPU scenes against the native kernel on the nodes as uploaded (measured over 80 seeds: none in most batches,
up to 1.2e-4 of a bounce batch in the street generator, whose walls carry coplanar window triangles) and at most 4 per launch for BVH8_CWBVH scenes
against the forced strict per-lane kernel (measured: none).  A sample of every batch is compared with the oracle (BVH::Intersect restated,
tiny_bvh.h:3222-3304) under the same rule.
A longer hunt: TBVH_RANDOM_LARGE_SEEDS=200 python -m pytest tests/test_random_large.py -m gpu"""
import os

import numpy as np
import pytest

import tinybvh_amd as tb
from tinybvh_amd import rays as R
from tinybvh_amd import scenes
from oracle_lib import compare_hits

pytestmark = pytest.mark.gpu

N_SEEDS = int(os.environ.get("TBVH_RANDOM_LARGE_SEEDS", "6"))
SCALE = int(os.environ.get("TBVH_RANDOM_LARGE_SCALE", "1"))     # 8: scenes of 0.3 .. 3.2 M triangles and batches of 1.6 .. 4.4 M rays — the class of the bench scene (probed two-flavor launches on
                                                                 # the incoherent-batch copies, three schedules to tune between); minutes per configuration on the host side
LAYOUTS = [tb.LAYOUT_BVH_GPU, tb.LAYOUT_BVH4_GPU, tb.LAYOUT_CWBVH]


def make_scene(rng):
    kind = int(rng.integers(0, 4))
    seed = int(rng.integers(1, 1 << 20))
    if kind == 0:
        return "atrium", scenes.atrium(SCALE * int(rng.integers(40_000, 300_000)), seed=seed)
    if kind == 1:
        return "street", scenes.street(SCALE * int(rng.integers(50_000, 400_000)), seed=seed)
    if kind == 2:
        return "blob", scenes.blob(SCALE * int(rng.integers(40_000, 200_000)), seed=seed)
    return "soup", scenes.soup(SCALE * int(rng.integers(33_000, 100_000)), seed=seed, extent=30.0 * SCALE ** (1 / 3), size=0.8)


def camera_rays(rng, lo, hi, n_min):
    """a pinhole camera somewhere around the scene looking at a point inside it, one sample per pixel, at least n_min rays"""
    c = 0.5 * (lo + hi)
    ext = hi - lo
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    eye = c + d * ext * float(rng.uniform(0.2, 0.9))
    tgt = c + rng.uniform(-0.2, 0.2, 3) * ext
    view = tgt - eye; view /= np.linalg.norm(view)
    if abs(view[1]) > 0.95:                                  # (the camera's up vector is y)
        view = np.array([0.8, 0.5, 0.33]); view /= np.linalg.norm(view)
    w = int(rng.choice([1024, 1280, 1536]))
    h = -(-n_min // w)
    return R.primary(R.camera(tuple(eye), tuple(view), w, h, 1, 1))


def classes(a, b, rays, verts):
    """how the differing rows of two record arrays split over the classes: for assertion messages"""
    diff, ok = knife_edge(a, b, rays, verts)
    x, y = a[diff], b[diff]
    closer = np.where(x["t"] <= y["t"], x, y)
    return {"differ": int(diff.size), "at_origin": int((np.abs(closer["t"]) < 1e-6).sum()), "same_or_near_t": int((np.abs(x["t"].astype(np.float64) - y["t"]) <= 4e-6 * np.abs(closer["t"])).sum()),
            "a_closer": int((x["t"] < y["t"]).sum()), "b_closer": int((x["t"] > y["t"]).sum()), "b_closer_beyond_origin": int(((x["t"] > y["t"]) & (np.abs(y["t"]) >= 1e-6)).sum()),
            "outside": int((~ok).sum())}


def knife_edge(a, b, rays, verts):
    """rows of two hit-record arrays that differ, and whether each difference is one of the rounding-level classes of the module text (judged on the CLOSER record)"""
    diff = np.nonzero((a.view(np.uint8).reshape(-1, 64)[:, 48:] != b.view(np.uint8).reshape(-1, 64)[:, 48:]).any(1))[0]
    x, y = a[diff], b[diff]
    closer = np.where(x["t"] <= y["t"], x, y)
    w = 1.0 - closer["u"].astype(np.float64) - closer["v"]
    edge = np.minimum(np.minimum(closer["u"], closer["v"]), w)
    near = np.abs(x["t"].astype(np.float64) - y["t"]) <= 4e-6 * np.maximum(np.abs(closer["t"]), 1e-30)
    tri = verts.reshape(-1, 3, 4)[np.minimum(closer["prim"], verts.shape[0] // 3 - 1), :, :3].astype(np.float64)
    nrm = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-300)
    skim = np.abs((rays["D"][diff].astype(np.float64) * nrm).sum(1)) < 1e-4
    ok = (edge < 2e-3) | (np.abs(closer["t"]) < 1e-6) | (near & (x["prim"] != y["prim"])) | skim
    return diff, ok


def assert_real_hits(oracle, verts, rays, recs, lo, hi):
    """each record is what the oracle's triangle test gives for that ray on that ONE triangle (a root leaf over the whole scene's box)"""
    box = np.zeros((1, 8), np.float32)
    box[0, 0:3] = lo - 0.5 * (hi - lo) - 1.0; box[0, 4:7] = hi + 0.5 * (hi - lo) + 1.0
    node = box.view(np.uint32).copy()
    node[0, 3] = 0; node[0, 7] = 1
    for r, g in zip(rays, recs):
        if g["prim"] == r["prim"] and g["t"] == r["t"]:
            continue                                        # (a miss: the record as uploaded)
        one = oracle.bvh2_intersect(node, np.array([g["prim"]], np.uint32), verts, np.array([r]))[0]
        assert one["t"] == g["t"] and one["u"] == g["u"] and one["v"] == g["v"] and one["prim"] == g["prim"], (r, g, one)


class Case:
    """one random configuration: scene, layout, library scene, batch kind, the camera the batch is made from"""
    def __init__(self, ctx, seed):
        self.ctx, self.seed = ctx, seed
        self.rng = rng = np.random.default_rng(9000 + seed)
        self.name, self.verts = make_scene(rng)
        self.layout = LAYOUTS[int(rng.integers(0, 3))]
        self.sc = tb.LAYOUT_CLASSES[self.layout](ctx).Build(self.verts)
        self.lo, self.hi = self.verts[:, :3].min(0), self.verts[:, :3].max(0)
        n_min = int(rng.choice([800_000, 1_100_000, 1_600_000, 2_200_000])) * (2 if SCALE > 1 else 1)
        self.kind = ["camera", "bounce", "shadow", "random"][int(rng.integers(0, 4))]
        self.cam = camera_rays(rng, self.lo, self.hi, n_min)
        self.n = self.cam.shape[0]
        self.d = ctx.malloc(self.n * 64)
        self.d_occ = ctx.malloc(self.n)
        self.forced = 72 if self.layout == tb.LAYOUT_CWBVH else 1         # the strict per-lane kernel / the native kernel on the nodes as uploaded

    def trace(self, rays, variant):
        ctx, sc, d, d_occ, n = self.ctx, self.sc, self.d, self.d_occ, self.n
        sc.set_variant(variant)
        ctx.to_device(d, rays); sc.intersect_device_fresh(d, n, 1e30)
        out = np.zeros(n, tb.RAY_DTYPE); ctx.from_device(out, d)
        ctx.to_device(d, rays); sc.occluded_device(d, n, d_occ)      # (the rays as generated: tmax = 1e30.  With tmax = the hit distance itself every flag would sit on the edge of the query interval)
        occ = np.zeros(n, np.uint8); ctx.from_device(occ, d_occ)
        sc.set_variant(0)
        return out, occ

    def rays(self):
        rng, lo, hi, n = self.rng, self.lo, self.hi, self.n
        if self.kind == "camera":
            return self.cam
        if self.kind == "random":
            pad = 0.05 * (hi - lo)
            return R.random_rays(n, lo - pad, hi + pad, seed=int(rng.integers(1, 1 << 20)))
        first, _ = self.trace(self.cam, self.forced)
        miss = first["t"] >= 1e30                            # camera rays that left the scene: continue from a random point inside it instead (R.bounce / R.shadow
        if miss.any():                                       # would start 20 / 1000 units out, where a float's spacing is a visible fraction of a small triangle)
            first["O"][miss] = (lo + rng.random((int(miss.sum()), 3)) * (hi - lo)).astype(np.float32)
            first["t"][miss] = 0.0
            first["prim"][miss] = 0
        if self.kind == "bounce":
            return R.bounce(first, self.verts, seed=int(rng.integers(1, 1 << 20)))
        light = 0.5 * (lo + hi) + np.array([0.0, 0.45 * (hi[1] - lo[1]), 0.0])
        rays = R.shadow(first, light, 1e-4 * float((hi - lo).max()))   # (the speedtest's 5e-7 of the extent, tiny_bvh_speedtest.cpp:851, is less than a float's spacing at these
        rays["t"] = np.float32(1e30)                                  # coordinates: up to 3 % of a batch off an axis-aligned floor then starts ON it, and every traversal tosses its own coin)
        return rays

    def free(self):
        self.ctx.free(self.d); self.ctx.free(self.d_occ); self.sc.free()


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_random_large_configuration(ctx, oracle, seed):
    case = Case(ctx, seed)
    name, verts, layout, kind, n, lo, hi, rng = case.name, case.verts, case.layout, case.kind, case.n, case.lo, case.hi, case.rng
    host = case.sc.host
    few = max(4, n // 4000)
    trace, forced = case.trace, case.forced
    rays = case.rays()
    want, want_occ = trace(rays, forced)
    for k in range(8):
        got, occ = trace(rays, 0)
        diff, ok = knife_edge(got, want, rays, verts)
        c = classes(got, want, rays, verts)
        assert ok.all() and c["differ"] - c["at_origin"] <= (4 if layout == tb.LAYOUT_CWBVH else few) and c["at_origin"] <= n // 1000, (seed, name, layout, kind, n, k, c)
        assert int((occ != want_occ).sum()) <= few + n // 1000, (seed, name, layout, kind, n, k, int((occ != want_occ).sum()))
        if diff.size and k in (0, 5):                      # (k = 5: the tuner is trying the packet kernel by then)
            assert_real_hits(oracle, verts, rays[diff], got[diff], lo, hi)
            assert_real_hits(oracle, verts, rays[diff], want[diff], lo, hi)
    idx = np.arange(int(rng.integers(0, 32)), n, 32)
    ref = oracle.bvh2_intersect(host.bvh2_nodes(), host.bvh2_prim_idx(), verts, rays[idx])
    diff, ok = knife_edge(got[idx], ref, rays[idx], verts)
    c = classes(got[idx], ref, rays[idx], verts)
    # (hits at the ray's origin are the frequent class: a bounce ray leaves an axis-aligned surface by 1e-3 along a random direction and, when that
    # direction is nearly tangent, lands within a float's spacing of the plane — up to 5e-4 of such a batch re-hit that surface at |t| < 1e-6 in one tree and not in another)
    assert ok.all() and c["differ"] - c["at_origin"] <= max(2, idx.size // 4000) and c["at_origin"] <= max(2, idx.size // 500), (seed, name, layout, kind, n, c)
    if diff.size:
        assert_real_hits(oracle, verts, rays[idx][diff], got[idx][diff], lo, hi)
    hit = (ref["prim"] != rays["prim"][idx]) | (ref["t"] != rays["t"][idx])
    assert int((occ[idx].astype(bool) != hit).sum()) <= max(2, idx.size // 2000), (seed, name, layout, kind, n)
    case.free()


N_REF_SEEDS = int(os.environ.get("TBVH_RANDOM_LARGE_REF_SEEDS", "4"))


@pytest.mark.parametrize("seed", range(N_REF_SEEDS))
def test_random_large_reference_blobs(ctx, oracle, reference, seed):
    """The same hunt on blobs the REAL reference builds (oracle/_ref: BVH::Build or BuildHQ — an SBVH, whose split references tile their triangles exactly —,
    then BVH_GPU / BVH4_GPU / BVH8_CWBVH::ConvertFrom), uploaded verbatim, against the reference's OWN traversal of that layout on that blob
    (BVH_GPU::Intersect tiny_bvh.h:4657-4712, BVH4_GPU::Intersect 5252-5343, BVH8_CWBVH::Intersect 7046-7154) and against BVH::Intersect on its BVH2.
    Exact-t ties are free here (the reference lets the later test win, the library the smaller index: DESIGN.md par. 4); everything else as above."""
    rng = np.random.default_rng(12000 + seed)
    kind_s = int(rng.integers(0, 3))
    s_seed = int(rng.integers(1, 1 << 20))
    if kind_s == 0:
        name, verts = "atrium", scenes.atrium(int(rng.integers(40_000, 200_000)), seed=s_seed)
    elif kind_s == 1:
        name, verts = "street", scenes.street(int(rng.integers(50_000, 250_000)), seed=s_seed)
    else:
        name, verts = "blob", scenes.blob(int(rng.integers(40_000, 150_000)), seed=s_seed)
    layout = LAYOUTS[int(rng.integers(0, 3))]
    hq = bool(rng.integers(0, 2))
    rs = reference.build(verts, hq=hq, threaded=True)
    if layout == tb.LAYOUT_BVH_GPU:
        sc = tb.BVH_GPU(ctx).Upload(rs.blob(5, 0, np.uint32, 16), rs.blob(5, 1, np.uint32, 1), verts)
    elif layout == tb.LAYOUT_BVH4_GPU:
        sc = tb.BVH4_GPU(ctx).Upload(rs.blob(8, 0, np.uint32, 4))
    else:
        sc = tb.BVH8_CWBVH(ctx).Upload(rs.blob(10, 0, np.uint32, 4), rs.blob(10, 1, np.uint32, 4))
    lo, hi = verts[:, :3].min(0), verts[:, :3].max(0)
    n_min = int(rng.choice([800_000, 1_100_000, 1_600_000]))
    kind = ["camera", "bounce", "random"][int(rng.integers(0, 3))]
    cam = camera_rays(rng, lo, hi, n_min)
    n = cam.shape[0]
    d = ctx.malloc(n * 64)
    forced = 72 if layout == tb.LAYOUT_CWBVH else 1

    def trace(rays, variant):
        sc.set_variant(variant)
        ctx.to_device(d, rays); sc.intersect_device_fresh(d, n, 1e30)
        out = np.zeros(n, tb.RAY_DTYPE); ctx.from_device(out, d)
        sc.set_variant(0)
        return out

    if kind == "camera":
        rays = cam
    elif kind == "random":
        pad = 0.05 * (hi - lo)
        rays = R.random_rays(n, lo - pad, hi + pad, seed=int(rng.integers(1, 1 << 20)))
    else:
        first = trace(cam, forced)
        miss = first["t"] >= 1e30
        if miss.any():
            first["O"][miss] = (lo + rng.random((int(miss.sum()), 3)) * (hi - lo)).astype(np.float32)
            first["t"][miss] = 0.0
            first["prim"][miss] = 0
        rays = R.bounce(first, verts, seed=int(rng.integers(1, 1 << 20)))
    want = trace(rays, forced)
    few = max(4, n // 4000)
    for k in range(8):
        got = trace(rays, 0)
        diff, ok = knife_edge(got, want, rays, verts)
        c = classes(got, want, rays, verts)
        assert ok.all() and c["differ"] - c["at_origin"] <= (4 if layout == tb.LAYOUT_CWBVH else few) and c["at_origin"] <= n // 1000, (seed, name, hq, layout, kind, n, k, c)
        if diff.size and k in (0, 5):
            assert_real_hits(oracle, verts, rays[diff], got[diff], lo, hi)
    idx = np.arange(int(rng.integers(0, 16)), n, 16)
    for what, ref in (("the reference's traversal of this layout", rs.intersect(layout, rays[idx])), ("BVH::Intersect", rs.intersect(1, rays[idx]))):
        diff, ok = knife_edge(got[idx], ref, rays[idx], verts)
        c = classes(got[idx], ref, rays[idx], verts)
        # (coplanar triangles at the same t or ulps apart — the street's windows lie IN its walls — are free here: at equal t the two tie rules differ, and ulps
        # apart the reference keeps the farther one when its box test, which has no slack, culls the closer one's flat box: seed 32, 0.4 % of a camera batch)
        assert ok.all() and c["differ"] - c["at_origin"] - c["same_or_near_t"] <= max(2, idx.size // 4000) and c["at_origin"] <= max(2, idx.size // 500), (seed, name, hq, layout, kind, n, what, c)
        assert c["b_closer_beyond_origin"] <= max(2, idx.size // 4000), (seed, name, hq, layout, kind, n, what, c)      # the library is rarely the one that loses a hit
        if diff.size:
            assert_real_hits(oracle, verts, rays[idx][diff], got[idx][diff], lo, hi)
        same = np.setdiff1d(np.arange(idx.size), diff)
        assert np.array_equal(got[idx][same].view(np.uint8).reshape(-1, 64)[:, 48:], ref[same].view(np.uint8).reshape(-1, 64)[:, 48:])    # (t, u, v, prim: the reference's bits)
    ctx.free(d); sc.free()


N_TLAS_SEEDS = int(os.environ.get("TBVH_RANDOM_LARGE_TLAS_SEEDS", "3"))


@pytest.mark.parametrize("seed", range(N_TLAS_SEEDS))
def test_random_large_tlas(ctx, oracle, seed):
    """Two-level scenes at the sizes tests/test_random_configs.py does not reach: 64 .. 1728 instances of 1-3 BLASes of 5 k .. 80 k triangles (one layout, or the
    reference's mix of layouts under one TLAS), host-built or rebuilt on the device, batches of 0.5 .. 2 M camera or random rays with masks on both sides;
    a 1 / 16 sample against BVH::IntersectTLAS restated (tiny_bvh.h:3306-3380), every launch of four leaving the same bytes."""
    from test_tlas import grid_instances, oracle_tlas
    rng = np.random.default_rng(15000 + seed)
    n_blas = int(rng.integers(1, 4))
    meshes = []
    for k in range(n_blas):
        if rng.random() < 0.6:
            m = scenes.blob(int(rng.integers(5_000, 80_000)), seed=int(rng.integers(1, 1 << 20)))
        else:
            m = scenes.soup(int(rng.integers(5_000, 40_000)), seed=int(rng.integers(1, 1 << 20)), extent=1.6, size=0.12)
        m = m.copy()
        c = 0.5 * (m[:, :3].min(0) + m[:, :3].max(0)); e = float((m[:, :3].max(0) - m[:, :3].min(0)).max())
        m[:, :3] = (m[:, :3] - c) * np.float32(1.6 / e)        # every BLAS in [-0.8, 0.8]^3
        meshes.append(np.ascontiguousarray(m))
    if rng.random() < 0.5:
        layouts = [[tb.LAYOUT_CWBVH, tb.LAYOUT_BVH_GPU, tb.LAYOUT_BVH4_GPU][int(rng.integers(0, 3))] for _ in range(n_blas)]
    else:
        layouts = [[tb.LAYOUT_CWBVH, tb.LAYOUT_BVH_GPU, tb.LAYOUT_BVH4_GPU][int(rng.integers(0, 3))]] * n_blas
    blas = [tb.LAYOUT_CLASSES[l](ctx).Build(meshes[i]) for i, l in enumerate(layouts)]
    side = int(rng.integers(4, 13))
    inst = grid_instances(side, float(rng.uniform(0.3, 0.7)), int(rng.integers(1, 1 << 20)), n_blas=n_blas)
    if rng.random() < 0.4:
        inst["mask"][:: int(rng.integers(2, 6))] = 0x0001
    tlas = tb.TLAS(ctx).Build(inst, blas)
    if rng.random() < 0.5:
        tlas.RebuildOnDevice()
    lo, hi = np.full(3, -1.0), np.full(3, 2.0 * side - 1.0)
    n_min = int(rng.choice([500_000, 1_000_000, 2_000_000]))
    if rng.random() < 0.6:
        rays = camera_rays(rng, lo, hi, n_min)
    else:
        rays = R.random_rays(n_min, lo - 1, hi + 1, seed=int(rng.integers(1, 1 << 20)))
    if rng.random() < 0.4:
        rays["mask"][::3] = 0x00F0
    n = rays.shape[0]
    first = tlas.Intersect(rays.copy())
    for k in range(3):
        again = tlas.Intersect(rays.copy())
        assert np.array_equal(again.view(np.uint8), first.view(np.uint8)), (seed, layouts, side, n, k)
    idx = np.arange(int(rng.integers(0, 16)), n, 16)
    want = oracle_tlas(oracle, tlas, blas, rays[idx])
    c = compare_hits(first[idx], want)
    assert c["hitmiss"] <= 1 and c["prim_real"] <= 1 and c["t_bad"] == 0 and c["uv_bad"] == 0 and c["tie"] == 0 and c["onsurf"] <= 4, (seed, layouts, side, n, c)
    assert c["bit_identical"] >= c["same_prim"] - 1, (seed, c)
    same = (first[idx]["t"] < 1e30) & (first[idx]["prim"] == want["prim"]) & (first[idx]["t"] == want["t"])
    assert np.array_equal(first[idx]["inst"][same], want["inst"][same])
    occ = tlas.IsOccluded(rays.copy())
    assert int((occ[idx].astype(bool) != (want["t"] < 1e30)).sum()) <= 2, (seed, layouts, side, n)
    # one BLAS is animated: a device refit (the copies the TLAS enters it through are refitted in place), then the host flow — the blob rebuilt and
    # re-uploaded (the copies are dropped and come back after four queries: every one of six queries must be right)
    from test_refit_device import deform
    k = int(rng.integers(0, n_blas))
    moved = deform(meshes[k], 0.01, seed=int(rng.integers(1, 1 << 20)))
    blas[k].Refit(moved)
    blas[k].host = tb.HostBVH(moved, layouts[k])
    want2 = oracle_tlas(oracle, tlas, blas, rays[idx])

    def right(got, wanted, what):
        c = compare_hits(got[idx], wanted)
        assert c["hitmiss"] <= 1 and c["prim_real"] <= 1 and c["t_bad"] == 0 and c["uv_bad"] == 0 and c["tie"] == 0 and c["onsurf"] <= 4, (seed, layouts, side, n, what, c)

    right(tlas.Intersect(rays.copy()), want2, "refit")
    h0 = tb.HostBVH(meshes[k], layouts[k])
    try:
        if layouts[k] == tb.LAYOUT_BVH_GPU:
            blas[k].Update(h0.blob(0, np.uint32, 16), h0.blob(1, np.uint32, 1), meshes[k])
        elif layouts[k] == tb.LAYOUT_BVH4_GPU:
            blas[k].Update(h0.blob(0, np.uint32, 4))
        else:
            blas[k].Update(h0.blob(0, np.uint32, 4), h0.blob(1, np.uint32, 4))
        blas[k].host = h0
        for q in range(6):
            right(tlas.Intersect(rays.copy()), want, ("update", q))
        assert int((tlas.IsOccluded(rays.copy())[idx].astype(bool) != (want["t"] < 1e30)).sum()) <= 2
    except tb.TbvhError as e:
        assert "larger than the one uploaded" in str(e)      # (cannot happen: the blob is the one first uploaded)
        raise
    tlas.free()
    for b in blas:
        b.free()


N_REFIT_SEEDS = int(os.environ.get("TBVH_RANDOM_LARGE_REFIT_SEEDS", "3"))


@pytest.mark.parametrize("seed", range(N_REFIT_SEEDS))
def test_random_large_refit(ctx, oracle, seed):
    """Animated meshes at these sizes: after tbvh_refit to smoothly displaced vertices (three frames, the last back at rest) a batch of 0.8 .. 1.6 M rays
    through the refitted tree — for BVH_GPU / BVH4_GPU scenes through their 8-wide copy, which the refit must carry along — gives the records of
    BVH::Intersect on a tree BUILT over the moved vertices (hit records do not depend on the tree, up to the classes above)."""
    from test_refit_device import deform
    rng = np.random.default_rng(18000 + seed)
    if rng.random() < 0.5:
        name, verts = "blob", scenes.blob(int(rng.integers(40_000, 150_000)), seed=int(rng.integers(1, 1 << 20)))
    else:
        name, verts = "atrium", scenes.rotate(scenes.rotate(scenes.atrium(int(rng.integers(40_000, 150_000)), seed=int(rng.integers(1, 1 << 20))), 0, 0.618), 1, 0.755)
    layout = LAYOUTS[int(rng.integers(0, 3))]
    sc = tb.LAYOUT_CLASSES[layout](ctx).Build(verts)
    lo, hi = verts[:, :3].min(0), verts[:, :3].max(0)
    n_min = int(rng.choice([800_000, 1_600_000]))
    rays = camera_rays(rng, lo, hi, n_min) if rng.random() < 0.5 else R.random_rays(n_min, lo - 0.05 * (hi - lo), hi + 0.05 * (hi - lo), seed=int(rng.integers(1, 1 << 20)))
    n = rays.shape[0]
    d = ctx.malloc(n * 64)
    ext = float((hi - lo).max())
    for frame, amount in enumerate((0.0, 0.004 * ext, 0.015 * ext, 0.0)):
        v2 = deform(verts, amount, seed=int(rng.integers(1, 1 << 20))) if amount else verts
        if frame:
            sc.Refit(v2)
        ctx.to_device(d, rays); sc.intersect_device_fresh(d, n, 1e30)
        got = np.zeros(n, tb.RAY_DTYPE); ctx.from_device(got, d)
        idx = np.arange(int(rng.integers(0, 32)), n, 32)
        h = tb.HostBVH(v2, tb.LAYOUT_BVH_GPU)
        ref = oracle.bvh2_intersect(h.bvh2_nodes(), h.bvh2_prim_idx(), v2, rays[idx])
        diff, ok = knife_edge(got[idx], ref, rays[idx], v2)
        c = classes(got[idx], ref, rays[idx], v2)
        assert ok.all() and c["differ"] - c["at_origin"] <= max(2, idx.size // 4000) and c["at_origin"] <= max(2, idx.size // 500), (seed, name, layout, n, frame, c)
        if diff.size:
            assert_real_hits(oracle, v2, rays[idx][diff], got[idx][diff], v2[:, :3].min(0), v2[:, :3].max(0))
        assert int((got["t"] < 1e30).sum()) > n // 50, (seed, name, layout, frame)
    ctx.free(d); sc.free()


N_REF_TLAS_SEEDS = int(os.environ.get("TBVH_RANDOM_LARGE_REF_TLAS_SEEDS", "2"))


@pytest.mark.parametrize("seed", range(N_REF_TLAS_SEEDS))
def test_random_large_tlas_of_reference_blobs(ctx, reference, seed):
    """Two-level scenes whose BLAS blobs the REAL reference built (Build or BuildHQ, any of the three layouts, also mixed under one TLAS) — so the 4-wide copies are
    decoded from the reference's own BVH8_CWBVH / BVH_GPU encodings (host_builder.cpp: cwbvh_to_bvh2, bvh_gpu_to_bvh2) — against the REAL BVH::IntersectTLAS
    (tiny_bvh.h:3306-3380) over the same instances: prim and instance exact, t / u / v the reference's bits up to its tie rule and the cull-bound class."""
    from oracle_lib import RefTlas, compare_with_real_reference
    from test_tlas import grid_instances
    rng = np.random.default_rng(21000 + seed)
    n_blas = int(rng.integers(1, 4))
    meshes, refs, blas, layouts = [], [], [], []
    for k in range(n_blas):
        m = scenes.blob(int(rng.integers(5_000, 60_000)), seed=int(rng.integers(1, 1 << 20))).copy()
        c = 0.5 * (m[:, :3].min(0) + m[:, :3].max(0)); e = float((m[:, :3].max(0) - m[:, :3].min(0)).max())
        m[:, :3] = (m[:, :3] - c) * np.float32(1.6 / e)
        m = np.ascontiguousarray(m)
        rs = reference.build(m, hq=bool(rng.integers(0, 2)), threaded=True)
        lay = LAYOUTS[int(rng.integers(0, 3))]
        if lay == tb.LAYOUT_BVH_GPU:
            b = tb.BVH_GPU(ctx).Upload(rs.blob(5, 0, np.uint32, 16), rs.blob(5, 1, np.uint32, 1), m)
        elif lay == tb.LAYOUT_BVH4_GPU:
            b = tb.BVH4_GPU(ctx).Upload(rs.blob(8, 0, np.uint32, 4))
        else:
            b = tb.BVH8_CWBVH(ctx).Upload(rs.blob(10, 0, np.uint32, 4), rs.blob(10, 1, np.uint32, 4))
        b._bounds = np.concatenate([m[:, :3].min(0), m[:, :3].max(0)]).astype(np.float32)
        meshes.append(m); refs.append(rs); blas.append(b); layouts.append(lay)
    side = int(rng.integers(4, 11))
    inst = grid_instances(side, float(rng.uniform(0.3, 0.7)), int(rng.integers(1, 1 << 20)), n_blas=n_blas)
    tlas = tb.TLAS(ctx).Build(inst.copy(), blas)
    rt = RefTlas(reference, inst, refs)
    lo, hi = np.full(3, -1.0), np.full(3, 2.0 * side - 1.0)
    n_min = int(rng.choice([500_000, 1_000_000]))
    rays = camera_rays(rng, lo, hi, n_min) if rng.random() < 0.6 else R.random_rays(n_min, lo - 1, hi + 1, seed=int(rng.integers(1, 1 << 20)))
    n = rays.shape[0]
    got = tlas.Intersect(rays.copy())
    idx = np.arange(int(rng.integers(0, 16)), n, 16)
    c = compare_with_real_reference(got[idx], rt.intersect(rays[idx]), check_inst=True)
    assert c["hitmiss"] <= 1 and c["prim_real"] <= 1 and c["t_bad"] == 0 and c["uv_differs"] == 0 and c["farther_by_ulps"] == 0, (seed, layouts, side, n, c)
    assert c["identical"] >= c["hits"] - c["tie_equal_t"] - c["closer_by_ulps"] - c["onsurf"] - 2, (seed, layouts, c)
    occ = tlas.IsOccluded(rays.copy())
    want_hit = rt.intersect(rays[idx])["t"] < 1e30
    assert int((occ[idx].astype(bool) != want_hit).sum()) <= 2, (seed, layouts, side, n)
    tlas.free()
    for b in blas:
        b.free()
