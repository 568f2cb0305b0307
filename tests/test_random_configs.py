"""Randomised parity: many small (scene, layout, builder option, batch shape, entry point) combinations against the oracle — the cheap way to
meet a corner no hand-written case names (ragged sizes around the wave and chunk boundaries, tmax classes, 64 / 128-byte strides, fresh and
in-place entry points, host and device builders, device-resident and host-array calls, closest-hit and any-hit on the same rays)."""
import os

import numpy as np
import pytest

import tinybvh_amd as tb
from tinybvh_amd import rays as R
from tinybvh_amd import scenes
from oracle_lib import compare_hits

pytestmark = pytest.mark.gpu

LAYOUTS = [tb.LAYOUT_BVH_GPU, tb.LAYOUT_BVH4_GPU, tb.LAYOUT_CWBVH]
N_SEEDS = int(os.environ.get("TBVH_RANDOM_SEEDS", "48"))   # a longer hunt: TBVH_RANDOM_SEEDS=2000 python -m pytest tests/test_random_configs.py -m gpu


def make_scene(rng):
    kind = int(rng.integers(0, 4))
    if kind == 0:
        return scenes.soup(int(rng.integers(1, 4000)), seed=int(rng.integers(1, 1 << 20)))
    if kind == 1:
        return scenes.blob(int(rng.integers(200, 6000)), seed=int(rng.integers(1, 1 << 20)))
    if kind == 2:
        return scenes.atrium(int(rng.integers(2000, 20000)), seed=int(rng.integers(1, 1 << 20)))
    v = scenes.soup(int(rng.integers(50, 1500)), seed=int(rng.integers(1, 1 << 20))).reshape(-1, 3, 4)
    return np.ascontiguousarray(np.concatenate([v, v[: max(1, v.shape[0] // 3)]]).reshape(-1, 4))   # duplicates: ties everywhere


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_random_configuration(ctx, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    verts = make_scene(rng)
    layout = LAYOUTS[int(rng.integers(0, 3))]
    opts = {}
    if layout != tb.LAYOUT_BVH_GPU and rng.random() < 0.4:
        opts["greedy_collapse"] = True
    if rng.random() < 0.3:
        opts["bins"] = int(rng.choice([4, 16, 32]))
    on_device = layout != tb.LAYOUT_BVH_GPU and rng.random() < 0.3
    cls = tb.LAYOUT_CLASSES[layout]
    if on_device:
        ploc = rng.random() < 0.5
        sc = cls(ctx).BuildOnDevice(verts, builder="ploc", radius=int(rng.choice([0, 8, 32]))) if ploc else cls(ctx).BuildOnDevice(verts)
        host = tb.HostBVH(verts, layout)
    else:
        sc = cls(ctx).Build(verts, **opts)
        host = sc.host
    lo, hi = verts[:, :3].min(0), verts[:, :3].max(0)
    pad = 0.1 * (hi - lo) + 0.01
    n = int(rng.choice([1, 63, 64, 65, 127, 129, 4095, 4097, 20000, 33333]))
    tmax = np.float32(rng.choice([1e30, float(np.linalg.norm(hi - lo)) * 0.3, 0.0]))
    rays = R.random_rays(n, lo - pad, hi + pad, seed=int(rng.integers(1, 1 << 20)), tmax=tmax)
    want = oracle.bvh2_intersect(host.bvh2_nodes(), host.bvh2_prim_idx(), verts, rays)

    mode = int(rng.integers(0, 4))
    if mode == 0:                                   # host array, packed
        got = sc.Intersect(rays.copy())
    elif mode == 1:                                 # host array, 128-byte stride in place
        wide = np.zeros((n, 2), dtype=tb.RAY_DTYPE); wide[:, 0] = rays; wide[:, 1]["t"] = 7.0
        flat = wide.reshape(-1)
        import ctypes as C
        tb.check(tb.lib.tbvh_intersect(sc._h, C.c_void_p(flat.ctypes.data), n, 128), "tbvh_intersect")
        got = wide[:, 0].copy()
        assert np.all(wide[:, 1]["t"] == 7.0)
    elif mode == 2:                                 # device-resident, in place
        d = ctx.malloc(n * 64); ctx.to_device(d, rays)
        sc.intersect_device(d, n)
        got = np.zeros(n, tb.RAY_DTYPE); ctx.from_device(got, d); ctx.free(d)
    else:                                           # device-resident, fresh entry point: every record written
        d = ctx.malloc(n * 64); scrambled = rays.copy(); scrambled["t"] = 123.0; scrambled["prim"] = 77
        ctx.to_device(d, scrambled)
        sc.intersect_device_fresh(d, n, float(tmax))
        got = np.zeros(n, tb.RAY_DTYPE); ctx.from_device(got, d); ctx.free(d)
        miss = want["t"] >= tmax if tmax < 1e30 else want["t"] >= 1e30
        miss &= want["prim"] == rays["prim"]
        want = want.copy(); want["u"][miss] = 0; want["v"][miss] = 0; want["prim"][miss] = 0; want["t"][miss] = tmax
    c = compare_hits(got, want, rtol=1e-5)
    assert c["hitmiss"] == 0 and c["prim_real"] == 0 and c["t_bad"] == 0 and c["uv_bad"] == 0 and c["tie"] == 0, (seed, layout, opts, on_device, n, float(tmax), mode, c)
    assert c["bit_identical"] == c["same_prim"], (seed, c)
    # any-hit on the same rays: occluded iff the closest-hit oracle changed the record (a hit within [0, tmax])
    ref = oracle.bvh2_intersect(host.bvh2_nodes(), host.bvh2_prim_idx(), verts, rays)
    hit = (ref["prim"] != rays["prim"]) | (ref["t"] != rays["t"])
    occ = sc.IsOccluded(rays.copy())
    assert int((occ.astype(bool) != hit).sum()) <= max(2, n // 2000), (seed, layout, n, float(tmax), mode)


@pytest.mark.parametrize("seed", range(max(N_SEEDS // 3, 1)))
def test_random_tlas_configuration(ctx, oracle, seed):
    """The same for two-level scenes: 1-3 BLASes of random layouts (also mixed under one TLAS), 1-200 instances with random rigid + non-uniform
    scale transforms and masks, random batch sizes; host TLAS build or the device rebuild; against BVH::IntersectTLAS restated."""
    from test_tlas import grid_instances, oracle_tlas, check
    rng = np.random.default_rng(5000 + seed)
    n_blas = int(rng.integers(1, 4))
    meshes = []
    for k in range(n_blas):
        m = scenes.blob(int(rng.integers(300, 4000)), seed=int(rng.integers(1, 1 << 20))) if rng.random() < 0.5 else scenes.soup(int(rng.integers(50, 1500)), seed=int(rng.integers(1, 1 << 20)), extent=1.6, size=0.25)
        if m[:, :3].min() >= 0:
            m = m.copy(); m[:, :3] -= 0.8
        meshes.append(m)
    mixed = rng.random() < 0.5
    if mixed:
        layouts = [[tb.LAYOUT_CWBVH, tb.LAYOUT_BVH_GPU, tb.LAYOUT_BVH4_GPU][int(rng.integers(0, 3))] for _ in range(n_blas)]
    else:
        layouts = [[tb.LAYOUT_CWBVH, tb.LAYOUT_BVH_GPU, tb.LAYOUT_BVH4_GPU][int(rng.integers(0, 3))]] * n_blas
    blas = [tb.LAYOUT_CLASSES[l](ctx).Build(meshes[i]) for i, l in enumerate(layouts)]
    side = int(rng.integers(1, 6))
    inst = grid_instances(side, float(rng.uniform(0.3, 0.7)), int(rng.integers(1, 1 << 20)), n_blas=n_blas)
    if rng.random() < 0.5:
        inst["mask"][:: int(rng.integers(2, 6))] = 0x0001
    tlas = tb.TLAS(ctx).Build(inst, blas)
    if rng.random() < 0.4:
        tlas.RebuildOnDevice()
    n = int(rng.choice([1, 65, 4097, 30000]))
    rays = R.random_rays(n, (-2, -2, -2), (2.0 * side + 1, 2.0 * side + 1, 2.0 * side + 1), seed=int(rng.integers(1, 1 << 20)))
    if rng.random() < 0.5:
        rays["mask"][::3] = 0x00F0
    want = oracle_tlas(oracle, tlas, blas, rays)
    check(tlas.Intersect(rays.copy()), want)
    occ = tlas.IsOccluded(rays.copy())
    assert int((occ.astype(bool) != (want["t"] < 1e30)).sum()) <= 2


@pytest.mark.parametrize("seed", range(max(N_SEEDS // 2, 1)))
def test_random_hybrid_copy_configuration(ctx, oracle, seed):
    """The placement incoherent batches of large scenes are traced on — priority-ordered node copy with a random split between packed and one-per-line
    nodes, one triangle embedded in every padded node's line, 64-byte triangle records — forced onto SMALL random scenes (tbvh_cwbvh_set_hybrid +
    variant 90), where the oracle is cheap: trees of the library's builder in both collapse flavours, of the device builders (level order: no
    renumbering), and leaves of up to three triangles (the embedded triangle is then the first of a multi-triangle leaf); ragged batch sizes,
    infinite and finite ranges, closest-hit and any-hit; a refit to moved vertices and an in-place tbvh_update_cwbvh keep the copies current."""
    rng = np.random.default_rng(9000 + seed)
    verts = make_scene(rng)
    opts = {}
    if rng.random() < 0.4:
        opts["greedy_collapse"] = True
    if rng.random() < 0.4:
        opts["max_leaf_tris"] = int(rng.choice([2, 3]))
    on_device = rng.random() < 0.25
    if on_device:
        sc = tb.BVH8_CWBVH(ctx).BuildOnDevice(verts, max_leaf_tris=int(rng.choice([0, 3])))
        host = tb.HostBVH(verts, tb.LAYOUT_CWBVH)
    else:
        sc = tb.BVH8_CWBVH(ctx).Build(verts, **opts)
        host = sc.host
    n_nodes = sc.download_blobs()[0].shape[0] // 5
    sc.set_hybrid(int(rng.choice([0, 8, 64, max(n_nodes // 2, 8), 10**9])))
    sc.set_variant(90)
    lo, hi = verts[:, :3].min(0), verts[:, :3].max(0)
    pad = 0.1 * (hi - lo) + 0.01
    n = int(rng.choice([1, 63, 65, 129, 4097, 20000, 33333]))
    tmax = np.float32(rng.choice([1e30, float(np.linalg.norm(hi - lo)) * 0.3]))
    rays = R.random_rays(n, lo - pad, hi + pad, seed=int(rng.integers(1, 1 << 20)), tmax=tmax)

    def check_against(h, v, what):
        want = oracle.bvh2_intersect(h.bvh2_nodes(), h.bvh2_prim_idx(), v, rays)
        c = compare_hits(sc.Intersect(rays.copy()), want, rtol=1e-5)
        assert c["hitmiss"] == 0 and c["prim_real"] == 0 and c["t_bad"] == 0 and c["uv_bad"] == 0 and c["tie"] == 0, (seed, what, opts, on_device, n, float(tmax), c)
        assert c["bit_identical"] == c["same_prim"], (seed, what, c)
        hit = (want["prim"] != rays["prim"]) | (want["t"] != rays["t"])
        occ = sc.IsOccluded(rays.copy())
        assert int((occ.astype(bool) != hit).sum()) <= max(2, n // 2000), (seed, what, n, float(tmax))

    check_against(host, verts, "as placed")
    moved = verts.copy(); moved[:, 2] += np.float32(0.03 * float((hi - lo).max())) * np.sin(verts[:, 0] * 3).astype(np.float32)
    h2 = tb.HostBVH(moved, tb.LAYOUT_BVH2_WALD)
    sc.Refit(moved)
    check_against(h2, moved, "refitted")
    if rng.random() < 0.5:       # hand the refitted blob back through the in-place update (same shape: the copies are re-derived on the device)
        nodes, tris = sc.download_blobs()
        sc.Refit(verts)
        sc.Update(nodes, tris)
        check_against(h2, moved, "updated in place")
    sc.free()
