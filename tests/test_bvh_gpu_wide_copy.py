"""A BVH_GPU / BVH4_GPU scene makes an 8-wide copy of its tree at its first query (tinybvh_amd/csrc/capi_scene.hip: makeWideCopy) and its queries trace that copy: the hit records
must be the ones the uploaded 2-wide nodes give — byte for byte under the library's tie rule (device_common.h: hit_wins), variant 1 = k_bvh2 on the
nodes as uploaded — and the reference's own (golden vectors from the real tiny_bvh.h: BVH::Intersect, tiny_bvh.h:3222-3304).  Blobs: the reference's
BVH_GPU::Build and BuildHQ (SBVH: clipped leaf boxes, primIdx with repeats and slack) from tests/golden, and the library's own builder; the copy
follows tbvh_update_bvh_gpu, tbvh_refit and tbvh_set_opacity_micromaps; small blobs do not get one."""
import os

import numpy as np
import pytest

import tinybvh_amd as tb
from tinybvh_amd import rays as R
from tinybvh_amd import scenes
from oracle_lib import compare_hits

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture
def wide_from_one_entry():
    old = os.environ.get("TBVH_WIDE_COPY_MIN")
    os.environ["TBVH_WIDE_COPY_MIN"] = "1"
    yield
    if old is None:
        os.environ.pop("TBVH_WIDE_COPY_MIN", None)
    else:
        os.environ["TBVH_WIDE_COPY_MIN"] = old


@pytest.mark.parametrize("name", ["soup_2k", "atrium_6k", "suzanne_decimated"])
def test_reference_blobs_through_the_wide_copy(ctx, wide_from_one_entry, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    verts, rays = g["verts"], g["rays"]
    want = rays.copy()
    want.view(np.uint32).reshape(-1, 16)[:, 12:16] = g["hits"]
    for k in (0, 1):                                      # BVH_GPU::Build and BuildHQ
        nodes, idx = g[f"bvhgpu_nodes_{k}"], g[f"bvhgpu_idx_{k}"].reshape(-1)
        sc = tb.BVH_GPU(ctx).Upload(nodes, idx, verts)
        plain_bytes = nodes.shape[0] * 64 + idx.shape[0] * 48
        assert sc.device_bytes == plain_bytes               # nothing until the scene is queried (a BLAS under a TLAS never is)
        wide = sc.Intersect(rays.copy())
        assert sc.device_bytes > plain_bytes                # the copy exists
        sc.set_variant(1)
        native = sc.Intersect(rays.copy())
        sc.set_variant(0)
        assert np.array_equal(wide.view(np.uint8), native.view(np.uint8)), (name, k)
        c = compare_hits(wide, want)
        assert c["hitmiss"] == 0 and c["prim_real"] == 0 and c["t_bad"] == 0 and c["uv_bad"] == 0 and c["tie"] <= 2, (name, k, c)
        occ = sc.IsOccluded(g["shadow_rays"].copy())
        assert int((occ != g["occluded"]).sum()) == 0
        sc.free()


def test_library_built_scene_update_refit_micromaps(ctx, oracle):
    verts = scenes.atrium(60_000, seed=3)
    sc = tb.BVH_GPU(ctx).Build(verts)
    h = sc.host
    rays = np.concatenate([R.random_rays(60_000, (-20, 0, -10), (20, 15, 10), seed=8), R.primary(R.camera(*scenes.SPONZA_CAMERAS[0], 256, 128, 1, 1))])
    want = oracle.bvh2_intersect(h.bvh2_nodes(), h.bvh2_prim_idx(), verts, rays)

    def both():
        a = sc.Intersect(rays.copy())
        assert sc.device_bytes > h.blob(0, np.uint32, 16).shape[0] * 64 + h.blob(1, np.uint32, 1).shape[0] * 48      # (made by the first query)
        sc.set_variant(1)
        b = sc.Intersect(rays.copy())
        sc.set_variant(0)
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
        return a
    c = compare_hits(both(), want)
    assert c["hits"] > 10_000 and c["hitmiss"] == 0 and c["prim_real"] == 0 and c["t_bad"] == 0 and c["tie"] == 0 and c["bit_identical"] == c["same_prim"], c
    # refit to moved vertices: both the uploaded nodes and the copy follow; the oracle's answer on a tree refitted the same way = the tree's own answer
    moved = verts.copy()
    moved[:, 1] += np.float32(0.05) * np.sin(verts[:, 0]).astype(np.float32)
    sc.Refit(moved)
    a = both()
    h2 = tb.HostBVH(moved, tb.LAYOUT_BVH2_WALD)
    want2 = oracle.bvh2_intersect(h2.bvh2_nodes(), h2.bvh2_prim_idx(), moved, rays)
    c = compare_hits(a, want2)
    assert c["hitmiss"] == 0 and c["prim_real"] == 0 and c["t_bad"] == 0, c
    # update with another tree of the same size class (rebuilt over the moved vertices with the same builder)
    hb = tb.HostBVH(moved, tb.LAYOUT_BVH_GPU)
    nodes, idx = hb.blob(0, np.uint32, 16), hb.blob(1, np.uint32, 1).reshape(-1)
    if nodes.shape[0] * 4 <= h.blob(0, np.uint32, 16).shape[0] * 4 and idx.shape[0] <= h.blob(1, np.uint32, 1).shape[0]:
        sc.Update(nodes, idx, moved)
        c = compare_hits(both(), want2)
        assert c["hitmiss"] == 0 and c["prim_real"] == 0 and c["t_bad"] == 0, c
    # opacity micromaps: every second triangle fully transparent — the copy must honour the same maps
    n_tris = verts.shape[0] // 3
    N = 2
    words = np.zeros((n_tris, 1), np.uint32)
    words[0::2] = 0xffffffff
    sc.SetOpacityMicroMaps(words, N)
    a = both()
    hit = a["t"] < 1e30
    assert hit.sum() > 1000 and np.all(a["prim"][hit] % 2 == 0)
    sc.SetOpacityMicroMaps(None, 0)
    sc.free()


def test_small_blobs_keep_the_two_wide_kernel(ctx):
    verts = scenes.soup(2_000, seed=2)
    sc = tb.BVH_GPU(ctx).Build(verts)
    sc.Intersect(R.random_rays(4096, (0, 0, 0), (10, 10, 10), seed=1))
    h = sc.host
    assert sc.device_bytes == h.blob(0, np.uint32, 16).shape[0] * 64 + h.blob(1, np.uint32, 1).shape[0] * 48
    sc.free()


def same_or_found_more(wide, native):
    """byte-identical, except rays the BVH4_GPU kernel lost to the reference encoder's slightly non-conservative quantisation (254.999 / extent, tiny_bvh.h:5196-5231: the
    decoded box can fall short by 4e-6 relative and cull a grazing hit — DESIGN.md par. 4) and the copy, whose boxes are re-quantised outward, finds: at most 2"""
    diff = np.nonzero((wide.view(np.uint8).reshape(-1, 64) != native.view(np.uint8).reshape(-1, 64)).any(1))[0]
    for i in diff:
        assert wide["t"][i] < native["t"][i], (i, wide[i], native[i])
    assert diff.size <= 2, diff.size
    return diff.size


@pytest.mark.parametrize("name", ["soup_2k", "atrium_6k", "suzanne_decimated"])
def test_reference_bvh4_streams_through_the_wide_copy(ctx, wide_from_one_entry, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    rays = g["rays"]
    want = rays.copy()
    want.view(np.uint32).reshape(-1, 16)[:, 12:16] = g["hits"]
    for k in (0, 1):                                      # BVH4_GPU::Build and BuildHQ
        blocks = g[f"bvh4_{k}"]
        sc = tb.BVH4_GPU(ctx).Upload(blocks)
        wide = sc.Intersect(rays.copy())
        assert sc.device_bytes > blocks.shape[0] * 16       # the copy exists (made by the first query)
        sc.set_variant(1)
        native = sc.Intersect(rays.copy())
        sc.set_variant(0)
        same_or_found_more(wide, native)
        c = compare_hits(wide, want)
        assert c["hitmiss"] <= 1 and c["prim_real"] == 0 and c["t_bad"] == 0 and c["uv_bad"] == 0 and c["tie"] <= 2, (name, k, c)
        occ = sc.IsOccluded(g["shadow_rays"].copy())
        assert int((occ != g["occluded"]).sum()) <= 1
        sc.free()


def test_library_built_bvh4_scene_refit_and_micromaps(ctx, oracle):
    verts = scenes.atrium(60_000, seed=3)
    sc = tb.BVH4_GPU(ctx).Build(verts)
    h = sc.host
    rays = np.concatenate([R.random_rays(60_000, (-20, 0, -10), (20, 15, 10), seed=8), R.primary(R.camera(*scenes.SPONZA_CAMERAS[0], 256, 128, 1, 1))])
    want = oracle.bvh2_intersect(h.bvh2_nodes(), h.bvh2_prim_idx(), verts, rays)

    def both():
        a = sc.Intersect(rays.copy())
        sc.set_variant(1)
        b = sc.Intersect(rays.copy())
        sc.set_variant(0)
        assert sc.device_bytes > h.blob(0, np.uint32, 4).shape[0] * 16
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))      # (the library's own encoder quantises conservatively: nothing for the copy to find in addition)
        return a
    c = compare_hits(both(), want)
    assert c["hits"] > 10_000 and c["hitmiss"] == 0 and c["prim_real"] == 0 and c["t_bad"] == 0 and c["tie"] == 0 and c["bit_identical"] == c["same_prim"], c
    moved = verts.copy()
    moved[:, 1] += np.float32(0.05) * np.sin(verts[:, 0]).astype(np.float32)
    sc.Refit(moved)
    h2 = tb.HostBVH(moved, tb.LAYOUT_BVH2_WALD)
    want2 = oracle.bvh2_intersect(h2.bvh2_nodes(), h2.bvh2_prim_idx(), moved, rays)
    c = compare_hits(both(), want2)
    assert c["hitmiss"] == 0 and c["prim_real"] == 0 and c["t_bad"] == 0, c
    n_tris = verts.shape[0] // 3
    words = np.zeros((n_tris, 1), np.uint32)
    words[0::2] = 0xffffffff
    sc.SetOpacityMicroMaps(words, 2)
    a = both()
    hit = a["t"] < 1e30
    assert hit.sum() > 1000 and np.all(a["prim"][hit] % 2 == 0)
    sc.SetOpacityMicroMaps(None, 0)
    sc.free()


def test_tlas_any_hit_queries_enter_bvh4_blases_through_their_copies(ctx, oracle):
    """BVH4_GPU BLASes under a TLAS: Intersect walks their own streams (k_tlas4), IsOccluded their 8-wide copies (k_tlas8; capi_scene.hip: reclassifyTlas) —
    made by the TLAS's first any-hit query for BLASes of 32 k triangles and more; a small BLAS next to a large one keeps every query on the uploaded streams."""
    from test_tlas import grid_instances, oracle_tlas, check
    mesh = scenes.blob(40_000, seed=7)
    mesh[:, :3] -= 0.5 * (mesh[:, :3].min(0) + mesh[:, :3].max(0))
    mesh[:, :3] *= np.float32(1.6 / float((mesh[:, :3].max(0) - mesh[:, :3].min(0)).max()))
    blas = tb.BVH4_GPU(ctx).Build(mesh)
    before = blas.device_bytes
    inst = grid_instances(4, 0.5, 3)
    tlas = tb.TLAS(ctx).Build(inst, [blas])
    rays = np.concatenate([R.random_rays(60_000, (-2, -2, -2), (9, 9, 9), seed=4), R.primary(R.camera((-3.0, 4.0, -5.0), (0.5, -0.2, 0.84), 256, 256, 1, 1))])
    want = oracle_tlas(oracle, tlas, [blas], rays)
    check(tlas.Intersect(rays.copy()), want)
    assert blas.device_bytes == before                     # a TLAS that only ever answers Intersect pays for no copy (nor for a second wide tree per rebuild)
    occ = tlas.IsOccluded(rays.copy())
    assert blas.device_bytes > before                      # the copy IsOccluded walks, made by the first such query
    assert int((occ.astype(bool) != (want["t"] < 1e30)).sum()) <= 2
    blas.set_variant(1)                                     # the uploaded stream for every query
    occ_native = tlas.IsOccluded(rays.copy())
    assert int((occ != occ_native).sum()) <= 2
    blas.set_variant(0)
    assert np.array_equal(tlas.IsOccluded(rays.copy()), occ)
    blas.Refit(mesh)                                        # both forms follow a refit
    assert np.array_equal(tlas.IsOccluded(rays.copy()), occ)
    check(tlas.Intersect(rays.copy()), want)
    tlas.free()
    small = tb.BVH4_GPU(ctx).Build(scenes.soup(2_000, seed=9, extent=1.6, size=0.2))
    small_before = small.device_bytes
    tlas2 = tb.TLAS(ctx).Build(grid_instances(3, 0.5, 5, n_blas=2), [blas, small])
    assert small.device_bytes == small_before
    want2 = oracle_tlas(oracle, tlas2, [blas, small], rays)
    check(tlas2.Intersect(rays.copy()), want2)
    assert int((tlas2.IsOccluded(rays.copy()).astype(bool) != (want2["t"] < 1e30)).sum()) <= 2
    tlas2.free(); blas.free(); small.free()


def test_tlas_enters_bvh_gpu_blases_through_their_wide_copies(ctx, oracle):
    """A TLAS over BVH_GPU BLASes (the reference's BLAS type for geometry that is refitted or rebuilt, traverse_tlas.cl:66-72) enters them through their
    8-wide copies, made at the TLAS upload (capi_scene.hip: reclassifyTlas, blasView): the records are BVH::IntersectTLAS's either way; tbvh_set_variant(blas, 1)
    puts the TLAS back on the uploaded nodes, a BLAS update is followed, a BLAS freed before its TLAS lives on with its copy."""
    from test_tlas import grid_instances, oracle_tlas, check
    meshes = [scenes.blob(40_000, seed=5), scenes.soup(3_000, seed=6, extent=1.6, size=0.2)]       # a small BLAS gets its copies too when a TLAS wants them: one kernel class for the TLAS
    for m in meshes:
        m[:, :3] -= 0.5 * (m[:, :3].min(0) + m[:, :3].max(0))
        m[:, :3] *= np.float32(1.6 / float((m[:, :3].max(0) - m[:, :3].min(0)).max()))
    blas = [tb.BVH_GPU(ctx).Build(m) for m in meshes]
    before = [b.device_bytes for b in blas]
    inst = grid_instances(4, 0.5, 3, n_blas=2)
    tlas = tb.TLAS(ctx).Build(inst, blas)
    assert blas[0].device_bytes > before[0] and blas[1].device_bytes > before[1]
    small_direct = blas[1].Intersect(R.random_rays(4096, (-1, -1, -1), (1, 1, 1), seed=2))         # ... while the small BLAS's OWN queries keep the uploaded nodes
    blas[1].set_variant(1)
    assert np.array_equal(small_direct.view(np.uint8), blas[1].Intersect(R.random_rays(4096, (-1, -1, -1), (1, 1, 1), seed=2)).view(np.uint8))
    blas[1].set_variant(0)
    rays = np.concatenate([R.random_rays(60_000, (-2, -2, -2), (9, 9, 9), seed=4), R.primary(R.camera((-3.0, 4.0, -5.0), (0.5, -0.2, 0.84), 256, 256, 1, 1))])
    want = oracle_tlas(oracle, tlas, blas, rays)
    through_copies = tlas.Intersect(rays.copy())
    check(through_copies, want)
    occ = tlas.IsOccluded(rays.copy())
    assert int((occ.astype(bool) != (want["t"] < 1e30)).sum()) <= 2
    blas[0].set_variant(1)                                   # the uploaded nodes again (k_tlas2)
    native = tlas.Intersect(rays.copy())
    check(native, want)
    assert np.array_equal(native.view(np.uint8), through_copies.view(np.uint8))
    blas[0].set_variant(0)
    assert np.array_equal(tlas.Intersect(rays.copy()).view(np.uint8), through_copies.view(np.uint8))
    # the BLAS is rebuilt over other vertices and updated in place: the copy is made again, the TLAS follows
    moved = np.ascontiguousarray(meshes[0][: 3 * 36_000]).copy(); moved[:, 1] *= np.float32(0.8)     # (fewer triangles: the blob fits the allocation; still enough for a copy)
    h2 = tb.HostBVH(moved, tb.LAYOUT_BVH_GPU)
    blas[0].Update(h2.blob(0, np.uint32, 16), h2.blob(1, np.uint32, 1), moved)
    blas[0].host = h2
    want2 = oracle_tlas(oracle, tlas, blas, rays)
    check(tlas.Intersect(rays.copy()), want2)
    # freed before its TLAS: the TLAS keeps what it traverses
    blas[0].free()
    check(tlas.Intersect(rays.copy()), want2)
    tlas.free(); blas[1].free()


def test_tlas_closest_hit_queries_enter_cwbvh_blases_through_4_wide_copies(ctx, oracle):
    """BVH8_CWBVH BLASes under a TLAS (the configuration of tiny_bvh_gpu2.cpp): Intersect walks 4-wide copies (k_tlas4; host_builder.cpp: cwbvh_to_bvh2 + the device
    converter in record mode), IsOccluded the uploaded nodes (k_tlas8).  Records are IntersectTLAS's; a forced variant on the BLAS pins the uploaded nodes for
    every query; tbvh_update_cwbvh and tbvh_refit are followed."""
    from test_tlas import grid_instances, oracle_tlas, check
    from test_refit_device import deform
    mesh = scenes.blob(40_000, seed=11)
    mesh[:, :3] -= 0.5 * (mesh[:, :3].min(0) + mesh[:, :3].max(0))
    mesh[:, :3] *= np.float32(1.6 / float((mesh[:, :3].max(0) - mesh[:, :3].min(0)).max()))
    blas = tb.BVH8_CWBVH(ctx).Build(mesh)
    before = blas.device_bytes
    tlas = tb.TLAS(ctx).Build(grid_instances(4, 0.5, 3), [blas])
    assert blas.device_bytes > before
    rays = np.concatenate([R.random_rays(60_000, (-2, -2, -2), (9, 9, 9), seed=4), R.primary(R.camera((-3.0, 4.0, -5.0), (0.5, -0.2, 0.84), 256, 256, 1, 1))])
    want = oracle_tlas(oracle, tlas, [blas], rays)
    via4 = tlas.Intersect(rays.copy())
    check(via4, want)
    occ = tlas.IsOccluded(rays.copy())
    assert int((occ.astype(bool) != (want["t"] < 1e30)).sum()) <= 2
    blas.set_variant(72)                                    # (any forced variant: the TLAS goes back to the uploaded nodes)
    native = tlas.Intersect(rays.copy())
    blas.set_variant(0)
    assert np.array_equal(native.view(np.uint8), via4.view(np.uint8))
    moved = deform(mesh, 0.05, seed=2)
    blas.Refit(moved)
    blas.host = tb.HostBVH(moved, tb.LAYOUT_CWBVH)
    check(tlas.Intersect(rays.copy()), oracle_tlas(oracle, tlas, [blas], rays))
    h2 = tb.HostBVH(mesh, tb.LAYOUT_CWBVH)
    blas.Update(h2.blob(0, np.uint32, 4), h2.blob(1, np.uint32, 4))
    blas.host = h2
    check(tlas.Intersect(rays.copy()), want)
    tlas.free(); blas.free()


def test_an_update_drops_the_copies_until_the_blob_has_settled(ctx, oracle):
    """tbvh_update_* is the reference's animation flow (BVH::Refit + ConvertFrom on the host, the blob re-uploaded every frame): making the copies again costs more
    than a frame's queries gain, so an update drops them and they come back after four queries without another update — and an update that arrives soon
    after they came back makes the scene wait four times as long (capi_internal.h: tbvh_scene::pendingCopies).  Results are right throughout."""
    verts = scenes.blob(40_000, seed=21)
    sc = tb.BVH_GPU(ctx).Build(verts)
    h = sc.host
    plain = h.blob(0, np.uint32, 16).shape[0] * 64 + h.blob(1, np.uint32, 1).shape[0] * 48
    lo, hi = verts[:, :3].min(0), verts[:, :3].max(0)
    rays = R.random_rays(20_000, lo - 0.1, hi + 0.1, seed=3)
    want = oracle.bvh2_intersect(h.bvh2_nodes(), h.bvh2_prim_idx(), verts, rays)

    def ask():
        c = compare_hits(sc.Intersect(rays.copy()), want)
        assert c["hitmiss"] == 0 and c["prim_real"] == 0 and c["t_bad"] == 0, c

    ask()
    assert sc.device_bytes > plain                        # the 8-wide copy of the first query
    nodes, idx = h.blob(0, np.uint32, 16), h.blob(1, np.uint32, 1).reshape(-1)
    sc.Update(nodes, idx, verts)
    assert sc.device_bytes == plain                       # dropped
    for k in range(3):
        ask()
        assert sc.device_bytes == plain, k
    ask()
    assert sc.device_bytes > plain                        # back with the fourth query
    sc.Update(nodes, idx, verts)                          # again, soon after: the scene now waits for 16 queries
    for k in range(15):
        ask()
        assert sc.device_bytes == plain, k
    ask()
    assert sc.device_bytes > plain
    sc.free()


def test_a_refit_keeps_the_copies_only_if_enough_rays_were_traced_since_the_last_one(ctx):
    """tbvh_refit refits a scene's copies in place (0.3-0.5 ms each per 100 k triangles): that pays from about 8 M rays per refit on.  A mesh refitted again
    after fewer rays loses its copies (they come back like after an update); one that traces a frame's worth of rays in between keeps them."""
    verts = scenes.blob(40_000, seed=23)
    sc = tb.BVH_GPU(ctx).Build(verts)
    h = sc.host
    plain = h.blob(0, np.uint32, 16).shape[0] * 64 + h.blob(1, np.uint32, 1).shape[0] * 48
    lo, hi = verts[:, :3].min(0), verts[:, :3].max(0)
    few = R.random_rays(4096, lo - 0.1, hi + 0.1, seed=3)
    n_many = 9 << 20
    d = ctx.malloc(n_many * 64)
    cam = R.camera(tuple(hi + (hi - lo)), tuple(-(hi - lo) / np.linalg.norm(hi - lo)), 3072, 3072, 1, 1)
    ctx.generate_primary(cam, d, 0, n_many)
    sc.Intersect(few.copy())
    assert sc.device_bytes > plain
    sc.Refit(verts)                                        # the first refit: nothing to compare with, the copy is refitted
    assert sc.device_bytes > plain
    sc.intersect_device_fresh(d, n_many, 1e30)             # 9.4 M rays
    sc.Refit(verts)
    assert sc.device_bytes > plain                         # kept
    sc.Intersect(few.copy())
    sc.Refit(verts)                                        # 4096 rays since the last refit
    assert sc.device_bytes == plain                        # dropped
    for k in range(4):
        sc.Intersect(few.copy())
    assert sc.device_bytes > plain                         # ... and back after four queries
    ctx.free(d); sc.free()
