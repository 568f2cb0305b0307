"""Boundary hardening of round 4: (1) host threads sharing ONE context and scene — the reference's per-ray API is const and its own
speedtest runs it from 8 threads on one BVH (tiny_bvh_speedtest.cpp:1077-1083), so a drop-in caller that keeps its thread loop must get
correct records: every entry point holds its context's lock; (2) tbvh_update_bvh_gpu / _bvh4_gpu / _cwbvh: a blob refitted and re-converted
on the host (BVH::Refit + ConvertFrom, tiny_bvh.h:3055-3093) goes into the existing scene in place."""
import threading

import numpy as np
import pytest

import tinybvh_amd as tb
from tinybvh_amd import rays as R
from tinybvh_amd import scenes
from oracle_lib import compare_hits

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("layout", [tb.LAYOUT_CWBVH, tb.LAYOUT_BVH4_GPU, tb.LAYOUT_BVH_GPU])
def test_threads_share_one_scene(ctx, oracle, layout):
    verts = scenes.atrium(40_000, seed=2)
    sc = tb.LAYOUT_CLASSES[layout](ctx).Build(verts)
    batches = [R.random_rays(20_000 + 3_000 * k, (-30, 0, -12), (30, 20, 12), seed=50 + k) for k in range(6)]
    batches.append(R.primary(R.camera(*scenes.SPONZA_CAMERAS[0], 256, 256, 1, 1)))       # one above the pinned-staging threshold (65 k rays)
    want = [sc.Intersect(b.copy()) for b in batches]                                        # single-threaded answers ...
    c = compare_hits(want[0], oracle.bvh2_intersect(sc.host.bvh2_nodes(), sc.host.bvh2_prim_idx(), verts, batches[0]))
    assert c["hits"] > 5000 and c["hitmiss"] == 0 and c["prim_real"] == 0 and c["tie"] == 0, c   # ... which are the oracle's
    occ_want = [sc.IsOccluded(b) for b in batches]
    errors = []

    def worker(k):
        try:
            for rep in range(4):
                j = (k + rep) % len(batches)
                got = sc.Intersect(batches[j].copy())
                if not np.array_equal(got.view(np.uint8), want[j].view(np.uint8)):
                    errors.append(f"thread {k} rep {rep}: Intersect records differ on batch {j}")
                if not np.array_equal(sc.IsOccluded(batches[j]), occ_want[j]):
                    errors.append(f"thread {k} rep {rep}: IsOccluded flags differ on batch {j}")
                ctx.time_last_ms()
        except Exception as e:   # noqa: BLE001
            errors.append(f"thread {k}: {e!r}")

    th = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:5]
    sc.free()


def _check(sc, oracle, host, verts, rays, what):
    got = sc.Intersect(rays.copy())
    c = compare_hits(got, oracle.bvh2_intersect(host.bvh2_nodes(), host.bvh2_prim_idx(), verts, rays))
    assert c["hits"] > 2000 and c["hitmiss"] == 0 and c["prim_real"] == 0 and c["t_bad"] == 0 and c["tie"] == 0, (what, c)


@pytest.mark.parametrize("layout", [tb.LAYOUT_CWBVH, tb.LAYOUT_BVH4_GPU, tb.LAYOUT_BVH_GPU])
def test_update_in_place(ctx, oracle, layout):
    """A re-converted blob (same size or smaller) replaces the scene's contents in place; a TLAS over the BLAS sees the new geometry without
    being told; a larger blob is refused with a status code and leaves the scene as it was."""
    verts = scenes.blob(30_000, seed=5)
    cls = tb.LAYOUT_CLASSES[layout]
    sc = cls(ctx).Build(verts)
    rays = R.random_rays(40_000, (-1.5, -1.5, -1.5), (1.5, 1.5, 1.5), seed=11)
    _check(sc, oracle, sc.host, verts, rays, "as uploaded")
    inst = tb.make_instances(np.eye(4, dtype=np.float32)[None], np.zeros(1, np.uint32))
    tlas = tb.TLAS(ctx).Build(inst, [sc])
    before = sc.device_bytes

    def blobs(h):
        if layout == tb.LAYOUT_CWBVH:
            return (h.blob(0, np.uint32, 4), h.blob(1, np.uint32, 4))
        if layout == tb.LAYOUT_BVH4_GPU:
            return (h.blob(0, np.uint32, 4),)
        return (h.blob(0, np.uint32, 16), h.blob(1, np.uint32, 1), h.verts)

    # (a) the same triangles, moved: what BVH::Refit + ConvertFrom hands over each frame
    moved = verts.copy(); moved[:, 0] += np.float32(0.05) * np.sin(verts[:, 1] * 4).astype(np.float32)
    h2 = tb.HostBVH(moved, layout)
    fits = all(b.nbytes <= a.nbytes for a, b in zip(blobs(sc.host), blobs(h2)))
    if fits:
        sc.Update(*blobs(h2))
        _check(sc, oracle, h2, moved, rays, "updated to moved vertices")
        got_t = tlas.Intersect(rays.copy())                          # the TLAS holds the BLAS's device pointers: same memory, new contents
        assert np.array_equal(got_t["t"], sc.Intersect(rays.copy())["t"])
    # (b) a smaller mesh in the same allocation
    small = scenes.blob(12_000, seed=6)
    h3 = tb.HostBVH(small, layout)
    sc.Update(*blobs(h3))
    _check(sc, oracle, h3, small, rays, "updated to a smaller mesh")
    assert sc.device_bytes <= before
    # (c) a larger one is refused; the scene keeps tracing what it holds
    big = tb.HostBVH(scenes.blob(60_000, seed=7), layout)
    with pytest.raises(tb.TbvhError, match="larger"):
        sc.Update(*blobs(big))
    _check(sc, oracle, h3, small, rays, "after the refused update")
    # (d) a malformed blob is refused like at upload
    if layout == tb.LAYOUT_CWBVH:
        bad = h3.blob(0, np.uint32, 4).copy(); bad[1, 0] = 0x7fffffff        # child base out of range
        with pytest.raises(tb.TbvhError):
            sc.Update(bad, h3.blob(1, np.uint32, 4))
    tlas.free(); sc.free()


def test_update_cwbvh_keeps_the_incoherent_copies_current(ctx, oracle):
    """With the hybrid node copy and the 64-byte triangle records in place (the incoherent flavor, forced by variant 90), an update with the
    same tree shape re-derives them on the device; one with another shape drops them (the scene falls back to the arrays as uploaded)."""
    verts = scenes.atrium(50_000, seed=4)
    sc = tb.BVH8_CWBVH(ctx).Build(verts)
    rays = R.random_rays(50_000, (-30, 0, -12), (30, 20, 12), seed=3)
    sc.set_hybrid(1024); sc.set_variant(90)
    _check(sc, oracle, sc.host, verts, rays, "incoherent flavor")
    # same topology: re-encode the SAME tree around moved vertices = refit on the host; the library's own device refit gives exactly such a blob
    moved = verts.copy(); moved[:, 1] += np.float32(0.02) * np.sin(verts[:, 0]).astype(np.float32)
    sc.set_variant(0)
    sc.Refit(moved)
    nodes, tris = sc.download_blobs()                                 # a refitted blob of the same shape
    sc.Refit(verts)                                                   # put the scene back ...
    sc.Update(nodes, tris)                                            # ... and hand the refitted blob over as a caller would
    sc.set_variant(90)
    h_moved = tb.HostBVH(moved, tb.LAYOUT_BVH2_WALD)
    _check(sc, oracle, h_moved, moved, rays, "updated, same shape, incoherent flavor")
    assert sc.device_bytes > (nodes.nbytes + tris.nbytes) * 2         # the copies are still there
    h3 = tb.HostBVH(scenes.atrium(20_000, seed=9), tb.LAYOUT_CWBVH)
    sc.set_variant(0)
    sc.Update(h3.blob(0, np.uint32, 4), h3.blob(1, np.uint32, 4))
    assert sc.device_bytes < (nodes.nbytes + tris.nbytes) * 1.1       # another tree: the copies went
    _check(sc, oracle, h3, h3.verts, rays, "updated, another tree")
    sc.free()


def test_time_history_reads_every_launch_once(ctx):
    """tbvh_time_history: launches enqueued back to back, their HIP-event durations read once afterwards (what bench.py's timed loop does)."""
    verts = scenes.soup(20_000, seed=2)
    sc = tb.BVH8_CWBVH(ctx).Build(verts)
    rays = R.random_rays(200_000, (0, 0, 0), (10, 10, 10), seed=1)
    d = ctx.malloc(rays.nbytes); ctx.to_device(d, rays)
    sizes = [200_000, 50_000, 200_000, 10_000, 120_000]
    for m in sizes:
        sc.intersect_device_fresh(d, m, 1e30)
    h = ctx.time_history(len(sizes))
    assert len(h) == len(sizes) and all(t > 0 for t in h), h
    assert abs(h[-1] - ctx.time_last_ms()) < 1e-6
    assert h[3] < h[0]                                   # 10 k rays take less than 200 k
    assert len(ctx.time_history(1000)) <= 256            # the ring remembers 256 operations
    ctx.free(d); sc.free()


def test_timing_can_be_switched_off(ctx):
    """tbvh_set_timing(0): queries enqueue their kernels only — the records are the same, the time calls keep the last TIMED operation; back on, they follow again."""
    verts = scenes.soup(20_000, seed=3)
    sc = tb.BVH8_CWBVH(ctx).Build(verts)
    rays = R.random_rays(100_000, (0, 0, 0), (10, 10, 10), seed=2)
    d = ctx.malloc(rays.nbytes); ctx.to_device(d, rays)
    sc.intersect_device_fresh(d, 100_000, 1e30)
    want = np.zeros_like(rays); ctx.from_device(want, d)
    t_timed = ctx.time_last_ms()
    n_hist = len(ctx.time_history(1000))
    try:
        ctx.set_timing(False)
        for m in (5_000, 100_000):
            sc.intersect_device_fresh(d, m, 1e30)
        got = np.zeros_like(rays); ctx.from_device(got, d)
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8))
        assert abs(ctx.time_last_ms() - t_timed) < 1e-6          # still the operation that was timed
        assert len(ctx.time_history(1000)) == n_hist
    finally:
        ctx.set_timing(True)
    sc.intersect_device_fresh(d, 5_000, 1e30)
    assert len(ctx.time_history(1000)) == min(n_hist + 1, 256) and 0 < ctx.time_last_ms() < t_timed
    ctx.free(d); sc.free()
