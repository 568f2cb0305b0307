"""One ray array over several devices through the C ABI (tbvh_intersect_sharded / tbvh_occluded_sharded):
the BVH replicated per context, contiguous wave-aligned shards, one host thread per device, results in place.  The test uses
every HIP device of the box, and at least two contexts (on a 1-GPU box both live on device 0, so the shard arithmetic, the
threads and the in-place write-back run exactly as they do with 8 devices); the sharded result must equal the
single-context result byte for byte."""
import ctypes as C

import numpy as np
import pytest

import tinybvh_amd as tb
from tinybvh_amd import rays as R
from tinybvh_amd import scenes
from tinybvh_amd.sharding import shard_range


def test_shard_range_matches_the_python_helper():
    b, e = C.c_uint64(), C.c_uint64()
    for n in (0, 1, 63, 64, 65, 1000, 16_777_216, 67_108_864 + 17):
        for world in (1, 2, 3, 8):
            for rank in range(world):
                tb.lib.tbvh_shard_range(n, rank, world, C.byref(b), C.byref(e))
                assert (b.value, e.value) == shard_range(n, rank, world), (n, rank, world)


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [tb.LAYOUT_BVH_GPU, tb.LAYOUT_BVH4_GPU, tb.LAYOUT_CWBVH])
@pytest.mark.parametrize("n_rays,dtype_bytes", [(100_003, 64), (40_000, 128), (77, 64)])
def test_sharded_equals_single_device(layout, n_rays, dtype_bytes):
    n_dev = tb.device_count()
    n_ctx = max(2, n_dev)
    verts = scenes.soup(8192, seed=7)
    ctxs = [tb.Context(i % n_dev) for i in range(n_ctx)]
    try:
        reps = [tb.LAYOUT_CLASSES[layout](c).Build(verts) for c in ctxs]
        rays = R.random_rays(n_rays, (0, 0, 0), (10, 10, 10), seed=3)
        if dtype_bytes == 128:   # a tinybvh::Ray[] passed in place: 64 bytes of record + 64 of user area
            wide = np.zeros(n_rays, np.dtype([("r", tb.RAY_DTYPE), ("user", "<u4", 16)]))
            wide["r"] = rays; wide["user"] = 0xABCD1234
            single = wide.copy(); shard = wide.copy()
            tb.check(tb.lib.tbvh_intersect(reps[0]._h, single.ctypes.data_as(C.c_void_p), n_rays, 128), "tbvh_intersect")
            tb.intersect_sharded(reps, shard)
            assert np.array_equal(single.view(np.uint8), shard.view(np.uint8))
            assert np.all(shard["user"] == 0xABCD1234)
            assert int((shard["r"]["t"] < 1e30).sum()) > n_rays // 20
            occ1 = reps[0].IsOccluded(rays.copy())
            assert np.array_equal(tb.occluded_sharded(reps, shard), occ1)
        else:
            single = reps[0].Intersect(rays.copy())
            shard = tb.intersect_sharded(reps, rays.copy())
            assert np.array_equal(single.view(np.uint8), shard.view(np.uint8))
            assert np.array_equal(tb.occluded_sharded(reps, rays.copy()), reps[0].IsOccluded(rays.copy()))
        # the same context twice is refused (one stream and staging area per shard)
        with pytest.raises(tb.TbvhError):
            tb.intersect_sharded([reps[0], reps[0]], rays.copy())
        for r in reps:
            r.free()
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [tb.LAYOUT_BVH4_GPU, tb.LAYOUT_CWBVH])
def test_device_resident_shards_equal_single_device(layout):
    """tbvh_intersect_sharded_device / tbvh_occluded_sharded_device: every shard generated, traced and kept on its own device (context); the
    records equal what ONE context produces for the whole batch, byte for byte."""
    n_dev = tb.device_count()
    n_ctx = max(2, n_dev)
    verts = scenes.atrium(60_000, seed=4)
    side = 512
    n = side * side
    cam = R.camera(*scenes.SPONZA_CAMERAS[0], side, side, 1, 1)
    ctxs = [tb.Context(i % n_dev) for i in range(n_ctx)]
    try:
        reps = [tb.LAYOUT_CLASSES[layout](c).Build(verts) for c in ctxs]
        # reference: everything on context 0
        d_all = ctxs[0].malloc(n * 64); d_occ_all = ctxs[0].malloc(n)
        ctxs[0].generate_primary(cam, d_all, 0, n)
        reps[0].intersect_device_fresh(d_all, n, 1e30)
        want = np.zeros(n, tb.RAY_DTYPE); ctxs[0].from_device(want, d_all)
        reps[0].occluded_device(d_all, n, d_occ_all)     # rays with hit.t = distance of the hit: occluded iff they hit (a triangle IN a box face may differ by an ulp)
        want_occ = np.zeros(n, np.uint8); ctxs[0].from_device(want_occ, d_occ_all)
        # shards: each context generates and traces its own range of the same batch
        ranges = [shard_range(n, r, n_ctx) for r in range(n_ctx)]
        d_rays, d_occ = [], []
        for c, (b, e) in zip(ctxs, ranges):
            d = c.malloc(max(e - b, 1) * 64); c.generate_primary(cam, d, b, e - b)
            d_rays.append(d); d_occ.append(c.malloc(max(e - b, 1)))
        km, dm = tb.intersect_sharded_device(reps, d_rays, [e - b for b, e in ranges], fresh=True)
        assert all(k > 0 for k in km) and all(d >= 0 for d in dm)
        got = np.zeros(n, tb.RAY_DTYPE)
        for c, d, (b, e) in zip(ctxs, d_rays, ranges):
            part = np.zeros(e - b, tb.RAY_DTYPE); c.from_device(part, d); got[b:e] = part
        assert int((got["t"] < 1e30).sum()) > n // 2
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8))
        tb.occluded_sharded_device(reps, d_rays, [e - b for b, e in ranges], d_occ)
        got_occ = np.zeros(n, np.uint8)
        for c, d, (b, e) in zip(ctxs, d_occ, ranges):
            part = np.zeros(e - b, np.uint8); c.from_device(part, d); got_occ[b:e] = part
        assert np.array_equal(got_occ, want_occ)
        with pytest.raises(tb.TbvhError):       # one scene per context
            tb.intersect_sharded_device([reps[0], reps[0]], d_rays[:2], [64, 64])
        for r in reps:
            r.free()
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.gpu
def test_wavefront_bands_over_several_contexts_render_the_same_image():
    """tbvh_wavefront_render_sharded: the image cut into bands of rows, one wavefront path tracer per device (context), the scene replicated;
    every band draws the random numbers the full frame draws for its pixels, so the gathered image is the one-context image up to the order of
    the float accumulations (a few ulps per pixel)."""
    n_dev = tb.device_count()
    n_ctx = max(2, n_dev)
    verts = scenes.atrium(60_000, seed=4)
    W, H = 256, 128
    cam = R.camera(*scenes.SPONZA_CAMERAS[0], W, H, 1, 1)
    light = (0.0, 0.9 * float(verts[:, 1].max()), 0.0)
    ctxs = [tb.Context(i % n_dev) for i in range(n_ctx + 1)]     # the last one renders the whole image
    try:
        reps = [tb.BVH8_CWBVH(c).Build(verts) for c in ctxs]
        dv = []
        for c in ctxs:
            d = c.malloc(verts.nbytes); c.to_device(d, verts); dv.append(d)
        full = tb.Wavefront(ctxs[-1], W, H)
        st_full = full.render(reps[-1], dv[-1], cam, light, (300.0, 300.0, 300.0), max_depth=3, seed=5)
        want = full.read()
        rows = [(H // 4) * r // n_ctx * 4 for r in range(n_ctx + 1)]
        wfs = [tb.Wavefront(ctxs[i], W, rows[i + 1] - rows[i]).set_band(rows[i], H) for i in range(n_ctx)]
        st = tb.wavefront_render_sharded(wfs, reps[:n_ctx], dv[:n_ctx], cam, light, (300.0, 300.0, 300.0), max_depth=3, seed=5)
        got = tb.wavefront_read_sharded(wfs, W, H)
        for d in range(3):   # the bands trace exactly the rays of the full frame
            assert sum(s["extend_rays"][d] for s in st) == st_full["extend_rays"][d]
            assert sum(s["shadow_rays"][d] for s in st) == st_full["shadow_rays"][d]
        assert want[..., :3].max() > 0
        assert np.allclose(got, want, rtol=1e-4, atol=1e-5 * float(want.max()))
        # bands out of order / not tiling the image are refused
        with pytest.raises(tb.TbvhError):
            tb.wavefront_render_sharded(wfs[::-1], reps[:n_ctx][::-1], dv[:n_ctx][::-1], cam, light)
        for w in wfs + [full]:
            w.close()
    finally:
        for c in ctxs:
            c.close()
