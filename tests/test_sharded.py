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
