"""The wave-packet traversal of both levels (tinybvh_amd/csrc/kernels_tlas8_packet.hip; the first kernel of a launch on a TLAS over BVH8_CWBVH BLASes, batches
of 64 k rays and more) returns the records of the per-lane kernel behind it — byte for byte under the library's tie rule — and the oracle's
(BVH::IntersectTLAS restated, tiny_bvh.h:3306-3380), with instance masks, non-uniform rotated instances, finite tmax, any-hit, and a batch that is
coherent only in its first half (waves of the packet kernel and of the per-lane kernel then share one pool).  Debug flag 16 = the packet kernel takes
the batch whatever its probe says (incoherent rays through it); TBVH_TLAS_PACKET=0 (read when a context is made) = never."""
import os

import numpy as np
import pytest

import tinybvh_amd as tb
from tinybvh_amd import rays as R
from tinybvh_amd import scenes
from test_tlas import check, grid_instances, oracle_tlas

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx_no_packet():
    old = os.environ.get("TBVH_TLAS_PACKET")
    os.environ["TBVH_TLAS_PACKET"] = "0"
    try:
        c = tb.Context(0)
    finally:
        if old is None:
            os.environ.pop("TBVH_TLAS_PACKET", None)
        else:
            os.environ["TBVH_TLAS_PACKET"] = old
    yield c
    c.close()


def make_scene(c):
    verts = scenes.blob(6000, seed=3)
    verts2 = scenes.soup(2000, seed=9, extent=1.6, size=0.25); verts2[:, :3] -= 0.8
    blas = [tb.BVH8_CWBVH(c).Build(verts), tb.BVH8_CWBVH(c).Build(verts2)]
    inst = grid_instances(4, 0.55, 2, n_blas=2)
    inst["mask"][::5] = 0x0001
    return blas, tb.TLAS(c).Build(inst, blas)


def camera_rays(w, h):
    return R.primary(R.camera((-3.0, 4.5, -5.0), (0.55, -0.25, 0.8), w, h, 1, 1))


def test_packet_equals_per_lane_and_oracle(ctx, ctx_no_packet, oracle_ties):
    blas, tlas = make_scene(ctx)
    blas0, tlas0 = make_scene(ctx_no_packet)
    cam = camera_rays(512, 256)                                   # 131 072 coherent rays: the packet kernel takes them
    cam["mask"][::3] = 0x00F0                                     # these rays skip the instances whose mask is 0x0001
    cam["inst"] = 0xDEAD
    rnd = R.random_rays(100_000, (-2, -2, -2), (8, 8, 8), seed=6)  # incoherent: the probe sends them to the per-lane kernel ...
    half = np.concatenate([cam[:65536], rnd[:65536]])             # (a batch whose probe pairs agree about half the time)
    for name, rays, force in (("camera", cam, False), ("random forced through the packet kernel", rnd, True), ("half and half", half, False),
                              ("camera, tmax 6", None, False)):
        if rays is None:
            rays = cam.copy(); rays["t"] = 6.0
        want = oracle_tlas(oracle_ties, tlas, blas, rays)
        if force:
            ctx.set_debug_flags(16)
        got = tlas.Intersect(rays.copy())
        occ = tlas.IsOccluded(rays.copy())
        if force:
            ctx.set_debug_flags(0)
        base = tlas0.Intersect(rays.copy())
        occ0 = tlas0.IsOccluded(rays.copy())
        assert np.array_equal(got.view(np.uint8), base.view(np.uint8)), name
        assert np.array_equal(occ, occ0), name
        c = check(got, want)
        assert c["hits"] > 2000, (name, c)
        miss = got["t"] >= rays["t"]
        assert np.all(got["inst"][miss & (rays["t"] >= 1e30)] == 0xDEAD)          # a miss leaves the record untouched


def test_the_probe_decides(ctx):
    blas, tlas = make_scene(ctx)
    cam = camera_rays(512, 256)
    tlas.Intersect(cam.copy())
    agree, pairs, verdict = ctx.last_probe()
    assert pairs > 0 and verdict == 2                              # coherent: traced by the packet kernel
    rnd = R.random_rays(131072, (-2, -2, -2), (8, 8, 8), seed=6)
    tlas.Intersect(rnd.copy())
    assert ctx.last_probe()[2] == 1                                # incoherent: left to the per-lane kernel
    tlas.Intersect(cam[:4096].copy())
    assert ctx.last_probe()[2] == 0                                # small batch: no probe, one kernel
