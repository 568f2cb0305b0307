"""Device-side per-frame TLAS rebuild (tbvh_rebuild_tlas_device): the instance
update must produce the same records as the host's BLASInstance::Update restatement, the LBVH must
be a valid BVH_GPU tree over all instances, and queries through it must return the reference's hit
records — checked against the restated BVH::IntersectTLAS both on the host-built (SAH) TLAS, which
proves the TLAS shape does not matter, and on the device-built tree itself (converted to the
32-byte node form the oracle walks)."""
import numpy as np
import pytest

import tinybvh_amd as tb
from tinybvh_amd import rays as R
from tinybvh_amd import scenes
from oracle_lib import tlas_intersect
from test_tlas import check, grid_instances, oracle_tlas


def al_to_wald(nodes64, n_inst):
    """BVH_GPU (Aila-Laine) TLAS nodes -> 32-byte BVHNode array with adjacent children (the form
    BVH::IntersectTLAS walks, tiny_bvh.h:3306-3380).  Also returns per-node boxes for validation."""
    f = nodes64.view(np.float32).reshape(-1, 16)
    u = nodes64.reshape(-1, 16)
    n_nodes = u.shape[0]
    out = np.zeros((n_nodes + 1, 8), np.uint32)          # aabbMin, leftFirst, aabbMax, triCount (node 1 stays unused)
    of = out.view(np.float32)
    if n_inst == 1:
        assert u[0, 11] == 1
        return out, None
    # (old index, new index, box) work list; the root has no box of its own: union of its children
    boxes = {}
    new_of = {0: 0}
    next_free = 2                                        # tinybvh keeps node 1 unused
    stack = [0]
    while stack:
        o = stack.pop()
        nn = new_of[o]
        if u[o, 11]:                                     # leaf: triCount, firstTri
            out[nn, 3] = u[o, 15]; out[nn, 7] = u[o, 11]
            continue
        l, r = int(u[o, 3]), int(u[o, 7])
        lmn, lmx, rmn, rmx = f[o, 0:3], f[o, 4:7], f[o, 8:11], f[o, 12:15]
        boxes[l] = (lmn, lmx); boxes[r] = (rmn, rmx)
        new_of[l], new_of[r] = next_free, next_free + 1
        out[nn, 3] = next_free; out[nn, 7] = 0
        next_free += 2
        for c in (l, r):
            of[new_of[c], 0:3] = boxes[c][0]; of[new_of[c], 4:7] = boxes[c][1]
        if o == 0:
            of[0, 0:3] = np.minimum(lmn, rmn); of[0, 4:7] = np.maximum(lmx, rmx)
        stack += [l, r]
    return out[:next_free], boxes


def validate_tree(nodes64, idx, inst):
    n = inst.shape[0]
    u = nodes64.reshape(-1, 16); f = nodes64.view(np.float32).reshape(-1, 16)
    assert u.shape[0] == 2 * n - 1
    assert sorted(idx.tolist()) == list(range(n))        # every instance exactly once
    if n == 1:
        assert u[0, 11] == 1 and u[0, 15] == 0
        return
    seen_leaves = []

    def box_of(o):
        if u[o, 11]:
            assert u[o, 11] == 1
            k = int(u[o, 15]); seen_leaves.append(k)
            i = int(idx[k])
            return inst["aabbMin"][i], inst["aabbMax"][i]
        l, r = int(u[o, 3]), int(u[o, 7])
        lb, rb = box_of(l), box_of(r)
        # the stored child boxes are exactly the children's boxes
        assert np.array_equal(f[o, 0:3], lb[0]) and np.array_equal(f[o, 4:7], lb[1]), o
        assert np.array_equal(f[o, 8:11], rb[0]) and np.array_equal(f[o, 12:15], rb[1]), o
        return np.minimum(lb[0], rb[0]), np.maximum(lb[1], rb[1])

    import sys
    sys.setrecursionlimit(10000)
    box_of(0)
    assert sorted(seen_leaves) == list(range(n))


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [tb.LAYOUT_CWBVH, tb.LAYOUT_BVH4_GPU])
def test_device_rebuild_parity(ctx, oracle, layout):
    verts = scenes.blob(6000, seed=3)
    verts2 = scenes.soup(2000, seed=9, extent=1.6, size=0.25); verts2[:, :3] -= 0.8
    blas = [tb.LAYOUT_CLASSES[layout](ctx).Build(verts), tb.LAYOUT_CLASSES[layout](ctx).Build(verts2)]
    inst = grid_instances(4, 0.55, 2, n_blas=2)
    inst["mask"][::5] = 0x0001
    tlas = tb.TLAS(ctx).Build(inst, blas)                # host SAH build: the comparison tree
    rays = R.random_rays(30_000, (-2, -2, -2), (8, 8, 8), seed=6)
    rays["mask"][::3] = 0x00F0
    want_host_tree = oracle_tlas(oracle, tlas, blas, rays)
    host_records = tlas.instances.copy()

    # frame 0: rebuild on the device from the transforms already in the records
    tlas.RebuildOnDevice()
    nodes, idx, dev_inst = tlas.Download()
    for fld in ("transform", "invTransform", "aabbMin", "aabbMax", "blasIdx", "mask"):
        assert np.array_equal(dev_inst[fld].view(np.uint32), host_records[fld].view(np.uint32)), fld   # bit-identical records
    validate_tree(nodes, idx, dev_inst)
    got = tlas.Intersect(rays.copy())
    check(got, want_host_tree)                           # TLAS shape does not change hit records
    wald, _ = al_to_wald(nodes, dev_inst.shape[0])
    bl = [(b.host.bvh2_nodes(), b.host.bvh2_prim_idx(), b.host.verts) for b in blas]
    want_dev_tree = tlas_intersect(oracle, wald, idx, dev_inst, bl, rays)
    c = check(got, want_dev_tree)
    assert c["hits"] > 3000

    # frame 1: new transforms from the host, rebuilt on the device; the host path gives the reference answer
    inst2 = inst.copy()
    inst2["transform"][:, 3] += 0.37; inst2["transform"][:, 11] -= 0.21; inst2["transform"][::7, 0] *= 1.3
    tlas.RebuildOnDevice(inst2["transform"])
    got2 = tlas.Intersect(rays.copy())
    occ2 = tlas.IsOccluded(rays.copy())
    ref = tb.TLAS(ctx).Build(inst2.copy(), blas)
    want2 = oracle_tlas(oracle, ref, blas, rays)
    check(got2, want2)
    assert not np.array_equal(want2["t"], want_host_tree["t"])
    assert int((occ2.astype(bool) != (want2["t"] < 1e30)).sum()) <= 2
    _, _, dev_inst2 = tlas.Download()
    for fld in ("transform", "invTransform", "aabbMin", "aabbMax"):
        assert np.array_equal(dev_inst2[fld].view(np.uint32), ref.instances[fld].view(np.uint32)), fld

    # frame 2: transforms that already live on the device
    d_t = ctx.malloc(inst.shape[0] * 64)
    ctx.to_device(d_t, np.ascontiguousarray(inst["transform"]))
    tlas.RebuildOnDevice(d_t, on_device=True)
    check(tlas.Intersect(rays.copy()), want_host_tree)
    ctx.free(d_t)


@pytest.mark.gpu
@pytest.mark.parametrize("n_side,dup", [(1, False), (2, False), (3, True), (10, False)])
def test_device_rebuild_shapes(ctx, oracle, n_side, dup):
    """1 instance (leaf root), 8, 27 with coincident instances (equal Morton codes), 1000."""
    verts = scenes.blob(1500, seed=5)
    blas = [tb.BVH8_CWBVH(ctx).Build(verts)]
    inst = grid_instances(n_side, 0.5, 4)
    if dup:
        inst["transform"][1::2] = inst["transform"][0::2][: inst["transform"][1::2].shape[0]]   # pairs of identical instances
    tlas = tb.TLAS(ctx).Build(inst, blas)
    span = 2.0 * n_side
    rays = R.random_rays(20_000, (-1.5, -1.5, -1.5), (span, span, span), seed=8)
    want = oracle_tlas(oracle, tlas, blas, rays)
    tlas.RebuildOnDevice()
    nodes, idx, dev_inst = tlas.Download()
    validate_tree(nodes, idx, dev_inst)
    got = tlas.Intersect(rays.copy())
    if dup:
        # coincident instances: which of the two identical copies reports the hit is a tie by construction
        hit = want["t"] < 1e30
        assert np.array_equal(got["t"][hit].view(np.uint32), want["t"][hit].view(np.uint32))
        assert np.array_equal(got["prim"][hit], want["prim"][hit])
        assert np.array_equal(got["t"] < 1e30, hit)
    else:
        check(got, want)


@pytest.mark.gpu
def test_device_rebuild_errors(ctx):
    verts = scenes.soup(500, seed=1)
    b = tb.BVH8_CWBVH(ctx).Build(verts)
    with pytest.raises(tb.TbvhError):
        tb.check(tb.lib.tbvh_rebuild_tlas_device(b._h, None, 0, None, 0), "rebuild on a BLAS")
    inst = grid_instances(2, 0.5, 1)
    tlas = tb.TLAS(ctx).Build(inst, [b])
    with pytest.raises(tb.TbvhError):
        tb.check(tb.lib.tbvh_rebuild_tlas_device(tlas._h, None, 0, None, 0), "first call without BLAS bounds")
    bounds = np.zeros((2, 6), np.float32)
    import ctypes as C
    with pytest.raises(tb.TbvhError):
        tb.check(tb.lib.tbvh_rebuild_tlas_device(tlas._h, None, 0, C.c_void_p(bounds.ctypes.data), 2), "wrong BLAS count")


@pytest.mark.gpu
def test_rebuild_after_the_tlas_grew(ctx, oracle):
    """A TLAS updated to more instances than it was created with, then rebuilt on the device from host transforms:
    the staging buffer of the transforms follows the new instance count (it used to keep its first size)."""
    verts = scenes.blob(1500, seed=5)
    blas = [tb.BVH8_CWBVH(ctx).Build(verts)]
    small = grid_instances(3, 0.5, 4)
    tlas = tb.TLAS(ctx).Build(small, blas)
    tlas.RebuildOnDevice(np.ascontiguousarray(small["transform"]))            # stages 27 transforms
    big = grid_instances(6, 0.5, 4)
    tlas.Build(big, blas)                                                     # update path: 216 instances
    tlas._bounds_sent = True
    tlas.RebuildOnDevice(np.ascontiguousarray(big["transform"]))              # stages 216
    rays = R.random_rays(20_000, (-2, -2, -2), (12, 12, 12), seed=8)
    ref = tb.TLAS(ctx).Build(big.copy(), blas)
    check(tlas.Intersect(rays.copy()), oracle_tlas(oracle, ref, blas, rays))


@pytest.mark.gpu
def test_tlas_blobs_are_validated(ctx):
    verts = scenes.blob(1500, seed=5)
    blas = [tb.BVH8_CWBVH(ctx).Build(verts)]
    inst = grid_instances(2, 0.5, 4)
    tlas = tb.TLAS(ctx).Build(inst, blas)
    nodes = tlas.host.blob(0, np.uint32, 16).copy(); idx = tlas.host.blob(1, np.uint32, 1).copy()
    bad = inst.copy(); bad["blasIdx"][3] = 7
    with pytest.raises(tb.TbvhError):
        tlas.Update(nodes, idx, bad)                                          # blasIdx beyond the BLAS list
    bad_idx = idx.copy(); bad_idx[0] = 1000
    with pytest.raises(tb.TbvhError):
        tlas.Update(nodes, bad_idx, inst)                                     # primIdx beyond the instances
    bad_nodes = nodes.copy()
    interior = np.flatnonzero(bad_nodes[:, 11] == 0)
    bad_nodes[interior[0], 3] = 0x7fffffff
    with pytest.raises(tb.TbvhError):
        tlas.Update(bad_nodes, idx, inst)                                     # child beyond the node array
    with pytest.raises(tb.TbvhError):
        tb.TLAS(ctx).Upload(nodes, idx, bad, blas)
    tlas.Update(nodes, idx, inst)                                             # the good blobs still go through


@pytest.mark.gpu
def test_blas_freed_or_remapped_under_a_live_tlas(ctx, oracle):
    """tbvh_free_scene on a BLAS that a TLAS still uses is deferred; opacity maps set on a BLAS after the TLAS upload
    reach the TLAS's descriptor."""
    verts = scenes.blob(1500, seed=5)
    blas = tb.BVH8_CWBVH(ctx).Build(verts)
    inst = grid_instances(2, 0.5, 4)
    tlas = tb.TLAS(ctx).Build(inst, [blas])
    rays = R.random_rays(20_000, (-2, -2, -2), (5, 5, 5), seed=8)
    want = oracle_tlas(oracle, tlas, [blas], rays)
    # all-clear maps: nothing can be hit any more
    n_tris = verts.shape[0] // 3
    blas.SetOpacityMicroMaps(np.zeros((n_tris, 1), np.uint32), 4)
    got = tlas.Intersect(rays.copy())
    assert np.all(got["t"] == rays["t"])
    blas.SetOpacityMicroMaps(None, 0)
    check(tlas.Intersect(rays.copy()), want)
    import ctypes as C
    tb.lib.tbvh_free_scene(blas._h)                                           # deferred: the TLAS still points at it
    for _ in range(4):                                                        # allocations in between would reuse freed memory
        tmp = ctx.malloc(verts.nbytes * 4); ctx.to_device(tmp, np.zeros(verts.nbytes, np.uint8)); ctx.free(tmp)
    check(tlas.Intersect(rays.copy()), want)
    blas._h = C.c_void_p()                                                    # the handle is gone for the wrapper
    tlas.free()


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [tb.LAYOUT_BVH4_GPU, tb.LAYOUT_CWBVH])
def test_wide_tlas_built_level_by_level_over_the_chip(ctx, oracle, layout):
    """TLASes of tens of thousands of instances collapse their wide tree with one launch per level over the whole chip instead of one workgroup
    (kernels_tlaswide.hip: k_tlas_wide_level; TBVH_TLAS_PAR_MIN = the instance count from which, 16384 by default).  Forced here on 1728 instances
    (both wide formats), and run at its real size on 35 937: the hit records are the one-workgroup build's, and IntersectTLAS's."""
    import os
    from test_tlas import grid_instances, oracle_tlas, check
    mesh = scenes.blob(3000, seed=4)
    mesh[:, :3] -= 0.5 * (mesh[:, :3].min(0) + mesh[:, :3].max(0))
    mesh[:, :3] *= np.float32(1.6 / float((mesh[:, :3].max(0) - mesh[:, :3].min(0)).max()))
    blas = tb.LAYOUT_CLASSES[layout](ctx).Build(mesh)
    old = os.environ.get("TBVH_TLAS_PAR_MIN")
    try:
        for side, force in ((12, True), (33, False)):
            inst = grid_instances(side, 0.5, 7)
            rays = R.random_rays(200_000, (-2, -2, -2), (2.0 * side, 2.0 * side, 2.0 * side), seed=5)
            os.environ["TBVH_TLAS_PAR_MIN"] = "1000000000"
            one = tb.TLAS(ctx).Build(inst.copy(), [blas])
            one.RebuildOnDevice()
            a = one.Intersect(rays.copy()); oa = one.IsOccluded(rays.copy())
            if force:
                os.environ["TBVH_TLAS_PAR_MIN"] = "1"
            else:
                os.environ.pop("TBVH_TLAS_PAR_MIN", None)
            par = tb.TLAS(ctx).Build(inst.copy(), [blas])
            par.RebuildOnDevice()
            b = par.Intersect(rays.copy()); ob = par.IsOccluded(rays.copy())
            assert (a["t"] < 1e30).sum() > 1000
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8)) and np.array_equal(oa, ob), (side, layout)
            idx = np.arange(0, rays.shape[0], 8)
            check(b[idx], oracle_tlas(oracle, par, [blas], rays[idx]))
            one.free(); par.free()
    finally:
        if old is None:
            os.environ.pop("TBVH_TLAS_PAR_MIN", None)
        else:
            os.environ["TBVH_TLAS_PAR_MIN"] = old
    blas.free()
