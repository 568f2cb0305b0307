"""A tree deeper than the traversal stack: the kernels must report it (status bit -> TBVH_E_FORMAT "traversal stack overflow"), never read or
write beyond their rows of the spill area.  The reference has no such path — BVH::Intersect walks a fixed stack[64] and simply overruns it
(tiny_bvh.h:3226) — so the expected records here are analytic, not the oracle's.  The tree is a caterpillar (chain_bvh2 below): collapsed 8-wide it is
~ D / 7 levels deep, and a ray travelling in -x leaves one pending group of interior children per level on the stack.  TBVH_SPILL_ENTRIES (read when a context is made) shrinks the spill area so that a few hundred levels
are enough.  Variant 92 = the wave-packet kernel (kernels_cwbvh_packet.hip) whatever the batch; variant 0 = the per-lane kernels."""
import os

import numpy as np
import pytest

import tinybvh_amd as tb

pytestmark = pytest.mark.gpu


CLUMP = 4     # triangles per clump: a BVH2 subtree of two 2-triangle leaves, so it stays an INTERIOR child of the wide node


def chain_bvh2(depth):
    """(nodes32 [N, 8] float32 in BVH::BVHNode layout (tiny_bvh.h:857-866), primIdx, verts): a caterpillar — BVH2 node k = { clump k, node k + 1 }, the
    last one { clump depth - 1, clump depth }; clump k = 4 large triangles perpendicular to x at x = k + 0.1 j.  Collapsed 8-wide (greedy: open the child
    with the largest area = the rest of the chain) a wide node holds 7 clumps and the rest of the chain, all INTERIOR children: a ray travelling in -x
    descends into the rest of the chain first and leaves the node's group of pending clumps on its stack — one entry per wide level."""
    n_clumps = depth + 1
    n_tris = CLUMP * n_clumps
    verts = np.zeros((3 * n_tris, 4), np.float32)
    x = (np.repeat(np.arange(n_clumps, dtype=np.float32), CLUMP) + np.tile(np.arange(CLUMP, dtype=np.float32) * np.float32(0.1), n_clumps)).astype(np.float32)
    verts[0::3, :3] = np.stack([x, np.full(n_tris, -1.0), np.full(n_tris, -1.0)], 1)
    verts[1::3, :3] = np.stack([x, np.full(n_tris, 3.0), np.full(n_tris, -1.0)], 1)
    verts[2::3, :3] = np.stack([x, np.full(n_tris, -1.0), np.full(n_tris, 3.0)], 1)
    rows = [[0.0] * 8, [0.0] * 8]                   # node 0 = root, node 1 unused (tiny_bvh.h:2277: children are allocated in pairs from index 2)
    words = [[0, 0], [0, 0]]                        # (leftFirst, triCount) per node

    def box(i, lo_x, hi_x):
        rows[i][0:3] = [lo_x, -1.0, -1.0]
        rows[i][4:7] = [hi_x, 3.0, 3.0]

    def pair():
        rows.extend([[0.0] * 8, [0.0] * 8]); words.extend([[0, 0], [0, 0]])
        return len(rows) - 2

    def clump(i, k):                                # node i becomes the 2-leaf subtree over clump k
        t0 = CLUMP * k
        box(i, float(x[t0]), float(x[t0 + CLUMP - 1]))
        c = pair()
        words[i] = [c, 0]
        for half in (0, 1):
            a = t0 + 2 * half
            box(c + half, float(x[a]), float(x[a + 1]))
            words[c + half] = [a, 2]
    me = 0
    for k in range(depth):
        box(me, float(x[CLUMP * k]), float(x[-1]))
        c = pair()
        words[me] = [c, 0]
        clump(c, k)
        if k == depth - 1:
            clump(c + 1, depth)
        me = c + 1
    nodes = np.array(rows, np.float32)
    u = nodes.view(np.uint32)
    w = np.array(words, np.uint32)
    u[:, 3] = w[:, 0]; u[:, 7] = w[:, 1]
    return nodes, np.arange(n_tris, dtype=np.uint32), verts


def rays_along_x(depth, n, sign):
    O = np.zeros((n, 3), np.float32)
    O[:, 0] = depth + 1.0 if sign < 0 else -1.0
    O[:, 1] = np.linspace(0.05, 0.6, n, dtype=np.float32)
    O[:, 2] = 0.2
    D = np.zeros((n, 3), np.float32); D[:, 0] = sign
    return tb.make_rays(O, D)


def expected(depth, sign):
    """(t, prim) of every ray: the first triangle met from either end"""
    if sign > 0:
        return np.float32(1.0), 0
    last = CLUMP * (depth + 1) - 1
    x_last = np.float32(depth) + np.float32(CLUMP - 1) * np.float32(0.1)
    return np.float32(depth + 1.0) - x_last, last


def check(got, depth, sign):
    t, prim = expected(depth, sign)
    assert np.all(got["prim"] == prim), (sign, got["prim"][:4], prim)
    assert np.allclose(got["t"], t, rtol=0, atol=2e-3), (sign, got["t"][:4], t)


@pytest.mark.parametrize("variant", [0, 92])
def test_moderately_deep_chain_is_traced_correctly(ctx, variant):
    depth = 700                                     # ~100 wide-tree levels: beyond the in-register / LDS part of every kernel's stack, within the spill area
    n2, pi, verts = chain_bvh2(depth)
    sc = tb.BVH8_CWBVH(ctx).ConvertFromBVH2(n2, pi, verts)
    sc.set_variant(variant)
    for sign in (-1.0, 1.0):
        check(sc.Intersect(rays_along_x(depth, 256, sign)), depth, sign)
        occ = sc.IsOccluded(rays_along_x(depth, 256, sign))
        assert occ.all()
    sc.free()


@pytest.mark.parametrize("variant", [0, 92])
def test_stack_overflow_is_an_error_not_a_wild_access(variant):
    depth = 2100                                    # ~300 wide-tree levels against a spill area of one 8-byte entry per lane
    n2, pi, verts = chain_bvh2(depth)
    old = os.environ.get("TBVH_SPILL_ENTRIES")
    os.environ["TBVH_SPILL_ENTRIES"] = "2"
    try:
        c = tb.Context(0)
    finally:
        if old is None:
            os.environ.pop("TBVH_SPILL_ENTRIES", None)
        else:
            os.environ["TBVH_SPILL_ENTRIES"] = old
    try:
        sc = tb.BVH8_CWBVH(c).ConvertFromBVH2(n2, pi, verts)
        sc.set_variant(variant)
        errors = 0
        for sign in (-1.0, 1.0):
            try:
                check(sc.Intersect(rays_along_x(depth, 256, sign)), depth, sign)                  # a direction that did not overflow is still right
            except tb.TbvhError as e:
                assert "stack overflow" in str(e), e
                errors += 1
        assert errors >= 1                                                                          # the interior-child-first direction must overflow
        # the context is usable afterwards: a shallow scene traces
        shallow = tb.BVH8_CWBVH(c).ConvertFromBVH2(*chain_bvh2(20))
        shallow.set_variant(variant)
        check(shallow.Intersect(rays_along_x(20, 64, -1.0)), 20, -1.0)
    finally:
        c.close()
