"""Host half of the 8-wide copies (tinybvh_amd/csrc/host_builder.cpp: bvh_gpu_to_bvh2, bvh4_gpu_to_bvh2; no GPU needed): an uploaded BVH_GPU blob or BVH4_GPU
stream is turned into a BVH2 in the reference's BVHNode layout (tiny_bvh.h:857-866) with leaves of at most 3 entries before the device converter collapses
it 8-wide.  Checked on the blobs the REAL tiny_bvh.h encoded (tests/golden: BVH_GPU / BVH4_GPU ::Build and ::BuildHQ): the result is a tree the oracle's
BVH::Intersect restatement can walk, it finds the reference's own hit records, no leaf exceeds 3 entries, every triangle record of a BVH4 stream is
carried over exactly once and bit for bit."""
import ctypes as C
import os

import numpy as np
import pytest

import tinybvh_amd as tb
from tinybvh_amd import _capi
from oracle_lib import compare_hits

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
NAMES = ["soup_2k", "atrium_6k", "suzanne_decimated"]


def decode(layout, blob, idx=None, verts=None, max_leaf=3, n_idx=None):
    lib = _capi.lib
    blob = np.ascontiguousarray(blob)
    n_blob = blob.shape[0]
    nn, nr = C.c_uint64(), C.c_uint64()
    args = [layout, C.c_void_p(blob.ctypes.data), n_blob,
            C.c_void_p(idx.ctypes.data) if idx is not None else None, (n_idx or 0) if idx is None else idx.shape[0],
            C.c_void_p(verts.ctypes.data) if verts is not None else None, 0 if verts is None else verts.shape[0] // 3, max_leaf]
    tb.check(lib.tbvh_debug_wide_copy_bvh2(*args, None, 0, C.byref(nn), None, 0, C.byref(nr)), "tbvh_debug_wide_copy_bvh2")
    nodes = np.zeros((nn.value, 8), np.float32)
    recs = np.zeros((max(nr.value, 1), 3, 4), np.float32)
    tb.check(lib.tbvh_debug_wide_copy_bvh2(*args, C.c_void_p(nodes.ctypes.data), nodes.shape[0], C.byref(nn), C.c_void_p(recs.ctypes.data), recs.shape[0], C.byref(nr)), "tbvh_debug_wide_copy_bvh2")
    return nodes, recs[: nr.value]


def leaf_sizes(nodes):
    u = nodes.view(np.uint32)
    reach, stack, sizes = set(), [0], []
    while stack:
        i = stack.pop()
        assert i not in reach
        reach.add(i)
        if u[i, 7]:
            sizes.append(int(u[i, 7]))
        else:
            stack += [int(u[i, 3]), int(u[i, 3]) + 1]
    return sizes


@pytest.mark.parametrize("name", NAMES)
def test_bvh_gpu_blob_to_bvh2(oracle_ref, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    verts, rays = g["verts"], g["rays"]
    want = rays.copy()
    want.view(np.uint32).reshape(-1, 16)[:, 12:16] = g["hits"]
    for k in (0, 1):
        blob, idx = g[f"bvhgpu_nodes_{k}"], np.ascontiguousarray(g[f"bvhgpu_idx_{k}"].reshape(-1))
        # (a) from primIdx + vertices, (b) RECORD MODE — what the library runs: the triangles as the gathered records it keeps on the device
        #     {v0|prim, e1, e2} per primIdx entry (slack entries of an SBVH array: zeros)
        p = np.minimum(idx, verts.shape[0] // 3 - 1)
        tv = verts.reshape(-1, 3, 4)
        recs = np.zeros((idx.shape[0], 3, 4), np.float32)
        recs[:, 0, :3] = tv[p, 0, :3]; recs[:, 0, 3] = p.astype(np.uint32).view(np.float32)
        recs[:, 1, :3] = tv[p, 1, :3] - tv[p, 0, :3]; recs[:, 2, :3] = tv[p, 2, :3] - tv[p, 0, :3]
        recs[idx >= verts.shape[0] // 3] = 0
        for nodes in (decode(tb.LAYOUT_BVH_GPU, blob, idx, verts)[0], decode(tb.LAYOUT_BVH_GPU, blob, None, np.ascontiguousarray(recs.reshape(-1, 4)), n_idx=idx.shape[0])[0]):
            sizes = leaf_sizes(nodes)
            assert max(sizes) <= 3 and sum(sizes) >= verts.shape[0] // 3
            got = oracle_ref.bvh2_intersect(nodes, idx, verts, rays.copy())
            c = compare_hits(got, want)
            assert c["hitmiss"] == 0 and c["prim_real"] == 0 and c["t_bad"] == 0 and c["uv_bad"] == 0 and c["tie"] <= 2, (name, k, c)
            assert c["bit_identical"] == c["same_prim"], (name, k, c)


def stream_records(blocks):
    """every inline triangle record of a BVH4_GPU stream, by walking it (tiny_bvh.h:1248-1266)"""
    blocks = blocks.view(np.float32)
    u = blocks.view(np.uint32)
    out, stack = [], [0]
    while stack:
        o = stack.pop()
        for info in u[o + 3]:
            info = int(info)
            if not info:
                continue
            if info & 0x80000000:
                cnt, rel = (info >> 16) & 0x7fff, info & 0xffff
                for j in range(cnt):
                    out.append(blocks[o + rel + 3 * j: o + rel + 3 * j + 3].copy())
            else:
                stack.append(info)
    return np.array(out, np.float32).reshape(-1, 3, 4)


@pytest.mark.parametrize("name", NAMES)
def test_bvh4_stream_to_bvh2(oracle_ref, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    rays = g["rays"]
    want = rays.copy()
    want.view(np.uint32).reshape(-1, 16)[:, 12:16] = g["hits"]
    for k in (0, 1):
        blocks = np.ascontiguousarray(g[f"bvh4_{k}"])
        nodes, recs = decode(tb.LAYOUT_BVH4_GPU, blocks)
        assert max(leaf_sizes(nodes)) <= 3
        # the stream's records, each exactly once, bit for bit
        theirs = stream_records(blocks)
        key = lambda a: sorted(map(bytes, a.reshape(a.shape[0], -1).view(np.uint8)))
        assert recs.shape == theirs.shape and key(recs) == key(theirs)
        # walk the BVH2 with the oracle: leaf entry i -> record i (vertices rebuilt as v0, v0 + e1, v0 + e2: t to 1e-5, the prim exact through the record's own word)
        verts = np.zeros((recs.shape[0] * 3, 4), np.float32)
        verts[0::3, :3] = recs[:, 0, :3]
        verts[1::3, :3] = recs[:, 0, :3] + recs[:, 1, :3]
        verts[2::3, :3] = recs[:, 0, :3] + recs[:, 2, :3]
        idx = np.arange(recs.shape[0], dtype=np.uint32)
        got = oracle_ref.bvh2_intersect(nodes, idx, verts, rays.copy())
        hit = got["t"] < 1e30
        got["prim"][hit] = recs[:, 0, 3].view(np.uint32)[got["prim"][hit]]
        c = compare_hits(got, want, rtol=1e-5)
        assert c["hitmiss"] <= 1 and c["prim_real"] == 0 and c["t_bad"] == 0, (name, k, c)


@pytest.mark.parametrize("name", NAMES)
def test_cwbvh_blob_to_bvh2(oracle_ref, name):
    """... and of the 4-wide copy a TLAS enters a BVH8_CWBVH BLAS through (host_builder.cpp: cwbvh_to_bvh2): the reference's own BVH8_CWBVH::ConvertFrom blobs
    (Build and BuildHQ) become a BVH2 the oracle can walk, every triangle record carried over once, bit for bit, as {v0|prim, e1, e2}."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    rays = g["rays"]
    want = rays.copy()
    want.view(np.uint32).reshape(-1, 16)[:, 12:16] = g["hits"]
    for k in (0, 1):
        cn = np.ascontiguousarray(g[f"cwbvh_nodes_{k}"]).view(np.float32).reshape(-1, 4)
        ct = np.ascontiguousarray(g[f"cwbvh_tris_{k}"]).view(np.float32).reshape(-1, 4)
        ct = ct[: ct.shape[0] // 3 * 3]
        nodes, recs = decode(tb.LAYOUT_CWBVH, cn, None, ct)
        assert max(leaf_sizes(nodes)) <= 3
        # the records the blob's leaves reference, each exactly once (an SBVH's triangle array has slack behind them): {e2, e1, v0|prim} -> {v0|prim, e1, e2}
        used = recs.reshape(-1, 3, 4)
        theirs = ct.reshape(-1, 3, 4)[:, ::-1, :]
        keys = set(map(bytes, theirs.reshape(theirs.shape[0], -1).view(np.uint8)))
        assert all(bytes(r) in keys for r in used.reshape(used.shape[0], -1).view(np.uint8))
        verts = np.zeros((used.shape[0] * 3, 4), np.float32)
        verts[0::3, :3] = used[:, 0, :3]
        verts[1::3, :3] = used[:, 0, :3] + used[:, 1, :3]
        verts[2::3, :3] = used[:, 0, :3] + used[:, 2, :3]
        idx = np.arange(used.shape[0], dtype=np.uint32)
        got = oracle_ref.bvh2_intersect(nodes, idx, verts, rays.copy())
        hit = got["t"] < 1e30
        got["prim"][hit] = used[:, 0, 3].view(np.uint32)[got["prim"][hit]]
        c = compare_hits(got, want, rtol=1e-5)
        assert c["hitmiss"] <= 1 and c["prim_real"] == 0 and c["t_bad"] == 0, (name, k, c)


def test_a_single_leaf_has_no_copy():
    verts = np.array([[0, 0, 0, 0], [1, 0, 0, 0], [0, 1, 0, 0]], np.float32)
    h = tb.HostBVH(verts, tb.LAYOUT_BVH_GPU)
    with pytest.raises(tb.TbvhError):
        decode(tb.LAYOUT_BVH_GPU, h.blob(0, np.uint32, 16), np.ascontiguousarray(h.blob(1, np.uint32, 1).reshape(-1)), verts)
