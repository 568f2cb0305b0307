"""Blob cache: files of BVH8_CWBVH::Save / Load (tiny_bvh.h:5786-5820) read and written through
tbvh_cwbvh_file_read / tbvh_cwbvh_file_write.  With the real reference behind oracle/_ref: a file it saves reads here
with byte-identical blobs, a file written here loads in BVH8_CWBVH::Load and traces there to the same hit records, and
the object-layout constants the file format depends on are those of the real header."""
import os

import numpy as np
import pytest

import tinybvh_amd as tb
from tinybvh_amd import rays as R
from tinybvh_amd import scenes
from oracle_lib import compare_hits


@pytest.fixture(scope="module")
def verts():
    return scenes.blob(12_000, seed=5)


def write_file(path, host, bounds=None):
    n, t = host.blob(0, np.uint32, 4), host.blob(1, np.uint32, 4)
    b = None if bounds is None else np.ascontiguousarray(bounds, np.float32)
    tb.check(tb.lib.tbvh_cwbvh_file_write(os.fsencode(path), n.ctypes.data, n.shape[0], t.ctypes.data, t.shape[0], host.n_tris,
                                          None if b is None else b.ctypes.data), "tbvh_cwbvh_file_write")


def test_object_layout_constants_are_the_real_headers(reference):
    """capi.hip hard-codes the offsets of tinybvh 1.6.7 / LP64; the real header must agree."""
    o = reference.cwbvh_object_layout()
    assert o == {"size": 560, "layout": 32, "triCount": 44, "idxCount": 48, "aabbMin": 72, "aabbMax": 84, "opmapN": 96, "opmap": 104,
                 "bvh8Data": 112, "bvh8Tris": 120, "allocatedBlocks": 128, "usedBlocks": 132, "bvh8.idxCount": 184, "ownBVH8": 552,
                 "c_trav": 52, "hqbvhbins": 64}, o
    assert reference.lib.ref_version() == b"tinybvh 1.6.7"


def test_write_then_read_round_trip(tmp_path, verts):
    host = tb.HostBVH(verts, tb.LAYOUT_CWBVH)
    path = str(tmp_path / "scene.cwbvh")
    write_file(path, host)
    n, t = host.blob(0, np.uint32, 4), host.blob(1, np.uint32, 4)
    assert os.path.getsize(path) == 8 + 560 + n.nbytes + (t.shape[0] // 3) * 64      # the reference's file length
    back = tb.HostBVH.from_cwbvh_file(path, expected_tris=host.n_tris)
    assert back.n_tris == host.n_tris and back.layout == tb.LAYOUT_CWBVH
    assert np.array_equal(back.blob(0, np.uint32, 4), n) and np.array_equal(back.blob(1, np.uint32, 4), t)
    assert tb.HostBVH.from_cwbvh_file(path).n_tris == host.n_tris                     # 0 = any triangle count


def test_read_refuses_what_load_refuses(tmp_path, verts):
    host = tb.HostBVH(verts, tb.LAYOUT_CWBVH)
    path = str(tmp_path / "scene.cwbvh")
    write_file(path, host)
    with pytest.raises(tb.TbvhError, match="expected"):
        tb.HostBVH.from_cwbvh_file(path, expected_tris=host.n_tris + 1)                # Load: fileTriCount != expectedTris
    raw = bytearray(open(path, "rb").read())
    for name, patch in (("version", (0, 8)), ("layout", (3, 6))):                      # sub-version byte, layout byte
        bad = bytearray(raw); bad[patch[0]] = patch[1]
        p2 = str(tmp_path / f"bad_{name}.cwbvh"); open(p2, "wb").write(bad)
        with pytest.raises(tb.TbvhError, match="header"):
            tb.HostBVH.from_cwbvh_file(p2)
    p3 = str(tmp_path / "short.cwbvh"); open(p3, "wb").write(raw[:-16])
    with pytest.raises(tb.TbvhError, match="length"):
        tb.HostBVH.from_cwbvh_file(p3)
    p4 = str(tmp_path / "tiny.cwbvh"); open(p4, "wb").write(raw[:100])
    with pytest.raises(tb.TbvhError, match="too short"):
        tb.HostBVH.from_cwbvh_file(p4)
    bad = bytearray(raw); bad[8 + 560 + 16 + 0] ^= 0xFF; bad[8 + 560 + 16 + 1] ^= 0xFF   # root node's child base index
    p5 = str(tmp_path / "corrupt.cwbvh"); open(p5, "wb").write(bad)
    with pytest.raises(tb.TbvhError):
        tb.HostBVH.from_cwbvh_file(p5)
    with pytest.raises(tb.TbvhError, match="cannot open"):
        tb.HostBVH.from_cwbvh_file(str(tmp_path / "missing.cwbvh"))


@pytest.mark.parametrize("hq", [False, True])
def test_reads_files_saved_by_the_reference(tmp_path, reference, verts, hq):
    rs = reference.build(verts, hq=hq)
    path = str(tmp_path / "ref.cwbvh")
    rs.cwbvh_save(path)
    back = tb.HostBVH.from_cwbvh_file(path, expected_tris=verts.shape[0] // 3)
    assert np.array_equal(back.blob(0, np.uint32, 4), rs.blob(10, 0, np.uint32, 4))
    assert np.array_equal(back.blob(1, np.uint32, 4), rs.blob(10, 1, np.uint32, 4))


def test_reference_loads_and_traces_files_written_here(tmp_path, reference, oracle, verts):
    host = tb.HostBVH(verts, tb.LAYOUT_CWBVH)
    path = str(tmp_path / "mine.cwbvh")
    write_file(path, host, bounds=np.concatenate([verts[:, :3].min(0), verts[:, :3].max(0)]))
    lo, hi = verts[:, :3].min(0), verts[:, :3].max(0)
    rays = R.random_rays(20_000, lo - 0.2, hi + 0.2, seed=17)
    got = reference.cwbvh_load_and_intersect(path, host.n_tris, rays)
    assert got is not None, "BVH8_CWBVH::Load refused a file written by tbvh_cwbvh_file_write"
    want = oracle.cwbvh_intersect(host.blob(0, np.uint32, 4), host.blob(1, np.uint32, 4), rays.copy())
    c = compare_hits(got, want)
    assert c["hits"] > 2000 and c["bit_identical"] == c["hits"] and c["hitmiss"] == 0 and c["prim_mismatch"] == 0, c
    assert reference.cwbvh_load_and_intersect(path, host.n_tris + 1, rays) is None     # Load's own triangle-count check
    write_file(path, host)                                                             # bounds from the root node
    assert reference.cwbvh_load_and_intersect(path, host.n_tris, rays) is not None


@pytest.mark.gpu
def test_save_load_through_the_device(ctx, oracle, tmp_path, verts):
    sc = tb.BVH8_CWBVH(ctx).Build(verts)
    lo, hi = verts[:, :3].min(0), verts[:, :3].max(0)
    rays = R.random_rays(30_000, lo - 0.2, hi + 0.2, seed=23)
    want = sc.Intersect(rays.copy())
    p1 = str(tmp_path / "a.cwbvh"); sc.Save(p1)
    loaded = tb.BVH8_CWBVH(ctx).Load(p1, expected_tris=verts.shape[0] // 3)
    assert np.array_equal(loaded.Intersect(rays.copy()).view(np.uint8), want.view(np.uint8))
    # a scene that only lives on the device (built there): Save reads the blobs back
    dev = tb.BVH8_CWBVH(ctx).BuildOnDevice(verts)
    p2 = str(tmp_path / "b.cwbvh"); dev.Save(p2, n_tris=verts.shape[0] // 3)
    again = tb.BVH8_CWBVH(ctx).Load(p2, expected_tris=verts.shape[0] // 3)
    host = tb.HostBVH(verts, tb.LAYOUT_BVH2_WALD)
    ref = oracle.bvh2_intersect(host.bvh2_nodes(), host.bvh2_prim_idx(), verts, rays.copy())
    c = compare_hits(again.Intersect(rays.copy()), ref)
    assert c["hitmiss"] == 0 and c["prim_real"] == 0 and c["t_bad"] == 0 and c["bit_identical"] == c["same_prim"], c
    for s in (sc, loaded, dev, again):
        s.free()
