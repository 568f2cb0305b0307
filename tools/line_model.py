#!/usr/bin/env python
"""CPU model of the cache lines an incoherent batch touches in the BVH8_CWBVH incoherent flavor (hybrid node copy + 64-byte triangle
records), and of candidate layouts for the leaf level — round 4, review item 4 ("fewer cache lines per ray").  Runs here (no GPU):
the oracle's CWBVH mirror records every node visit and triangle test of a bounce-ray sample (oracle/tbvh_oracle.c: orc_cwbvh_trace),
this script prices them in 128-byte lines.

usage: tools/line_model.py [--scene bistro] [--side 192]
"""
import argparse, ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import tinybvh_amd as tb
from tinybvh_amd import rays as R, scenes
from oracle_lib import Oracle, _p


def bounce(verts, hits, rng):
    """k_gen_bounce on the host: uniform direction in the hemisphere about the hit triangle's normal."""
    ok = hits["t"] < 1e30
    h = hits[ok]
    I = h["O"] + h["t"][:, None] * h["D"]
    a, b, c = (verts[h["prim"] * 3 + k, :3] for k in range(3))
    N = np.cross(b - a, c - a); N /= np.maximum(np.linalg.norm(N, axis=1, keepdims=True), 1e-20)
    N[(N * h["D"]).sum(1) > 0] *= -1
    Rd = rng.random((h.shape[0], 3), dtype=np.float32) - 0.5; Rd /= np.linalg.norm(Rd, axis=1, keepdims=True)
    Rd[(N * Rd).sum(1) < 0] *= -1
    return tb.make_rays((I + 1e-3 * Rd).astype(np.float32), Rd.astype(np.float32))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="bistro"); ap.add_argument("--side", type=int, default=160)
    a = ap.parse_args()
    verts, label = scenes.get(a.scene)
    t0 = time.time()
    h = tb.HostBVH(verts, tb.LAYOUT_CWBVH)
    nodes, tris = h.blob(0, np.uint32, 4), h.blob(1, np.uint32, 4)
    nN = nodes.shape[0] // 5
    print(f"{label}: {nN} nodes, {tris.shape[0] // 3} triangle records, built in {time.time() - t0:.1f}s")
    orc = Oracle(1)
    cams = scenes.STREET_CAMERAS if a.scene.startswith(("bistro", "street")) else scenes.SPONZA_CAMERAS
    prim = R.primary(R.camera(*cams[0], a.side, a.side, 1, 1))
    rng = np.random.default_rng(5)
    h1 = orc.cwbvh_intersect(nodes, tris, prim)
    b1 = bounce(verts, h1, rng); h2 = orc.cwbvh_intersect(nodes, tris, b1)
    b2 = bounce(verts, h2, rng); h3 = orc.cwbvh_intersect(nodes, tris, b2)
    b3 = bounce(verts, h3, rng)
    batch = np.concatenate([b1[: b1.shape[0] // 3], b2[: b2.shape[0] // 3], b3[: b3.shape[0] // 3]])   # depths 1-3 in thirds, like bench.py
    L = orc.lib
    L.orc_cwbvh_trace.restype = C.c_uint64
    L.orc_cwbvh_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64]
    cap = batch.shape[0] * 400
    out = np.zeros(cap, np.uint32)
    r = batch.copy()
    L.orc_set_tie_rule(1)
    nw = L.orc_cwbvh_trace(_p(nodes), _p(tris), _p(r), r.shape[0], r.strides[0], _p(out), cap)
    assert nw < cap
    ev = out[:nw]
    nrays = batch.shape[0]
    isnode = (ev & 0x80000000) != 0
    issep = ev == 0xFFFFFFFF
    isnode &= ~issep
    istri = ~isnode & ~issep
    S = isnode.sum() / nrays; T = istri.sum() / nrays
    print(f"{nrays} bounce rays: S = {S:.2f} node visits, T = {T:.2f} triangle tests per ray")
    # ---- per node: leaf children (meta) and their box areas -------------------------------------------------------------------------
    nd = nodes.reshape(nN, 5, 4)
    meta = np.stack([(nd[:, 1, 2] >> (8 * i)) & 255 for i in range(4)] + [(nd[:, 1, 3] >> (8 * i)) & 255 for i in range(4)], 1)   # (nN, 8)
    imask = nd[:, 0, 3] >> 24
    is_inner = ((imask[:, None] >> np.arange(8)[None, :]) & 1).astype(bool)
    is_leaf = (~is_inner) & (meta != 0)
    leaf_off = meta & 31
    leaf_cnt = np.array([0, 1, 0, 2, 0, 0, 0, 3])[(meta >> 5) & 7] * is_leaf     # unary count bits 001 / 011 / 111
    ex = [((nd[:, 0, 3] >> (8 * k)) & 255).astype(np.int8).astype(np.float64) for k in range(3)]
    def q(word_row, word_col):
        return np.stack([(nd[:, word_row, word_col] >> (8 * i)) & 255 for i in range(4)] + [(nd[:, word_row, word_col + 1] >> (8 * i)) & 255 for i in range(4)], 1).astype(np.float64)
    lo = [q(2, 0), q(2, 2), q(3, 0)]; hi = [q(3, 2), q(4, 0), q(4, 2)]
    d = [np.maximum(hi[k] - lo[k], 0) * np.exp2(ex[k])[:, None] for k in range(3)]
    area = d[0] * d[1] + d[1] * d[2] + d[2] * d[0]
    area = np.where(is_leaf, area, -1.0)
    tri_per_node = leaf_cnt.sum(1)
    print(f"nodes with triangles: {(tri_per_node > 0).sum()} ({(tri_per_node > 0).mean() * 100:.0f} %), triangles per such node {tri_per_node[tri_per_node > 0].mean():.2f}, "
          f"leaf children per such node {is_leaf.sum(1)[tri_per_node > 0].mean():.2f}")
    best_slot = area.argmax(1)                                   # rule A: the leaf child with the largest box
    embedA = np.where(tri_per_node > 0, leaf_off[np.arange(nN), best_slot], 255)
    # ---- events -> per (visit) triangle tests ----------------------------------------------------------------------------------------
    tnode = (ev[istri] >> 5).astype(np.int64); ti = (ev[istri] & 31).astype(np.int64)
    hitA = (embedA[tnode] == ti).sum() / nrays
    # upper bound: per node the relative index tested most often in this trace
    key = tnode * 32 + ti
    uk, cnt = np.unique(key, return_counts=True)
    order = np.lexsort((-cnt, uk // 32))
    first = np.ones(uk.shape[0], bool); nn_ = (uk // 32)[order]; first[1:] = nn_[1:] != nn_[:-1]
    hitBest = cnt[order][first].sum() / nrays
    print(f"triangle tests served by ONE triangle embedded in its node's line: rule 'largest leaf box' {hitA:.2f} / ray ({hitA / T * 100:.0f} % of T); "
          f"best possible choice (oracle over this trace) {hitBest:.2f} ({hitBest / T * 100:.0f} %)")
    # ---- lines per ray under layouts ---------------------------------------------------------------------------------------------------
    # visit id of every triangle test = index of the last node event before it
    vid = np.cumsum(isnode)[istri]
    triBase = nd[:, 1, 1].astype(np.int64) // 3                  # triangle record index of the node's first triangle
    rec = triBase[tnode] + ti
    def distinct(lines):   # distinct (visit, line) pairs per ray
        return np.unique(vid * (1 << 34) + lines).shape[0] / nrays
    l64 = distinct(rec // 2)                                      # 64-byte records, two per line (shipped)
    l48 = distinct((rec * 48) // 128) + ((((rec * 48) % 128) > 80).sum() / nrays)   # packed 48-byte records: + straddlers
    not_emb = embedA[tnode] != ti
    # embedded layout: the embedded triangle lives in the node line; the node's OTHER triangles follow in their own array, 64 bytes each, two per line
    rel_other = ti - (ti > embedA[tnode])                         # index among the node's other triangles
    # (the per-node base of the compacted array is line-aligned per pair here: an optimistic but simple model)
    lE = np.unique((vid[not_emb] * (1 << 34)) + (rel_other[not_emb] // 2 + (tnode[not_emb] << 4))).shape[0] / nrays
    print(f"triangle lines per ray: shipped (64-byte records) {l64:.2f}; packed 48-byte {l48:.2f}; one triangle embedded per node + the others 64-byte, node-aligned {lE:.2f}")
    print(f"=> of about {S + l64:.1f} line fetches per ray (S + triangle lines), the embedded-triangle layout removes {l64 - lE:.2f} ({(l64 - lE) / (S + l64) * 100:.1f} %)")
    # tests per leaf-bearing visit
    nv = np.unique(vid).shape[0] / nrays
    print(f"node visits that test at least one triangle: {nv:.2f} / ray; triangle tests per such visit {T / nv:.2f}")

if __name__ == "__main__":
    main()
