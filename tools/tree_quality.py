#!/usr/bin/env python
"""Tree quality of builder settings, on the CPU: node visits S and triangle tests T per ray of BVH8_CWBVH trees built with different
host-builder parameters, counted by the oracle's CWBVH mirror (oracle/tbvh_oracle.c: orc_cwbvh_trace) on ONE fixed set of camera and
bounce rays of the bench scene.  The bounce kernel's L1 lookups per ray are 5 S + 3 T + 4 (DESIGN.md §5 "Round 4").

usage: tools/tree_quality.py [--scene bistro] [--side 160] name=bins:max_leaf:collapse(0 default, 1 optimal, 2 greedy):c_prim[:split_budget] | name=ref | name=refhq ...
"""
import argparse, ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import tinybvh_amd as tb
from tinybvh_amd import rays as R, scenes
from oracle_lib import Oracle, _p
from line_model import bounce


def count(orc, nodes, tris, batch):
    L = orc.lib
    L.orc_cwbvh_trace.restype = C.c_uint64
    L.orc_cwbvh_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64]
    cap = batch.shape[0] * 600
    out = np.zeros(cap, np.uint32)
    r = batch.copy()
    L.orc_set_tie_rule(1)
    nw = L.orc_cwbvh_trace(_p(nodes), _p(tris), _p(r), r.shape[0], r.strides[0], _p(out), cap)
    assert nw < cap
    ev = out[:nw]
    sep = ev == 0xFFFFFFFF
    node = ((ev & 0x80000000) != 0) & ~sep
    tri = ~node & ~sep
    return node.sum() / batch.shape[0], tri.sum() / batch.shape[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*")
    ap.add_argument("--scene", default="bistro"); ap.add_argument("--side", type=int, default=160)
    a = ap.parse_args()
    verts, label = scenes.get(a.scene)
    h = tb.HostBVH(verts, tb.LAYOUT_CWBVH)
    nodes, tris = h.blob(0, np.uint32, 4), h.blob(1, np.uint32, 4)
    orc = Oracle(1)
    cams = scenes.cameras(a.scene)
    prim = R.primary(R.camera(*cams[0], a.side, a.side, 1, 1))
    rng = np.random.default_rng(5)
    h1 = orc.cwbvh_intersect(nodes, tris, prim)
    b1 = bounce(verts, h1, rng); h2 = orc.cwbvh_intersect(nodes, tris, b1)
    b2 = bounce(verts, h2, rng); h3 = orc.cwbvh_intersect(nodes, tris, b2)
    b3 = bounce(verts, h3, rng)
    batch = np.concatenate([b1[: b1.shape[0] // 3], b2[: b2.shape[0] // 3], b3[: b3.shape[0] // 3]])
    print(f"{label}; {prim.shape[0]} camera rays, {batch.shape[0]} bounce rays (depths 1-3)")
    print(f"{'config':28s} {'nodes':>8s} {'trirec':>8s} {'build s':>8s} | camera S      T  | bounce S      T   lookups")
    for c in ["default=0:0:0:0"] + a.configs:
        name, spec = c.split("=")
        if spec in ("ref", "refhq"):     # the reference's own builder + ConvertFrom (oracle/_ref)
            from oracle_lib import Reference
            t0 = time.time()
            rs = Reference().build(verts, hq=spec == "refhq", threaded=True)
            n, t = rs.blob(10, 0, np.uint32, 4), rs.blob(10, 1, np.uint32, 4)
            dt = time.time() - t0
            sp, tp = count(orc, n, t, prim)
            sb, tbn = count(orc, n, t, batch)
            print(f"{name:28s} {n.shape[0] // 5:8d} {t.shape[0] // 3:8d} {dt:8.1f} | {sp:7.2f} {tp:6.2f}  | {sb:7.2f} {tbn:6.2f}  {5 * sb + 3 * tbn + 4:7.1f}", flush=True)
            continue
        f = spec.split(":")
        bins, leaf, opt, cp = int(f[0]), int(f[1]), int(f[2]), float(f[3])
        sb_ = float(f[4]) if len(f) > 4 else 0.0   # triangle split budget (fraction of the triangle count)
        t0 = time.time()
        hb = tb.HostBVH(verts, tb.LAYOUT_CWBVH, bins=bins, max_leaf_tris=leaf, optimal_collapse=opt == 1, greedy_collapse=opt == 2, c_prim=cp, split_budget=sb_)
        dt = time.time() - t0
        n, t = hb.blob(0, np.uint32, 4), hb.blob(1, np.uint32, 4)
        sp, tp = count(orc, n, t, prim)
        sb, tbn = count(orc, n, t, batch)
        print(f"{name:28s} {n.shape[0] // 5:8d} {t.shape[0] // 3:8d} {dt:8.1f} | {sp:7.2f} {tp:6.2f}  | {sb:7.2f} {tbn:6.2f}  {5 * sb + 3 * tbn + 4:7.1f}", flush=True)


if __name__ == "__main__":
    main()
