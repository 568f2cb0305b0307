#!/usr/bin/env python
"""CPU only: why does the BVH8_CWBVH traversal of a library-built blob not reach a given primitive for a given ray?  Finds the chain of nodes above the
primitive's leaf slot and evaluates each one's child-box test for the ray the way cwbvh_node.h: cw_test_node does (float32, fma emulated in float64).
usage: cw_trace.py SEED RAY PRIM   (seed / ray of tests/test_random_large.py, camera kind)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import tinybvh_amd as tb
import test_random_large as T

f32 = np.float32
def fma(a, b, c): return f32(np.float64(a) * np.float64(b) + np.float64(c))

seed, ri, prim = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(9000 + seed)
name, verts = T.make_scene(rng)
layout = T.LAYOUTS[int(rng.integers(0, 3))]
n_min = int(rng.choice([800_000, 1_100_000, 1_600_000, 2_200_000]))
kind = ["camera", "bounce", "shadow", "random"][int(rng.integers(0, 4))]
lo, hi = verts[:, :3].min(0), verts[:, :3].max(0)
cam = T.camera_rays(rng, lo, hi, n_min)
r = cam[ri]
O, D, rD = r["O"], r["D"], r["rD"]
host = tb.HostBVH(verts, tb.LAYOUT_CWBVH)
nodes = host.blob(0, np.float32, 4).reshape(-1, 5, 4)
tris = host.blob(1, np.float32, 4).reshape(-1, 3, 4)
nu = nodes.view(np.uint32)
tu = tris.view(np.uint32)
recs = np.nonzero(tu[:, 2, 3] == prim)[0]
print(f"{name}, {nodes.shape[0]} nodes, {tris.shape[0]} triangle records; prim {prim} sits in records {recs.tolist()}")
tri = verts.reshape(-1, 3, 4)[prim, :, :3]
print("triangle", tri.tolist())

def children(ni):
    """(slot, kind, index / (first record, count), box lo, box hi) of node ni"""
    n = nodes[ni]; u = nu[ni]
    ex = np.array([np.int8(u[0, 3] & 255), np.int8((u[0, 3] >> 8) & 255), np.int8((u[0, 3] >> 16) & 255)]).astype(np.int32)
    imask = u[0, 3] >> 24
    cb, tbase = u[1, 0], u[1, 1]
    meta = np.frombuffer(u[1, 2:4].tobytes(), np.uint8)
    q = np.frombuffer(u[2:5].tobytes(), np.uint8).reshape(6, 8)      # qlox qloy qloz qhix qhiy qhiz
    out = []
    for s in range(8):
        m = int(meta[s])
        if m == 0: continue
        blo = n[0, :3].astype(np.float64) + q[0:3, s].astype(np.float64) * np.exp2(ex.astype(np.float64))
        bhi = n[0, :3].astype(np.float64) + q[3:6, s].astype(np.float64) * np.exp2(ex.astype(np.float64))
        if (m & 0x18) == 0x18:
            slot = (m & 31) - 24
            idx = int(cb) + bin(int(imask) & ((1 << slot) - 1)).count("1")
            out.append((s, "node", idx, blo, bhi, q[:, s], ex))
        else:
            out.append((s, "leaf", (int(tbase) // 3 + (m & 31), bin(m >> 5).count("1")), blo, bhi, q[:, s], ex))
    return out

# parent map
parent = {}
leaf_of = {}
for ni in range(nodes.shape[0]):
    for c in children(ni):
        if c[1] == "node": parent[c[2]] = (ni, c[0])
        else:
            for k in range(c[2][1]): leaf_of[c[2][0] + k] = (ni, c[0])
for rec in recs:
    ni, slot = leaf_of[int(rec)]
    chain = [(ni, slot)]
    while ni in parent:
        ni, slot = parent[ni]
        chain.append((ni, slot))
    chain.reverse()
    print(f"record {rec}: chain of (node, slot) from the root: {chain}")
    for ni, slot in chain[-2:]:
        c = [x for x in children(ni) if x[0] == slot][0]
        n0 = nodes[ni, 0]
        q, ex = c[5], c[6]
        a = [f32(np.ldexp(rD[k], int(ex[k]))) for k in range(3)]
        o = [f32(f32(n0[k] - O[k]) * rD[k]) for k in range(3)]
        tn, tf = [], []
        for k in range(3):
            ql, qh = (q[3 + k], q[k]) if rD[k] < 0 else (q[k], q[3 + k])
            tn.append(fma(f32(ql), a[k], o[k])); tf.append(fma(f32(qh), a[k], o[k]))
        cmin = max(max(tn), f32(0)); cmax = min(tf)
        # the exact distances to the (decoded, exact) planes
        ex_n = [((c[4][k] if rD[k] < 0 else c[3][k]) - np.float64(O[k])) / np.float64(D[k]) for k in range(3)]
        ex_f = [((c[3][k] if rD[k] < 0 else c[4][k]) - np.float64(O[k])) / np.float64(D[k]) for k in range(3)]
        inside = np.all(c[3] <= tri.min(0) + 0) and np.all(c[4] >= tri.max(0))
        print(f"  node {ni} slot {slot} ({c[1]}): box {c[3].tolist()} .. {c[4].tolist()}  holds the triangle: {bool(inside)}")
        print(f"      computed near {[float(x) for x in tn]} far {[float(x) for x in tf]}  ->  cmin {float(cmin):.9g} cmax {float(cmax):.9g}  {'ENTERED' if cmin <= cmax else 'MISSED'}")
        print(f"      exact    near {[float(x) for x in ex_n]} far {[float(x) for x in ex_f]}  ->  {max(max(ex_n), 0):.9g} .. {min(ex_f):.9g}")
