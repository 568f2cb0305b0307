#!/usr/bin/env python
"""For seeds of tests/test_random_large.py: which rays differ between the default path and the forced kernel (and from launch to launch), with the classes of
that test, and what the oracle says about the first few.   usage: [REPEAT=1] python tools/debug/large_diff.py 54 34 6"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import tinybvh_amd as tb
from oracle_lib import Oracle
import test_random_large as T

ctx = tb.Context(0)
orc = Oracle(tie_rule=1)
np.set_printoptions(precision=9, floatmode="unique")
for seed in [int(a) for a in sys.argv[1:]]:
    case = T.Case(ctx, seed)
    rays = case.rays()
    host = case.sc.host
    want, want_occ = case.trace(rays, case.forced)
    got, occ = case.trace(rays, 0)
    print(f"seed {seed}: {case.name} {case.verts.shape[0] // 3} tris, layout {case.layout}, {case.kind}, {case.n} rays: {T.classes(got, want, rays, case.verts)}, {int((occ != want_occ).sum())} flags differ")
    diff, ok = T.knife_edge(got, want, rays, case.verts)
    show = np.concatenate([diff[~ok][:12], diff[ok][:4]])
    ref = orc.bvh2_intersect(host.bvh2_nodes(), host.bvh2_prim_idx(), case.verts, rays[show])
    tri = case.verts.reshape(-1, 3, 4)[:, :, :3]
    for j, i in enumerate(show):
        g, w, r = got[i], want[i], ref[j]
        print(f"  ray {i} ({'in a class' if ok[np.nonzero(diff == i)[0][0]] else 'OUTSIDE the classes'}): O {rays['O'][i]} D {rays['D'][i]}")
        print(f"     default t {g['t']:.9g} u {g['u']:.6g} v {g['v']:.6g} prim {g['prim']} | forced t {w['t']:.9g} u {w['u']:.6g} v {w['v']:.6g} prim {w['prim']} | oracle t {r['t']:.9g} u {r['u']:.6g} v {r['v']:.6g} prim {r['prim']}")
        for p in {int(g['prim']), int(w['prim'])}:
            if p < tri.shape[0]:
                print(f"      prim {p}: {tri[p].tolist()}")
    if os.environ.get("REPEAT"):
        prev, prev_occ = got, occ
        for k in range(1, 8):
            g2, o2 = case.trace(rays, 0)
            dr = np.nonzero((g2.view(np.uint8).reshape(-1, 64) != prev.view(np.uint8).reshape(-1, 64)).any(1))[0]
            print(f"   launch {k}: {dr.size} records differ from launch {k - 1}, {int((o2 != prev_occ).sum())} flags; schedule {case.sc.coherent_schedule(False)} {case.sc.coherent_schedule(True)} probe {ctx.last_probe()}")
            for i in dr[:4]:
                a, b = prev[i], g2[i]
                print(f"      ray {i} (of {case.n}): before t {a['t']:.9g} u {a['u']:.6g} v {a['v']:.6g} prim {a['prim']} | now t {b['t']:.9g} u {b['u']:.6g} v {b['v']:.6g} prim {b['prim']}")
            prev, prev_occ = g2, o2
    case.free()
