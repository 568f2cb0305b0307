#!/usr/bin/env python
"""The non-axis-aligned gap (round 5, review item 1): the SAME 2.83 M triangles straight and rotated by irrational angles about two axes, the
same camera carried along, 16.7 M camera / bounce (depth 1-3) / shadow rays each, traced on

    lib whole    the library's own host builder (binned SAH + SAH-optimal collapse) over whole triangles (TBVH_BUILD_WHOLE_TRIANGLES: round 4's default)
    lib +30%     ... with triangles split ahead of the build, 30 % extra references (host_builder.cpp: presplit): the default since round 5
    lib +100%    ... 100 % extra references
    ref Build    the real tinybvh BVH8_CWBVH::Build blob (oracle/_ref), uploaded verbatim
    ref BuildHQ  the real tinybvh BVH8_CWBVH::BuildHQ blob (spatial splits, tiny_bvh.h:2623-3040), uploaded verbatim

with node visits S and triangle tests T per ray counted by the oracle's mirror of the layout (tiny_bvh.h:7046-7154 restated) on a strided
sample of the very batches, and 5 S + 3 T + 4 = the L1 lookups per ray of the strict schedule (DESIGN.md par. 5).  The batches are made
once per scene on the library tree (hit records do not depend on the tree).

usage: tools/rotated_table.py [--side 4096] [--passes 5] [--scenes bistro,street_rot] > profiles/r05_rotated.txt"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import tinybvh_amd as tb  # noqa: E402
from tinybvh_amd import rays as R, scenes  # noqa: E402
from ab_probe import make_batches  # noqa: E402
from oracle_lib import Oracle, Reference, have_reference  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", type=int, default=4096)
    ap.add_argument("--passes", type=int, default=5)
    ap.add_argument("--scenes", default="bistro,street_rot")
    ap.add_argument("--sample", type=int, default=16384)
    a = ap.parse_args()
    orc = Oracle()
    n = a.side * a.side
    for name in a.scenes.split(","):
        verts, label = scenes.get(name)
        ctx = tb.Context(0)
        base = tb.BVH8_CWBVH(ctx).Build(verts)
        cam = R.camera(*scenes.cameras(name)[0], a.side, a.side, 1, 1)
        d_prim, d_diff, d_shad = make_batches(ctx, base, verts, cam, n)
        d_occ = ctx.malloc(n)
        samples = {}
        for kind, d in (("camera", d_prim), ("bounce", d_diff)):
            full = np.zeros(n, dtype=tb.RAY_DTYPE); ctx.from_device(full, d)
            s = full[:: max(n // a.sample, 1)][: a.sample].copy(); s["t"] = 1e30
            samples[kind] = s
            del full
        print(f"\n{label}; {n} rays per batch, median of {a.passes} launches; S / T on {a.sample} strided rays of each batch")
        print(f"{'tree':14s} {'nodes':>8s} {'tri rec':>8s} {'build s':>7s} | {'camera S':>8s} {'T':>6s} {'lookups':>7s} {'MRays/s':>8s} | {'bounce S':>8s} {'T':>6s} {'lookups':>7s} {'MRays/s':>8s} | {'shadow MRays/s':>14s}")
        trees = [("lib whole", 0.0), ("lib +30%", 0.3), ("lib +100%", 1.0)]
        if have_reference():
            trees += [("ref Build", "ref"), ("ref BuildHQ", "refhq")]
        first = {}
        for tname, spec in trees:
            t0 = time.time()
            if spec in ("ref", "refhq"):
                rs = Reference().build(verts, hq=spec == "refhq", threaded=True)
                nodes, tris = rs.blob(10, 0, np.uint32, 4), rs.blob(10, 1, np.uint32, 4)
                used = int(np.flatnonzero(tris.any(1)).max() + 1) // 3 if tris.shape[0] else 0   # (BuildHQ sizes bvh8Tris for 1.5 x the triangles; the tail is slack)
            else:
                host = tb.HostBVH(verts, tb.LAYOUT_CWBVH, split_budget=spec)
                nodes, tris = host.blob(0, np.uint32, 4), host.blob(1, np.uint32, 4)
                used = tris.shape[0] // 3
            dt = time.time() - t0
            sc = tb.BVH8_CWBVH(ctx).Upload(nodes, tris)
            row = {}
            for kind, d, fn in (("camera", d_prim, None), ("bounce", d_diff, None), ("shadow", d_shad, None)):
                ms = []
                for p_ in range(a.passes + 2):
                    if kind == "shadow":
                        sc.occluded_device(d, n, d_occ)
                    else:
                        sc.intersect_device_fresh(d, n, 1e30)
                    ctx.synchronize()
                    if p_ >= 2:
                        ms.append(ctx.time_last_ms())
                row[kind] = n / (float(np.median(ms)) * 1e-3) / 1e6
            st = {}
            for kind in ("camera", "bounce"):
                _, cnt = orc.cwbvh_intersect(nodes, tris, samples[kind].copy(), counts=True)
                st[kind] = (cnt[0] / samples[kind].shape[0], cnt[1] / samples[kind].shape[0])
            if not first:
                first = dict(row)
            cells = " | ".join(f"{st[k][0]:8.2f} {st[k][1]:6.2f} {5 * st[k][0] + 3 * st[k][1] + 4:7.1f} {row[k]:8.0f}" for k in ("camera", "bounce"))
            print(f"{tname:14s} {nodes.shape[0] // 5:8d} {used:8d} {dt:7.1f} | {cells} | {row['shadow']:14.0f}", flush=True)
            sc.free()
        for p_ in (d_prim, d_diff, d_shad, d_occ):
            ctx.free(p_)
        base.free(); ctx.close()


if __name__ == "__main__":
    main()
