"""A/B of kernel variants on the contract bench's own batches (not the contract benchmark — that is bench.py): the scene is
built once, the primary / diffuse (bounce depths 1-3 in thirds) / shadow batches are generated once exactly as bench.py
does, then every requested variant traces them PASSES times (tbvh_intersect_device_fresh, HIP events).  Variants: the
diagnostic ones the library still holds (kernels_cwbvh.hip: launch_cwbvh; 0 = as shipped).
    python tools/ab_probe.py --variants 0,52,72 [--scene bistro --side 4096 --layout 10 --stats 59,61]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tinybvh_amd as tb  # noqa: E402
from tinybvh_amd import rays as R  # noqa: E402
from tinybvh_amd import scenes  # noqa: E402


def make_batches(ctx, sc, verts, cam, n, seed=1000):
    d_verts = ctx.malloc(verts.nbytes); ctx.to_device(d_verts, verts)
    d_prim, d_diff, d_shad = (ctx.malloc(n * 64) for _ in range(3))
    ext = float((verts[:, :3].max(0) - verts[:, :3].min(0)).max())
    light = (0.0, 0.9 * float(verts[:, 1].max()), 0.0)
    third = n // 3
    ctx.generate_primary(cam, d_prim, 0, n)
    sc.intersect_device(d_prim, n)
    ctx.generate_shadow(d_prim, d_shad, n, light, ext * 5e-7)
    ctx.generate_bounce(d_verts, d_prim, d_diff, n, seed + 1)
    sc.intersect_device(d_diff + third * 64, n - third)
    ctx.generate_bounce(d_verts, d_diff + third * 64, d_diff + third * 64, n - third, seed + 2)
    sc.intersect_device(d_diff + 2 * third * 64, n - 2 * third)
    ctx.generate_bounce(d_verts, d_diff + 2 * third * 64, d_diff + 2 * third * 64, n - 2 * third, seed + 3)
    ctx.reset_hits(d_prim, n)
    ctx.synchronize()
    ctx.free(d_verts)
    return d_prim, d_diff, d_shad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="bistro")
    ap.add_argument("--side", type=int, default=4096)
    ap.add_argument("--layout", type=int, default=10)
    ap.add_argument("--variants", default="0")
    ap.add_argument("--stats", default="", help="statistics variants (lane utilisation counters)")
    ap.add_argument("--hist", default="", help="histogram variants (layout 10: 79 wave ends, 80 pool dry, 81 wave ends with stealing)")
    ap.add_argument("--timeline", default="", help="wave timeline variants (layout 8: 13)")
    ap.add_argument("--passes", type=int, default=4)
    ap.add_argument("--out", default="")
    ap.add_argument("--device-build", action="store_true")
    ap.add_argument("--flags", type=int, default=0, help="tbvh_debug_set_flags for the measured launches")
    a = ap.parse_args()
    verts, label = scenes.get(a.scene)
    ctx = tb.Context(0)
    t0 = time.time()
    cls = tb.LAYOUT_CLASSES[a.layout]
    sc = cls(ctx).BuildOnDevice(verts) if a.device_build else cls(ctx).Build(verts)
    print(f"scene: {label}: {verts.shape[0] // 3} tris; layout {a.layout}; build+upload {time.time() - t0:.1f}s; {sc.device_bytes / 1e6:.0f} MB", flush=True)
    n = a.side * a.side
    cams = scenes.STREET_CAMERAS if a.scene == "bistro" else scenes.SPONZA_CAMERAS
    cam = R.camera(*cams[0], a.side, a.side, 1, 1)
    d_prim, d_diff, d_shad = make_batches(ctx, sc, verts, cam, n)
    d_occ = ctx.malloc(n)
    ctx.set_debug_flags(a.flags)
    res = {}
    ref_hits = {}
    for v in [int(x) for x in a.variants.split(",") if x]:
        try:
            sc.set_variant(v)
        except tb.TbvhError as e:
            print(f"variant {v}: {e}", flush=True)
            continue
        row = {}
        for kind, d in (("primary", d_prim), ("diffuse", d_diff), ("shadow", d_shad)):
            ms = []
            for p in range(a.passes + 1):
                if kind == "shadow":
                    sc.occluded_device(d, n, d_occ)
                else:
                    sc.intersect_device_fresh(d, n, 1e30)
                t = ctx.time_last_ms()
                if p:
                    ms.append(t)
            row[kind] = n / (float(np.mean(ms)) * 1e-3) / 1e6
            # cheap cross-check between variants: checksum of the hit prims (ties may differ) and number of hits
            if kind != "shadow":
                buf = np.zeros(min(n, 1 << 20), tb.RAY_DTYPE); ctx.from_device(buf, d)
                key = (kind,)
                sig = (int((buf["t"] < 1e30).sum()), int(buf["prim"].astype(np.uint64).sum()))
                if v == 0 or key not in ref_hits:
                    ref_hits.setdefault(key, (buf["prim"].copy(), buf["t"].copy()))
                pr, tt = ref_hits[key]
                row[kind + "_prim_diff"] = int((pr != buf["prim"]).sum())
                row[kind + "_t_diff"] = int((tt != buf["t"]).sum())
        res[v] = row
        print(f"variant {v:3d}: primary {row['primary']:7.1f}  diffuse {row['diffuse']:7.1f}  shadow {row['shadow']:7.1f} MRays/s   "
              f"primary+diffuse {2 * n / (n / row['primary'] + n / row['diffuse']):7.1f}   [vs first variant, first 1M rays: prim differs {row['primary_prim_diff']}/{row['diffuse_prim_diff']}, "
              f"t differs {row['primary_t_diff']}/{row['diffuse_t_diff']}]", flush=True)
    for v in [int(x) for x in a.stats.split(",") if x]:
        try:
            sc.set_variant(v)
        except tb.TbvhError as e:
            print(f"stats variant {v}: {e}", flush=True)
            continue
        for kind, d in (("primary", d_prim), ("diffuse", d_diff)):
            st = (C.c_uint64 * 8)()
            tb.lib.tbvh_debug_stats(ctx._h, st, 1)
            sc.intersect_device_fresh(d, n, 1e30)
            ms = ctx.time_last_ms()
            tb.lib.tbvh_debug_stats(ctx._h, st, 1)
            it, act, node, titer, tri, rf, rfd, niter = [int(x) for x in st]
            it = max(it, 1)
            print(f"stats {v} [{kind}] {ms:.2f} ms: wave-iterations {it}  per ray {it * 64 / n:.1f}  active/64 {act / it / 64:.3f}  "
                  f"node phases/iter {niter / it:.3f} at {node / max(niter, 1) / 64:.3f} lanes  ({node / n:.2f} node visits/ray)  "
                  f"tri phases/iter {titer / it:.3f} at {tri / max(titer, 1) / 64:.3f} lanes ({tri / n:.2f} tri tests/ray)  "
                  f"uniform node phases (one node, one octant) {rf / max(niter, 1):.3f} of all, holding {rfd / max(node, 1):.3f} of the node visits", flush=True)
            res[f"stats{v}_{kind}"] = dict(ms=ms, iters=it, active=act, node_lanes=node, node_iters=niter, tri_iters=titer, tri_lanes=tri, refills=rf)
    for v in [int(x) for x in a.timeline.split(",") if x]:
        # wave timeline variants (k_bvh4 variant 13): 10 ns ticks of the constant clock
        try:
            sc.set_variant(v)
        except tb.TbvhError as e:
            print(f"timeline variant {v}: {e}", flush=True)
            continue
        for kind, d in (("primary", d_prim), ("diffuse", d_diff)):
            st = (C.c_uint64 * 8)()
            for _ in range(3):
                tb.lib.tbvh_debug_stats(ctx._h, st, 1)
                sc.intersect_device_fresh(d, n, 1e30)
                ms = ctx.time_last_ms()
                tb.lib.tbvh_debug_stats(ctx._h, st, 1)
            nmin0, max0, sum0, nmin1, max1, sum1, sumdry, cnt = [int(x) for x in st]
            min0 = (~nmin0) & (2**64 - 1); min1 = (~nmin1) & (2**64 - 1)
            us = lambda t: t / 100.0
            print(f"timeline {v} [{kind}] kernel {ms * 1e3:.0f} us, {cnt} waves: starts {us(max0 - min0):.1f} us apart (mean +{us(sum0 / cnt - min0):.1f}); "
                  f"pool dry at mean +{us(sumdry / cnt - min0):.1f} us; ends first +{us(min1 - min0):.1f}, mean +{us(sum1 / cnt - min0):.1f}, last +{us(max1 - min0):.1f} us; "
                  f"mean wave life {us((sum1 - sum0) / cnt):.1f} us", flush=True)
    for v in [int(x) for x in a.hist.split(",") if x]:
        try:
            sc.set_variant(v)
        except tb.TbvhError as e:
            print(f"histogram variant {v}: {e}", flush=True)
            continue
        for kind, d in (("primary", d_prim), ("diffuse", d_diff)):
            st = (C.c_uint64 * 8)()
            for _ in range(2):
                tb.lib.tbvh_debug_stats(ctx._h, st, 1)
                sc.intersect_device_fresh(d, n, 1e30)
                ms = ctx.time_last_ms()
                tb.lib.tbvh_debug_stats(ctx._h, st, 1)
            if v in (82, 83, 86):
                mx, it, act, waves, steals = [int(x) for x in st][:5]
                print(f"tail {v} [{kind}] kernel {ms * 1e3:.0f} us: after the pool ran dry the waves ran {it / max(waves, 1):.0f} passes on average with {act / max(it, 1):.1f} lanes busy; "
                      f"the longest tail ran {mx >> 32} passes with {(mx & 0xFFFFFFFF) / max(mx >> 32, 1):.1f} lanes busy; subtrees taken over: {steals}", flush=True)
                continue
            print(f"histogram {v} [{kind}] kernel {ms * 1e3:.0f} us; waves per 64 us bin (last = 448 us and later): {[int(x) for x in st]}", flush=True)
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)
    sc.free(); ctx.close()


if __name__ == "__main__":
    main()
