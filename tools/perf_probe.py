"""Kernel iteration probe (not the contract benchmark — that is bench.py): MRays/s of every
layout on a scene, primary / diffuse-bounce / shadow batches generated on the device."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tinybvh_amd as tb  # noqa: E402
from tinybvh_amd import rays as R  # noqa: E402
from tinybvh_amd import scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="sponza")
    ap.add_argument("--layouts", default="5,8,10")
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=1)
    ap.add_argument("--passes", type=int, default=5)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--cam", type=int, default=0)
    ap.add_argument("--optimal", action="store_true")
    ap.add_argument("--cprim", type=float, default=0.0)
    a = ap.parse_args()
    verts, label = scenes.get(a.scene)
    print(f"scene: {label}: {verts.shape[0] // 3} tris", flush=True)
    cams = scenes.SPONZA_CAMERAS if a.scene == "sponza" else scenes.STREET_CAMERAS
    eye, view = cams[a.cam]
    cam = R.camera(eye, view, a.width, a.height, a.spp, a.spp)
    n = a.width * a.height * a.spp * a.spp
    ctx = tb.Context(0)
    d_verts = ctx.malloc(verts.nbytes); ctx.to_device(d_verts, verts)
    d_prim = ctx.malloc(n * 64); d_work = ctx.malloc(n * 64); d_b = ctx.malloc(n * 64); d_occ = ctx.malloc(n)
    ext = float((verts[:, :3].max(0) - verts[:, :3].min(0)).max())
    for layout in [int(x) for x in a.layouts.split(",")]:
        t0 = time.time()
        sc = tb.LAYOUT_CLASSES[layout](ctx).Build(verts, optimal_collapse=a.optimal, c_prim=a.cprim)
        if a.variant:
            sc.set_variant(a.variant)
        tb_build = time.time() - t0
        ctx.generate_primary(cam, d_prim, 0, n)
        res = {}
        for kind in ("primary", "bounce1", "bounce2", "shadow"):
            ms = []
            for p in range(a.passes + 1):
                if kind == "primary":
                    ctx.generate_primary(cam, d_work, 0, n)
                    sc.intersect_device(d_work, n)
                elif kind == "bounce1":
                    ctx.generate_bounce(d_verts, d_prim, d_work, n, 1)
                    sc.intersect_device(d_work, n)
                elif kind == "bounce2":
                    ctx.generate_bounce(d_verts, d_b, d_work, n, 2)
                    sc.intersect_device(d_work, n)
                else:
                    ctx.generate_shadow(d_prim, d_work, n, (0.0, 0.9 * float(verts[:, 1].max()), 0.0), ext * 5e-7)
                    sc.occluded_device(d_work, n, d_occ)
                t = ctx.time_last_ms()
                if p:
                    ms.append(t)
            if kind == "primary":
                ctx.generate_primary(cam, d_prim, 0, n); sc.intersect_device(d_prim, n); ctx.synchronize()
            if kind == "bounce1":
                ctx.generate_bounce(d_verts, d_prim, d_b, n, 1); sc.intersect_device(d_b, n); ctx.synchronize()
            res[kind] = n / (np.mean(ms) * 1e-3) / 1e6
            if a.variant == 46:
                import ctypes as C
                st = (C.c_uint64 * 8)()
                tb.lib.tbvh_debug_stats(ctx._h, st, 1)
                tot = max(sum(int(x) for x in st), 1)
                print(f"   [{kind}] generation cohesion histogram (<.5 .5-.6 .6-.7 .7-.75 .75-.8 .8-.85 .85-.9 >=.9): " + " ".join(f"{int(x) / tot:.3f}" for x in st), flush=True)
            if a.variant == 48:
                import ctypes as C
                st = (C.c_uint64 * 8)()
                tb.lib.tbvh_debug_stats(ctx._h, st, 1)
                it, act, nlanes, titer, tri, niter, nuni = [int(x) for x in st[:7]]
                print(f"   [{kind}] lockstep: node-visit iterations {niter}, uniform {nuni} ({nuni / max(niter, 1):.3f}), lanes per node iteration {nlanes / max(niter, 1):.1f}", flush=True)
            if a.variant in (7, 9):
                import ctypes as C
                st = (C.c_uint64 * 8)()
                tb.lib.tbvh_debug_stats(ctx._h, st, 1)
                it, act, node, titer, tri, rf, rfd = [int(x) for x in st[:7]]
                if it:
                    print(f"   [{kind}] iters/wave-launch {it}  active/64 {act/it/64:.3f}  node-lanes/64 {node/it/64:.3f}  "
                          f"tri-iters per iter {titer/it:.3f}  tri-lanes/64 {tri/max(titer,1)/64:.3f}  refills {rf} ({rfd/max(rf,1):.1f} rays each)", flush=True)
        print(f"layout {layout}: host build+upload {tb_build:.2f}s  " + "  ".join(f"{k} {v:8.1f} MRays/s" for k, v in res.items()), flush=True)
        sc.free()
    ctx.close()


if __name__ == "__main__":
    main()
