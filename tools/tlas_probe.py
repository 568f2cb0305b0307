"""BASELINE config 5 probe: TLAS over N^3 instances of the Dragon stand-in, 8 M primary rays per
frame, per-frame host TLAS rebuild + update, BLAS layout BVH4_GPU or CWBVH."""
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


def instances(n_side, t, scale=0.07 * 10, n_blas=1):
    g = np.stack(np.meshgrid(np.arange(n_side), np.arange(n_side), np.arange(n_side), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    k = g.shape[0]
    ang = (t * 0.5 + np.arange(k) * 0.37).astype(np.float32)
    c, s = np.cos(ang), np.sin(ang)
    T = np.zeros((k, 4, 4), np.float32)
    T[:, 0, 0] = c * scale; T[:, 0, 2] = s * scale; T[:, 1, 1] = scale; T[:, 2, 0] = -s * scale; T[:, 2, 2] = c * scale; T[:, 3, 3] = 1
    T[:, :3, 3] = g * 2.0
    return tb.make_instances(T, (np.arange(k) % n_blas).astype(np.uint32))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layout", type=int, default=8)
    ap.add_argument("--side", type=int, default=10)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--layout2", type=int, default=0, help="a second BLAS of this layout: every other instance uses it (mixed BLAS layouts under one TLAS)")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--random", type=int, default=0, help="also trace this many incoherent rays (random origins and directions inside the grid)")
    a = ap.parse_args()
    verts, label = scenes.get("dragon")
    ctx = tb.Context(0)
    blas = tb.LAYOUT_CLASSES[a.layout](ctx).Build(verts)
    blases = [blas] + ([tb.LAYOUT_CLASSES[a.layout2](ctx).Build(verts)] if a.layout2 else [])
    W, H = a.width, a.height
    n = W * H
    ext = 2.0 * a.side
    cam = R.camera((-0.6 * ext, 0.8 * ext, -0.9 * ext), (0.62, -0.38, 0.68), W, H, 1, 1)
    d_rays = ctx.malloc(n * 64)
    ctx.generate_primary(cam, d_rays, 0, n)
    tlas = tb.TLAS(ctx)
    for f in range(a.frames):
        inst = instances(a.side, float(f), n_blas=len(blases))
        t0 = time.perf_counter()
        tlas.Build(inst, blases)          # host: BLASInstance update + TLAS build + upload/update
        t_host = time.perf_counter() - t0
        if a.variant:
            tlas.set_variant(a.variant)
        tlas.intersect_device_fresh(d_rays, n, 1e30)
        ms = ctx.time_last_ms()
        print(f"frame {f}: {inst.shape[0]} instances of {label} ({verts.shape[0] // 3} tris), BLAS layout {a.layout}: host TLAS rebuild+upload {t_host * 1e3:.2f} ms, "
              f"trace {n} rays {ms:.3f} ms = {n / ms / 1e3:.1f} MRays/s", flush=True)
    # the same frames with the TLAS rebuilt on the device (tbvh_rebuild_tlas_device): transforms go up (64 B per
    # instance), instance update + LBVH build run on the GPU
    for f in range(a.frames):
        inst = instances(a.side, float(f), n_blas=len(blases))
        xf = np.ascontiguousarray(inst["transform"])
        ctx.synchronize()
        t0 = time.perf_counter()
        tlas.RebuildOnDevice(xf)
        t_call = time.perf_counter() - t0
        ctx.synchronize()
        t_sync = time.perf_counter() - t0
        ms_build = ctx.time_last_ms()
        if a.variant:
            tlas.set_variant(a.variant)
        tlas.intersect_device_fresh(d_rays, n, 1e30)
        ms = ctx.time_last_ms()
        print(f"frame {f}: DEVICE TLAS rebuild: host call {t_call * 1e3:.3f} ms (returns before the GPU is done), until done {t_sync * 1e3:.3f} ms, "
              f"device time {ms_build:.3f} ms; trace {ms:.3f} ms = {n / ms / 1e3:.1f} MRays/s", flush=True)
    if a.variant == 12:
        import ctypes as C
        st = (C.c_uint64 * 8)()
        tb.lib.tbvh_debug_stats(ctx._h, st, 1)
        tot = max(sum(int(x) for x in st), 1)
        print("   camera rays: generation cohesion histogram (<.25 -.375 -.5 -.6 -.7 -.8 -.9 >=.9): " + " ".join(f"{int(x) / tot:.3f}" for x in st), flush=True)
    if a.random:
        m = a.random
        rr = R.random_rays(m, (-1.0, -1.0, -1.0), (ext, ext, ext), seed=9)
        d_rr = ctx.malloc(m * 64); ctx.to_device(d_rr, rr)
        d_occ = ctx.malloc(m)
        ms = []
        for k in range(4):
            tlas.intersect_device_fresh(d_rr, m, 1e30); t = ctx.time_last_ms()
            if k:
                ms.append(t)
        mo = []
        for k in range(4):
            tlas.occluded_device(d_rr, m, d_occ); t = ctx.time_last_ms()
            if k:
                mo.append(t)
        if a.variant == 12:
            import ctypes as C
            st = (C.c_uint64 * 8)()
            tb.lib.tbvh_debug_stats(ctx._h, st, 1)
            tot = max(sum(int(x) for x in st), 1)
            print("   incoherent: generation cohesion histogram (<.25 -.375 -.5 -.6 -.7 -.8 -.9 >=.9): " + " ".join(f"{int(x) / tot:.3f}" for x in st), flush=True)
        if a.variant in (15, 26) and a.layout != 5:   # statistics variants of the flat loop (15) and of k_tlas4 / k_tlas8 (26)
            import ctypes as C
            st = (C.c_uint64 * 8)()
            tb.lib.tbvh_debug_stats(ctx._h, st, 1)
            tlas.intersect_device_fresh(d_rr, m, 1e30); ctx.synchronize()
            tb.lib.tbvh_debug_stats(ctx._h, st, 1)
            it, act, pa, la, pb, lb, pc, lc = [int(x) for x in st]
            print(f"   incoherent flat-loop statistics: wave iterations {it} ({it * 64 / m:.1f} per ray), active lanes/64 {act / max(it, 1) / 64:.3f}; "
                  f"TLAS (26: node) phases {pa / it:.3f}/iter at {la / max(pa, 1):.1f} lanes ({la / m:.1f} steps/ray); instance phases {pb / it:.3f}/iter at {lb / max(pb, 1):.1f} lanes ({lb / m:.1f}/ray); "
                  f"BLAS phases {pc / it:.3f}/iter at {lc / max(pc, 1):.1f} lanes ({lc / m:.1f} steps/ray)", flush=True)
        print(f"incoherent: {m} random rays: Intersect {np.mean(ms):.3f} ms = {m / np.mean(ms) / 1e3:.1f} MRays/s, IsOccluded {np.mean(mo):.3f} ms = {m / np.mean(mo) / 1e3:.1f} MRays/s", flush=True)
    hits = np.zeros(n, tb.RAY_DTYPE); ctx.from_device(hits, d_rays)
    print("hit fraction", float((hits["t"] < 1e30).mean()), "distinct instances hit", len(np.unique(hits["inst"][hits["t"] < 1e30])))
    ctx.close()


if __name__ == "__main__":
    main()
