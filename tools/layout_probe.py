"""Round-3 probe of the two levers DESIGN.md §5 points at for incoherent batches — what the kernel fetches, and in which order:
  (a) node placement: surface-area priority order with the first K nodes packed and the rest one per 128-byte line
      (tbvh_cwbvh_set_hybrid), K swept;
  (b) ray order: the diffuse batch binned by (origin cell, direction octant) (tbvh_bin_rays_device), cell bits and key form swept,
      the binning's own device time reported next to the traversal it speeds up;
on the contract bench's own batches (tools/ab_probe.py: make_batches).  Records of every placement are compared byte for byte with
the default's (the placement must not change a hit record); binned batches are compared through the permutation.
    python tools/layout_probe.py [--scene bistro --side 4096 --passes 3 --hybrid 0,8192,32768,131072,all --bins 4:0,5:0,5:1,5:2,6:1]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import tinybvh_amd as tb  # noqa: E402
from tinybvh_amd import rays as R  # noqa: E402
from tinybvh_amd import scenes  # noqa: E402
from ab_probe import make_batches  # noqa: E402


def rate(ctx, fn, n, passes):
    ms = []
    for p in range(passes + 1):
        fn()
        t = ctx.time_last_ms()
        if p:
            ms.append(t)
    return n / (float(np.mean(ms)) * 1e-3) / 1e6, float(np.mean(ms))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="bistro")
    ap.add_argument("--side", type=int, default=4096)
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--hybrid", default="all,0,8192,32768,131072")
    ap.add_argument("--bins", default="4:0,5:0,5:1,5:2,6:0,6:1")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--flags", default="", help="experiment flag sets to A/B on the default placement and on the best hybrid, e.g. 1,2,3")
    ap.add_argument("--no-bins", action="store_true")
    a = ap.parse_args()
    verts, label = scenes.get(a.scene)
    ctx = tb.Context(0)
    t0 = time.time()
    sc = tb.BVH8_CWBVH(ctx).Build(verts)
    print(f"scene: {label}: {verts.shape[0] // 3} tris; build+upload {time.time() - t0:.1f}s; {sc.device_bytes / 1e6:.0f} MB", flush=True)
    n = a.side * a.side
    cams = scenes.STREET_CAMERAS if a.scene == "bistro" else scenes.SPONZA_CAMERAS
    cam = R.camera(*cams[0], a.side, a.side, 1, 1)
    d_prim, d_diff, d_shad = make_batches(ctx, sc, verts, cam, n)
    d_occ = ctx.malloc(n)
    if a.variant:
        sc.set_variant(a.variant)
    n_nodes = sc.host.blob(0, np.uint32, 4).shape[0] // 5
    lo, hi = verts[:, :3].min(0), verts[:, :3].max(0)
    bounds = [float(x) for x in lo] + [float(x) for x in hi]

    def measure(tag):
        row = {}
        row["primary"], _ = rate(ctx, lambda: sc.intersect_device_fresh(d_prim, n, 1e30), n, a.passes)
        row["diffuse"], _ = rate(ctx, lambda: sc.intersect_device_fresh(d_diff, n, 1e30), n, a.passes)
        row["shadow"], _ = rate(ctx, lambda: sc.occluded_device(d_shad, n, d_occ), n, a.passes)
        print(f"{tag:34s} primary {row['primary']:7.1f}  diffuse {row['diffuse']:7.1f}  shadow {row['shadow']:7.1f} MRays/s   "
              f"primary+diffuse {2 * n / (n / row['primary'] + n / row['diffuse']):7.1f}", flush=True)
        return row

    def records(d, m=1 << 21):
        buf = np.zeros(min(n, m), tb.RAY_DTYPE)
        ctx.from_device(buf, d)
        return buf

    base = measure("uploaded array (default)")
    ref_p, ref_d = records(d_prim), records(d_diff)
    occ0 = np.zeros(n, np.uint8); ctx.from_device(occ0, d_occ)
    for k in [x for x in a.hybrid.split(",") if x]:
        K = n_nodes if k == "all" else int(k)
        t0 = time.time()
        sc.set_hybrid(K)
        dt = time.time() - t0
        measure(f"hybrid: {k:>7s} of {n_nodes} packed")
        gp, gd = records(d_prim), records(d_diff)
        occ = np.zeros(n, np.uint8); ctx.from_device(occ, d_occ)
        same = all(np.array_equal(gp[f].view(np.uint32), ref_p[f].view(np.uint32)) and np.array_equal(gd[f].view(np.uint32), ref_d[f].view(np.uint32)) for f in ("t", "u", "v", "prim"))
        print(f"    records identical to the default's: {same}; occlusion flags identical: {bool(np.array_equal(occ, occ0))}; set_hybrid {dt * 1e3:.0f} ms; {sc.device_bytes / 1e6:.0f} MB", flush=True)
    sc.set_hybrid(-1)
    for fl in [int(x) for x in a.flags.split(",") if x]:
        ctx.set_debug_flags(fl)
        measure(f"flags {fl}, uploaded array")
        gp, gd = records(d_prim), records(d_diff)
        same = all(np.array_equal(gp[f].view(np.uint32), ref_p[f].view(np.uint32)) and np.array_equal(gd[f].view(np.uint32), ref_d[f].view(np.uint32)) for f in ("t", "u", "v", "prim"))
        sc.set_hybrid(8192)
        measure(f"flags {fl}, hybrid 8192")
        sc.set_hybrid(-1)
        print(f"    records identical to the default's: {same}", flush=True)
    ctx.set_debug_flags(0)
    if a.no_bins:
        ctx.close()
        return

    # ---- ray order --------------------------------------------------------------------------------------------------------------
    d_sorted = ctx.malloc(n * 64)
    d_perm = ctx.malloc(n * 4)
    best = None
    for spec in [x for x in a.bins.split(",") if x]:
        bits, flags = (int(v) for v in spec.split(":"))
        bin_ms = []
        for p in range(3):
            ctx.bin_rays(d_diff, d_sorted, n, bounds, bits, flags, d_perm)
            t = ctx.time_last_ms()
            if p:
                bin_ms.append(t)
        r, ms = rate(ctx, lambda: sc.intersect_device_fresh(d_sorted, n, 1e30), n, a.passes)
        # the same rays, the same records: compare through the permutation on a sample
        perm = np.zeros(n, np.uint32); ctx.from_device(perm, d_perm)
        got = np.zeros(n, tb.RAY_DTYPE); ctx.from_device(got, d_sorted)
        full = np.zeros(n, tb.RAY_DTYPE); ctx.from_device(full, d_diff)
        ok = bool(np.array_equal(np.sort(perm), np.arange(n, dtype=np.uint32)))
        same = all(np.array_equal(got[f].view(np.uint32), full[f][perm].view(np.uint32)) for f in ("t", "u", "v", "prim"))
        del perm, got, full
        bm = float(np.mean(bin_ms))
        print(f"diffuse binned {bits} bits/axis, key form {flags}: traversal {r:7.1f} MRays/s ({ms:.3f} ms; pixel order {base['diffuse']:.1f}), binning {bm:.3f} ms "
              f"-> {n / ((ms + bm) * 1e-3) / 1e6:7.1f} MRays/s with the binning counted; permutation valid {ok}, records identical {same}", flush=True)
        if best is None or r > best[0]:
            best = (r, bits, flags)
    if best:
        # the best ray order on the best placements
        ctx.bin_rays(d_diff, d_sorted, n, bounds, best[1], best[2], d_perm)
        for k in [x for x in a.hybrid.split(",") if x]:
            K = n_nodes if k == "all" else int(k)
            sc.set_hybrid(K)
            r, ms = rate(ctx, lambda: sc.intersect_device_fresh(d_sorted, n, 1e30), n, a.passes)
            print(f"binned {best[1]}:{best[2]} + hybrid {k:>7s}: diffuse {r:7.1f} MRays/s", flush=True)
        sc.set_hybrid(-1)
    ctx.close()


if __name__ == "__main__":
    main()
