"""Leave the Infinity Cache: the same street generator at growing triangle counts (BVH 0.2 -> several GB), CWBVH built on
the device (tbvh_build_device: LBVH, one triangle per leaf — the host SAH builder would take minutes at these sizes; the
2.8 M point is also traced with the host-built tree for reference), camera rays and a bounce depth 1-3 mix.  Prints
MRays/s per size and variant and writes the launch order of the traversal kernel to a JSON file so that a
`rocprofv3 --pmc FETCH_SIZE` pass over the same command can be matched dispatch by dispatch (tools/size_sweep_pmc.py).
    python tools/size_sweep.py --sizes 2.8,12,30,60 --variants 0,62,64 [--side 2048] [--order order.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import tinybvh_amd as tb  # noqa: E402
from tinybvh_amd import rays as R  # noqa: E402
from tinybvh_amd import scenes  # noqa: E402
from ab_probe import make_batches  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="2.8,12,30,60", help="million triangles")
    ap.add_argument("--variants", default="0")
    ap.add_argument("--side", type=int, default=2048)
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--order", default="")
    ap.add_argument("--host-tree", action="store_true", help="also the host SAH tree at the first size")
    ap.add_argument("--st", action="store_true", help="node visits / triangle tests per ray of every tree (variant 59)")
    a = ap.parse_args()
    ctx = tb.Context(0)
    n = a.side * a.side
    cam = R.camera(*scenes.STREET_CAMERAS[0], a.side, a.side, 1, 1)
    order = []   # one entry per launch of the nearest-hit traversal kernel, in launch order
    variants = [int(x) for x in a.variants.split(",")]
    copy_gbps = None
    for k, m in enumerate([float(x) for x in a.sizes.split(",")]):
        nt = int(m * 1e6) if m != 2.8 else 2_832_120
        t0 = time.time()
        verts = scenes.street(nt, seed=2)
        t_gen = time.time() - t0
        trees = [("device LBVH", lambda: tb.BVH8_CWBVH(ctx).BuildOnDevice(verts))]
        if a.host_tree and k == 0:
            trees.append(("host SAH", lambda: tb.BVH8_CWBVH(ctx).Build(verts)))
        for tree, make in trees:
            t0 = time.time()
            sc = make()
            t_build = time.time() - t0
            nb = sc.download_sizes() if hasattr(sc, "download_sizes") else None
            d_prim, d_diff, d_shad = make_batches(ctx, sc, verts, cam, n)
            order += [dict(tris=nt, tree=tree, kind="prep", variant=0)] * 3
            ctx.free(d_shad)
            print(f"== {nt} triangles ({tree}): generated in {t_gen:.1f}s, built in {t_build:.2f}s, BVH {sc.device_bytes / 1e6:.0f} MB", flush=True)
            for v in variants:
                try:
                    sc.set_variant(v)
                except tb.TbvhError as e:
                    print(f"   variant {v}: {e}"); continue
                row = {}
                for kind, d in (("primary", d_prim), ("diffuse", d_diff)):
                    ms = []
                    for p in range(a.passes + 1):
                        sc.intersect_device_fresh(d, n, 1e30)
                        order.append(dict(tris=nt, tree=tree, kind=kind, variant=v, bytes=int(sc.device_bytes)))
                        t = ctx.time_last_ms()
                        if p:
                            ms.append(t)
                    row[kind] = n / (float(np.mean(ms)) * 1e-3) / 1e6
                print(f"   variant {v:3d}: BVH+copies {sc.device_bytes / 1e6:7.0f} MB  primary {row['primary']:7.1f}  diffuse {row['diffuse']:7.1f} MRays/s", flush=True)
            if a.st:
                # node visits (S) and triangle tests (T) per ray from the instrumented strict kernel: the algorithmic bytes of SURVEY §8(d)
                import ctypes as C
                sc.set_variant(59)
                for kind, d in (("primary", d_prim), ("diffuse", d_diff)):
                    st = (C.c_uint64 * 8)()
                    tb.lib.tbvh_debug_stats(ctx._h, st, 1)
                    sc.intersect_device_fresh(d, n, 1e30)
                    order.append(dict(tris=nt, tree=tree, kind="prep", variant=59))
                    tb.lib.tbvh_debug_stats(ctx._h, st, 1)
                    S, T = int(st[2]) / n, int(st[4]) / n
                    alg = S * 80 + T * 48 + 64 + 16
                    print(f"   S/T {kind:8s}: {S:.2f} node visits, {T:.2f} triangle tests per ray -> algorithmic {alg:.0f} B/ray (80 S + 48 T + 64 + 16)", flush=True)
                sc.set_variant(0)
            sc.free(); ctx.free(d_prim); ctx.free(d_diff)
        del verts
    if a.order:
        json.dump(order, open(a.order, "w"))
    ctx.close()


if __name__ == "__main__":
    main()
