"""The drop-in case timed: layouts built by the REAL tiny_bvh.h (oracle/_ref: BVH_GPU / BVH4_GPU / BVH8_CWBVH ::Build and
::BuildHQ, tiny_bvh.h:4551-4560, 5059-5070, 5822-5835) uploaded verbatim and traced by the HIP kernels, next to the
library's own host builder on the same scene and the same rays (primary, bounce depth 1-3 mix, shadow — the batches of
bench.py).  Not part of bench.py's value path.  Also counts node visits S / triangle tests T per ray of the CWBVH blobs
with the restated CPU mirror, which is what explains a rate difference between two trees.
    python tools/ref_blobs_probe.py --scenes bistro,bunny,head,suzanne,legocar [--side 2048]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import tinybvh_amd as tb  # noqa: E402
from tinybvh_amd import rays as R  # noqa: E402
from tinybvh_amd import scenes  # noqa: E402
from ab_probe import make_batches  # noqa: E402
from oracle_lib import Oracle, Reference, have_reference  # noqa: E402


def time_scene(ctx, sc, batches, n, passes=3):
    d_prim, d_diff, d_shad, d_occ = batches
    out = {}
    for kind, d in (("primary", d_prim), ("diffuse", d_diff), ("shadow", d_shad)):
        ms = []
        for p in range(passes + 1):
            if kind == "shadow":
                sc.occluded_device(d, n, d_occ)
            else:
                sc.intersect_device_fresh(d, n, 1e30)
            t = ctx.time_last_ms()
            if p:
                ms.append(t)
        out[kind] = n / (float(np.mean(ms)) * 1e-3) / 1e6
    return out


def mesh_camera(verts, side):
    lo, hi = verts[:, :3].min(0), verts[:, :3].max(0)
    c = (lo + hi) / 2; ext = float((hi - lo).max())
    eye = c + np.array([0.9, 0.5, -1.3], np.float32) * ext
    view = (c - eye) / np.linalg.norm(c - eye)
    return R.camera(tuple(float(x) for x in eye), tuple(float(x) for x in view), side, side, 1, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", default="bistro,bunny,head,suzanne,legocar")
    ap.add_argument("--side", type=int, default=2048)
    ap.add_argument("--layouts", default="10,8,5")
    ap.add_argument("--hq", type=int, default=1)
    ap.add_argument("--optimize", default="", help="also Build + Optimize(n) trees of the reinsertion optimiser (BVH8_CWBVH only), e.g. 25,50")
    a = ap.parse_args()
    assert have_reference(), "oracle/_ref/libtinybvh_ref.so missing (built by oracle/Makefile where /root/reference exists)"
    ref = Reference(); orc = Oracle()
    ctx = tb.Context(0)
    n = a.side * a.side
    shim = {5: 4, 8: 6, 10: 9}   # oracle/ref_shim.cpp's own layout selectors
    for name in a.scenes.split(","):
        if name in ("bistro", "sponza", "dragon"):
            verts, label = scenes.get(name)
            cam = R.camera(*(scenes.STREET_CAMERAS if name == "bistro" else scenes.SPONZA_CAMERAS)[0], a.side, a.side, 1, 1)
        else:
            path = scenes.find_real(name + ".bin") or os.path.join(ROOT, "gpurun_in", name + ".bin")
            if not path or not os.path.exists(path):
                print(f"{name}: mesh file not found, skipped", flush=True)
                continue
            verts, label = scenes.load_bin(path), name + ".bin (reference testdata)"
            cam = mesh_camera(verts, a.side)
        print(f"== {label}: {verts.shape[0] // 3} triangles, {n} rays per batch", flush=True)
        own = {L: tb.LAYOUT_CLASSES[L](ctx).Build(verts) for L in [int(x) for x in a.layouts.split(",")]}
        # the batches come from the library-built CWBVH (any correct BVH gives the same rays)
        first = own[10] if 10 in own else next(iter(own.values()))
        d_prim, d_diff, d_shad = make_batches(ctx, first, verts, cam, n)
        d_occ = ctx.malloc(n)
        batches = (d_prim, d_diff, d_shad, d_occ)
        sample = np.zeros(min(n, 65536), tb.RAY_DTYPE)
        stride = max(n // sample.shape[0], 1)
        full = np.zeros(n, tb.RAY_DTYPE); ctx.from_device(full, d_diff); dsample = full[::stride][: sample.shape[0]].copy(); dsample["t"] = 1e30
        ctx.from_device(full, d_prim); psample = full[::stride][: sample.shape[0]].copy(); psample["t"] = 1e30
        del full
        rows = []
        for hq in ([0, 1] if a.hq else [0]) + [100 + int(x) for x in a.optimize.split(",") if x]:
            t0 = time.time()
            rs = ref.build(verts, hq=hq, threaded=True)
            tag = f"tiny_bvh.h Build+Optimize({hq - 100})" if hq >= 100 else "tiny_bvh.h BuildHQ" if hq else "tiny_bvh.h Build"
            for L in own:
                if hq >= 100 and L != 10:
                    continue
                t1 = time.time()
                if L == 5:
                    sc = tb.BVH_GPU(ctx).Upload(rs.blob(5, 0, np.uint32, 16), rs.blob(5, 1, np.uint32, 1), verts)
                elif L == 8:
                    sc = tb.BVH4_GPU(ctx).Upload(rs.blob(8, 0, np.uint32, 4))
                else:
                    nodes, tris = rs.blob(10, 0, np.uint32, 4), rs.blob(10, 1, np.uint32, 4)
                    sc = tb.BVH8_CWBVH(ctx).Upload(nodes, tris)
                r = time_scene(ctx, sc, batches, n)
                r.update(layout=L, tree=tag + f" [{time.time() - t0:.1f} s]", mb=sc.device_bytes / 1e6, build_s=time.time() - t1)
                if L == 10:
                    for kind, smp in (("primary", psample), ("diffuse", dsample)):
                        _, cnt = orc.cwbvh_intersect(nodes, tris, smp, counts=True)
                        r[kind + "_S"] = cnt[0] / smp.shape[0]; r[kind + "_T"] = cnt[1] / smp.shape[0]
                rows.append(r)
                sc.free()
            del rs
        for L, sc in own.items():
            r = time_scene(ctx, sc, batches, n)
            r.update(layout=L, tree="library host builder", mb=sc.device_bytes / 1e6, build_s=0.0)
            if L == 10:
                h = sc.host
                for kind, smp in (("primary", psample), ("diffuse", dsample)):
                    _, cnt = orc.cwbvh_intersect(h.blob(0, np.uint32, 4), h.blob(1, np.uint32, 4), smp, counts=True)
                    r[kind + "_S"] = cnt[0] / smp.shape[0]; r[kind + "_T"] = cnt[1] / smp.shape[0]
            rows.append(r)
        names = {5: "BVH_GPU", 8: "BVH4_GPU", 10: "BVH8_CWBVH"}
        for L in own:
            for r in [x for x in rows if x["layout"] == L]:
                extra = f"  S/T primary {r['primary_S']:.1f}/{r['primary_T']:.1f}  diffuse {r['diffuse_S']:.1f}/{r['diffuse_T']:.1f}" if "diffuse_S" in r else ""
                print(f"  {names[L]:11s} {r['tree']:32s} {r['mb']:7.1f} MB  primary {r['primary']:7.1f}  diffuse {r['diffuse']:7.1f}  shadow {r['shadow']:7.1f} MRays/s{extra}", flush=True)
        for d in batches:
            ctx.free(d)
        for sc in own.values():
            sc.free()
    ctx.close()


if __name__ == "__main__":
    main()
