"""Interleaved A/B of run-time configurations of the BVH8_CWBVH kernel on the contract bench's own batches: every round runs every configuration
once (so clock drift and box-to-box differences hit all of them alike), the report is the MEDIAN over the rounds.
A configuration is  name=hybridK:flags:variant  with hybridK = keep (the copies made at upload) | -1 (drop the node copy) | all | <n>, flags = tbvh_debug_set_flags bits, variant =
tbvh_set_variant.  Flag 8 = derive the hybrid copy without embedded triangles (round 4 A/B).
    python tools/ab_configs.py --side 4096 --rounds 7 base=-1:0:0 hy8k=8192:0:0 hy8k_nt=8192:1:0"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="+")
    ap.add_argument("--scene", default="bistro")
    ap.add_argument("--side", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--check", action="store_true", help="after the timing: the bounce batch's records and the shadow flags of every configuration byte-compared with the first one's")
    a = ap.parse_args()
    verts, label = scenes.get(a.scene)
    ctx = tb.Context(0)
    sc = tb.BVH8_CWBVH(ctx).Build(verts)
    n = a.side * a.side
    cams = scenes.cameras(a.scene)
    cam = R.camera(*cams[0], a.side, a.side, 1, 1)
    d_prim, d_diff, d_shad = make_batches(ctx, sc, verts, cam, n)
    d_occ = ctx.malloc(n)
    n_nodes = sc.host.blob(0, np.uint32, 4).shape[0] // 5
    cfgs = []
    for c in a.configs:
        name, spec = c.split("=")
        k, fl, v = spec.split(":")
        cfgs.append((name, None if k == "keep" else n_nodes if k == "all" else int(k), int(fl), int(v)))
    print(f"{label}; {n} rays per batch; {a.rounds} rounds, medians", flush=True)
    ms = {c[0]: {"primary": [], "diffuse": [], "shadow": []} for c in cfgs}
    cur_k = None
    for r in range(a.rounds + 1):
        for name, k, fl, v in cfgs:
            ctx.set_debug_flags(fl)
            if k is not None and (k, fl & 8) != cur_k:      # (flag 8: the hybrid copy is derived WITHOUT a triangle in each node's line)
                sc.set_hybrid(k); cur_k = (k, fl & 8)
            sc.set_variant(v)
            for kind, fn in (("primary", lambda: sc.intersect_device_fresh(d_prim, n, 1e30)), ("diffuse", lambda: sc.intersect_device_fresh(d_diff, n, 1e30)),
                             ("shadow", lambda: sc.occluded_device(d_shad, n, d_occ))):
                fn(); fn()
                t = ctx.time_last_ms()
                if r:
                    ms[name][kind].append(t)
    base = None
    for name, k, fl, v in cfgs:
        med = {kind: float(np.median(x)) for kind, x in ms[name].items()}
        rate = {kind: n / (m * 1e-3) / 1e6 for kind, m in med.items()}
        if base is None:
            base = rate
        print(f"{name:16s} hybrid {str(k):>7s} flags {fl} variant {v:3d}: primary {rate['primary']:7.1f} ({rate['primary'] / base['primary'] - 1:+.1%})  diffuse {rate['diffuse']:7.1f} ({rate['diffuse'] / base['diffuse'] - 1:+.1%})  "
              f"shadow {rate['shadow']:7.1f} ({rate['shadow'] / base['shadow'] - 1:+.1%})   primary+diffuse {2 * n / ((med['primary'] + med['diffuse']) * 1e-3) / 1e6:7.1f}", flush=True)
    if a.check:
        want = None
        for name, k, fl, v in cfgs:
            ctx.set_debug_flags(fl)
            if k is not None and (k, fl & 8) != cur_k:
                sc.set_hybrid(k); cur_k = (k, fl & 8)
            sc.set_variant(v)
            sc.intersect_device_fresh(d_diff, n, 1e30); sc.occluded_device(d_shad, n, d_occ)
            got = np.zeros((n, 16), np.uint32); occ = np.zeros(n, np.uint8)
            ctx.from_device(got, d_diff); ctx.from_device(occ, d_occ)
            if want is None:
                want = (got[:, 12:].copy(), occ)
                print(f"check: {name} is the yardstick ({int((got[:, 12].view(np.float32) < 1e30).sum())} bounce hits, {int(occ.sum())} occluded)")
            else:
                print(f"check: {name}: {int((got[:, 12:] != want[0]).any(1).sum())} bounce records differ, {int((occ != want[1]).sum())} shadow flags differ", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
