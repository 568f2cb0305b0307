"""Interleaved A/B of k_tlas4's phase thresholds / refill threshold (experiment build: debug flags bits 16..19, kernels_tlas4.hip) on config 5's TLAS:
camera, shadow and random rays, medians over rounds.   usage: TBVH_LIB_OVERRIDE=.../libtinybvh_amd_exp.so python tools/ab_tlas4.py [rounds]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tinybvh_amd as tb
from tinybvh_amd import rays as R, scenes
import bench_detail as bd

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ctx = tb.Context(0)
dlabel, blas, tlas, cam, nt = bd.config5_setup(tb, R, scenes, ctx, 8)
d = ctx.malloc(nt * 64); d_sh = ctx.malloc(nt * 64); d_occ = ctx.malloc(nt)
ctx.generate_primary(cam, d, 0, nt)
tlas.intersect_device_fresh(d, nt, 1e30)
ctx.generate_shadow(d, d_sh, nt, (10.0, 40.0, 10.0), 20.0 * 5e-7)
rr = R.random_rays(1 << 22, (-1.0, -1.0, -1.0), (20.0, 20.0, 20.0), seed=9)
d_r = ctx.malloc(rr.shape[0] * 64); ctx.to_device(d_r, rr)
names = {0: "shipped 24/8/8 refill 16", 1: "32/8/8", 2: "16/8/8", 3: "24/4/4", 4: "24/16/16", 5: "24/8/16", 6: "24/16/8", 7: "refill 8", 8: "refill 32", 9: "40/8/8"}
res = {k: ([], [], []) for k in names}
ref = None
for r in range(rounds):
    for k in names:
        ctx.set_debug_flags(k << 16)
        for j, fn in enumerate((lambda: tlas.intersect_device_fresh(d, nt, 1e30), lambda: tlas.occluded_device(d_sh, nt, d_occ), lambda: tlas.intersect_device_fresh(d_r, rr.shape[0], 1e30))):
            fn(); ctx.synchronize(); fn(); ctx.synchronize()
            res[k][j].append(ctx.time_last_ms())
        if r == 0:
            rec = np.zeros(nt, tb.RAY_DTYPE); ctx.from_device(rec, d)
            if ref is None: ref = rec
            else: assert np.array_equal(rec.view(np.uint8), ref.view(np.uint8)), k
base = [float(np.median(x)) for x in res[0]]
for k, nm in names.items():
    m = [float(np.median(x)) for x in res[k]]
    print(f"{nm:26s} camera {nt / m[0] / 1e3:7.1f} ({(base[0] / m[0] - 1) * 100:+.1f} %)  shadow {nt / m[1] / 1e3:7.1f} ({(base[1] / m[1] - 1) * 100:+.1f} %)  random {rr.shape[0] / m[2] / 1e3:7.1f} ({(base[2] / m[2] - 1) * 100:+.1f} %)")
