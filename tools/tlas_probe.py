"""Config 5 probe (not the contract bench): 1000 instances of the Dragon stand-in (10 x 10 x 10 grid), 3840 x 2160 camera rays, 4.2 M random rays, shadow rays
of the camera hits — MRays/s by HIP events for BLAS layout 8 (BVH4_GPU) or 10 (BVH8_CWBVH).  TBVH_TLAS_PACKET=0 in the environment turns the wave-packet
kernel of kernels_tlas8_packet.hip off (A/B)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tinybvh_amd as tb  # noqa: E402
from tinybvh_amd import rays as R  # noqa: E402
from tinybvh_amd import scenes  # noqa: E402
import bench_detail as bd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--layout", type=int, default=10)
ap.add_argument("--passes", type=int, default=6)
ap.add_argument("--random", type=int, default=1 << 22)
a = ap.parse_args()
ctx = tb.Context(0)
dlabel, blas, tlas, cam, nt = bd.config5_setup(tb, R, scenes, ctx, a.layout)
d = ctx.malloc(nt * 64)
d_sh = ctx.malloc(nt * 64)
d_occ = ctx.malloc(nt)
ctx.generate_primary(cam, d, 0, nt)


def timed(fn):
    ms = []
    for p in range(a.passes):
        fn(); ctx.synchronize()
        if p:
            ms.append(ctx.time_last_ms())
    return float(np.median(ms))


ms_cam = timed(lambda: tlas.intersect_device_fresh(d, nt, 1e30))
ext = 20.0
ctx.generate_shadow(d, d_sh, nt, (10.0, 40.0, 10.0), ext * 5e-7)
ms_sh = timed(lambda: tlas.occluded_device(d_sh, nt, d_occ))
rr = R.random_rays(a.random, (-1.0, -1.0, -1.0), (ext, ext, ext), seed=9)
d_r = ctx.malloc(rr.shape[0] * 64); ctx.to_device(d_r, rr)
ms_r = timed(lambda: tlas.intersect_device_fresh(d_r, rr.shape[0], 1e30))
rec = np.zeros(nt, tb.RAY_DTYPE); ctx.from_device(rec, d)
print(f"BLAS layout {a.layout} ({dlabel}), TBVH_TLAS_PACKET={os.environ.get('TBVH_TLAS_PACKET', '1')}: camera {nt / ms_cam / 1e3:8.1f} MRays/s ({ms_cam:.3f} ms)  "
      f"shadow {nt / ms_sh / 1e3:8.1f}  random {rr.shape[0] / ms_r / 1e3:8.1f}   hits {int((rec['t'] < 1e30).sum())}  checksum {int(rec['prim'][rec['t'] < 1e30].astype(np.uint64).sum())} {float(rec['t'][rec['t'] < 1e30].astype(np.float64).sum()):.6f}", flush=True)
ctx.close()
