"""Round 5, review item 1, the remedy the reference's own API offers for geometry that has a frame of its own: the rotated street as ONE instance of the
STRAIGHT street's BLAS under a TLAS whose instance transform is the rotation (tiny_bvh.h:3306-3380: the ray is taken into the BLAS's frame, t is preserved).
Same world-space rays (camera + bounce batches of the flat rotated scene), three ways: the flat rotated scene (BVH8_CWBVH over rotated vertices), the
1-instance TLAS over the straight BLAS (BVH8_CWBVH and BVH4_GPU BLAS), and — for scale — the straight street with the straight camera.
usage: tools/rotated_as_instance.py [--side 4096] > profiles/r05_rotated_as_instance.txt"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import tinybvh_amd as tb
from tinybvh_amd import rays as R, scenes
from ab_probe import make_batches

ap = argparse.ArgumentParser(); ap.add_argument("--side", type=int, default=4096); ap.add_argument("--passes", type=int, default=5)
a = ap.parse_args()
n = a.side * a.side
ctx = tb.Context(0)
straight, _ = scenes.get("bistro")
rot, label = scenes.get("street_rot")


def rate(fn):
    ms = []
    for p_ in range(a.passes + 2):
        fn(); ctx.synchronize()
        if p_ >= 2:
            ms.append(ctx.time_last_ms())
    return n / (float(np.median(ms)) * 1e-3) / 1e6


flat = tb.BVH8_CWBVH(ctx).Build(rot)
cam_rot = R.camera(*scenes.street_rot_camera(0), a.side, a.side, 1, 1)
d_prim, d_diff, d_shad = make_batches(ctx, flat, rot, cam_rot, n)
ctx.free(d_shad)
print(f"{label}; {n} world-space rays per batch, median of {a.passes} launches")
r_flat = {k: rate(lambda d=d: flat.intersect_device_fresh(d, n, 1e30)) for k, d in (("camera", d_prim), ("bounce", d_diff))}
ref = {}
for k, d in (("camera", d_prim), ("bounce", d_diff)):
    full = np.zeros(n, tb.RAY_DTYPE); ctx.from_device(full, d); ref[k] = full[:: n // 65536][:65536].copy()
print(f"  flat BVH8_CWBVH over the rotated vertices:            camera {r_flat['camera']:7.0f}  bounce {r_flat['bounce']:7.0f} MRays/s")
flat.free()
# the instance transform = the rotation scenes.street_rot applies (object -> world), as a row-major 4 x 4
M = np.eye(4, dtype=np.float32)
basis = np.eye(4, dtype=np.float32)[:3]
for ax, ang in scenes.STREET_ROT_ANGLES:
    basis = scenes.rotate(basis, ax, ang)
M[:3, :3] = basis[:, :3].T
for lay, name in ((10, "BVH8_CWBVH"), (8, "BVH4_GPU")):
    blas = tb.LAYOUT_CLASSES[lay](ctx).Build(straight)
    tlas = tb.TLAS(ctx).Build(tb.make_instances(M[None], np.zeros(1, np.uint32)), [blas])
    r = {k: rate(lambda d=d: tlas.intersect_device_fresh(d, n, 1e30)) for k, d in (("camera", d_prim), ("bounce", d_diff))}
    agree = {}
    for k, d in (("camera", d_prim), ("bounce", d_diff)):
        full = np.zeros(n, tb.RAY_DTYPE); ctx.from_device(full, d); got = full[:: n // 65536][:65536]
        w = ref[k]
        both = (got["t"] < 1e30) & (w["t"] < 1e30)
        rel = np.abs(got["t"][both] - w["t"][both]) / np.maximum(w["t"][both], 1e-20)
        agree[k] = (int(((got["t"] < 1e30) != (w["t"] < 1e30)).sum()), int((got["prim"][both] != w["prim"][both]).sum()), float(rel.max()) if rel.size else 0.0)
    print(f"  TLAS, ONE instance (the rotation) of the straight {name:10s}: camera {r['camera']:7.0f}  bounce {r['bounce']:7.0f} MRays/s"
          f"   [vs the flat scene's records, 65 536 sampled: hit/miss differs {agree['camera'][0]} / {agree['bounce'][0]}, prim differs {agree['camera'][1]} / {agree['bounce'][1]}, max relative t difference {max(agree['camera'][2], agree['bounce'][2]):.1e}]", flush=True)
    tlas.free(); blas.free()
for p_ in (d_prim, d_diff):
    ctx.free(p_)
sc = tb.BVH8_CWBVH(ctx).Build(straight)
cam = R.camera(*scenes.STREET_CAMERAS[0], a.side, a.side, 1, 1)
d_prim, d_diff, d_shad = make_batches(ctx, sc, straight, cam, n)
r = {k: rate(lambda d=d: sc.intersect_device_fresh(d, n, 1e30)) for k, d in (("camera", d_prim), ("bounce", d_diff))}
print(f"  (for scale) the straight street, flat BVH8_CWBVH:     camera {r['camera']:7.0f}  bounce {r['bounce']:7.0f} MRays/s")
ctx.close()
