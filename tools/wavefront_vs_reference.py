"""Frame time of the path tracer against the reference's own wavefront.cl on the same GPU: the demo's scene set-up
(tiny_bvh_gpu.cpp:93-94, 128-158: Sponza stand-in + the 9 x 5 light quad), the demo's resolution, the same CWBVH blob, the same
camera and blue-noise table; the reference kernels run through ROCm OpenCL (oracle/ref_ocl.cpp), this library in
TBVH_WF_REFERENCE_LETTER mode (the same estimator).  Per-frame time = (time of F2 frames - time of F1 frames) / (F2 - F1), wall clock
around the whole call including the final read-back, so set-up, kernel compilation and upload cancel out."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import tinybvh_amd as tb  # noqa: E402
from tinybvh_amd import rays as R  # noqa: E402
from tinybvh_amd import scenes  # noqa: E402
from oracle_lib import ReferenceOpenCL  # noqa: E402
from test_wavefront_reference import ATOMIC_CONNECT, blue_noise, compare, u2f  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=1280)
ap.add_argument("--height", type=int, default=720)
ap.add_argument("--f1", type=int, default=8)
ap.add_argument("--f2", type=int, default=40)
a = ap.parse_args()
verts, label = scenes.get("sponza")
verts = verts.copy(); verts[:, 3] = u2f(0x00C0C0C0)
w, d, pos = 9.0, 5.0, np.array([-22.0, 12.0, 2.0], np.float32)
q = np.array([[-w, 0, -d], [w, 0, -d], [w, 0, d], [-w, 0, -d], [w, 0, d], [-w, 0, d]], np.float32) * 0.5 + pos
quad = np.zeros((6, 4), np.float32); quad[:, :3] = q; quad[:, 3] = u2f(0x01FFFFFF)
verts = np.ascontiguousarray(np.concatenate([quad, verts]))
W, H = a.width, a.height
ctx = tb.Context(0)
ocl = ReferenceOpenCL()
sc = tb.BVH8_CWBVH(ctx).Build(verts)
cam = R.camera(*scenes.SPONZA_CAMERAS[0], W, H, 1, 1)
noise = blue_noise()
d_verts = ctx.malloc(verts.nbytes); ctx.to_device(d_verts, verts)
wf = tb.Wavefront(ctx, W, H); wf.set_blue_noise(noise)
h = sc.host


def reference(frames, patch):
    t0 = time.perf_counter()
    img = ocl.wavefront(h.blob(0, np.uint32, 4), h.blob(1, np.uint32, 4), verts, noise, list(cam.eye), list(cam.p1), list(cam.p2), list(cam.p3), W, H, frames, 3, patch)
    return img, time.perf_counter() - t0


def mine(frames):
    t0 = time.perf_counter()
    for f in range(frames):
        wf.render(sc, d_verts, cam, (-22.0, 12.0, 2.0), (25.0, 25.0, 22.0), sky_lo=(0.7, 0.7, 1.2), sky_hi=(0.7, 0.7, 1.2), eps=1e-4, max_depth=3, seed=1000 + f, clear=(f == 0),
                  stats=False, light_size=(9.0, 5.0), one_diffuse_bounce=True, reference_letter=True, sample_index=f)
    img = wf.read() / frames
    return img, time.perf_counter() - t0


print(f"{label} + light quad: {verts.shape[0] // 3} triangles, {W} x {H}, 3 iterations per frame; OpenCL device {ocl.device}")
for tag, patch in (("wavefront.cl as shipped", ""), ("wavefront.cl with Connect's accumulation made atomic", ATOMIC_CONNECT)):
    reference(2, patch)
    _, t1 = reference(a.f1, patch); ref_img, t2 = reference(a.f2, patch)
    print(f"  reference, {tag}: {(t2 - t1) / (a.f2 - a.f1) * 1e3:.3f} ms per frame")
mine(2)
_, t1 = mine(a.f1); img, t2 = mine(a.f2)
print(f"  this library (device-resident queues, reference-letter shading): {(t2 - t1) / (a.f2 - a.f1) * 1e3:.3f} ms per frame")
rel, bias = compare(ref_img, img)
print(f"  images after {a.f2} frames: mean relative difference of 8 x 8 blocks {rel:.4f}, bias {bias:+.4f}")
