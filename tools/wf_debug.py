"""Bisecting aid for tests/test_wavefront_reference.py: the reference's wavefront.cl (oracle/ref_ocl.cpp) next to this library's path tracer
per iteration count, with terms switched off on both sides (REFOCL_WF_PATCH edits the .cl text, WF_SKY / WF_LCOL set the library's sky and light)."""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import tinybvh_amd as tb
from tinybvh_amd import rays as R, scenes
from oracle_lib import ReferenceOpenCL
from test_wavefront_reference import demo_scene, blue_noise, blocks
ctx = tb.Context(0)
ocl = ReferenceOpenCL()
verts = demo_scene()
W, H, frames = 256, 128, 256
sc = tb.BVH8_CWBVH(ctx).Build(verts); h = sc.host
cam = R.camera(*scenes.SPONZA_CAMERAS[0], W, H, 1, 1)
noise = blue_noise()
d_verts = ctx.malloc(verts.nbytes); ctx.to_device(d_verts, verts)
wf = tb.Wavefront(ctx, W, H); wf.set_blue_noise(noise)
import sys as _s
SKY = tuple(float(x) for x in os.environ.get("WF_SKY", "0.7,0.7,1.2").split(","))
LCOL = tuple(float(x) for x in os.environ.get("WF_LCOL", "25,25,22").split(","))
for it in (1, 2):
    ref = ocl.wavefront(h.blob(0, np.uint32, 4), h.blob(1, np.uint32, 4), verts, noise, list(cam.eye), list(cam.p1), list(cam.p2), list(cam.p3), W, H, frames, it, os.environ.get('REFOCL_WF_PATCH', ''))
    for f in range(frames):
        wf.render(sc, d_verts, cam, (-22.0, 12.0, 2.0), LCOL, sky_lo=SKY, sky_hi=SKY, eps=1e-4, max_depth=it, seed=1000 + f, clear=(f == 0), stats=False,
                  light_size=(9.0, 5.0), one_diffuse_bounce=True, reference_letter=True, sample_index=f)
    mine = wf.read() / frames
    a, b = blocks(ref), blocks(mine)
    print(f"iterations {it}: ref mean {a.mean():.5f} mine {b.mean():.5f} ratio {b.mean()/a.mean():.4f} rel block diff {np.abs(a-b).mean()/a.mean():.4f}  ref.w mean {ref[...,3].mean():.3f} mine.w {mine[...,3].mean():.3f}")
    # ratio by brightness quartile of ref
    q = np.quantile(a.mean(-1), [0.25, 0.5, 0.75])
    lum = a.mean(-1)
    for lo, hi in ((0, q[0]), (q[0], q[1]), (q[1], q[2]), (q[2], 1e9)):
        m = (lum >= lo) & (lum < hi)
        print(f"    ref luminance [{lo:.3f},{hi:.3f}): ratio {b[m].mean()/a[m].mean():.4f}")
