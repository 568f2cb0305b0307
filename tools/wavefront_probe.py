"""Whole-frame wavefront path tracer on the Bistro stand-in: rays per stage and MRays/s."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tinybvh_amd as tb  # noqa: E402
from tinybvh_amd import rays as R  # noqa: E402
from tinybvh_amd import scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "bistro"
W = H = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
verts, label = scenes.get(name)
ctx = tb.Context(0)
sc = tb.BVH8_CWBVH(ctx).Build(verts)
d_verts = ctx.malloc(verts.nbytes); ctx.to_device(d_verts, verts)
cams = scenes.STREET_CAMERAS if name == "bistro" else scenes.SPONZA_CAMERAS
cam = R.camera(*cams[0], W, H, 1, 1)
wf = tb.Wavefront(ctx, W, H)
light = (0.0, 0.9 * float(verts[:, 1].max()), 0.0)
for f in range(int(os.environ.get("FRAMES", "4"))):
    st = wf.render(sc, d_verts, cam, light, (3000.0, 3000.0, 3000.0), max_depth=3, seed=f + 1)
    total = sum(st["extend_rays"]) + sum(st["shadow_rays"])
    print(f"frame {f}: {label}: extend {st['extend_rays']} shadow {st['shadow_rays']}  {st['frame_ms']:.2f} ms  -> {total / st['frame_ms'] / 1e3:.0f} MRays/s all stages", flush=True)
img = wf.read()
print("mean radiance", img[..., :3].mean())
