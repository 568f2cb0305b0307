"""One frame split into horizontal BANDS rendered by several tbvh_wavefront objects on several contexts of ONE device (tbvh_wavefront_set_band):
the stages of a band are dependent launches, each half tail; the bands' chains are independent, so one band's launches fill the other's tails.
Compared with one wavefront for the whole frame and with two whole frames in flight (tools/wavefront_two_lanes.py).  The image is the same
for any number of bands (pixel and ray indices in the seeds are global: tests/test_sharded.py)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinybvh_amd as tb
from tinybvh_amd import rays as R, scenes
name = sys.argv[1] if len(sys.argv) > 1 else "sponza"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1280, 720)
verts, label = scenes.get(name)
cams = scenes.SPONZA_CAMERAS if name == "sponza" else scenes.STREET_CAMERAS
cam = R.camera(*cams[0], W, H, 1, 1)
host = tb.HostBVH(verts, 10)
light = ((-22.0, 12.0, 2.0), (25.0, 25.0, 22.0)) if name == "sponza" else ((40.0, 60.0, 20.0), (9000.0, 9000.0, 8000.0))
kw = dict(sky_lo=(0.7, 0.7, 1.2), sky_hi=(0.7, 0.7, 1.2), eps=1e-4, max_depth=3, stats=False, light_size=(9.0, 5.0), one_diffuse_bounce=True)


def make(nb):
    lanes = []
    rows = [H // nb // 4 * 4] * nb
    rows[-1] = H - sum(rows[:-1])
    first = 0
    for i in range(nb):
        ctx = tb.Context(0); sc = tb.BVH8_CWBVH(ctx); sc.host = host; sc.Upload(host.blob(0, np.uint32, 4), host.blob(1, np.uint32, 4))
        dv = ctx.malloc(verts.nbytes); ctx.to_device(dv, verts)
        wf = tb.Wavefront(ctx, W, rows[i])
        if nb > 1: wf.set_band(first, H)
        first += rows[i]
        lanes.append((ctx, sc, dv, wf))
    return lanes


def run(lanes, frames):
    for l in lanes: l[0].synchronize()
    t0 = time.perf_counter()
    for f in range(frames):
        for ctx, sc, dv, wf in lanes:
            wf.render(sc, dv, cam, light[0], light[1], seed=1000 + f, clear=(f == 0), **kw)
    for l in lanes: l[0].synchronize()
    return (time.perf_counter() - t0) / frames * 1e3


ref = None
for nb in (1, 2, 3, 4):
    lanes = make(nb)
    run(lanes, 10)
    ms = run(lanes, 200)
    img = np.concatenate([l[3].read() for l in lanes], 0)
    if ref is None: ref = img
    print(f"{label[:24]} {W} x {H}, 3 bounces, 200 frames: {nb} band(s) on {nb} context(s): {ms:.3f} ms per frame; image identical to one band: {bool(np.array_equal(img, ref))}", flush=True)
    for l in lanes: l[0].close()
