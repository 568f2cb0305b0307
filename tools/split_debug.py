"""Which rays differ between a kernel variant with split rays and one without (experiment build)?  Prints the hit records of both."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import tinybvh_amd as tb
from tinybvh_amd import rays as R, scenes
from ab_probe import make_batches

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="sponza"); ap.add_argument("--side", type=int, default=1024)
ap.add_argument("--layout", type=int, default=5); ap.add_argument("--a", type=int, default=12); ap.add_argument("--b", type=int, default=13)
ap.add_argument("--reps", type=int, default=4)
a = ap.parse_args()
verts, label = scenes.get(a.scene)
ctx = tb.Context(0)
sc = tb.LAYOUT_CLASSES[a.layout](ctx).Build(verts)
n = a.side * a.side
cams = scenes.STREET_CAMERAS if a.scene == "bistro" else scenes.SPONZA_CAMERAS
cam = R.camera(*cams[0], a.side, a.side, 1, 1)
d_prim, d_diff, d_shad = make_batches(ctx, sc, verts, cam, n)
for kind, d in (("primary", d_prim), ("diffuse", d_diff)):
    sc.set_variant(a.a); sc.intersect_device_fresh(d, n, 1e30)
    ref = np.zeros(n, tb.RAY_DTYPE); ctx.from_device(ref, d)
    for rep in range(a.reps):
        sc.set_variant(a.b); sc.intersect_device_fresh(d, n, 1e30)
        got = np.zeros(n, tb.RAY_DTYPE); ctx.from_device(got, d)
        bad = np.nonzero((got["t"] != ref["t"]) | (got["prim"] != ref["prim"]))[0]
        print(kind, "rep", rep, "differing rays:", len(bad))
        for i in bad[:6]:
            print("  ray", i, "O", ref["O"][i], "D", ref["D"][i], "\n     ref", ref["t"][i], ref["u"][i], ref["v"][i], ref["prim"][i], " got", got["t"][i], got["u"][i], got["v"][i], got["prim"][i],
                  " t bits", hex(ref["t"][i].view(np.uint32)), hex(got["t"][i].view(np.uint32)))
