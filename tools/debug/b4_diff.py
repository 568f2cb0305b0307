"""debug aid: the one ray of tests/test_full_size.py::test_bistro_16m_properties on which BVH8_CWBVH and BVH4_GPU (through its 8-wide copy) disagree"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import tinybvh_amd as tb
from tinybvh_amd import rays as R, scenes
from oracle_lib import Oracle
verts, label = scenes.get("bistro")
side = 4096; n = side * side
cam = R.camera(*scenes.STREET_CAMERAS[1], side, side, 1, 1)
ctx = tb.Context(0)
cw = tb.BVH8_CWBVH(ctx).Build(verts)
b4 = tb.BVH4_GPU(ctx).Build(verts)
d_verts = ctx.malloc(verts.nbytes); ctx.to_device(d_verts, verts)
d_p, d_b = ctx.malloc(n * 64), ctx.malloc(n * 64)
ctx.generate_primary(cam, d_p, 0, n); cw.intersect_device(d_p, n)
ctx.generate_bounce(d_verts, d_p, d_b, n, 99)
rays0 = np.zeros(n, tb.RAY_DTYPE); ctx.from_device(rays0, d_b)
def trace(sc):
    sc.intersect_device_fresh(d_b, n, 1e30); out = np.zeros(n, tb.RAY_DTYPE); ctx.from_device(out, d_b); return out
a = trace(cw); b = trace(b4); b4.set_variant(1); c = trace(b4); b4.set_variant(0)
bad = np.nonzero((a["t"] != b["t"]) | (a["prim"] != b["prim"]))[0]
print("cw vs b4(copy) differ on", bad.size, "rays; cw vs b4(native):", int(((a["t"] != c["t"]) | (a["prim"] != c["prim"])).sum()))
orc = Oracle(tie_rule=1)
h = cw.host
for i in bad[:5]:
    r = rays0[i:i + 1].copy(); r["t"] = 1e30
    w = orc.bvh2_intersect(h.bvh2_nodes(), h.bvh2_prim_idx(), verts, r)
    print(i, "O", rays0["O"][i], "D", rays0["D"][i])
    for name, x in (("cw", a), ("b4 copy", b), ("b4 native", c), ("oracle", w if True else None)):
        rec = x[i] if name != "oracle" else x[0]
        print(f"   {name:10s} t {float(rec['t']).hex()} ({rec['t']:.9g}) prim {rec['prim']} u {rec['u']:.7g} v {rec['v']:.7g}")
    for p in {int(a["prim"][i]), int(b["prim"][i])}:
        print("   tri", p, verts[3 * p:3 * p + 3, :3].tolist())
