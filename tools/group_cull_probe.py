"""CPU analysis (the oracle's CWBVH mirror): how many node visits does ONE entry distance per node group save — the minimum over the group's interior
children, kept with the group on the stack, never refreshed (what fits in the 16 spare bits of the kernels' 8-byte stack word)?  Camera rays and
cosine-ish bounce rays of the bench scene.  profiles/r06_diffuse.txt, attempt 2."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import tinybvh_amd as tb  # noqa: E402
from tinybvh_amd import rays as R, scenes  # noqa: E402
from oracle_lib import Oracle, _p  # noqa: E402

verts, label = scenes.get(sys.argv[1] if len(sys.argv) > 1 else "bistro")
h = tb.HostBVH(verts, tb.LAYOUT_CWBVH)
nodes, tris = h.blob(0, np.uint32, 4), h.blob(1, np.uint32, 4)
orc = Oracle(tie_rule=1)
orc.lib.orc_cwbvh_group_cull.restype = C.c_uint64
cam = R.primary(R.camera(*scenes.cameras("bistro")[0], 192, 128, 1, 1))
w = orc.cwbvh_intersect(nodes, tris, cam.copy())
hit = w["t"] < 1e30
rng = np.random.default_rng(1)
batches = {"camera": cam}
src = w
for depth in (1, 2):
    hit = src["t"] < 1e30
    P = (src["O"] + src["D"] * src["t"][:, None])[hit]
    D = rng.normal(size=P.shape).astype(np.float32); D /= np.linalg.norm(D, axis=1, keepdims=True)
    b = tb.make_rays((P - 1e-3 * src["D"][hit]).astype(np.float32), D)
    batches[f"bounce {depth}"] = b
    src = orc.cwbvh_intersect(nodes, tris, b.copy())
print(label)
for name, rays in batches.items():
    base, c0 = orc.cwbvh_intersect(nodes, tris, rays.copy(), counts=True)
    r = np.ascontiguousarray(rays.copy()); c1 = np.zeros(2, np.uint64)
    orc.lib.orc_set_tie_rule(1)
    dropped = orc.lib.orc_cwbvh_group_cull(_p(nodes), _p(tris), _p(r), r.shape[0], r.strides[0], _p(c1))
    same = np.array_equal(base.view(np.uint8), r.view(np.uint8))
    n = rays.shape[0]
    print(f"  {name:9s} {n:6d} rays: node visits per ray {c0[0] / n:6.2f} -> {c1[0] / n:6.2f} ({c1[0] / c0[0] - 1:+.1%}), triangle tests {c0[1] / n:5.2f} -> {c1[1] / n:5.2f}; children dropped per ray {dropped / n:.2f}; records identical: {same}")
