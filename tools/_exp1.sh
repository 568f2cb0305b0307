cd $GRAFT_REPO_ROOT
python - <<'PY'
import numpy as np, tinybvh_amd as tb
from tinybvh_amd import rays as R, scenes
verts,_=scenes.get("sponza")
ctx=tb.Context(0)
sc=tb.BVH8_CWBVH(ctx).Build(verts)
cam=R.camera(*scenes.SPONZA_CAMERAS[0],512,512,1,1)
rays=R.primary(cam)
a=sc.Intersect(rays.copy()); sc.set_variant(49); b=sc.Intersect(rays.copy())
print("variant 49 vs default identical:", np.array_equal(a.view(np.uint8), b.view(np.uint8)), int((a["t"]<1e30).sum()))
oa=sc.IsOccluded(rays.copy()); sc.set_variant(0); ob=sc.IsOccluded(rays.copy()); print("occluded identical:", np.array_equal(oa,ob))
PY
for v in 0 49 0 49; do
  echo "cwbvh sponza variant $v"; timeout 200 python tools/perf_probe.py --scene sponza --layouts 9 --variant $v --passes 15 2>&1 | grep layout
done
for v in 0 49 0 49; do
  echo "cwbvh bistro variant $v"; timeout 200 python tools/perf_probe.py --scene bistro --width 4096 --height 4096 --layouts 9 --variant $v --passes 4 2>&1 | grep layout
done
