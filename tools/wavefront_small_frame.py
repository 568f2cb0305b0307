"""Frame time of the path tracer at the reference demo resolution (1280 x 720, 3 bounces): wall clock of back-to-back frames against the GPU events of
single frames (are the ~13 launches of a frame launch-bound?  no: 0.91 ms either way)."""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinybvh_amd as tb
from tinybvh_amd import rays as R, scenes
verts,_=scenes.get("sponza")
ctx=tb.Context(0); sc=tb.BVH8_CWBVH(ctx).Build(verts)
W,H=1280,720
cam=R.camera(*scenes.SPONZA_CAMERAS[0],W,H,1,1)
d_verts=ctx.malloc(verts.nbytes); ctx.to_device(d_verts,verts)
wf=tb.Wavefront(ctx,W,H)
def frame(f, stats):
    return wf.render(sc,d_verts,cam,(-22.0,12.0,2.0),(25.0,25.0,22.0),sky_lo=(0.7,0.7,1.2),sky_hi=(0.7,0.7,1.2),eps=1e-4,max_depth=3,seed=1000+f,clear=(f==0),stats=stats,light_size=(9.0,5.0),one_diffuse_bounce=True)
for f in range(5): frame(f, False)
ctx.synchronize()
t0=time.perf_counter()
for f in range(100): frame(f, False)
ctx.synchronize()
wall=(time.perf_counter()-t0)/100*1e3
st=[frame(f, True) for f in range(10)]
ev=np.mean([s["frame_ms"] for s in st]) if isinstance(st[0], dict) else float('nan')
print(f"1280x720, 3 bounces: wall {wall:.3f} ms per frame (100 frames back to back), GPU events {ev:.3f} ms per frame")
