"""Round 5: what ONE traversal per wave of 64 consecutive camera rays would visit: the union of the node visits and triangle tests of its rays (CPU, the oracle's
mirror), against one ray's own visits — the work model behind kernels_cwbvh_packet.hip."""
import sys, ctypes as C, numpy as np
sys.path[:0]=['/root/repo','/root/repo/tests','/root/repo/tools']
import tinybvh_amd as tb
from tinybvh_amd import rays as R, scenes
from oracle_lib import Oracle, _p
orc=Oracle(1)
def events(nodes,tris,batch):
    L=orc.lib
    L.orc_cwbvh_trace.restype=C.c_uint64
    L.orc_cwbvh_trace.argtypes=[C.c_void_p,C.c_void_p,C.c_void_p,C.c_uint64,C.c_uint32,C.c_void_p,C.c_uint64]
    cap=batch.shape[0]*1200; out=np.zeros(cap,np.uint32); r=batch.copy(); L.orc_set_tie_rule(1)
    nw=L.orc_cwbvh_trace(_p(nodes),_p(tris),_p(r),r.shape[0],r.strides[0],_p(out),cap); assert nw<cap
    return out[:nw], r
for name in ("bistro","street_rot","sponza"):
    verts,_=scenes.get(name)
    h=tb.HostBVH(verts,tb.LAYOUT_CWBVH); nodes,tris=h.blob(0,np.uint32,4),h.blob(1,np.uint32,4)
    side=4096 if name!="sponza" else 1024
    cam=R.camera(*scenes.cameras(name)[0],side,side,1,1)
    allr=R.primary(cam)   # full image in tile order
    n=allr.shape[0]
    rng=np.random.default_rng(3)
    chunks=rng.choice(n//64, 400, replace=False)
    sel=np.concatenate([np.arange(c*64,c*64+64) for c in chunks])
    ev,_=events(nodes,tris,allr[sel])
    sep=np.flatnonzero(ev==0xFFFFFFFF)
    starts=np.concatenate([[0],sep[:-1]+1])
    Sray=[];Tray=[];Sw=[];Tw=[]; oct_mixed=0
    D=allr[sel]["D"]
    for ci in range(len(chunks)):
        un=set(); ut=set()
        for k in range(64):
            a,b=starts[ci*64+k],sep[ci*64+k]
            e=ev[a:b]; nd=e[(e&0x80000000)!=0]&0x7FFFFFFF; tr=e[(e&0x80000000)==0]
            Sray.append(len(nd)); Tray.append(len(tr)); un.update(nd.tolist()); ut.update(tr.tolist())
        Sw.append(len(un)); Tw.append(len(ut))
        d=D[ci*64:ci*64+64]; o=(d[:,0]<0)*4+(d[:,1]<0)*2+(d[:,2]<0)
        oct_mixed+= int(len(set(o.tolist()))>1)
    print(f"{name}: {side}x{side} camera rays, 400 random 64-ray chunks: S per ray {np.mean(Sray):.1f}, union per chunk {np.mean(Sw):.1f} (x{np.mean(Sw)/np.mean(Sray):.2f}); T per ray {np.mean(Tray):.2f}, union of triangle tests per chunk {np.mean(Tw):.1f}; chunks with mixed octants {oct_mixed}/400")
