"""device TLAS rebuild and camera trace against the number of instances"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tinybvh_amd as tb
from tinybvh_amd import rays as R, scenes
ctx = tb.Context(0)
dv, _ = scenes.get("dragon")
blas = tb.BVH4_GPU(ctx).Build(dv)
for side in (10, 20, 40, 64):
    g = np.stack(np.meshgrid(np.arange(side), np.arange(side), np.arange(side), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    ang = (np.arange(g.shape[0]) * 0.37).astype(np.float32)
    T = np.zeros((g.shape[0], 4, 4), np.float32)
    T[:, 0, 0] = np.cos(ang) * 0.7; T[:, 0, 2] = np.sin(ang) * 0.7; T[:, 1, 1] = 0.7; T[:, 2, 0] = -np.sin(ang) * 0.7; T[:, 2, 2] = np.cos(ang) * 0.7; T[:, 3, 3] = 1
    T[:, :3, 3] = g * 2.0
    inst = tb.make_instances(T, np.zeros(g.shape[0], np.uint32))
    tlas = tb.TLAS(ctx).Build(inst, [blas])
    ms = []
    for k in range(4):
        tlas.RebuildOnDevice(np.ascontiguousarray(inst["transform"])); ctx.synchronize(); ms.append(ctx.time_last_ms())
    ext = 2.0 * side
    W_, H_ = 3840, 2160
    cam = R.camera((-0.6 * ext, 0.8 * ext, -0.9 * ext), (0.62, -0.38, 0.68), W_, H_, 1, 1)
    d = ctx.malloc(W_ * H_ * 64); ctx.generate_primary(cam, d, 0, W_ * H_)
    tr = []
    for k in range(4):
        tlas.intersect_device_fresh(d, W_ * H_, 1e30); ctx.synchronize(); tr.append(ctx.time_last_ms())
    print(f"{g.shape[0]:7d} instances: device rebuild {np.median(ms[1:]):8.3f} ms   8.3 M camera rays {np.median(tr[1:]):7.3f} ms = {W_ * H_ / np.median(tr[1:]) / 1e3:7.1f} MRays/s", flush=True)
    ctx.free(d); tlas.free()
