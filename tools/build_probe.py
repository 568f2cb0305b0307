"""Device builds (tbvh_build_device: LBVH; tbvh_build_device_ploc: PLOC at several radii; both + collapse + encode) vs the host SAH
build: build time and the trace rate through each tree.
    python tools/build_probe.py [bistro|sponza] [side] [radii, e.g. 8,16,32]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tinybvh_amd as tb  # noqa: E402
from tinybvh_amd import rays as R  # noqa: E402
from tinybvh_amd import scenes  # noqa: E402



def cwbvh_sah(nodes16, tris16):
    """Surface-area expectation of a BVH8_CWBVH blob for random long rays: (node visits, triangle tests) per ray that hits the root box,
    and the deepest level.  nodes16: (5 n, 4) uint32 as downloaded."""
    raw = np.ascontiguousarray(nodes16).view(np.uint8).reshape(-1, 80)
    e = raw[:, 12:15].view(np.int8).astype(np.int32)
    scale = np.ldexp(np.ones_like(e, dtype=np.float64), e)                       # 2^e per axis
    meta = raw[:, 24:32]
    q = raw[:, 32:80].reshape(-1, 6, 8).astype(np.float64)                       # qlo x,y,z, qhi x,y,z
    ext = np.maximum(q[:, 3:6, :] - q[:, 0:3, :], 0.0) * scale[:, :, None]       # (n, 3, 8)
    area = ext[:, 0] * ext[:, 1] + ext[:, 1] * ext[:, 2] + ext[:, 2] * ext[:, 0]  # (n, 8)
    used = meta != 0
    internal = used & ((meta & 0x1f) >= 24)
    leaf = used & ~internal
    ntri = np.zeros(meta.shape, np.int64)
    for b in range(3):
        ntri += ((meta >> (5 + b)) & 1)
    ntri = np.where(leaf, ntri, 0)
    # root box: union of its children
    lo = (q[0, 0:3, :] * scale[0][:, None])[:, used[0]].min(1); hi = (q[0, 3:6, :] * scale[0][:, None])[:, used[0]].max(1)
    d = hi - lo
    a_root = d[0] * d[1] + d[1] * d[2] + d[2] * d[0]
    s_cost = 1.0 + float((area * internal).sum()) / a_root
    t_cost = float((area * ntri).sum()) / a_root
    # depth: children of node i are consecutive from childBase in slot order of the internal ones
    child_base = raw[:, 16:20].copy().view(np.uint32)[:, 0].astype(np.int64)
    depth = np.zeros(raw.shape[0], np.int32)
    level = np.array([0], np.int64); dmax = 0
    while level.size:
        cnt = internal[level].sum(1)
        nxt = np.concatenate([child_base[n] + np.arange(c) for n, c in zip(level, cnt) if c]) if cnt.sum() else np.array([], np.int64)
        dmax += 1
        level = nxt
        if dmax > 200: break
    return s_cost, t_cost, dmax, raw.shape[0]


name = sys.argv[1] if len(sys.argv) > 1 else "bistro"
side = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
verts, label = scenes.get(name)
ctx = tb.Context(0)
t0 = time.perf_counter(); host = tb.BVH8_CWBVH(ctx).Build(verts); ctx.synchronize(); t_host = time.perf_counter() - t0
for it in range(3):
    t0 = time.perf_counter(); dev = tb.BVH8_CWBVH(ctx).BuildOnDevice(verts); t_call = time.perf_counter() - t0
    ms = ctx.time_last_ms()
    print(f"{label}: device build {ms:.2f} ms on the GPU ({verts.shape[0] // 3 / ms / 1e3:.0f} Mtris/s), {t_call * 1e3:.1f} ms wall incl. uploading {verts.nbytes / 1e6:.0f} MB of vertices; "
          f"{dev.device_bytes / 1e6:.0f} MB  (host SAH build + encode + upload: {t_host * 1e3:.0f} ms, {host.device_bytes / 1e6:.0f} MB)", flush=True)
    if it < 2: dev.free()
radii = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "8,16,32").split(",")]
trees = [("host SAH tree", host), ("device LBVH tree", dev)]
for r in radii:
    for it in range(2):
        t0 = time.perf_counter(); pl = tb.BVH8_CWBVH(ctx).BuildOnDevice(verts, builder="ploc", radius=r); t_call = time.perf_counter() - t0
        ms = ctx.time_last_ms()
        if it == 0: pl.free()
    print(f"{label}: PLOC radius {r}: {ms:.2f} ms on the GPU ({verts.shape[0] // 3 / ms / 1e3:.0f} Mtris/s), {t_call * 1e3:.1f} ms wall; {pl.device_bytes / 1e6:.0f} MB", flush=True)
    trees.append((f"device PLOC tree, radius {r}", pl))
n = side * side
cams = scenes.SPONZA_CAMERAS if name == "sponza" else scenes.STREET_CAMERAS
cam = R.camera(*cams[0], side, side, 1, 1)
d = ctx.malloc(n * 64); d_b = ctx.malloc(n * 64); d_verts = ctx.malloc(verts.nbytes); ctx.to_device(d_verts, verts)
if os.environ.get('PROBE_REUPLOAD'):
    trees += [(nm + ' re-uploaded', tb.BVH8_CWBVH(ctx).Upload(*sc.download_blobs())) for nm, sc in trees[1:]]
if os.environ.get('PROBE_REVERSE'): trees = trees[::-1]
for nm, sc in trees:
    s_c, t_c, dmax, nn_ = cwbvh_sah(*sc.download_blobs())
    print(f"  {nm:42s}: {nn_} wide nodes, {dmax} levels; surface-area expectation per ray through the root box: {s_c:.2f} node visits, {t_c:.2f} triangle tests", flush=True)
    ts, tb_ = [], []
    for p in range(4):
        ctx.generate_primary(cam, d, 0, n); sc.intersect_device(d, n); ts.append(ctx.time_last_ms())
        ctx.generate_bounce(d_verts, d, d_b, n, 1); sc.intersect_device(d_b, n); tb_.append(ctx.time_last_ms())
    import ctypes as C
    sc.set_variant(59)
    st = (C.c_uint64 * 8)()
    tb.lib.tbvh_debug_stats(ctx._h, st, 1)
    sc.intersect_device(d_b, n)
    tb.lib.tbvh_debug_stats(ctx._h, st, 1)
    st2 = (C.c_uint64 * 8)()
    ctx.generate_primary(cam, d, 0, n)
    tb.lib.tbvh_debug_stats(ctx._h, st2, 1)
    sc.intersect_device(d, n)
    tb.lib.tbvh_debug_stats(ctx._h, st2, 1)
    sc.set_variant(0)
    print(f"  {nm:42s}: measured on the camera batch: {int(st2[2]) / n:.2f} node visits, {int(st2[4]) / n:.2f} triangle tests per ray, {int(st2[0]) * 64 / n:.1f} wave-iterations x 64 per ray, lanes active {int(st2[1]) / max(int(st2[0]), 1) / 64:.2f}")
    print(f"  {nm:42s}: measured on the bounce batch: {int(st[2]) / n:.2f} node visits, {int(st[4]) / n:.2f} triangle tests per ray, {int(st[0]) * 64 / n:.1f} wave-iterations x 64 per ray")
    print(f"  {nm:42s}: camera rays {n / np.mean(ts[1:]) / 1e3:.0f} MRays/s, bounce rays {n / np.mean(tb_[1:]) / 1e3:.0f} MRays/s")
ctx.close()
