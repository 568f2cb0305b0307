"""Device-resident wavefront path tracer: queue bookkeeping and a
depth-1 frame re-computed on the host with the oracle (same RNG, same shading formulas)."""
import numpy as np
import pytest

import tinybvh_amd as tb
from tinybvh_amd import rays as R
from tinybvh_amd import scenes

pytestmark = pytest.mark.gpu


def wang(s):
    s = (s ^ np.uint32(61)) ^ (s >> np.uint32(16)); s = s * np.uint32(9); s = s ^ (s >> np.uint32(4)); s = s * np.uint32(0x27d4eb2d)
    return s ^ (s >> np.uint32(15))


def xorshift(s):
    s = s ^ (s << np.uint32(13)); s = s ^ (s >> np.uint32(17)); s = s ^ (s << np.uint32(5))
    return s


def test_frame_against_host_recomputation(ctx, oracle):
    verts = scenes.atrium(40_000, seed=1)
    verts[0::3, 3] = np.frombuffer(np.array([0x00C08040], np.uint32).tobytes(), np.float32)[0]   # albedo RGB8 in v0.w
    W, H = 256, 128
    sc = tb.BVH8_CWBVH(ctx).Build(verts)
    d_verts = ctx.malloc(verts.nbytes); ctx.to_device(d_verts, verts)
    eye, view = scenes.SPONZA_CAMERAS[0]
    cam = R.camera(eye, view, W, H, 1, 1)
    wf = tb.Wavefront(ctx, W, H)
    light, lcol, lo, hi, eps, seed = (0.0, 24.0, 0.0), (300.0, 280.0, 260.0), (0.6, 0.7, 0.8), (0.2, 0.4, 0.9), 1e-3, 5
    st = wf.render(sc, d_verts, cam, light, lcol, lo, hi, eps, max_depth=1, seed=seed)
    img = wf.read()
    n = W * H
    assert st["extend_rays"][0] == n
    # ---- host recomputation of the same frame -------------------------------------------------------
    with np.errstate(over="ignore"):
        i = np.arange(n, dtype=np.uint32)
        s = wang(np.uint32(seed) * np.uint32(9781) + i * np.uint32(6271) + np.uint32(1)); s = np.where(s == 0, np.uint32(1), s)
        s = xorshift(s); r0 = (s >> np.uint32(8)).astype(np.float32) * np.float32(1 / 16777216)
        s = xorshift(s); r1 = (s >> np.uint32(8)).astype(np.float32) * np.float32(1 / 16777216)
    in_tile = i & 15; tile = i >> 4; tiles_x = W // 4
    px = (tile % tiles_x) * 4 + (in_tile & 3); py = (tile // tiles_x) * 4 + (in_tile >> 2)
    u = (px.astype(np.float32) + r0) / np.float32(W); v = (py.astype(np.float32) + r1) / np.float32(H)
    e = np.array(cam.eye, np.float32); p1 = np.array(cam.p1, np.float32); p2 = np.array(cam.p2, np.float32); p3 = np.array(cam.p3, np.float32)
    P = p1 + u[:, None] * (p2 - p1) + v[:, None] * (p3 - p1)
    rays = tb.make_rays(np.broadcast_to(e, P.shape), P - e)
    h = sc.host
    hits = oracle.bvh2_intersect(h.bvh2_nodes(), h.bvh2_prim_idx(), verts, rays)
    hit = hits["t"] < 1e30
    want = np.zeros((n, 3), np.float64)
    k = 0.5 * (rays["D"][:, 1] + 1.0)
    sky = np.array(lo)[None, :] + k[:, None] * (np.array(hi) - np.array(lo))[None, :]
    want[~hit] = sky[~hit]
    tri = verts.reshape(-1, 3, 4)
    pr = hits["prim"][hit]
    v0, v1, v2 = tri[pr, 0, :3], tri[pr, 1, :3], tri[pr, 2, :3]
    N = np.cross(v1 - v0, v2 - v0); N /= np.linalg.norm(N, axis=1, keepdims=True)
    D = rays["D"][hit]
    N = np.where(((N * D).sum(1) > 0)[:, None], -N, N)
    I = rays["O"][hit] + hits["t"][hit][:, None] * D
    L = np.array(light)[None, :] - I; dist = np.linalg.norm(L, axis=1); L = L / dist[:, None]
    ndl = (N * L).sum(1)
    sh = tb.make_rays(I + L * eps, L, (dist - 2 * eps).astype(np.float32))
    occ = oracle.bvh2_occluded(h.bvh2_nodes(), h.bvh2_prim_idx(), verts, sh).astype(bool)
    albedo = np.array([0xC0, 0x80, 0x40], np.float64) * 0.00392
    g = ndl / dist ** 2 * 0.31830988
    c = albedo[None, :] * np.array(lcol)[None, :] * g[:, None]
    c[(ndl <= 0) | occ | (dist <= 2 * eps)] = 0
    want[hit] = c
    pix = (py * W + px).astype(np.int64)
    ref = np.zeros((n, 3)); np.add.at(ref, pix, want)
    got = img.reshape(-1, 4)[:, :3].astype(np.float64)
    # same pixels lit, same values up to float rounding of the shading math (sqrt / division order);
    # a few pixels may differ when a shadow ray grazes geometry
    bad = np.abs(got - ref).max(1) > 1e-3 * (1 + np.abs(ref).max(1))
    assert bad.sum() <= n // 2000, int(bad.sum())
    assert abs(got.mean() - ref.mean()) < 1e-3 * ref.mean()
    assert st["shadow_rays"][0] == int(((ndl > 0) & (dist > 2 * eps)).sum()) or abs(st["shadow_rays"][0] - int((ndl > 0).sum())) < 20
    wf.close(); ctx.free(d_verts)


def test_multi_bounce_bookkeeping(ctx):
    verts = scenes.atrium(40_000, seed=1)
    W, H = 512, 256
    sc = tb.BVH8_CWBVH(ctx).Build(verts)
    d_verts = ctx.malloc(verts.nbytes); ctx.to_device(d_verts, verts)
    eye, view = scenes.SPONZA_CAMERAS[1]
    cam = R.camera(eye, view, W, H, 1, 1)
    wf = tb.Wavefront(ctx, W, H)
    st = wf.render(sc, d_verts, cam, (0.0, 24.0, 0.0), (300.0, 300.0, 300.0), max_depth=3, seed=3)
    e, s = st["extend_rays"], st["shadow_rays"]
    assert e[0] == W * H and e[0] >= e[1] >= e[2] > 0          # only hit paths continue
    assert all(s[d] <= e[d] for d in range(3)) and s[0] > 0    # at most one shadow ray per live path
    a = wf.read()
    assert np.isfinite(a).all() and a[..., :3].min() >= 0 and a[..., :3].mean() > 0.01
    # accumulation over frames: a second frame without clearing roughly doubles the image
    wf.render(sc, d_verts, cam, (0.0, 24.0, 0.0), (300.0, 300.0, 300.0), max_depth=3, seed=4, clear=False)
    b = wf.read()
    assert 1.8 < b[..., :3].mean() / a[..., :3].mean() < 2.2
    wf.close(); ctx.free(d_verts)
