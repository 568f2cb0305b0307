"""Host-side (numpy) ray generators: the ray batches of tiny_bvh_speedtest.cpp, for tests and
small runs.  The benchmark generates the same batches on the device (kernels_raygen.hip).
"""
from __future__ import annotations

import numpy as np

from . import BVH_FAR, RAY_DTYPE, Camera, make_rays
from .scenes import view_pyramid


def camera(eye, view, width: int, height: int, spp_x: int = 4, spp_y: int = 4) -> Camera:
    e, p1, p2, p3 = view_pyramid(eye, view)
    cam = Camera()
    cam.eye[:] = [float(x) for x in e]
    cam.p1[:] = [float(x) for x in p1]
    cam.p2[:] = [float(x) for x in p2]
    cam.p3[:] = [float(x) for x in p3]
    cam.width, cam.height, cam.spp_x, cam.spp_y = width, height, spp_x, spp_y
    return cam


def primary(cam: Camera, first: int = 0, n: int | None = None) -> np.ndarray:
    """Primary rays in the speedtest's order: 4x4-pixel tiles, spp samples per pixel
    (tiny_bvh_speedtest.cpp:526-549)."""
    spp = cam.spp_x * cam.spp_y
    total = cam.width * cam.height * spp
    if n is None:
        n = total - first
    i = np.arange(first, first + n, dtype=np.int64)
    s = i % spp
    pix = i // spp
    in_tile = pix & 15
    tile = pix >> 4
    tiles_x = cam.width // 4
    px = (tile % tiles_x) * 4 + (in_tile & 3)
    py = (tile // tiles_x) * 4 + (in_tile >> 2)
    u = ((px * cam.spp_x + (s % cam.spp_x)).astype(np.float32) / np.float32(cam.width * cam.spp_x)).astype(np.float32)
    v = ((py * cam.spp_y + (s // cam.spp_x)).astype(np.float32) / np.float32(cam.height * cam.spp_y)).astype(np.float32)
    eye = np.array(cam.eye, np.float32); p1 = np.array(cam.p1, np.float32); p2 = np.array(cam.p2, np.float32); p3 = np.array(cam.p3, np.float32)
    P = p1[None, :] + u[:, None] * (p2 - p1)[None, :] + v[:, None] * (p3 - p1)[None, :]
    return make_rays(np.broadcast_to(eye, P.shape), P - eye[None, :])


def random_rays(n: int, lo, hi, seed: int = 11, tmax=BVH_FAR) -> np.ndarray:
    """Incoherent rays: origins uniform in [lo, hi], directions uniform on the sphere."""
    rng = np.random.default_rng(seed)
    lo = np.asarray(lo, np.float32); hi = np.asarray(hi, np.float32)
    O = lo + rng.random((n, 3), dtype=np.float32) * (hi - lo)
    D = rng.normal(size=(n, 3)).astype(np.float32)
    return make_rays(O, D, tmax)


def bounce(rays: np.ndarray, verts: np.ndarray, seed: int = 5) -> np.ndarray:
    """One diffuse bounce from traced rays (tiny_bvh_speedtest.cpp:561-587)."""
    rng = np.random.default_rng(seed)
    n = rays.shape[0]
    R = (rng.random((n, 3), dtype=np.float32) - np.float32(0.5))
    R /= np.maximum(np.linalg.norm(R, axis=1, keepdims=True), 1e-12).astype(np.float32)
    hit = rays["t"] < BVH_FAR
    O = rays["O"]; D = rays["D"]
    I = np.where(hit[:, None], O + rays["t"][:, None] * D, O + np.float32(20) * D).astype(np.float32)
    tri = verts.reshape(-1, 3, 4)[:, :, :3]
    p = np.where(hit, rays["prim"], 0)
    N = np.cross(tri[p, 1] - tri[p, 0], tri[p, 2] - tri[p, 0])
    N /= np.maximum(np.linalg.norm(N, axis=1, keepdims=True), 1e-20)
    N = np.where(((N * D).sum(1) > 0)[:, None], -N, N)
    flip = hit & ((N * R).sum(1) < 0)
    R = np.where(flip[:, None], -R, R).astype(np.float32)
    return make_rays(I + np.float32(0.001) * R, R)


def shadow(rays: np.ndarray, light, eps: float) -> np.ndarray:
    """Shadow rays toward a point light (tiny_bvh_speedtest.cpp:851-865)."""
    t = np.minimum(np.float32(1000), rays["t"])
    I = (rays["O"] + t[:, None] * rays["D"]).astype(np.float32)
    L = (np.asarray(light, np.float32)[None, :] - I).astype(np.float32)
    dist = np.linalg.norm(L, axis=1).astype(np.float32)
    Ld = L / np.maximum(dist, 1e-20)[:, None]
    return make_rays(I + Ld * np.float32(eps), Ld, (dist - np.float32(eps)).astype(np.float32))
