"""tinybvh_amd — MI355X-native batched ray traversal behind tinybvh's GPU-layout API.

Thin Python mirror of the reference's host interface for this one path (class and method
names follow tiny_bvh.h: ``BVH_GPU`` / ``BVH4_GPU`` / ``BVH8_CWBVH`` with ``Build``,
``Intersect``, ``IsOccluded``), over the C ABI in ``include/tinybvh_amd.h``.  All compute
is in the HIP library; this package holds no traversal code and no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import os

import numpy as np

from . import _capi
from ._capi import BuildParams, Camera, TbvhError, check, lib

LAYOUT_BVH2_WALD = 1
LAYOUT_BVH_GPU = 5
LAYOUT_BVH4_GPU = 8
LAYOUT_CWBVH = 10
# wavefront materials: v0.w of a triangle's first vertex = type << 24 | 0xRRGGBB (wavefront.cl:12-13, 160)
MATERIAL_DIFFUSE, MATERIAL_LIGHT, MATERIAL_SPECULAR = 0, 1, 2

BVH_FAR = np.float32(1e30)

# first 64 bytes of tinybvh::Ray (tiny_bvh.h:689-709) == device struct Ray (traverse.cl:11-17)
RAY_DTYPE = np.dtype([
    ("O", "<f4", 3), ("mask", "<u4"),
    ("D", "<f4", 3), ("instIdx", "<u4"),
    ("rD", "<f4", 3), ("inst", "<u4"),
    ("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("prim", "<u4"),
])
assert RAY_DTYPE.itemsize == 64


def safercp(x: np.ndarray) -> np.ndarray:
    """tinybvh_safercp (tiny_bvh.h:442): 1/x, or +-1e30 when |x| <= 1e-12."""
    x = np.asarray(x, dtype=np.float32)
    big = np.abs(x) > np.float32(1e-12)
    with np.errstate(divide="ignore"):
        r = np.where(big, np.float32(1.0) / np.where(big, x, np.float32(1.0)), np.where(x >= 0, BVH_FAR, -BVH_FAR))
    return r.astype(np.float32)


def make_rays(O: np.ndarray, D: np.ndarray, tmax=BVH_FAR, normalize: bool = True) -> np.ndarray:
    """Build ray records the way the tinybvh::Ray constructor does (tiny_bvh.h:695-703)."""
    O = np.ascontiguousarray(O, dtype=np.float32).reshape(-1, 3)
    D = np.ascontiguousarray(D, dtype=np.float32).reshape(-1, 3)
    if normalize:
        l = np.sqrt((D * D).sum(axis=1, dtype=np.float32)).astype(np.float32)
        rl = np.where(l == 0, np.float32(0), np.float32(1) / np.where(l == 0, np.float32(1), l)).astype(np.float32)
        D = (D * rl[:, None]).astype(np.float32)
    rays = np.zeros(O.shape[0], dtype=RAY_DTYPE)
    rays["O"] = O
    rays["D"] = D
    rays["rD"] = safercp(D)
    rays["mask"] = 0xFFFF
    rays["t"] = tmax
    return rays


def _ptr(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


class Context:
    """One HIP device (replaces tinyocl's process-global InitCL, tiny_ocl.h:945-1139)."""

    def __init__(self, device: int = 0):
        h = C.c_void_p()
        check(lib.tbvh_init(device, C.byref(h)), "tbvh_init")
        self._h = h
        self.device = device

    def close(self):
        if self._h:
            lib.tbvh_shutdown(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        check(lib.tbvh_synchronize(self._h), "tbvh_synchronize")

    def set_stream(self, hip_stream: Optional[int]):
        check(lib.tbvh_set_stream(self._h, C.c_void_p(hip_stream or 0)), "tbvh_set_stream")

    def set_timing(self, enabled: bool):
        """Per-operation HIP-event timing on / off (tbvh_set_timing): off saves two event records per query."""
        check(lib.tbvh_set_timing(self._h, 1 if enabled else 0), "tbvh_set_timing")

    def copy_bandwidth_gbps(self, nbytes: int = 1 << 30, reps: int = 3) -> float:
        """Measured streaming-copy bandwidth of this device (tbvh_measure_copy_bandwidth), GB/s read + written."""
        g = C.c_double(0)
        check(lib.tbvh_measure_copy_bandwidth(self._h, nbytes, reps, C.byref(g)), "tbvh_measure_copy_bandwidth")
        return float(g.value)

    def read_bandwidth_gbps(self, nbytes: int = 1 << 30, reps: int = 3) -> float:
        """Measured read-only streaming bandwidth of this device (tbvh_measure_read_bandwidth), GB/s."""
        g = C.c_double(0)
        check(lib.tbvh_measure_read_bandwidth(self._h, nbytes, reps, C.byref(g)), "tbvh_measure_read_bandwidth")
        return float(g.value)

    def valu_issue_ginstr(self, reps: int = 3) -> float:
        """Measured VALU issue ceiling (1e9 wave64 instructions per second, whole chip) for the CWBVH node test's mix (tbvh_measure_valu_issue)."""
        g = C.c_double(0)
        check(lib.tbvh_measure_valu_issue(self._h, reps, C.byref(g)), "tbvh_measure_valu_issue")
        return float(g.value)

    def link_bandwidth_gbps(self, nbytes: int = 1 << 28, reps: int = 3):
        """Measured host link rates (tbvh_measure_link_bandwidth): (host-to-device, device-to-host) GB/s of a pinned hipMemcpyAsync."""
        up, down = C.c_double(0), C.c_double(0)
        check(lib.tbvh_measure_link_bandwidth(self._h, nbytes, reps, C.byref(up), C.byref(down)), "tbvh_measure_link_bandwidth")
        return float(up.value), float(down.value)

    def pinned_array(self, shape, dtype) -> np.ndarray:
        """A numpy array in page-locked host memory of the library's (tbvh_pinned_malloc): a packed RAY_DTYPE array that lives there goes up by DMA
        without the packing pass.  Give it back with pinned_free(array) (or it goes with the context)."""
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) * dt.itemsize
        p = C.c_void_p()
        check(lib.tbvh_pinned_malloc(self._h, max(n, 1), C.byref(p)), "tbvh_pinned_malloc")
        buf = (C.c_char * max(n, 1)).from_address(p.value)
        a = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[a.ctypes.data] = p.value
        return a

    def pinned_free(self, a: np.ndarray):
        p = getattr(self, "_pinned", {}).pop(a.ctypes.data, None)
        check(lib.tbvh_pinned_free(self._h, C.c_void_p(p if p is not None else a.ctypes.data)), "tbvh_pinned_free")

    def time_last_ms(self) -> float:
        return float(lib.tbvh_time_last_ms(self._h))

    def time_history(self, k: int):
        """HIP-event durations (ms) of the last k timed operations on this context, oldest first (tbvh_time_history): what a
        caller that enqueues its launches back to back reads ONCE instead of synchronizing after every launch."""
        buf = (C.c_float * max(int(k), 1))()
        cnt = C.c_uint32(0)
        check(lib.tbvh_time_history(self._h, buf, int(k), C.byref(cnt)), "tbvh_time_history")
        return [float(buf[i]) for i in range(cnt.value)]

    def set_debug_flags(self, flags: int):
        check(lib.tbvh_debug_set_flags(self._h, int(flags)), "tbvh_debug_set_flags")

    def last_probe(self):
        """(agreeing pairs, pairs, verdict) of the coherence probe of the most recent query: verdict 0 = no probe ran,
        1 = incoherent (strict schedule), 2 = coherent (tbvh_debug_last_probe)."""
        out = (C.c_uint32 * 3)()
        check(lib.tbvh_debug_last_probe(self._h, out), "tbvh_debug_last_probe")
        return int(out[0]), int(out[1]), int(out[2])

    # device buffers (replace tinyocl::Buffer for resident rays)
    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        check(lib.tbvh_device_malloc(self._h, nbytes, C.byref(p)), "tbvh_device_malloc")
        return p.value

    def free(self, dptr: int):
        check(lib.tbvh_device_free(self._h, C.c_void_p(dptr)), "tbvh_device_free")

    def to_device(self, dptr: int, a: np.ndarray):
        a = np.ascontiguousarray(a)
        check(lib.tbvh_copy_to_device(self._h, C.c_void_p(dptr), _ptr(a), a.nbytes), "tbvh_copy_to_device")

    def from_device(self, a: np.ndarray, dptr: int):
        assert a.flags["C_CONTIGUOUS"]
        check(lib.tbvh_copy_from_device(self._h, _ptr(a), C.c_void_p(dptr), a.nbytes), "tbvh_copy_from_device")

    def reset_hits(self, d_rays: int, n: int, tmax: float = 1e30):
        check(lib.tbvh_reset_hits_device(self._h, C.c_void_p(d_rays), n, float(tmax)), "tbvh_reset_hits_device")

    # ray generators
    def generate_primary(self, cam: Camera, d_rays: int, first: int, n: int):
        check(lib.tbvh_generate_primary_device(self._h, C.byref(cam), C.c_void_p(d_rays), first, n), "tbvh_generate_primary_device")

    def generate_bounce(self, d_verts: int, d_in: int, d_out: int, n: int, seed: int):
        check(lib.tbvh_generate_bounce_device(self._h, C.c_void_p(d_verts), C.c_void_p(d_in), C.c_void_p(d_out), n, seed), "tbvh_generate_bounce_device")

    def bin_rays(self, d_in: int, d_out: int, n: int, bounds6, cell_bits: int = 5, flags: int = 1, d_perm: int = 0):
        """Counting sort of a resident batch by (Morton code of the origin's cell, direction octant): tbvh_bin_rays_device."""
        b = (C.c_float * 6)(*[float(x) for x in bounds6])
        check(lib.tbvh_bin_rays_device(self._h, C.c_void_p(d_in), C.c_void_p(d_out), n, b, cell_bits, flags, C.c_void_p(d_perm or 0)), "tbvh_bin_rays_device")

    def generate_shadow(self, d_in: int, d_out: int, n: int, light, eps: float):
        l = (C.c_float * 3)(*[float(x) for x in light])
        check(lib.tbvh_generate_shadow_device(self._h, C.c_void_p(d_in), C.c_void_p(d_out), n, l, float(eps)), "tbvh_generate_shadow_device")


class HostBVH:
    """Blobs built on the host by the library's own builder (tbvh_host_build)."""

    def __init__(self, verts: np.ndarray, layout: int, bins: int = 0, max_leaf_tris: int = 0, threads: int = 0,
                 optimal_collapse: bool = False, c_prim: float = 0.0, greedy_collapse: bool = False, split_budget: Optional[float] = None):
        verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 4)
        assert verts.shape[0] % 3 == 0
        self.verts = verts
        self.n_tris = verts.shape[0] // 3
        self.layout = layout
        flags = (2 if optimal_collapse else 0) | (4 if greedy_collapse else 0) | (int(round(c_prim * 100)) << 8)
        if split_budget is not None:   # None: the layout's default (BVH8_CWBVH: 30 % extra references; the others: whole triangles)
            if split_budget > 0:       # TBVH_BUILD_SPLIT_TRIANGLES, budget in per cent of the triangle count
                flags |= 8 | (min(max(int(round(split_budget * 100)), 1), 255) << 24)
            else:
                flags |= 16            # TBVH_BUILD_WHOLE_TRIANGLES
        bp = BuildParams(bins, max_leaf_tris, threads, flags)
        h = C.c_void_p()
        check(lib.tbvh_host_build(_ptr(verts), self.n_tris, layout, C.byref(bp), C.byref(h)), "tbvh_host_build")
        self._h = h

    @classmethod
    def from_cwbvh_file(cls, path: str, expected_tris: int = 0) -> "HostBVH":
        """The blobs of a BVH8_CWBVH::Save file (tbvh_cwbvh_file_read); no BVH2 and no vertices come with it."""
        self = cls.__new__(cls)
        h = C.c_void_p(); n = C.c_uint64(0)
        self._h = None
        check(lib.tbvh_cwbvh_file_read(os.fsencode(path), expected_tris, C.byref(h), C.byref(n)), "tbvh_cwbvh_file_read")
        self._h = h
        self.verts = None
        self.n_tris = int(n.value)
        self.layout = LAYOUT_CWBVH
        return self

    def save_cwbvh(self, path: str) -> None:
        """Write this BVH8_CWBVH's blobs as a file BVH8_CWBVH::Load accepts (tbvh_cwbvh_file_write; tiny_bvh.h:5786-5795)."""
        assert self.layout == LAYOUT_CWBVH
        nodes, tris = self.blob(0, np.uint32, 4), self.blob(1, np.uint32, 4)
        check(lib.tbvh_cwbvh_file_write(os.fsencode(path), _ptr(nodes), nodes.shape[0], _ptr(tris), tris.shape[0], self.n_tris or tris.shape[0] // 3, None), "tbvh_cwbvh_file_write")

    def blob(self, which: int, dtype, width: int) -> np.ndarray:
        """Zero-copy numpy view of blob `which` (the view keeps this object alive)."""
        p = lib.tbvh_host_blob(self._h, which)
        n = lib.tbvh_host_blob_count(self._h, which)
        if not p or n == 0:
            return np.zeros((0, width), dtype=dtype)
        nbytes = n * np.dtype(dtype).itemsize * width
        buf = (C.c_char * nbytes).from_address(p)
        buf._owner = self   # the view keeps this object (and so the native blob) alive
        a = np.frombuffer(buf, dtype=dtype).reshape(n, width)
        a.flags.writeable = False
        return a

    # the Wald BVH2 every layout was encoded from (for the oracle)
    def bvh2_nodes(self) -> np.ndarray:
        return self.blob(0 if self.layout == LAYOUT_BVH2_WALD else 2, np.uint32, 8)

    def bvh2_prim_idx(self) -> np.ndarray:
        return self.blob(1 if self.layout in (LAYOUT_BVH2_WALD, LAYOUT_BVH_GPU) else 3, np.uint32, 1).reshape(-1)

    def __del__(self):
        try:
            if self._h:
                lib.tbvh_host_free(self._h)
                self._h = None
        except Exception:
            pass


class _Scene:
    """An uploaded layout.  Intersect / IsOccluded mirror X::Intersect(Ray&) /
    X::IsOccluded(const Ray&) of the reference, but over whole ray arrays."""

    layout = 0

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self._h = C.c_void_p()

    def free(self):
        if self._h and self.ctx._h:
            lib.tbvh_free_scene(self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    @property
    def device_bytes(self) -> int:
        return int(lib.tbvh_scene_device_bytes(self._h))

    def Refit(self, verts, on_device: bool = False):
        """Refit this BLAS on the device to moved vertices (tbvh_refit; BVH::Refit, tiny_bvh.h:3055-3093):
        verts is the (3 n_tris, 4) float32 vertex array (host) or a device pointer with n_tris = on_device."""
        if on_device:
            ptr, n_tris = C.c_void_p(int(verts[0])), int(verts[1])
        else:
            verts = np.ascontiguousarray(verts, np.float32)
            assert verts.ndim == 2 and verts.shape[1] == 4 and verts.shape[0] % 3 == 0
            ptr, n_tris = _ptr(verts), verts.shape[0] // 3
        check(lib.tbvh_refit(self._h, ptr, n_tris, 1 if on_device else 0), "tbvh_refit")
        return self

    def SetOpacityMicroMaps(self, map_data, N: int):
        """BVHBase::SetOpacityMicroMaps (tiny_bvh.h:826): map_data = uint32 array of n_tris * ceil(N*N/32) words, or None to clear."""
        if map_data is None or N == 0:
            check(lib.tbvh_set_opacity_micromaps(self._h, None, 0, 0, 0), "tbvh_set_opacity_micromaps")
            return self
        m = np.ascontiguousarray(map_data, np.uint32).reshape(-1)
        wpt = (N * N + 31) // 32
        assert m.size % wpt == 0
        check(lib.tbvh_set_opacity_micromaps(self._h, _ptr(m), N, m.size // wpt, 0), "tbvh_set_opacity_micromaps")
        return self

    def download_blobs(self):
        """(nodes, triangle records) as (n, 4) uint32 arrays of 16-byte blocks, read back from the device."""
        out = []
        for which in (0, 1):
            nb = C.c_uint64(0)
            check(lib.tbvh_scene_download(self._h, which, None, 0, C.byref(nb)), "tbvh_scene_download")
            a = np.zeros((nb.value // 16, 4), np.uint32)
            if nb.value:
                check(lib.tbvh_scene_download(self._h, which, _ptr(a), nb.value, C.byref(nb)), "tbvh_scene_download")
            out.append(a)
        return tuple(out)

    def coherent_schedule(self, anyhit: bool = False):
        """(decision, samples of the deferred schedule, samples of the strict one, 1000 x strict / deferred time per ray): tbvh_debug_coherent_schedule."""
        out = (C.c_uint32 * 4)()
        check(lib.tbvh_debug_coherent_schedule(self._h, 1 if anyhit else 0, out), "tbvh_debug_coherent_schedule")
        return tuple(int(x) for x in out)

    def schedule_hint(self):
        """tbvh_scene_get_schedule_hint: {"closest_hit": [c0, c1, c2], "any_hit": [...]} per batch-size class (< 6 M, < 12 M, more rays); 0 undecided, 1 deferred + gated, 2 strict."""
        b = (C.c_uint8 * 8)()
        check(lib.tbvh_scene_get_schedule_hint(self._h, C.cast(b, C.c_void_p)), "tbvh_scene_get_schedule_hint")
        return {"closest_hit": [int(b[0]), int(b[1]), int(b[2])], "any_hit": [int(b[3]), int(b[4]), int(b[5])], "small_batches": [int(b[6]), int(b[7])]}   # small_batches: 768 k .. 1.5 M rays on a scene under 48 MB (closest-hit, any-hit)

    def set_schedule_hint(self, hint) -> None:
        """tbvh_scene_set_schedule_hint: pins the non-zero entries of a dict as schedule_hint() returns it; zero entries go back to measuring."""
        b = (C.c_uint8 * 8)(*(list(hint["closest_hit"]) + list(hint["any_hit"]) + list(hint.get("small_batches", [0, 0]))))
        check(lib.tbvh_scene_set_schedule_hint(self._h, C.cast(b, C.c_void_p)), "tbvh_scene_set_schedule_hint")

    def set_variant(self, v: int):
        check(lib.tbvh_set_variant(self._h, v), "tbvh_set_variant")

    def set_hybrid(self, packed_nodes: int):
        """BVH8_CWBVH: traverse a priority-ordered copy of the nodes whose first `packed_nodes` are packed and the others
        one per 128-byte line (tbvh_cwbvh_set_hybrid); < 0: back to the uploaded array."""
        check(lib.tbvh_cwbvh_set_hybrid(self._h, int(packed_nodes)), "tbvh_cwbvh_set_hybrid")

    def Intersect(self, rays: np.ndarray) -> np.ndarray:
        """rays: structured RAY_DTYPE array (64-byte records) or a (n, 128)-byte host Ray[] view;
        updated in place (bytes 44..63 of records that hit) and returned."""
        assert rays.flags["C_CONTIGUOUS"] and rays.flags["WRITEABLE"]
        stride = rays.strides[0] if rays.shape[0] else max(rays.dtype.itemsize, 64)
        check(lib.tbvh_intersect(self._h, _ptr(rays), rays.shape[0], stride), "tbvh_intersect")
        return rays

    def IsOccluded(self, rays: np.ndarray) -> np.ndarray:
        assert rays.flags["C_CONTIGUOUS"]
        out = np.zeros(rays.shape[0], dtype=np.uint8)
        stride = rays.strides[0] if rays.shape[0] else max(rays.dtype.itemsize, 64)
        check(lib.tbvh_occluded(self._h, _ptr(rays), rays.shape[0], stride, _ptr(out)), "tbvh_occluded")
        return out

    # device-resident, asynchronous
    def intersect_device(self, d_rays: int, n: int):
        check(lib.tbvh_intersect_device(self._h, C.c_void_p(d_rays), n), "tbvh_intersect_device")

    def intersect_device_fresh(self, d_rays: int, n: int, tmax: float = 1e30):
        check(lib.tbvh_intersect_device_fresh(self._h, C.c_void_p(d_rays), n, float(tmax)), "tbvh_intersect_device_fresh")

    def occluded_device(self, d_rays: int, n: int, d_out: int):
        check(lib.tbvh_occluded_device(self._h, C.c_void_p(d_rays), n, C.c_void_p(d_out)), "tbvh_occluded_device")


class BVH_GPU(_Scene):
    """Aila-Laine 2-wide layout (tiny_bvh.h:1092-1127)."""
    layout = LAYOUT_BVH_GPU

    def Build(self, verts: np.ndarray, **kw) -> "BVH_GPU":
        self.host = HostBVH(verts, LAYOUT_BVH_GPU, **kw)
        return self.Upload(self.host.blob(0, np.uint32, 16), self.host.blob(1, np.uint32, 1), self.host.verts)

    def Upload(self, nodes64: np.ndarray, prim_idx: np.ndarray, verts: np.ndarray) -> "BVH_GPU":
        nodes64 = np.ascontiguousarray(nodes64); prim_idx = np.ascontiguousarray(prim_idx, dtype=np.uint32)
        verts = np.ascontiguousarray(verts, dtype=np.float32)
        check(lib.tbvh_upload_bvh_gpu(self.ctx._h, _ptr(nodes64), nodes64.nbytes // 64, _ptr(prim_idx), prim_idx.size,
                                      _ptr(verts), verts.size // 12, C.byref(self._h)), "tbvh_upload_bvh_gpu")
        return self

    def Update(self, nodes64: np.ndarray, prim_idx: np.ndarray, verts: np.ndarray) -> "BVH_GPU":
        """In-place re-upload of a blob refitted / re-converted on the host (tbvh_update_bvh_gpu): same handle, same device memory."""
        nodes64 = np.ascontiguousarray(nodes64); prim_idx = np.ascontiguousarray(prim_idx, dtype=np.uint32)
        verts = np.ascontiguousarray(verts, dtype=np.float32)
        check(lib.tbvh_update_bvh_gpu(self._h, _ptr(nodes64), nodes64.nbytes // 64, _ptr(prim_idx), prim_idx.size, _ptr(verts), verts.size // 12), "tbvh_update_bvh_gpu")
        return self


class BVH4_GPU(_Scene):
    """Quantized 4-wide layout with inline triangles (tiny_bvh.h:1245-1289)."""
    layout = LAYOUT_BVH4_GPU

    def Build(self, verts: np.ndarray, **kw) -> "BVH4_GPU":
        self.host = HostBVH(verts, LAYOUT_BVH4_GPU, **kw)
        return self.Upload(self.host.blob(0, np.uint32, 4))

    def BuildOnDevice(self, verts: np.ndarray, max_leaf_tris: int = 4, builder: str = "lbvh", radius: int = 0) -> "BVH4_GPU":
        """LBVH (tbvh_build_device) or PLOC (tbvh_build_device_ploc) build + 4-wide collapse + encode on the GPU."""
        verts = np.ascontiguousarray(verts, np.float32)
        if builder == "ploc":
            check(lib.tbvh_build_device_ploc(self.ctx._h, _ptr(verts), verts.shape[0] // 3, 0, LAYOUT_BVH4_GPU, radius, C.byref(self._h)), "tbvh_build_device_ploc")
        else:
            check(lib.tbvh_build_device(self.ctx._h, _ptr(verts), verts.shape[0] // 3, 0, LAYOUT_BVH4_GPU, max_leaf_tris, C.byref(self._h)), "tbvh_build_device")
        return self

    def ConvertFromBVH2(self, nodes32: np.ndarray, prim_idx: np.ndarray, verts: np.ndarray) -> "BVH4_GPU":
        """BVH4_GPU::ConvertFrom on the device (tbvh_convert_bvh2_device)."""
        nodes32 = np.ascontiguousarray(nodes32); prim_idx = np.ascontiguousarray(prim_idx, np.uint32); verts = np.ascontiguousarray(verts, np.float32)
        check(lib.tbvh_convert_bvh2_device(self.ctx._h, _ptr(nodes32), nodes32.nbytes // 32, _ptr(prim_idx), prim_idx.size, _ptr(verts), verts.shape[0] // 3,
                                           0, LAYOUT_BVH4_GPU, C.byref(self._h)), "tbvh_convert_bvh2_device")
        return self

    def Upload(self, blocks16: np.ndarray) -> "BVH4_GPU":
        blocks16 = np.ascontiguousarray(blocks16)
        check(lib.tbvh_upload_bvh4_gpu(self.ctx._h, _ptr(blocks16), blocks16.nbytes // 16, C.byref(self._h)), "tbvh_upload_bvh4_gpu")
        return self

    def Update(self, blocks16: np.ndarray) -> "BVH4_GPU":
        """In-place re-upload of a blob refitted / re-converted on the host (tbvh_update_bvh4_gpu)."""
        blocks16 = np.ascontiguousarray(blocks16)
        check(lib.tbvh_update_bvh4_gpu(self._h, _ptr(blocks16), blocks16.nbytes // 16), "tbvh_update_bvh4_gpu")
        return self


class BVH8_CWBVH(_Scene):
    """Compressed wide BVH (tiny_bvh.h:1334-1362)."""
    layout = LAYOUT_CWBVH

    def Build(self, verts: np.ndarray, **kw) -> "BVH8_CWBVH":
        self.host = HostBVH(verts, LAYOUT_CWBVH, **kw)
        return self.Upload(self.host.blob(0, np.uint32, 4), self.host.blob(1, np.uint32, 4))

    def BuildOnDevice(self, verts: np.ndarray, max_leaf_tris: int = 0, builder: str = "lbvh", radius: int = 0) -> "BVH8_CWBVH":
        """LBVH (tbvh_build_device) or PLOC (tbvh_build_device_ploc) build + wide collapse + encode on the GPU; nothing is built on the host."""
        verts = np.ascontiguousarray(verts, np.float32)
        if builder == "ploc":
            check(lib.tbvh_build_device_ploc(self.ctx._h, _ptr(verts), verts.shape[0] // 3, 0, LAYOUT_CWBVH, radius, C.byref(self._h)), "tbvh_build_device_ploc")
        else:
            check(lib.tbvh_build_device(self.ctx._h, _ptr(verts), verts.shape[0] // 3, 0, LAYOUT_CWBVH, max_leaf_tris, C.byref(self._h)), "tbvh_build_device")
        return self

    def ConvertFromBVH2(self, nodes32: np.ndarray, prim_idx: np.ndarray, verts: np.ndarray) -> "BVH8_CWBVH":
        """BVH8_CWBVH::ConvertFrom on the device (tbvh_convert_bvh2_device): a plain BVH2 (32-byte BVHNode
        array with leaves of at most 3 triangles, primIdx, vertices) goes up, the GPU collapses and encodes."""
        nodes32 = np.ascontiguousarray(nodes32); prim_idx = np.ascontiguousarray(prim_idx, np.uint32); verts = np.ascontiguousarray(verts, np.float32)
        check(lib.tbvh_convert_bvh2_device(self.ctx._h, _ptr(nodes32), nodes32.nbytes // 32, _ptr(prim_idx), prim_idx.size, _ptr(verts), verts.shape[0] // 3,
                                           0, LAYOUT_CWBVH, C.byref(self._h)), "tbvh_convert_bvh2_device")
        return self

    def Save(self, path: str, n_tris: int = 0, bounds=None) -> None:
        """BVH8_CWBVH::Save (tiny_bvh.h:5786-5795): a file BVH8_CWBVH::Load accepts.  The blobs come from the host
        copy if this scene was built here, else they are read back from the device."""
        host = getattr(self, "host", None)
        if host is not None:
            nodes, tris = host.blob(0, np.uint32, 4), host.blob(1, np.uint32, 4)
            n_tris = n_tris or host.n_tris
        else:
            nodes, tris = self.download_blobs()
        n_tris = n_tris or tris.shape[0] // 3
        b = None if bounds is None else np.ascontiguousarray(bounds, np.float32).reshape(6)
        check(lib.tbvh_cwbvh_file_write(os.fsencode(path), _ptr(nodes), nodes.shape[0], _ptr(tris), tris.shape[0], n_tris,
                                        None if b is None else _ptr(b)), "tbvh_cwbvh_file_write")

    def Load(self, path: str, expected_tris: int = 0) -> "BVH8_CWBVH":
        """BVH8_CWBVH::Load (tiny_bvh.h:5797-5820) + upload: also reads files written by the reference itself."""
        self.host = HostBVH.from_cwbvh_file(path, expected_tris)
        return self.Upload(self.host.blob(0, np.uint32, 4), self.host.blob(1, np.uint32, 4))

    def Upload(self, nodes16: np.ndarray, tris16: np.ndarray) -> "BVH8_CWBVH":
        nodes16 = np.ascontiguousarray(nodes16); tris16 = np.ascontiguousarray(tris16)
        check(lib.tbvh_upload_cwbvh(self.ctx._h, _ptr(nodes16), nodes16.nbytes // 16, _ptr(tris16), tris16.nbytes // 16,
                                    C.byref(self._h)), "tbvh_upload_cwbvh")
        return self

    def Update(self, nodes16: np.ndarray, tris16: np.ndarray) -> "BVH8_CWBVH":
        """In-place re-upload of a blob refitted / re-converted on the host (tbvh_update_cwbvh): BVH::Refit + ConvertFrom of the
        reference's animation flow without freeing the scene; TLASes over this BLAS keep working."""
        nodes16 = np.ascontiguousarray(nodes16); tris16 = np.ascontiguousarray(tris16)
        check(lib.tbvh_update_cwbvh(self._h, _ptr(nodes16), nodes16.nbytes // 16, _ptr(tris16), tris16.nbytes // 16), "tbvh_update_cwbvh")
        return self


LAYOUT_CLASSES = {LAYOUT_BVH_GPU: BVH_GPU, LAYOUT_BVH4_GPU: BVH4_GPU, LAYOUT_CWBVH: BVH8_CWBVH}


def intersect_sharded(replicas: list, rays: np.ndarray) -> np.ndarray:
    """ONE ray array over several devices (tbvh_intersect_sharded): replicas[i] is the same BVH uploaded
    through its own Context; contiguous wave-aligned shards, one host thread per device, results written in place."""
    assert rays.flags["C_CONTIGUOUS"] and rays.dtype.itemsize in (64, 128)
    arr = (C.c_void_p * len(replicas))(*[r._h for r in replicas])
    check(lib.tbvh_intersect_sharded(arr, len(replicas), _ptr(rays), rays.shape[0], rays.dtype.itemsize), "tbvh_intersect_sharded")
    return rays


def occluded_sharded(replicas: list, rays: np.ndarray) -> np.ndarray:
    assert rays.flags["C_CONTIGUOUS"] and rays.dtype.itemsize in (64, 128)
    occ = np.zeros(rays.shape[0], np.uint8)
    arr = (C.c_void_p * len(replicas))(*[r._h for r in replicas])
    check(lib.tbvh_occluded_sharded(arr, len(replicas), _ptr(rays), rays.shape[0], rays.dtype.itemsize, _ptr(occ)), "tbvh_occluded_sharded")
    return occ


def intersect_sharded_device(replicas: list, d_rays: list, n_rays: list, fresh: bool = True, tmax: float = 1e30):
    """Device-resident shards (tbvh_intersect_sharded_device): d_rays[i] / n_rays[i] live on the device of replicas[i]; one host thread
    enqueues all launches, then waits.  Returns (kernel ms per device, host dispatch ms per device)."""
    k = len(replicas)
    sc = (C.c_void_p * k)(*[r._h for r in replicas])
    dp = (C.c_void_p * k)(*[int(p) for p in d_rays])
    nn = (C.c_uint64 * k)(*[int(x) for x in n_rays])
    km, dm = (C.c_float * k)(), (C.c_float * k)()
    check(lib.tbvh_intersect_sharded_device(sc, k, dp, nn, 1 if fresh else 0, float(tmax), km, dm), "tbvh_intersect_sharded_device")
    return [float(x) for x in km], [float(x) for x in dm]


def occluded_sharded_device(replicas: list, d_rays: list, n_rays: list, d_occ: list):
    k = len(replicas)
    sc = (C.c_void_p * k)(*[r._h for r in replicas])
    dp = (C.c_void_p * k)(*[int(p) for p in d_rays])
    do = (C.c_void_p * k)(*[int(p) for p in d_occ])
    nn = (C.c_uint64 * k)(*[int(x) for x in n_rays])
    km, dm = (C.c_float * k)(), (C.c_float * k)()
    check(lib.tbvh_occluded_sharded_device(sc, k, dp, nn, do, km, dm), "tbvh_occluded_sharded_device")
    return [float(x) for x in km], [float(x) for x in dm]


def wavefront_render_sharded(wfs: list, scenes_: list, d_verts: list, cam: Camera, light_pos, light_color=(1.0, 1.0, 1.0), sky_lo=(0.6, 0.7, 0.8), sky_hi=(0.2, 0.4, 0.9),
                             eps: float = 1e-3, max_depth: int = 3, seed: int = 1, clear: bool = True, light_size=(0.0, 0.0)):
    """One frame over several devices (tbvh_wavefront_render_sharded): wfs[i] is the band of the image (Wavefront.set_band) rendered on
    the device of scenes_[i].  Returns per band {"extend_rays", "shadow_rays", "frame_ms", "dispatch_ms"}."""
    k = len(wfs)
    p = _capi.WfParams()
    p.light_pos[:] = [float(x) for x in light_pos]; p.light_color[:] = [float(x) for x in light_color]
    p.sky_lo[:] = [float(x) for x in sky_lo]; p.sky_hi[:] = [float(x) for x in sky_hi]
    p.eps, p.max_depth, p.seed, p.clear = float(eps), int(max_depth), int(seed), int(clear)
    p.light_size[:] = [float(x) for x in light_size]; p.flags = 0; p.sample_index = 0xFFFFFFFF
    wa = (C.c_void_p * k)(*[w._h for w in wfs])
    sa = (C.c_void_p * k)(*[s._h for s in scenes_])
    va = (C.c_void_p * k)(*[int(v) if v else None for v in d_verts])
    st = (_capi.WfStats * k)()
    dm = (C.c_float * k)()
    check(lib.tbvh_wavefront_render_sharded(wa, sa, va, k, C.byref(cam), C.byref(p), st, dm), "tbvh_wavefront_render_sharded")
    return [{"extend_rays": [int(x) for x in st[i].extend_rays[:max_depth]], "shadow_rays": [int(x) for x in st[i].shadow_rays[:max_depth]],
             "frame_ms": float(st[i].frame_ms), "dispatch_ms": float(dm[i])} for i in range(k)]


def wavefront_read_sharded(wfs: list, width: int, full_height: int) -> np.ndarray:
    img = np.zeros((full_height, width, 4), np.float32)
    wa = (C.c_void_p * len(wfs))(*[w._h for w in wfs])
    check(lib.tbvh_wavefront_read_sharded(wa, len(wfs), _ptr(img)), "tbvh_wavefront_read_sharded")
    return img


def device_count() -> int:
    return int(lib.tbvh_device_count())


# BLASInstance, 192 bytes (tiny_bvh.h:1443-1457); transform is row-major with the translation in
# elements 3, 7, 11 (tiny_bvh.h:513-528)
INSTANCE_DTYPE = np.dtype([
    ("transform", "<f4", 16), ("invTransform", "<f4", 16),
    ("aabbMin", "<f4", 3), ("blasIdx", "<u4"), ("aabbMax", "<f4", 3), ("mask", "<u4"), ("pad", "<u4", 8),
])
assert INSTANCE_DTYPE.itemsize == 192


def make_instances(transforms: np.ndarray, blas_idx, mask: int = 0xFFFF) -> np.ndarray:
    """BLASInstance records from (n, 4, 4) row-major transforms; invTransform and the bounds are
    filled by TLAS.Build (BLASInstance::Update, tiny_bvh.h:8386-8400)."""
    t = np.ascontiguousarray(transforms, np.float32).reshape(-1, 16)
    inst = np.zeros(t.shape[0], INSTANCE_DTYPE)
    inst["transform"] = t
    inst["invTransform"] = np.eye(4, dtype=np.float32).reshape(16)
    inst["blasIdx"] = blas_idx
    inst["mask"] = mask
    return inst


class TLAS(_Scene):
    """Top-level BVH over BLAS instances: BVH_GPU nodes over BLASInstance records
    (BVH_GPU::Build(BLASInstance*, ...), tiny_bvh.h:4575-4581; tiny_bvh_gpu2.cpp:108-136)."""
    layout = LAYOUT_BVH_GPU

    def Build(self, instances: np.ndarray, blas: list) -> "TLAS":
        """instances: INSTANCE_DTYPE array with transform/blasIdx/mask set (updated in place);
        blas: uploaded BLAS scenes (BVH8_CWBVH, BVH4_GPU or BVH_GPU, also mixed) built with .Build()."""
        assert instances.dtype == INSTANCE_DTYPE and instances.flags["C_CONTIGUOUS"]
        bounds = np.zeros((len(blas), 6), np.float32)
        for i, b in enumerate(blas):
            if getattr(b, "_bounds", None) is None:      # root box of the BLAS, computed once
                v = b.host.verts[:, :3]
                b._bounds = np.concatenate([v.min(0), v.max(0)]).astype(np.float32)
            bounds[i] = b._bounds
        h = C.c_void_p()
        check(lib.tbvh_host_build_tlas(_ptr(instances), instances.shape[0], _ptr(bounds), len(blas), C.byref(h)), "tbvh_host_build_tlas")
        host = HostBVH.__new__(HostBVH)
        host._h = h; host.layout = LAYOUT_BVH_GPU; host.verts = None; host.n_tris = instances.shape[0]
        self.host = host
        self.instances = instances
        self.blas = list(blas)
        nodes = host.blob(0, np.uint32, 16); idx = host.blob(1, np.uint32, 1)
        if self._h:
            return self.Update(nodes, idx, instances)
        return self.Upload(nodes, idx, instances, blas)

    def Upload(self, nodes64: np.ndarray, tlas_idx: np.ndarray, instances: np.ndarray, blas: list) -> "TLAS":
        nodes64 = np.ascontiguousarray(nodes64); tlas_idx = np.ascontiguousarray(tlas_idx, np.uint32)
        arr = (C.c_void_p * len(blas))(*[b._h for b in blas])
        check(lib.tbvh_upload_tlas(self.ctx._h, _ptr(nodes64), nodes64.nbytes // 64, _ptr(tlas_idx), tlas_idx.size, _ptr(instances),
                                   instances.shape[0], arr, len(blas), C.byref(self._h)), "tbvh_upload_tlas")
        self.blas = list(blas)  # BLAS scenes must outlive the TLAS
        return self

    def Update(self, nodes64: np.ndarray, tlas_idx: np.ndarray, instances: np.ndarray) -> "TLAS":
        nodes64 = np.ascontiguousarray(nodes64); tlas_idx = np.ascontiguousarray(tlas_idx, np.uint32)
        check(lib.tbvh_update_tlas(self._h, _ptr(nodes64), nodes64.nbytes // 64, _ptr(tlas_idx), tlas_idx.size, _ptr(instances), instances.shape[0]),
              "tbvh_update_tlas")
        return self


    def _blas_bounds(self, blas: list) -> np.ndarray:
        bounds = np.zeros((len(blas), 6), np.float32)
        for i, b in enumerate(blas):
            if getattr(b, "_bounds", None) is None:      # root box of the BLAS, computed once
                v = b.host.verts[:, :3]
                b._bounds = np.concatenate([v.min(0), v.max(0)]).astype(np.float32)
            bounds[i] = b._bounds
        return bounds

    def RebuildOnDevice(self, transforms=None, on_device: bool = False) -> "TLAS":
        """Per-frame rebuild on the GPU (tbvh_rebuild_tlas_device): instance update + LBVH TLAS, no host
        build and no node upload.  transforms: (n, 16) float32 array (host), a device pointer (int,
        on_device=True), or None to keep the transforms already in the device records."""
        bounds = None
        if not getattr(self, "_bounds_sent", False):
            bounds = self._blas_bounds(self.blas)
        if transforms is None:
            t = None
        elif on_device:
            t = C.c_void_p(int(transforms))
        else:
            transforms = np.ascontiguousarray(transforms, np.float32)
            assert transforms.size == self.instances.shape[0] * 16
            t = _ptr(transforms)
        check(lib.tbvh_rebuild_tlas_device(self._h, t, 1 if on_device else 0, _ptr(bounds) if bounds is not None else None,
                                           len(self.blas) if bounds is not None else 0), "tbvh_rebuild_tlas_device")
        self._bounds_sent = True
        return self

    def Download(self):
        """(nodes64 as (n,16) uint32, tlas_idx, instances) currently on the device."""
        n = self.instances.shape[0]
        nn = C.c_uint64(0)
        check(lib.tbvh_tlas_download(self._h, None, 0, None, 0, None, 0, C.byref(nn)), "tbvh_tlas_download")
        nodes = np.zeros((nn.value, 16), np.uint32); idx = np.zeros(n, np.uint32); inst = np.zeros(n, INSTANCE_DTYPE)
        check(lib.tbvh_tlas_download(self._h, _ptr(nodes), nn.value, _ptr(idx), n, _ptr(inst), n, C.byref(nn)), "tbvh_tlas_download")
        return nodes, idx, inst


class Wavefront:
    """Device-resident wavefront path tracer (tbvh_wavefront_*): one call enqueues a whole frame
    (Generate, {Extend, Shade} x depth, Connect) with all queues and counters on the device."""

    def __init__(self, ctx: Context, width: int, height: int):
        self.ctx, self.width, self.height = ctx, width, height
        h = C.c_void_p()
        check(lib.tbvh_wavefront_create(ctx._h, width, height, C.byref(h)), "tbvh_wavefront_create")
        self._h = h

    def render(self, scene: _Scene, d_verts: int, cam: Camera, light_pos, light_color=(1.0, 1.0, 1.0), sky_lo=(0.6, 0.7, 0.8), sky_hi=(0.2, 0.4, 0.9),
               eps: float = 1e-3, max_depth: int = 3, seed: int = 1, clear: bool = True, stats: bool = True,
               light_size=(0.0, 0.0), one_diffuse_bounce: bool = False, reference_letter: bool = False, sample_index: int = 0xFFFFFFFF):
        p = _capi.WfParams()
        p.light_pos[:] = [float(x) for x in light_pos]; p.light_color[:] = [float(x) for x in light_color]
        p.sky_lo[:] = [float(x) for x in sky_lo]; p.sky_hi[:] = [float(x) for x in sky_hi]
        p.eps, p.max_depth, p.seed, p.clear = float(eps), int(max_depth), int(seed), int(clear)
        p.light_size[:] = [float(x) for x in light_size]; p.flags = (1 if one_diffuse_bounce else 0) | (2 if reference_letter else 0)
        p.sample_index = int(sample_index) & 0xFFFFFFFF
        st = _capi.WfStats()
        check(lib.tbvh_wavefront_render(self._h, scene._h, C.c_void_p(d_verts) if d_verts else None, C.byref(cam), C.byref(p), C.byref(st) if stats else None), "tbvh_wavefront_render")
        if not stats:
            return None
        return {"extend_rays": [int(x) for x in st.extend_rays[:max_depth]], "shadow_rays": [int(x) for x in st.shadow_rays[:max_depth]], "frame_ms": float(st.frame_ms)}

    def set_band(self, first_row: int, full_height: int):
        """This object renders rows [first_row, first_row + height) of an image of full_height rows (tbvh_wavefront_set_band)."""
        check(lib.tbvh_wavefront_set_band(self._h, int(first_row), int(full_height)), "tbvh_wavefront_set_band")
        return self

    def read(self) -> np.ndarray:
        img = np.zeros((self.height, self.width, 4), np.float32)
        check(lib.tbvh_wavefront_read(self._h, _ptr(img)), "tbvh_wavefront_read")
        return img

    def set_blue_noise(self, table) -> None:
        """The demos' 128 x 128 x 8 blue-noise table (uint32 words), or None to remove it (tbvh_wavefront_set_blue_noise)."""
        if table is None:
            check(lib.tbvh_wavefront_set_blue_noise(self._h, None, 0), "tbvh_wavefront_set_blue_noise")
            return
        t = np.ascontiguousarray(table, np.uint32).reshape(-1)
        check(lib.tbvh_wavefront_set_blue_noise(self._h, _ptr(t), t.size), "tbvh_wavefront_set_blue_noise")

    def set_blas_vertices(self, d_verts_per_blas: list) -> None:
        """TLAS scenes: the device vertex array of every BLAS, in blasIdx order (tbvh_wavefront_set_blas_vertices)."""
        arr = (C.c_void_p * len(d_verts_per_blas))(*[int(p) for p in d_verts_per_blas])
        check(lib.tbvh_wavefront_set_blas_vertices(self._h, arr, len(d_verts_per_blas)), "tbvh_wavefront_set_blas_vertices")

    def finalize(self, scale: float = 1.0) -> np.ndarray:
        """Finalize of wavefront.cl:275-286: (height, width) uint32 0x00RRGGBB."""
        px = np.zeros((self.height, self.width), np.uint32)
        check(lib.tbvh_wavefront_finalize(self._h, float(scale), _ptr(px)), "tbvh_wavefront_finalize")
        return px

    def close(self):
        if self._h and self.ctx._h:
            lib.tbvh_wavefront_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
