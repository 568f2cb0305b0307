"""Loaders for the CHECKERS under oracle/ (test infrastructure): the plain-C restatement
(oracle/liborc.so) and, when present, the real reference behind its C shim
(oracle/_ref/libtinybvh_ref.so).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_PATH = os.path.join(ROOT, "oracle", "liborc.so")
REF_PATH = os.path.join(ROOT, "oracle", "_ref", "libtinybvh_ref.so")

_vp, _u64, _u32 = C.c_void_p, C.c_uint64, C.c_uint32


def _p(a):
    return C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)


def build_oracle():
    if not os.path.exists(ORC_PATH) or os.path.getmtime(ORC_PATH) < os.path.getmtime(os.path.join(ROOT, "oracle", "tbvh_oracle.c")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liborc.so"], stdout=subprocess.DEVNULL)


class Oracle:
    """The C restatement.  All functions take/return numpy arrays of RAY_DTYPE records."""

    def __init__(self, tie_rule: int = 1):
        """tie_rule 1 (default): the library's (at exactly equal t the smaller prim, then the smaller instance, wins:
        order-independent; under it the GPU parity tests demand the exact prim); 0: the reference's (the later test wins;
        what the pinned comparisons against oracle/_ref and the golden vectors need).  The rule is process-wide state of
        liborc, so every call sets it."""
        build_oracle()
        self.lib = C.CDLL(ORC_PATH)
        self.tie_rule = int(tie_rule)
        L = self.lib
        assert L.orc_abi_version() >= 2, "oracle/liborc.so is stale: make -C oracle"
        L.orc_bvh2_intersect.argtypes = [_vp, _vp, _vp, _vp, _u64, _u32, _vp]
        L.orc_bvh2_occluded.argtypes = [_vp, _vp, _vp, _vp, _u64, _u32, _vp]
        L.orc_bvhgpu_intersect.argtypes = [_vp, _vp, _vp, _vp, _u64, _u32, _vp]
        L.orc_bvh4_intersect.argtypes = [_vp, _vp, _u64, _u32, _vp]
        L.orc_cwbvh_intersect.argtypes = [_vp, _vp, _vp, _u64, _u32, _vp]
        L.orc_tlas_intersect.argtypes = [_vp, _vp, _vp, _vp, _vp, _u64, _u32]
        for f in (L.orc_bvh2_intersect, L.orc_bvh2_occluded, L.orc_bvhgpu_intersect, L.orc_bvh4_intersect, L.orc_cwbvh_intersect, L.orc_tlas_intersect):
            f.restype = None

    @staticmethod
    def _prep(rays):
        r = np.ascontiguousarray(rays).copy()
        return r

    def set_opmap(self, map_data, n):
        """Opacity micromaps for ALL following oracle queries (None clears); the array must stay alive."""
        self._opmap = None if map_data is None else np.ascontiguousarray(map_data, np.uint32)
        self.lib.orc_set_opmap(_p(self._opmap) if self._opmap is not None else None, int(n))

    def bvh2_intersect(self, nodes32, prim_idx, verts, rays, counts=False):
        r = self._prep(rays)
        c = np.zeros(2, np.uint64) if counts else None
        self.lib.orc_set_tie_rule(self.tie_rule)
        self.lib.orc_bvh2_intersect(_p(nodes32), _p(prim_idx), _p(verts), _p(r), r.shape[0], r.strides[0], _p(c))
        return (r, c) if counts else r

    def bvh2_occluded(self, nodes32, prim_idx, verts, rays):
        r = np.ascontiguousarray(rays)
        out = np.zeros(r.shape[0], np.uint8)
        self.lib.orc_bvh2_occluded(_p(nodes32), _p(prim_idx), _p(verts), _p(r), r.shape[0], r.strides[0], _p(out))
        return out

    def bvhgpu_intersect(self, nodes64, prim_idx, verts, rays, counts=False):
        r = self._prep(rays)
        c = np.zeros(2, np.uint64) if counts else None
        self.lib.orc_set_tie_rule(self.tie_rule)
        self.lib.orc_bvhgpu_intersect(_p(nodes64), _p(prim_idx), _p(verts), _p(r), r.shape[0], r.strides[0], _p(c))
        return (r, c) if counts else r

    def bvh4_intersect(self, blocks16, rays, counts=False):
        r = self._prep(rays)
        c = np.zeros(2, np.uint64) if counts else None
        self.lib.orc_set_tie_rule(self.tie_rule)
        self.lib.orc_bvh4_intersect(_p(blocks16), _p(r), r.shape[0], r.strides[0], _p(c))
        return (r, c) if counts else r

    def cwbvh_intersect(self, nodes16, tris16, rays, counts=False):
        r = self._prep(rays)
        c = np.zeros(2, np.uint64) if counts else None
        self.lib.orc_set_tie_rule(self.tie_rule)
        self.lib.orc_cwbvh_intersect(_p(nodes16), _p(tris16), _p(r), r.shape[0], r.strides[0], _p(c))
        return (r, c) if counts else r


class OrcBlas(C.Structure):
    _fields_ = [("nodes32", C.c_void_p), ("primIdx", C.c_void_p), ("verts16", C.c_void_p)]


def tlas_intersect(orc: "Oracle", tlas_nodes32, tlas_idx, instances, blas_list, rays):
    """BVH::IntersectTLAS restated (oracle/tbvh_oracle.c: orc_tlas_intersect).  blas_list:
    [(bvh2_nodes, prim_idx, verts), ...] indexed by BLASInstance::blasIdx."""
    keep = []
    arr = (OrcBlas * len(blas_list))()
    for i, (n, p, v) in enumerate(blas_list):
        n = np.ascontiguousarray(n); p = np.ascontiguousarray(p); v = np.ascontiguousarray(v)
        keep += [n, p, v]
        arr[i] = OrcBlas(n.ctypes.data, p.ctypes.data, v.ctypes.data)
    r = np.ascontiguousarray(rays).copy()
    tn = np.ascontiguousarray(tlas_nodes32); ti = np.ascontiguousarray(tlas_idx, np.uint32); inst = np.ascontiguousarray(instances)
    orc.lib.orc_set_tie_rule(orc.tie_rule)
    orc.lib.orc_tlas_intersect(_p(tn), _p(ti), _p(inst), C.cast(arr, C.c_void_p), _p(r), r.shape[0], r.strides[0])
    return r


def have_reference() -> bool:
    return os.path.exists(REF_PATH)


class Reference:
    """The real tiny_bvh.h behind oracle/ref_shim.cpp."""

    def __init__(self):
        self.lib = C.CDLL(REF_PATH)
        L = self.lib
        L.ref_selfcheck.restype = C.c_int
        L.ref_version.restype = C.c_char_p
        L.ref_build.restype = _vp; L.ref_build.argtypes = [_vp, _u32, C.c_int, C.c_int]
        L.ref_free.argtypes = [_vp]; L.ref_free.restype = None
        L.ref_blob.restype = _u64; L.ref_blob.argtypes = [_vp, C.c_int, C.c_int, C.POINTER(_vp)]
        L.ref_verts.restype = _vp; L.ref_verts.argtypes = [_vp]
        L.ref_intersect.restype = C.c_int; L.ref_intersect.argtypes = [_vp, C.c_int, _vp, _u64, _u32]
        L.ref_occluded.restype = C.c_int; L.ref_occluded.argtypes = [_vp, C.c_int, _vp, _u64, _u32, _vp]
        L.ref_counts.restype = C.c_int; L.ref_counts.argtypes = [_vp, C.c_int, _vp, _u64, _u32, C.POINTER(_u64), C.POINTER(_u64)]
        L.ref_time_mt.restype = C.c_double; L.ref_time_mt.argtypes = [_vp, C.c_int, _vp, _u64, _u32, C.c_int, C.c_int, C.POINTER(_u64)]
        L.ref_set_opmap.restype = None; L.ref_set_opmap.argtypes = [_vp, _vp, _u32]
        L.ref_tlas_build.restype = _vp; L.ref_tlas_build.argtypes = [_vp, _u32, C.POINTER(_vp), _u32]
        L.ref_tlas_free.argtypes = [_vp]; L.ref_tlas_free.restype = None
        L.ref_tlas_intersect.restype = C.c_int; L.ref_tlas_intersect.argtypes = [_vp, _vp, _u64, _u32]
        L.ref_tlas_blob.restype = _u64; L.ref_tlas_blob.argtypes = [_vp, C.c_int, C.POINTER(_vp)]
        L.ref_cwbvh_object_layout.restype = None; L.ref_cwbvh_object_layout.argtypes = [C.POINTER(_u32)]
        L.ref_cwbvh_default_image.restype = None; L.ref_cwbvh_default_image.argtypes = [_vp, _u32]
        L.ref_cwbvh_save.restype = C.c_int; L.ref_cwbvh_save.argtypes = [_vp, C.c_char_p]
        L.ref_cwbvh_load_and_intersect.restype = C.c_int; L.ref_cwbvh_load_and_intersect.argtypes = [C.c_char_p, _u32, _vp, _u64, _u32]
        assert L.ref_selfcheck() == 0, "tinybvh::Ray layout differs from the 64-byte record"

    def build(self, verts, hq=False, threaded=False):
        return RefScene(self, verts, hq, threaded)

    def cwbvh_object_layout(self):
        """sizeof(BVH8_CWBVH) and the offsets a Save / Load compatible file needs (ref_shim.cpp)."""
        o = (_u32 * 16)()
        self.lib.ref_cwbvh_object_layout(o)
        keys = ("size", "layout", "triCount", "idxCount", "aabbMin", "aabbMax", "opmapN", "opmap", "bvh8Data", "bvh8Tris",
                "allocatedBlocks", "usedBlocks", "bvh8.idxCount", "ownBVH8", "c_trav", "hqbvhbins")
        return dict(zip(keys, [int(x) for x in o]))

    def cwbvh_load_and_intersect(self, path, expected_tris, rays):
        """BVH8_CWBVH::Load on a fresh object + BVH8_CWBVH::Intersect; None if Load refused the file."""
        r = np.ascontiguousarray(rays).copy()
        rc = self.lib.ref_cwbvh_load_and_intersect(os.fsencode(path), expected_tris, _p(r), r.shape[0], r.strides[0])
        return None if rc else r


class RefScene:
    def __init__(self, ref: Reference, verts, hq, threaded):
        self.ref = ref
        self.verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 4)
        self.h = ref.lib.ref_build(_p(self.verts), self.verts.shape[0] // 3, int(hq), int(threaded))

    def __del__(self):
        try:
            if self.h:
                self.ref.lib.ref_free(self.h)
                self.h = None
        except Exception:
            pass

    def blob(self, layout, which, dtype, width):
        """layout: BVHBase::BVHType as everywhere (1 BVH, 5 BVH_GPU, 8 BVH4_GPU, 10 CWBVH; 110 = the BVH2 behind the CWBVH)."""
        assert layout in (1, 5, 8, 10, 110), layout
        p = _vp()
        n = self.ref.lib.ref_blob(self.h, layout, which, C.byref(p))
        nbytes = n * np.dtype(dtype).itemsize * width
        if not p.value or not n:
            return np.zeros((0, width), dtype)
        return np.frombuffer((C.c_char * nbytes).from_address(p.value), dtype=dtype).reshape(n, width).copy()

    def cwbvh_save(self, path):
        assert self.ref.lib.ref_cwbvh_save(self.h, os.fsencode(path)) == 0

    def intersect(self, layout, rays):
        r = np.ascontiguousarray(rays).copy()
        assert self.ref.lib.ref_intersect(self.h, layout, _p(r), r.shape[0], r.strides[0]) == 0
        return r

    def occluded(self, layout, rays):
        r = np.ascontiguousarray(rays)
        out = np.zeros(r.shape[0], np.uint8)
        assert self.ref.lib.ref_occluded(self.h, layout, _p(r), r.shape[0], r.strides[0], _p(out)) == 0
        return out

    def counts(self, layout, rays):
        r = np.ascontiguousarray(rays)
        s, t = _u64(), _u64()
        assert self.ref.lib.ref_counts(self.h, layout, _p(r), r.shape[0], r.strides[0], C.byref(s), C.byref(t)) == 0
        return s.value, t.value

    def time_mt(self, layout, rays, threads=0, shadow=False):
        r = np.ascontiguousarray(rays)
        hits = _u64()
        sec = self.ref.lib.ref_time_mt(self.h, layout, _p(r), r.shape[0], r.strides[0], threads, int(shadow), C.byref(hits))
        assert sec >= 0, f"ref_time_mt: layout {layout} is not timed (1 = BVH, 11 = BVH8_CPU)"
        return sec, hits.value


class RefTlas:
    """The real BVH::Build(BLASInstance*, ...) + BVH::IntersectTLAS (tiny_bvh.h:2221-2259, 3306-3380) over RefScene BLASes.
    `instances` (192-byte BLASInstance records with transform / blasIdx / mask set) are copied; .inst holds them as the
    reference updated them (BLASInstance::Update)."""

    def __init__(self, ref: Reference, instances, ref_scenes):
        self.ref = ref
        self.scenes = list(ref_scenes)        # keep the BLASes alive
        self.inst = np.ascontiguousarray(instances).copy()
        arr = (_vp * len(self.scenes))(*[s.h for s in self.scenes])
        self.h = ref.lib.ref_tlas_build(_p(self.inst), self.inst.shape[0], arr, len(self.scenes))

    def intersect(self, rays):
        r = np.ascontiguousarray(rays).copy()
        assert self.ref.lib.ref_tlas_intersect(self.h, _p(r), r.shape[0], r.strides[0]) == 0
        return r

    def __del__(self):
        try:
            if self.h:
                self.ref.lib.ref_tlas_free(self.h)
                self.h = None
        except Exception:
            pass


# ---- comparison with the parity contract -----------------------------------------------------------

def compare_hits(got: np.ndarray, want: np.ndarray, rtol: float = 1e-5):
    """Per-ray comparison of hit records under the contract of BASELINE.json: hit/miss and
    prim exact, t/u/v within rtol (relative for t, absolute-on-[0,1] for u,v).  Returns a dict
    of counts.  Classes of disagreement, all inherited from the reference (its own layouts
    disagree with BVH::Intersect in exactly these ways, SURVEY.md §7 "Bit-exact prim"):
      tie    prim differs while t agrees to rtol: two triangles at the same distance, the
             winner depends on visit order;
      onsurf one side reports a hit at t == +-0 (ray origin exactly on a triangle's plane,
             e.g. a bounce ray leaving an axis-aligned wall whose 1e-3 offset rounds away).
             Whether that triangle is even tested depends on how the slab test of the
             layout rounds (fma(b, rD, -O*rD) in BVH::Intersect vs (b - O) * rD in the wide
             layouts), so BVH2 and CWBVH/BVH4 traversals of the reference itself differ here.
    Everything else is a real error: hitmiss, prim_real (different triangle at a different
    distance), t_bad / uv_bad (same triangle, values off)."""
    far = np.float32(1e30)
    gh, wh = got["t"] < far, want["t"] < far
    onsurf = ((got["t"] == 0) | (want["t"] == 0)) & ((got["prim"] != want["prim"]) | (gh != wh))
    res = {"n": int(got.shape[0]), "hits": int(wh.sum()), "onsurf": int(onsurf.sum())}
    res["hitmiss"] = int(((gh != wh) & ~onsurf).sum())
    both = gh & wh & ~onsurf
    dt = np.abs(got["t"][both].astype(np.float64) - want["t"][both].astype(np.float64))
    rel = dt / np.maximum(np.abs(want["t"][both].astype(np.float64)), 1e-30)
    same_prim = got["prim"][both] == want["prim"][both]
    close_t = rel <= rtol
    res["prim_mismatch"] = int((~same_prim).sum())
    res["tie"] = int((~same_prim & close_t).sum())
    res["prim_real"] = int((~same_prim & ~close_t).sum())
    res["t_bad"] = int((same_prim & ~close_t).sum())
    sp = same_prim
    du = np.abs(got["u"][both][sp].astype(np.float64) - want["u"][both][sp].astype(np.float64))
    dv = np.abs(got["v"][both][sp].astype(np.float64) - want["v"][both][sp].astype(np.float64))
    res["max_rel_t"] = float(rel[same_prim].max()) if same_prim.any() else 0.0
    res["max_abs_uv"] = float(max(du.max(), dv.max())) if sp.any() else 0.0
    res["uv_bad"] = int(((du > 1e-5) | (dv > 1e-5)).sum()) if sp.any() else 0
    bi = sp.copy()
    for f in ("t", "u", "v"):
        bi &= got[f][both].view(np.uint32) == want[f][both].view(np.uint32)
    res["bit_identical"] = int(bi.sum())
    res["same_prim"] = int(sp.sum())
    return res


def compare_with_real_reference(got: np.ndarray, want: np.ndarray, check_inst: bool = False, ulp_window: int = 64):
    """GPU hit records against records of the REAL reference (oracle/_ref: BVH::Intersect / IntersectTLAS of tiny_bvh.h, run under ITS
    tie rule — among exactly equal t the later test wins, tiny_bvh.h:1656).  The library deviates from that rule in two deliberate,
    documented ways (DESIGN.md par. 4), and this comparison COUNTS each instead of folding them into a tolerance:
      tie_equal_t      both report a hit at the bit-identical t but name different primitives (or instances): the library's
                       order-independent rule (smaller prim, then smaller instance) against the reference's traversal-order-dependent one;
      closer_by_ulps   the library's t is SMALLER by 1..ulp_window ulps (whatever the prim): box tests cull 2^-20 beyond the closest hit
                       (device_common.h: cull_bound), so a candidate within a few ulps behind a face the reference's exact bound culled
                       is still tested and may win;
      farther_by_ulps  the library's t is LARGER by 1..ulp_window ulps (expected 0: tri_test rejects t > hit.t exactly).
    Real errors: hitmiss, prim_real (different primitive, t apart by more than the window), t_bad (same primitive, t apart by more than
    the window), uv_differs (same primitive, same t bits, u or v bits differ).  onsurf (one side reports t == +-0: ray origin exactly on a
    triangle's plane, where the reference's own layouts disagree with each other) is counted and excluded, as in compare_hits."""
    far = np.float32(1e30)
    gh, wh = got["t"] < far, want["t"] < far
    onsurf = ((got["t"] == 0) | (want["t"] == 0)) & ((got["prim"] != want["prim"]) | (gh != wh))
    both = gh & wh & ~onsurf
    gt = got["t"][both].view(np.int32).astype(np.int64); wt = want["t"][both].view(np.int32).astype(np.int64)   # positive floats order like their bits
    ulps = gt - wt
    same = got["prim"][both] == want["prim"][both]
    if check_inst:
        same &= got["inst"][both] == want["inst"][both]
    uv_same = (got["u"][both].view(np.uint32) == want["u"][both].view(np.uint32)) & (got["v"][both].view(np.uint32) == want["v"][both].view(np.uint32))
    near = np.abs(ulps) <= ulp_window
    res = {"n": int(got.shape[0]), "hits": int(wh.sum()), "onsurf": int(onsurf.sum()),
           "hitmiss": int(((gh != wh) & ~onsurf).sum()),
           "identical": int((same & (ulps == 0) & uv_same).sum()),
           "tie_equal_t": int((~same & (ulps == 0)).sum()),
           "closer_by_ulps": int(((ulps < 0) & near).sum()),
           "closer_by_ulps_same_prim": int(((ulps < 0) & near & same).sum()),
           "farther_by_ulps": int(((ulps > 0) & near).sum()),
           "max_ulps": int(np.abs(ulps[near]).max()) if near.any() else 0,
           "prim_real": int((~same & ~near).sum()),
           "t_bad": int((same & ~near).sum()),
           "uv_differs": int((same & (ulps == 0) & ~uv_same).sum())}
    res["differ_from_reference"] = res["tie_equal_t"] + res["closer_by_ulps"] + res["farther_by_ulps"]
    return res


# ---- the reference's own OpenCL kernels through ROCm OpenCL (oracle/ref_ocl.cpp) ----------------
REFOCL_PATH = os.path.join(ROOT, "oracle", "_ref", "libtinybvh_refocl.so")


class ReferenceOpenCL:
    """batch_ailalaine / batch_gpu4way / batch_cwbvh of the reference, compiled by the OpenCL
    driver at run time from the source text embedded at build time.  Measurement aid: lets the
    HIP kernels be timed next to the kernels they replace on the same GPU."""

    def __init__(self):
        if not os.path.exists(REFOCL_PATH):
            raise RuntimeError("oracle/_ref/libtinybvh_refocl.so not built (needs the reference checkout at build time)")
        self.lib = C.CDLL(REFOCL_PATH)
        L = self.lib
        L.refocl_init.restype = C.c_int
        L.refocl_error.restype = C.c_char_p
        L.refocl_device.restype = C.c_char_p
        L.refocl_run.restype = C.c_double
        L.refocl_run.argtypes = [C.c_int, _vp, _u64, _vp, _u64, _vp, _u64, _vp, _u64, C.c_int]
        L.refocl_wavefront.restype = C.c_int
        L.refocl_wavefront.argtypes = [_vp, _u64, _vp, _u64, _vp, _u64, _vp, _vp, _vp, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_char_p, _vp]
        L.refocl_tlas_extend.restype = C.c_double
        L.refocl_tlas_extend.argtypes = [_vp, _u64, _vp, _u64, _vp, _u64, _vp, _u64, _vp, _u64, _vp, _u64, C.c_int, _vp]
        if L.refocl_init() != 0:
            raise RuntimeError("reference OpenCL kernels unavailable: " + L.refocl_error().decode(errors="replace")[:2000])
        self.device = L.refocl_device().decode()

    def wavefront(self, nodes, tris, verts, noise, eye, p0, p1, p2, width, height, frames, iterations=3, patch=""):
        """The reference's wavefront.cl path tracer, driven like tiny_bvh_gpu.cpp:128-158 for `frames` frames; returns the
        averaged accumulator as (height, width, 4) float32."""
        nodes, tris, verts = (np.ascontiguousarray(x) for x in (nodes, tris, verts))
        noise = np.ascontiguousarray(noise, np.uint32).reshape(-1)
        assert noise.size == 128 * 128 * 8
        v4 = [np.ascontiguousarray(list(x) + [0.0], np.float32) for x in (eye, p0, p1, p2)]
        out = np.zeros((height, width, 4), np.float32)
        r = self.lib.refocl_wavefront(_p(nodes), nodes.nbytes, _p(tris), tris.nbytes, _p(verts), verts.nbytes, _p(noise), _p(v4[0]), _p(v4[1]), _p(v4[2]), _p(v4[3]),
                                      width, height, frames, iterations, patch.encode(), _p(out))
        if r != 0:
            raise RuntimeError(f"refocl_wavefront: {r}: " + self.lib.refocl_error().decode(errors="replace")[:3000])
        return out

    def tlas_extend(self, tlas_nodes, tlas_idx, instances, blas_nodes, blas_tris, rays, passes=3):
        """The reference's traverse_tlas (TLAS in BVH_GPU format over instances of ONE BVH8_CWBVH BLAS) through wavefront2.cl's Extend
        kernel, launched as tiny_bvh_gpu2.cpp:191 does.  Returns ((n, 4) float32 hits: t, u, v, prim + (inst << 24) as bits; mean kernel ms)."""
        tlas_nodes, instances, blas_nodes, blas_tris = (np.ascontiguousarray(x) for x in (tlas_nodes, instances, blas_nodes, blas_tris))
        tlas_idx = np.ascontiguousarray(tlas_idx, np.uint32)
        r = np.ascontiguousarray(rays)
        n = r.shape[0]
        out = np.zeros((n, 4), np.float32)
        ms = self.lib.refocl_tlas_extend(_p(tlas_nodes), tlas_nodes.nbytes, _p(tlas_idx), tlas_idx.size, _p(instances), instances.shape[0], _p(blas_nodes), blas_nodes.nbytes,
                                         _p(blas_tris), blas_tris.nbytes, _p(r), n, passes, _p(out))
        if ms < 0:
            raise RuntimeError("refocl_tlas_extend: " + self.lib.refocl_error().decode(errors="replace")[:3000])
        return out, ms

    def run(self, layout, blobs, rays, passes=3):
        """blobs: list of numpy arrays in kernel-argument order.  Returns (rays_out, mean_ms)."""
        r = np.ascontiguousarray(rays).copy()
        n = (r.shape[0] // 64) * 64
        b = [np.ascontiguousarray(x) for x in blobs] + [None, None]
        ms = self.lib.refocl_run(layout, _p(b[0]), b[0].nbytes, _p(b[1]), b[1].nbytes if b[1] is not None else 0,
                                 _p(b[2]), b[2].nbytes if b[2] is not None else 0, _p(r), n, passes)
        if ms < 0:
            raise RuntimeError("refocl_run: " + self.lib.refocl_error().decode(errors="replace")[:2000])
        return r[:n], ms
