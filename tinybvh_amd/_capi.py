"""ctypes binding of the C ABI declared in include/tinybvh_amd.h.

The shared library is built in-tree (tinybvh_amd/csrc/Makefile ->
tinybvh_amd/libtinybvh_amd.so).  There is no CPU fallback: if the library is missing or a
symbol is absent, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TBVH_LIB_OVERRIDE") or os.path.join(_HERE, "libtinybvh_amd.so")   # override: A/B builds in tools/

# every symbol include/tinybvh_amd.h declares: name -> (restype, argtypes)
_u64, _u32, _vp, _i = C.c_uint64, C.c_uint32, C.c_void_p, C.c_int
_pp = C.POINTER(C.c_void_p)


class BuildParams(C.Structure):
    _fields_ = [("bins", _u32), ("max_leaf_tris", _u32), ("threads", _u32), ("flags", _u32)]


class Camera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("p1", C.c_float * 3), ("p2", C.c_float * 3), ("p3", C.c_float * 3),
                ("width", _u32), ("height", _u32), ("spp_x", _u32), ("spp_y", _u32)]


class WfParams(C.Structure):
    _fields_ = [("light_pos", C.c_float * 3), ("light_color", C.c_float * 3), ("sky_lo", C.c_float * 3), ("sky_hi", C.c_float * 3),
                ("eps", C.c_float), ("max_depth", _u32), ("seed", _u32), ("clear", _u32), ("light_size", C.c_float * 2), ("flags", _u32), ("sample_index", _u32)]


class WfStats(C.Structure):
    _fields_ = [("extend_rays", _u64 * 8), ("shadow_rays", _u64 * 8), ("frame_ms", C.c_float)]


SYMBOLS = {
    "tbvh_debug_wide_copy_bvh2": (_i, [_i, _vp, _u64, _vp, _u64, _vp, _u64, _u32, _vp, _u64, C.POINTER(_u64), _vp, _u64, C.POINTER(_u64)]),
    "tbvh_wavefront_create": (_i, [_vp, _u32, _u32, _pp]),
    "tbvh_wavefront_destroy": (None, [_vp]),
    "tbvh_wavefront_render": (_i, [_vp, _vp, _vp, C.POINTER(Camera), C.POINTER(WfParams), C.POINTER(WfStats)]),
    "tbvh_wavefront_read": (_i, [_vp, _vp]),
    "tbvh_wavefront_set_blas_vertices": (_i, [_vp, _vp, _u64]),
    "tbvh_wavefront_finalize": (_i, [_vp, C.c_float, _vp]),
    "tbvh_wavefront_set_blue_noise": (_i, [_vp, _vp, _u64]),
    "tbvh_abi_version": (_i, []),
    "tbvh_last_error": (C.c_char_p, []),
    "tbvh_device_count": (_i, []),
    "tbvh_init": (_i, [_i, _pp]),
    "tbvh_shutdown": (None, [_vp]),
    "tbvh_synchronize": (_i, [_vp]),
    "tbvh_set_stream": (_i, [_vp, _vp]),
    "tbvh_set_timing": (_i, [_vp, _i]),
    "tbvh_upload_bvh_gpu": (_i, [_vp, _vp, _u64, _vp, _u64, _vp, _u64, _pp]),
    "tbvh_upload_bvh4_gpu": (_i, [_vp, _vp, _u64, _pp]),
    "tbvh_upload_cwbvh": (_i, [_vp, _vp, _u64, _vp, _u64, _pp]),
    "tbvh_upload_tlas": (_i, [_vp, _vp, _u64, _vp, _u64, _vp, _u64, _pp, _u64, _pp]),
    "tbvh_update_tlas": (_i, [_vp, _vp, _u64, _vp, _u64, _vp, _u64]),
    "tbvh_update_bvh_gpu": (_i, [_vp, _vp, _u64, _vp, _u64, _vp, _u64]),
    "tbvh_update_bvh4_gpu": (_i, [_vp, _vp, _u64]),
    "tbvh_update_cwbvh": (_i, [_vp, _vp, _u64, _vp, _u64]),
    "tbvh_rebuild_tlas_device": (_i, [_vp, _vp, _i, _vp, _u64]),
    "tbvh_refit": (_i, [_vp, _vp, _u64, _i]),
    "tbvh_set_opacity_micromaps": (_i, [_vp, _vp, _u32, _u64, _i]),
    "tbvh_scene_download": (_i, [_vp, _i, _vp, _u64, _vp]),
    "tbvh_build_device": (_i, [_vp, _vp, _u64, _i, _i, _u32, _vp]),
    "tbvh_build_device_ploc": (_i, [_vp, _vp, _u64, _i, _i, _u32, _vp]),
    "tbvh_convert_bvh2_device": (_i, [_vp, _vp, _u64, _vp, _u64, _vp, _u64, _i, _i, _vp]),
    "tbvh_tlas_download": (_i, [_vp, _vp, _u64, _vp, _u64, _vp, _u64, _vp]),
    "tbvh_free_scene": (None, [_vp]),
    "tbvh_scene_layout": (_i, [_vp]),
    "tbvh_scene_device_bytes": (_u64, [_vp]),
    "tbvh_intersect": (_i, [_vp, _vp, _u64, _u32]),
    "tbvh_occluded": (_i, [_vp, _vp, _u64, _u32, _vp]),
    "tbvh_intersect_sharded": (_i, [_vp, _u32, _vp, _u64, _u32]),
    "tbvh_occluded_sharded": (_i, [_vp, _u32, _vp, _u64, _u32, _vp]),
    "tbvh_intersect_sharded_device": (_i, [_vp, _u32, _vp, _vp, _i, C.c_float, _vp, _vp]),
    "tbvh_occluded_sharded_device": (_i, [_vp, _u32, _vp, _vp, _vp, _vp, _vp]),
    "tbvh_wavefront_set_band": (_i, [_vp, _u32, _u32]),
    "tbvh_wavefront_render_sharded": (_i, [_vp, _vp, _vp, _u32, C.POINTER(Camera), _vp, _vp, _vp]),
    "tbvh_wavefront_read_sharded": (_i, [_vp, _u32, _vp]),
    "tbvh_shard_range": (None, [_u64, _u32, _u32, C.POINTER(_u64), C.POINTER(_u64)]),
    "tbvh_intersect_device": (_i, [_vp, _vp, _u64]),
    "tbvh_occluded_device": (_i, [_vp, _vp, _u64, _vp]),
    "tbvh_intersect_device_fresh": (_i, [_vp, _vp, _u64, C.c_float]),
    "tbvh_reset_hits_device": (_i, [_vp, _vp, _u64, C.c_float]),
    "tbvh_time_last_ms": (C.c_float, [_vp]),
    "tbvh_debug_coherent_schedule": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_uint32)]),
    "tbvh_time_history": (C.c_int, [_vp, C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32)]),
    "tbvh_measure_copy_bandwidth": (_i, [_vp, _u64, _u32, C.POINTER(C.c_double)]),
    "tbvh_measure_read_bandwidth": (_i, [_vp, _u64, _u32, C.POINTER(C.c_double)]),
    "tbvh_measure_valu_issue": (_i, [_vp, _u32, C.POINTER(C.c_double)]),
    "tbvh_measure_link_bandwidth": (_i, [_vp, _u64, _u32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "tbvh_scene_get_schedule_hint": (_i, [_vp, _vp]),
    "tbvh_scene_set_schedule_hint": (_i, [_vp, _vp]),
    "tbvh_pinned_malloc": (_i, [_vp, _u64, _pp]),
    "tbvh_pinned_free": (_i, [_vp, _vp]),
    "tbvh_set_variant": (_i, [_vp, _i]),
    "tbvh_debug_stats": (_i, [_vp, _vp, _i]),
    "tbvh_debug_last_probe": (_i, [_vp, _vp]),
    "tbvh_debug_set_flags": (_i, [_vp, _u32]),
    "tbvh_cwbvh_set_hybrid": (_i, [_vp, C.c_int64]),
    "tbvh_bin_rays_device": (_i, [_vp, _vp, _vp, _u64, C.POINTER(C.c_float), _u32, _u32, _vp]),
    "tbvh_generate_primary_device": (_i, [_vp, C.POINTER(Camera), _vp, _u64, _u64]),
    "tbvh_generate_bounce_device": (_i, [_vp, _vp, _vp, _vp, _u64, _u32]),
    "tbvh_generate_shadow_device": (_i, [_vp, _vp, _vp, _u64, C.POINTER(C.c_float), C.c_float]),
    "tbvh_device_malloc": (_i, [_vp, _u64, _pp]),
    "tbvh_device_free": (_i, [_vp, _vp]),
    "tbvh_copy_to_device": (_i, [_vp, _vp, _vp, _u64]),
    "tbvh_copy_from_device": (_i, [_vp, _vp, _vp, _u64]),
    "tbvh_host_build": (_i, [_vp, _u64, _i, C.POINTER(BuildParams), _pp]),
    "tbvh_host_build_tlas": (_i, [_vp, _u64, _vp, _u64, _pp]),
    "tbvh_host_free": (None, [_vp]),
    "tbvh_cwbvh_file_write": (_i, [C.c_char_p, _vp, _u64, _vp, _u64, _u64, _vp]),
    "tbvh_cwbvh_file_read": (_i, [C.c_char_p, _u64, _pp, C.POINTER(_u64)]),
    "tbvh_host_layout": (_i, [_vp]),
    "tbvh_host_blob": (_vp, [_vp, _i]),
    "tbvh_host_blob_count": (_u64, [_vp, _i]),
    "tbvh_upload_host": (_i, [_vp, _vp, _vp, _u64, _pp]),
}


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C tinybvh_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and the header disagree
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class TbvhError(RuntimeError):
    def __init__(self, code: int, where: str):
        msg = lib.tbvh_last_error()
        super().__init__(f"{where}: error {code}: {msg.decode() if msg else ''}")
        self.code = code


def check(code: int, where: str) -> None:
    if code != 0:
        raise TbvhError(code, where)
