"""The two-level kernels next to the reference's own GPU TLAS traversal (traverse_tlas.cl:13-107, reached through wavefront2.cl's
Extend kernel on ROCm OpenCL — oracle/ref_ocl.cpp: refocl_tlas_extend): same TLAS nodes, instance records, BLAS blobs and rays.
The CPU oracle (BVH::IntersectTLAS, bit-exact) is what test_tlas.py pins the records to; this one shows the same agreement with the kernel
the library replaces.  The reference derives rD with native_recip on the device and folds the instance into prim (<< 24), so the
comparison is hit / miss on every ray and t to 1e-4 relative."""
import os

import numpy as np
import pytest

import tinybvh_amd as tb
from tinybvh_amd import rays as R
from tinybvh_amd import scenes
from oracle_lib import REFOCL_PATH, ReferenceOpenCL

pytestmark = pytest.mark.gpu


def test_tlas_over_cwbvh_instances_agrees_with_traverse_tlas_cl(ctx):
    if not os.path.exists(REFOCL_PATH):
        pytest.skip("oracle/_ref/libtinybvh_refocl.so not built (needs the reference checkout at build time)")
    try:
        ocl = ReferenceOpenCL()
    except (RuntimeError, OSError) as e:
        pytest.skip(f"no OpenCL runtime for the reference kernels: {e}")
    verts = scenes.blob(8_000, seed=4)
    blas = tb.BVH8_CWBVH(ctx).Build(verts)
    g = np.stack(np.meshgrid(np.arange(4), np.arange(4), np.arange(4), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    k = g.shape[0]
    ang = (0.3 + np.arange(k) * 0.37).astype(np.float32)
    T = np.zeros((k, 4, 4), np.float32)
    T[:, 0, 0] = np.cos(ang) * 0.7; T[:, 0, 2] = np.sin(ang) * 0.7; T[:, 1, 1] = 0.7; T[:, 2, 0] = -np.sin(ang) * 0.7; T[:, 2, 2] = np.cos(ang) * 0.7; T[:, 3, 3] = 1
    T[:, :3, 3] = g * 2.0
    inst = tb.make_instances(T, np.zeros(k, np.uint32))
    tlas = tb.TLAS(ctx).Build(inst, [blas])
    nodes, idx, irec = tlas.Download()
    rays = R.random_rays(65536, (-1.0, -1.0, -1.0), (8.0, 8.0, 8.0), seed=21)
    mine = tlas.Intersect(rays.copy())
    h = blas.host
    ref, _ = ocl.tlas_extend(nodes, idx, irec, h.blob(0, np.uint32, 4), h.blob(1, np.uint32, 4), rays, passes=1)
    mh, rh = mine["t"] < 1e30, ref[:, 0] < 1e30
    assert mh.sum() > 5000
    assert int((mh != rh).sum()) <= 2
    both = mh & rh
    rel = np.abs(mine["t"][both] - ref[both, 0]) / np.maximum(np.abs(ref[both, 0]), 1e-20)
    assert int((rel > 1e-4).sum()) <= max(4, int(both.sum()) // 2000), int((rel > 1e-4).sum())
    # the instance the reference folded into prim's top byte is the library's hit.inst (64 instances fit in 8 bits)
    pr = ref[both, 3].view(np.uint32)
    same_t = rel <= 1e-6
    assert np.array_equal((pr >> 24)[same_t], mine["inst"][both][same_t] & 0xFF)
    tlas.free(); blas.free()
