// bvh4_encode.h — device-side quantisation of a BVH4_GPU node (tiny_bvh.h:5196-5231; host_builder.cpp: encode_bvh4_gpu), shared
// by the BVH2 -> BVH4_GPU conversion (kernels_convert.hip) and the 4-wide TLAS builder (kernels_tlas4.hip).
//
// Node = 4 blocks: {bmin.xyz | qxmin[4]} {ext/255 .xyz | qxmax[4]} {qymin, qymax, qzmin, qzmax} {childInfo[4]}; a child plane decodes
// as bmin + (ext/255) * q.  The reference scales with 254.999 / extent, which can fall 4e-6 relative short of the far face; here the
// decode step is nudged up until 255 steps really reach it and every plane is verified against the decode (conservative boxes).
#pragma once
#include "device_common.h"

namespace tbvh {

struct Bvh4Frame { float bmn[3], e255[3], scale[3], guard[3]; };

__device__ __forceinline__ Bvh4Frame bvh4_frame(float3 mn, float3 mx) {
    Bvh4Frame f;
    const float bmx[3] = {mx.x, mx.y, mx.z};
    f.bmn[0] = mn.x; f.bmn[1] = mn.y; f.bmn[2] = mn.z;
    for (int a = 0; a < 3; a++) {
        const float ext = bmx[a] - f.bmn[a];
        f.scale[a] = ext > 1e-10f ? 254.999f / ext : 0.f;
        f.e255[a] = ext * (1.0f / 255.0f);
        f.guard[a] = 4e-7f * fmaxf(fmaxf(fabsf(f.bmn[a]), fabsf(bmx[a])), ext);
        if (ext > 0) {   // the decode step must carry 255 steps past the far face: jump there, then settle ulp by ulp
            const float need = ((bmx[a] + f.guard[a]) - f.bmn[a]) * (1.0f / 255.0f);
            if (need > f.e255[a]) f.e255[a] = need;
            while (f.bmn[a] + f.e255[a] * 255.0f < bmx[a] + f.guard[a]) f.e255[a] = nextafterf(f.e255[a], 1e30f);
        }
    }
    return f;
}

// quantised planes of child i (byte i of the six words q: xmin, xmax, ymin, ymax, zmin, zmax)
__device__ __forceinline__ void bvh4_quantize_child(const Bvh4Frame& f, float3 cmn3, float3 cmx3, uint32_t i, uint32_t q[6]) {
    const float cmn[3] = {cmn3.x, cmn3.y, cmn3.z}, cmx[3] = {cmx3.x, cmx3.y, cmx3.z};
    for (int a = 0; a < 3; a++) {
        int lo = (int)floorf((cmn[a] - f.bmn[a]) * f.scale[a]), hi = (int)ceilf((cmx[a] - f.bmn[a]) * f.scale[a]);
        lo = lo < 0 ? 0 : (lo > 255 ? 255 : lo); hi = hi < 0 ? 0 : (hi > 255 ? 255 : hi);
        while (lo > 0 && f.bmn[a] + f.e255[a] * (float)lo > cmn[a] - f.guard[a]) lo--;
        while (hi < 255 && f.bmn[a] + f.e255[a] * (float)hi < cmx[a] + f.guard[a]) hi++;
        q[2 * a] |= (uint32_t)lo << (8 * i); q[2 * a + 1] |= (uint32_t)hi << (8 * i);
    }
}

__device__ __forceinline__ void bvh4_write_node(float4* nb, const Bvh4Frame& f, const uint32_t q[6], const uint32_t info[4]) {
    nb[0] = make_float4(f.bmn[0], f.bmn[1], f.bmn[2], as_f32(q[0]));
    nb[1] = make_float4(f.e255[0], f.e255[1], f.e255[2], as_f32(q[1]));
    nb[2] = make_float4(as_f32(q[2]), as_f32(q[3]), as_f32(q[4]), as_f32(q[5]));
    nb[3] = make_float4(as_f32(info[0]), as_f32(info[1]), as_f32(info[2]), as_f32(info[3]));
}

// whole node at once: frame [mn, mx], children i with used[i] get their boxes quantised, the others stay 0 (info 0 = no child)
__device__ __forceinline__ void bvh4_quantize_write(float4* nb, float3 mn, float3 mx, const float3 cmn[4], const float3 cmx[4], const bool used[4], const uint32_t info[4]) {
    const Bvh4Frame f = bvh4_frame(mn, mx);
    uint32_t q[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t i = 0; i < 4; i++) if (used[i]) bvh4_quantize_child(f, cmn[i], cmx[i], i, q);
    bvh4_write_node(nb, f, q, info);
}

}  // namespace tbvh
