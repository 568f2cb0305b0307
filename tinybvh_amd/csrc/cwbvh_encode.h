// cwbvh_encode.h — device-side BVH8_CWBVH node quantiser shared by the refit and the conversion kernels.
// Same arithmetic as the host encoder (host_builder.cpp: encode_cwbvh), which follows
// BVH8_CWBVH::ConvertFrom (tiny_bvh.h:5940-5967): origin = node box minimum, per-axis exponent = smallest e
// with extent <= 255 * 2^e, child planes floor / ceil in units of 2^e — and then every plane is checked
// against the decode lo + q * 2^e the traversal kernels use, so float rounding can never shrink a box.
#pragma once
#include "device_common.h"

namespace tbvh {

__device__ __forceinline__ float3 min3(float3 a, float3 b) { return make_float3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); }
__device__ __forceinline__ float3 max3(float3 a, float3 b) { return make_float3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }

// Octant slots of the up to 8 children of a node: greedy on cost[s][i] = dot(centroid_i - centroid_node, dir_s), the assignment of
// BVH8_CWBVH::ConvertFrom (tiny_bvh.h:5906-5938) and of the host encoder.  slotOf[i] = slot of child i, childIn[s] = child in slot s or -1.
__device__ inline void cw_assign_slots(uint32_t nk, float3 mn, float3 mx, const float3* cmn, const float3* cmx, int* slotOf, int* childIn) {
    const float3 nc = make_float3(0.5f * (mn.x + mx.x), 0.5f * (mn.y + mx.y), 0.5f * (mn.z + mx.z));
    float cost[8][8];
    for (int s = 0; s < 8; s++) { childIn[s] = -1; slotOf[s] = -1; }
    for (uint32_t i = 0; i < nk; i++) {
        const float dx = 0.5f * (cmn[i].x + cmx[i].x) - nc.x, dy = 0.5f * (cmn[i].y + cmx[i].y) - nc.y, dz = 0.5f * (cmn[i].z + cmx[i].z) - nc.z;
        for (int s = 0; s < 8; s++) cost[s][i] = ((s & 4) ? -dx : dx) + ((s & 2) ? -dy : dy) + ((s & 1) ? -dz : dz);
    }
    for (uint32_t k = 0; k < nk; k++) {
        float best = 1e30f; int bs = -1, bi = -1;
        for (int s = 0; s < 8; s++) if (childIn[s] < 0)
            for (uint32_t i = 0; i < nk; i++) if (slotOf[i] < 0 && cost[s][i] < best) { best = cost[s][i]; bs = s; bi = (int)i; }
        if (bs < 0) { for (int s = 0; s < 8 && bs < 0; s++) if (childIn[s] < 0) for (uint32_t i = 0; i < nk; i++) if (slotOf[i] < 0) { bs = s; bi = (int)i; break; } }   // NaN boxes: any free pair
        slotOf[bi] = bs; childIn[bs] = bi;
    }
}

// Writes the five float4 of one node.  cmn/cmx/used: the child boxes per slot; triBase in float4 blocks.
__device__ inline void cw_quantize_write(float4* __restrict__ np, float3 mn, float3 mx, const float3* cmn, const float3* cmx, const bool* used,
                                         uint32_t imask, uint32_t childBase, uint32_t triBase, uint32_t meta0, uint32_t meta1) {
    const float lo[3] = {mn.x, mn.y, mn.z}, hi[3] = {mx.x, mx.y, mx.z};
    int e[3];
    uint32_t q[6][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};   // qlo_x qlo_y qlo_z qhi_x qhi_y qhi_z, 8 bytes each
    for (int a = 0; a < 3; a++) {
        const float ext = hi[a] - lo[a];
        // smallest e with extent <= 255 * 2^e: start at floor(log2(extent / 255)) and let the same guard as the host
        // encoder (no child plane may need more than 255 steps, the far face must reach the box) raise it
        int ea = ext > 0 ? ilogbf(ext * (1.0f / 255.0f)) : -126;
        if (ea < -126) ea = -126;
        for (;;) {
            const float sc = ldexpf(1.0f, -ea);
            bool ok = true;
            for (int s = 0; s < 8; s++) if (used[s]) {
                const float cm = a == 0 ? cmx[s].x : a == 1 ? cmx[s].y : cmx[s].z;
                if (ceilf((cm - lo[a]) * sc) > 255.f) ok = false;
            }
            if (lo[a] + ldexpf(255.0f, ea) < hi[a]) ok = false;
            if (ok || ea >= 127) break;
            ea++;
        }
        e[a] = ea;
        const float inv = ldexpf(1.0f, -ea), sc = ldexpf(1.0f, ea);
        for (int s = 0; s < 8; s++) if (used[s]) {
            const float cl = a == 0 ? cmn[s].x : a == 1 ? cmn[s].y : cmn[s].z;
            const float ch = a == 0 ? cmx[s].x : a == 1 ? cmx[s].y : cmx[s].z;
            int ql = (int)floorf((cl - lo[a]) * inv), qh = (int)ceilf((ch - lo[a]) * inv);
            ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql); qh = qh < 0 ? 0 : (qh > 255 ? 255 : qh);
            while (ql > 0 && lo[a] + sc * (float)ql > cl) ql--;
            while (qh < 255 && lo[a] + sc * (float)qh < ch) qh++;
            q[a][s >> 2] |= (uint32_t)ql << (8 * (s & 3));
            q[3 + a][s >> 2] |= (uint32_t)qh << (8 * (s & 3));
        }
    }
    const uint32_t eim = ((uint32_t)(uint8_t)(int8_t)e[0]) | ((uint32_t)(uint8_t)(int8_t)e[1] << 8) | ((uint32_t)(uint8_t)(int8_t)e[2] << 16) | (imask << 24);
    np[0] = make_float4(lo[0], lo[1], lo[2], as_f32(eim));
    np[1] = make_float4(as_f32(childBase), as_f32(triBase), as_f32(meta0), as_f32(meta1));
    np[2] = make_float4(as_f32(q[0][0]), as_f32(q[0][1]), as_f32(q[1][0]), as_f32(q[1][1]));
    np[3] = make_float4(as_f32(q[2][0]), as_f32(q[2][1]), as_f32(q[3][0]), as_f32(q[3][1]));
    np[4] = make_float4(as_f32(q[4][0]), as_f32(q[4][1]), as_f32(q[5][0]), as_f32(q[5][1]));
}

}  // namespace tbvh
