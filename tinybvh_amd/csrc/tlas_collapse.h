// tlas_collapse.h — reading a caller's TLAS (BVH_GPU / Aila-Laine nodes over BLASInstance records, tiny_bvh.h:4575-4581) as a tree
// of "kids" for the wide-TLAS builders (kernels_tlas4.hip: 4-wide in the BVH4_GPU node format; kernels_tlas8.hip: 8-wide in the
// BVH8_CWBVH node format).  A kid is a subtree with its box: an interior AL node, or a range of the instance index list (an AL leaf
// with more than one instance — the reference's builder may leave several — is split in halves until one instance is left).
#pragma once
#include "device_common.h"

namespace tbvh {

struct Kid { float3 mn, mx; uint32_t ref, cnt; };   // cnt == 0xffffffff: AL interior node `ref`; else instances idx[ref .. ref + cnt)

__device__ __forceinline__ float kid_area(const Kid& k) {
    const float ex = k.mx.x - k.mn.x, ey = k.mx.y - k.mn.y, ez = k.mx.z - k.mn.z;
    return ex * ey + ey * ez + ez * ex;
}
__device__ __forceinline__ Kid range_kid(const uint32_t* __restrict__ idx, const float4* __restrict__ inst, uint32_t first, uint32_t cnt) {
    Kid k; k.ref = first; k.cnt = cnt;
    k.mn = make_float3(1e30f, 1e30f, 1e30f); k.mx = make_float3(-1e30f, -1e30f, -1e30f);
    for (uint32_t j = 0; j < cnt; j++) {
        const float4* ip = inst + (size_t)idx[first + j] * 12;
        const float4 a = ip[8], b = ip[9];
        k.mn = make_float3(fminf(k.mn.x, a.x), fminf(k.mn.y, a.y), fminf(k.mn.z, a.z));
        k.mx = make_float3(fmaxf(k.mx.x, b.x), fmaxf(k.mx.y, b.y), fmaxf(k.mx.z, b.z));
    }
    return k;
}
// the two children of AL node `a` as kids (a child that is an AL leaf becomes the instance range it lists)
__device__ __forceinline__ void al_children(const float4* __restrict__ al, uint32_t nAL, uint32_t a, Kid& l, Kid& r) {
    const float4 n0 = al[(size_t)a * 4], n1 = al[(size_t)a * 4 + 1], n2 = al[(size_t)a * 4 + 2], n3 = al[(size_t)a * 4 + 3];
    l.mn = make_float3(n0.x, n0.y, n0.z); l.mx = make_float3(n1.x, n1.y, n1.z); l.ref = as_u32(n0.w); l.cnt = 0xffffffffu;
    r.mn = make_float3(n2.x, n2.y, n2.z); r.mx = make_float3(n3.x, n3.y, n3.z); r.ref = as_u32(n1.w); r.cnt = 0xffffffffu;
    for (Kid* k : {&l, &r}) {
        if (k->ref >= nAL) { k->ref = 0; k->cnt = 0; continue; }   // malformed: an empty child
        const uint32_t tc = as_u32(al[(size_t)k->ref * 4 + 2].w);
        if (tc) { k->cnt = tc; k->ref = as_u32(al[(size_t)k->ref * 4 + 3].w); }
    }
}

}  // namespace tbvh
