// kernels_tlas.hip — two-level (TLAS / BLAS) Intersect and IsOccluded for gfx950.
//
// Replaces traverse_tlas / isoccluded_tlas (traverse_tlas.cl:13-193) from scratch, with the
// semantics of the CPU oracle BVH::IntersectTLAS (tiny_bvh.h:3306-3380):
//   * TLAS = BVH_GPU (Aila-Laine) nodes over BLASInstance records (192 bytes, tiny_bvh.h:1443-1457);
//     a TLAS leaf lists instance indices (tlas.bvh.primIdx);
//   * per instance: skip unless (inst.mask & ray.mask) (3326); O' = invTransform * O with the
//     w divide of tinybvh_transform_point (tiny_bvh.h:513-522), D' = invTransform3x3 * D, NOT
//     re-normalised, so t stays in world units (3329-3333); rD' = tinybvh_safercp(D');
//   * the BLAS is traversed with the transformed ray and the current hit carried in and out;
//     on acceptance hit.inst = instance index (INST_IDX_BITS == 32: byte 44 of the ray record,
//     tiny_bvh.h:665, 8526) — a full 32-bit id, unlike the reference device code's
//     prim | inst << 24 packing (traverse_tlas.cl:77) that aliases above 256 instances.
// BLAS layouts: BVH8_CWBVH and BVH4_GPU (all BLASes of one TLAS share a layout).  The point
// and vector transforms are written as the same unfused mul/add chains as the oracle
// (oracle/tbvh_oracle.c: orc_xform_point / orc_xform_vec), and this file is built with
// -ffp-contract=off, so the transformed ray — and therefore t,u,v — match bit for bit.
#include "device_common.h"
#include "lane_stack.h"
#include "ray_pool.h"
#include "kernels.h"

namespace tbvh {

namespace {

constexpr int WG = 64;

__device__ __forceinline__ float fmin3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
__device__ __forceinline__ float fmax3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ uint32_t sext_s8x4(uint32_t i) { return ((i >> 7) & 0x01010101u) * 0xffu; }
__device__ __forceinline__ float safercp(float x) {
    if (x > 1e-12f || x < -1e-12f) return 1.0f / x;
    return x >= 0 ? kFar : -kFar;
}

template <int LDS_N> using StackT = LaneStack<uint2, LDS_N, WG>;   // LDS top + global spill (lane_stack.h)

struct RayL {  // a ray in some space + its current best hit
    float3 O, D, rD;
    float4 hit;   // t, u, v, prim
    bool found;
};

// ---- BLAS traversals; each runs until the stack is back at `base` ------------------------------

template <bool ANYHIT, int LDS_N>
__device__ __forceinline__ void blas_cwbvh(const GlobalF4 nodes, const GlobalF4 tris, RayL& r, StackT<LDS_N>& st, const Omm om) {
    const int base = st.sp;
    const uint32_t oct = 7u - ((r.D.x < 0 ? 4u : 0u) | (r.D.y < 0 ? 2u : 0u) | (r.D.z < 0 ? 1u : 0u));
    const uint32_t octinv4 = oct * 0x01010101u;
    uint2 ng = make_uint2(0u, 0x80000000u), tg = make_uint2(0u, 0u);
    for (;;) {
        if (ng.y > 0x00FFFFFFu) {
            const uint32_t imask = ng.y;
            const uint32_t bit = 31u - (uint32_t)__clz(ng.y);
            const uint32_t cbase = ng.x;
            ng.y &= ~(1u << bit);
            if (ng.y > 0x00FFFFFFu) st.push(ng);
            const uint32_t slot = (bit - 24u) ^ oct;
            const uint32_t ci = (cbase + __popc(imask & ~(0xFFFFFFFFu << slot))) * 5u;
            const float4 n0 = nodes[ci], n1 = nodes[ci + 1], n2 = nodes[ci + 2], n3 = nodes[ci + 3], n4 = nodes[ci + 4];
            const uint32_t ew = as_u32(n0.w);
            const float ax = ldexpf(r.rD.x, (int)(int8_t)(ew)), ay = ldexpf(r.rD.y, (int)(int8_t)(ew >> 8)), az = ldexpf(r.rD.z, (int)(int8_t)(ew >> 16));
            const float ox = (n0.x - r.O.x) * r.rD.x, oy = (n0.y - r.O.y) * r.rD.y, oz = (n0.z - r.O.z) * r.rD.z;
            uint32_t hitmask = 0;
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const uint32_t meta4 = half ? as_u32(n1.w) : as_u32(n1.z);
                const uint32_t inner4 = (meta4 & (meta4 << 1)) & 0x10101010u;
                const uint32_t imask4 = sext_s8x4(inner4 << 3);
                const uint32_t bitidx4 = (meta4 ^ (octinv4 & imask4)) & 0x1F1F1F1Fu;
                const uint32_t bits4 = (meta4 >> 5) & 0x07070707u;
                const uint32_t qlx = half ? as_u32(n2.y) : as_u32(n2.x), qhx = half ? as_u32(n3.w) : as_u32(n3.z);
                const uint32_t qly = half ? as_u32(n2.w) : as_u32(n2.z), qhy = half ? as_u32(n4.y) : as_u32(n4.x);
                const uint32_t qlz = half ? as_u32(n3.y) : as_u32(n3.x), qhz = half ? as_u32(n4.w) : as_u32(n4.z);
                const uint32_t lox = r.rD.x < 0 ? qhx : qlx, hix = r.rD.x < 0 ? qlx : qhx;
                const uint32_t loy = r.rD.y < 0 ? qhy : qly, hiy = r.rD.y < 0 ? qly : qhy;
                const uint32_t loz = r.rD.z < 0 ? qhz : qlz, hiz = r.rD.z < 0 ? qlz : qhz;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int sh = 8 * i;
                    const float tnx = __builtin_fmaf((float)((lox >> sh) & 255), ax, ox), tfx = __builtin_fmaf((float)((hix >> sh) & 255), ax, ox);
                    const float tny = __builtin_fmaf((float)((loy >> sh) & 255), ay, oy), tfy = __builtin_fmaf((float)((hiy >> sh) & 255), ay, oy);
                    const float tnz = __builtin_fmaf((float)((loz >> sh) & 255), az, oz), tfz = __builtin_fmaf((float)((hiz >> sh) & 255), az, oz);
                    const float cmin = __builtin_fmaxf(fmax3(tnx, tny, tnz), 0.0f);
                    const float cmax = __builtin_fminf(fmin3(tfx, tfy, tfz), r.hit.x);
                    if (cmin <= cmax) hitmask |= ((bits4 >> sh) & 255u) << ((bitidx4 >> sh) & 255u);
                }
            }
            ng.x = as_u32(n1.x); tg.x = as_u32(n1.y);
            ng.y = (hitmask & 0xFF000000u) | (ew >> 24);
            tg.y = hitmask & 0x00FFFFFFu;
        } else {
            tg = ng;
            ng = make_uint2(0u, 0u);
        }
        while (tg.y != 0) {
            const uint32_t ti = 31u - (uint32_t)__clz(tg.y);
            tg.y &= ~(1u << ti);
            const uint32_t ta = tg.x + ti * 3u;
            const float4 e2 = tris[ta], e1 = tris[ta + 1], v0 = tris[ta + 2];
            TriHit h;
            if (tri_test(r.O, r.D, xyz(v0), xyz(e1), xyz(e2), r.hit.x, h, om, as_u32(v0.w))) {
                r.found = true;
                if (ANYHIT) { st.sp = base; return; }
                r.hit = make_float4(h.t, h.u, h.v, v0.w);
            }
        }
        if (ng.y > 0x00FFFFFFu) continue;
        if (st.sp == base) return;
        ng = st.pop();
    }
}

template <bool ANYHIT, int LDS_N>
__device__ __forceinline__ void blas_bvh4(const GlobalF4 data, RayL& r, StackT<LDS_N>& st, const Omm om) {
    const int base = st.sp;
    uint32_t offset = 0;
    for (;;) {
        const float4 d0 = data[offset], d1 = data[offset + 1], d2 = data[offset + 2], d3 = data[offset + 3];
        const float sx = d1.x * r.rD.x, sy = d1.y * r.rD.y, sz = d1.z * r.rD.z;
        const float bx = (d0.x - r.O.x) * r.rD.x, by = (d0.y - r.O.y) * r.rD.y, bz = (d0.z - r.O.z) * r.rD.z;
        const uint32_t qx0 = as_u32(d0.w), qx1 = as_u32(d1.w);
        const uint32_t qy0 = as_u32(d2.x), qy1 = as_u32(d2.y), qz0 = as_u32(d2.z), qz1 = as_u32(d2.w);
        float dist[4];
        uint32_t info[4] = { as_u32(d3.x), as_u32(d3.y), as_u32(d3.z), as_u32(d3.w) };
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int sh = 8 * i;
            const float x1 = __builtin_fmaf((float)((qx0 >> sh) & 255), sx, bx), x2 = __builtin_fmaf((float)((qx1 >> sh) & 255), sx, bx);
            const float y1 = __builtin_fmaf((float)((qy0 >> sh) & 255), sy, by), y2 = __builtin_fmaf((float)((qy1 >> sh) & 255), sy, by);
            const float z1 = __builtin_fmaf((float)((qz0 >> sh) & 255), sz, bz), z2 = __builtin_fmaf((float)((qz1 >> sh) & 255), sz, bz);
            const float tmin = __builtin_fmaxf(fmax3(__builtin_fminf(x1, x2), __builtin_fminf(y1, y2), __builtin_fminf(z1, z2)), 0.0f);
            const float tmax = __builtin_fminf(fmin3(__builtin_fmaxf(x1, x2), __builtin_fmaxf(y1, y2), __builtin_fmaxf(z1, z2)), r.hit.x);
            dist[i] = (tmin > tmax || info[i] == 0) ? kFar : tmin;
        }
#define TBVH_CSWAP(a, b) if (dist[a] < dist[b]) { const float tf = dist[a]; dist[a] = dist[b]; dist[b] = tf; const uint32_t tu = info[a]; info[a] = info[b]; info[b] = tu; }
        TBVH_CSWAP(0, 2) TBVH_CSWAP(1, 3) TBVH_CSWAP(0, 1) TBVH_CSWAP(2, 3) TBVH_CSWAP(1, 2)
#undef TBVH_CSWAP
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (dist[i] < kFar && !(info[i] & 0x80000000u)) st.push(make_uint2(info[i], 0u));
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (!(dist[i] < kFar) || !(info[i] & 0x80000000u)) continue;
            const uint32_t N = (info[i] >> 16) & 0x7fff;
            uint32_t ta = offset + (info[i] & 0xffff);
            for (uint32_t j = 0; j < N; j++, ta += 3) {
                const float4 v0 = data[ta], e1 = data[ta + 1], e2 = data[ta + 2];
                TriHit h;
                if (tri_test(r.O, r.D, xyz(v0), xyz(e1), xyz(e2), r.hit.x, h, om, as_u32(v0.w))) {
                    r.found = true;
                    if (ANYHIT) { st.sp = base; return; }
                    r.hit = make_float4(h.t, h.u, h.v, v0.w);
                }
            }
        }
        if (st.sp == base) return;
        offset = st.pop().x;
    }
}

// instance record = BLASInstance, 12 float4 (192 bytes)
template <bool ANYHIT, int BLAS_LAYOUT, int LDS_N>
__device__ __forceinline__ void tlas_body(const float4* __restrict__ tlasNodes, const uint32_t* __restrict__ tlasIdx,
                                          const float4* __restrict__ instances, const BlasDesc* __restrict__ blas, const QueryArgs& q,
                                          uint32_t* __restrict__ status) {
    __shared__ uint2 stk[LDS_N][WG];
    StackT<LDS_N> st;
    st.init(&stk[0][threadIdx.x], (uint2*)q.spill + (blockIdx.x * WG + threadIdx.x), (size_t)gridDim.x * WG, q.spillStride);
    RayPool<64> pool;
    const uint64_t nRaysTotal = q.nRaysDev ? *q.nRaysDev : q.nRays;   // batch size may live on the device (wavefront queues)
    pool.init(q.poolParts);
    for (;;) {
        // whole-wave batches here: the nested TLAS/BLAS loops keep per-lane state in registers
        uint64_t ri = 0;
        const bool got = pool.acquire(true, q.counter, nRaysTotal, ri);
        if (__ballot(got) == 0) break;
        if (!got) continue;
        RayRec* rp = q.rays + ri;
        const float3 O = xyz(rp->O), D = xyz(rp->D), rD = xyz(rp->rD);
        const uint32_t rayMask = as_u32(rp->O.w);
        float4 hit = q.fresh ? make_float4(q.freshTmax, 0.f, 0.f, 0.f) : rp->hit;
        uint32_t hitInst = as_u32(rp->rD.w);
        bool found = false;
        const float3 ro = make_float3(O.x * rD.x, O.y * rD.y, O.z * rD.z);
        st.sp = 0;
        uint32_t node = 0;
        for (;;) {
            const float4 n0 = tlasNodes[node * 4], n1 = tlasNodes[node * 4 + 1], n2 = tlasNodes[node * 4 + 2], n3 = tlasNodes[node * 4 + 3];
            const uint32_t cnt = as_u32(n2.w);
            if (cnt) {
                const uint32_t first = as_u32(n3.w);
                for (uint32_t i = 0; i < cnt; i++) {
                    const uint32_t ii = tlasIdx[first + i];
                    const float4* ip = instances + (size_t)ii * 12;
                    const float4 b0 = ip[8], b1 = ip[9];           // aabbMin|blasIdx, aabbMax|mask
                    if (!(as_u32(b1.w) & rayMask)) continue;       // tiny_bvh.h:3326
                    const float4 r0 = ip[4], r1 = ip[5], r2 = ip[6], r3 = ip[7];   // invTransform rows
                    RayL rl;
                    {   // tinybvh_transform_point (tiny_bvh.h:512-522) with the reference build's contraction (first product fused
                        // into the first addition, the second rounded, the third fused, translation added last): the oracle
                        // restates it the same way and is pinned bit for bit against the real IntersectTLAS
                        const float px = __builtin_fmaf(r0.z, O.z, __builtin_fmaf(r0.x, O.x, r0.y * O.y)) + r0.w;
                        const float py = __builtin_fmaf(r1.z, O.z, __builtin_fmaf(r1.x, O.x, r1.y * O.y)) + r1.w;
                        const float pz = __builtin_fmaf(r2.z, O.z, __builtin_fmaf(r2.x, O.x, r2.y * O.y)) + r2.w;
                        const float w = __builtin_fmaf(r3.z, O.z, __builtin_fmaf(r3.x, O.x, r3.y * O.y)) + r3.w;
                        if (w == 1) rl.O = make_float3(px, py, pz);
                        else { const float iw = 1.f / w; rl.O = make_float3(px * iw, py * iw, pz * iw); }
                    }
                    rl.D = make_float3(__builtin_fmaf(r0.z, D.z, __builtin_fmaf(r0.x, D.x, r0.y * D.y)), __builtin_fmaf(r1.z, D.z, __builtin_fmaf(r1.x, D.x, r1.y * D.y)),
                                       __builtin_fmaf(r2.z, D.z, __builtin_fmaf(r2.x, D.x, r2.y * D.y)));   // tinybvh_transform_vector (523-528), same contraction
                    rl.rD = make_float3(safercp(rl.D.x), safercp(rl.D.y), safercp(rl.D.z));
                    rl.hit = hit; rl.found = false;
                    const BlasDesc bd = blas[as_u32(b0.w)];
                    if (BLAS_LAYOUT == 9) blas_cwbvh<ANYHIT, LDS_N>(GlobalF4(bd.nodes), GlobalF4(bd.tris), rl, st, Omm{bd.opmap, bd.opmapN});
                    else blas_bvh4<ANYHIT, LDS_N>(GlobalF4(bd.nodes), rl, st, Omm{bd.opmap, bd.opmapN});
                    if (rl.found) { found = true; hit = rl.hit; hitInst = ii; if (ANYHIT) break; }
                }
                if (ANYHIT && found) break;
                if (st.sp == 0) break;
                node = st.pop().x;
                continue;
            }
            // SLAB_TEST_TWO_NODES form (tiny_bvh.h:3202-3220), as in the BVH_GPU kernel
            const float lx1 = __builtin_fmaf(n0.x, rD.x, -ro.x), lx2 = __builtin_fmaf(n1.x, rD.x, -ro.x);
            const float ly1 = __builtin_fmaf(n0.y, rD.y, -ro.y), ly2 = __builtin_fmaf(n1.y, rD.y, -ro.y);
            const float lz1 = __builtin_fmaf(n0.z, rD.z, -ro.z), lz2 = __builtin_fmaf(n1.z, rD.z, -ro.z);
            const float rx1 = __builtin_fmaf(n2.x, rD.x, -ro.x), rx2 = __builtin_fmaf(n3.x, rD.x, -ro.x);
            const float ry1 = __builtin_fmaf(n2.y, rD.y, -ro.y), ry2 = __builtin_fmaf(n3.y, rD.y, -ro.y);
            const float rz1 = __builtin_fmaf(n2.z, rD.z, -ro.z), rz2 = __builtin_fmaf(n3.z, rD.z, -ro.z);
            const float tminL = __builtin_fmaxf(fmax3(__builtin_fminf(lx1, lx2), __builtin_fminf(ly1, ly2), __builtin_fminf(lz1, lz2)), 0.0f);
            const float tmaxL = __builtin_fminf(fmin3(__builtin_fmaxf(lx1, lx2), __builtin_fmaxf(ly1, ly2), __builtin_fmaxf(lz1, lz2)), hit.x);
            const float tminR = __builtin_fmaxf(fmax3(__builtin_fminf(rx1, rx2), __builtin_fminf(ry1, ry2), __builtin_fminf(rz1, rz2)), 0.0f);
            const float tmaxR = __builtin_fminf(fmin3(__builtin_fmaxf(rx1, rx2), __builtin_fmaxf(ry1, ry2), __builtin_fmaxf(rz1, rz2)), hit.x);
            const bool hL = tmaxL >= tminL, hR = tmaxR >= tminR;
            uint32_t l = as_u32(n0.w), r = as_u32(n1.w);
            if (hL && hR) {
                if (tminL > tminR) { const uint32_t t = l; l = r; r = t; }
                st.push(make_uint2(r, 0u));
                node = l;
            } else if (hL) node = l;
            else if (hR) node = r;
            else {
                if (st.sp == 0) break;
                node = st.pop().x;
            }
        }
        if (ANYHIT) q.occluded[ri] = found ? 1 : 0;
        else if (found) { rp->hit = hit; ((uint32_t*)rp)[11] = hitInst; }   // byte 44 = hit.inst
        else if (q.fresh) rp->hit = hit;
    }
    if (st.overflow) atomicOr(status, 1u);
}

template <bool ANYHIT, int BLAS_LAYOUT, int LDS_N = 16>
__global__ __launch_bounds__(WG) void k_tlas(const float4* __restrict__ tlasNodes, const uint32_t* __restrict__ tlasIdx,
                                             const float4* __restrict__ instances, const BlasDesc* __restrict__ blas, QueryArgs q,
                                             uint32_t* __restrict__ status) {
    tlas_body<ANYHIT, BLAS_LAYOUT, LDS_N>(tlasNodes, tlasIdx, instances, blas, q, status);
}
// the same with the register budget of 5 waves per SIMD (<= 96 VGPRs; left alone the compiler takes 88-115)
template <bool ANYHIT, int BLAS_LAYOUT, int LDS_N = 16>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_tlas_w5(const float4* __restrict__ tlasNodes, const uint32_t* __restrict__ tlasIdx,
                                             const float4* __restrict__ instances, const BlasDesc* __restrict__ blas, QueryArgs q,
                                             uint32_t* __restrict__ status) {
    tlas_body<ANYHIT, BLAS_LAYOUT, LDS_N>(tlasNodes, tlasIdx, instances, blas, q, status);
}

}  // namespace

void launch_tlas(bool anyhit, int blasLayout, int variant, const float4* tlasNodes, const uint32_t* tlasIdx, const float4* instances,
                 const BlasDesc* blas, const QueryArgs& q, uint32_t* status, uint32_t blocks, hipStream_t s) {
#define TBVH_LT(K, ...)                                                                                                                      \
    do {                                                                                                                                \
        if (blasLayout == 9) {                                                                                                          \
            if (anyhit) hipLaunchKernelGGL((K<true, 9, ##__VA_ARGS__>), dim3(blocks), dim3(WG), 0, s, tlasNodes, tlasIdx, instances, blas, q, status); \
            else hipLaunchKernelGGL((K<false, 9, ##__VA_ARGS__>), dim3(blocks), dim3(WG), 0, s, tlasNodes, tlasIdx, instances, blas, q, status);       \
        } else {                                                                                                                        \
            if (anyhit) hipLaunchKernelGGL((K<true, 6, ##__VA_ARGS__>), dim3(blocks), dim3(WG), 0, s, tlasNodes, tlasIdx, instances, blas, q, status); \
            else hipLaunchKernelGGL((K<false, 6, ##__VA_ARGS__>), dim3(blocks), dim3(WG), 0, s, tlasNodes, tlasIdx, instances, blas, q, status);       \
        }                                                                                                                               \
    } while (0)
    // Default: the register budget of 5 waves per SIMD (<= 96 VGPRs; left alone the compiler takes 88-115 and runs 4) and a
    // 12-entry LDS stack top (6 KB per workgroup: 20 workgroups per CU fit the 160 KB).  1000 instances, 8.3 M camera rays:
    // BVH4_GPU BLASes 2.10 -> 1.95 ms, CWBVH BLASes 2.73 -> 2.56 ms.
    if (variant == 2) TBVH_LT(k_tlas_w5, 8);
    else if (variant == 4) TBVH_LT(k_tlas, 8);
    else if (variant == 5) TBVH_LT(k_tlas, 16);        // the former default
    else if (variant == 1) TBVH_LT(k_tlas_w5, 16);
    else TBVH_LT(k_tlas_w5, 12);
#undef TBVH_LT
}

}  // namespace tbvh
