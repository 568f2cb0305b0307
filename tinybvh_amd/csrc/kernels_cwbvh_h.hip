// kernels_cwbvh_h.hip — BVH8_CWBVH traversal on a gfx950-friendly internal node layout.
//
// The caller's blob stays the reference format (tiny_bvh.h:5884-6018).  At upload every
// 80-byte node is re-laid-out ONCE into a 128-byte, 128-byte-aligned record ("H node"):
//
//   +0   lo.xyz | ex,ey,ez,imask            (blk0 verbatim)
//   +16  childBase | triBase | meta[8]       (blk1 verbatim)
//   +32  qlo_x[8] as fp16   +48 qlo_y[8]   +64  qlo_z[8]
//   +80  qhi_x[8] as fp16   +96 qhi_y[8]   +112 qhi_z[8]
//
// Why (measured on MI355X, tools/ubench/valu_rate.hip: a wave64 VALU op issues in ~4 cycles
// and the traversal kernel sits on that VALU-issue roofline):
//   * 0..255 is exact in fp16, and v_fma_mix_f32 converts an fp16 operand inside the FMA, so
//     the 48 v_cvt_f32_ubyte of the byte format disappear: t = fma_mix(q_h, 2^e*rD, (lo-O)*rD);
//   * one plane (8 children) is exactly one 16-byte load, so "near"/"far" plane selection by
//     ray direction sign is a per-ray load OFFSET instead of 12 v_cndmask per node;
//   * one node = one 128-byte cache line (80-byte nodes straddle two lines half the time).
// Decoded values are identical to the byte format (same integers), so results are too.
//
// Schedule: persistent one-wave workgroups, one lane = one ray, per-lane ray replacement
// (ray_pool.h), at most one triangle test + one node visit per lane and iteration, traversal
// stack in LDS with a global spill area.
#include "device_common.h"
#include "lane_stack.h"
#include "ray_pool.h"
#include "kernels.h"

namespace tbvh {

namespace {

constexpr int WG = 64;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float fmin3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
__device__ __forceinline__ float fmax3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ uint32_t sext_s8x4(uint32_t i) { return ((i >> 7) & 0x01010101u) * 0xffu; }

__device__ __forceinline__ float4 ld_f4(const char* base, uint32_t off) { return *(const float4*)(base + off); }
__device__ __forceinline__ uint4 ld_u4(const char* base, uint32_t off) { return *(const uint4*)(base + off); }

// t = q * a + o with q an exact small integer held as fp16: one v_fma_mix_f32
__device__ __forceinline__ float tplane(uint32_t w, int hi, float a, float o) {
    const half2_t h = __builtin_bit_cast(half2_t, w);
    return __builtin_fmaf((float)(hi ? h.y : h.x), a, o);
}

struct NodeOut { uint32_t childBase, triBase, hitmask, imask; };

// offN*/offF* : per-ray byte offsets of the near / far plane of each axis inside an H node.
__device__ __forceinline__ NodeOut visit_node_h(const char* __restrict__ nodes, uint32_t nodeIdx, float3 O, float3 rD,
                                                float tmax, uint32_t octinv4, uint32_t offNX, uint32_t offNY, uint32_t offNZ) {
    const uint32_t b = nodeIdx << 7;
    const float4 n0 = ld_f4(nodes, b), n1 = ld_f4(nodes, b + 16);
    const uint4 nx = ld_u4(nodes, b + offNX), fx = ld_u4(nodes, b + (112u - offNX));
    const uint4 ny = ld_u4(nodes, b + offNY), fy = ld_u4(nodes, b + (144u - offNY));
    const uint4 nz = ld_u4(nodes, b + offNZ), fz = ld_u4(nodes, b + (176u - offNZ));
    const uint32_t ew = as_u32(n0.w);
    const float ax = ldexpf(rD.x, (int)(int8_t)(ew)), ay = ldexpf(rD.y, (int)(int8_t)(ew >> 8)), az = ldexpf(rD.z, (int)(int8_t)(ew >> 16));
    const float ox = (n0.x - O.x) * rD.x, oy = (n0.y - O.y) * rD.y, oz = (n0.z - O.z) * rD.z;
    const uint32_t nxw[4] = {nx.x, nx.y, nx.z, nx.w}, fxw[4] = {fx.x, fx.y, fx.z, fx.w};
    const uint32_t nyw[4] = {ny.x, ny.y, ny.z, ny.w}, fyw[4] = {fy.x, fy.y, fy.z, fy.w};
    const uint32_t nzw[4] = {nz.x, nz.y, nz.z, nz.w}, fzw[4] = {fz.x, fz.y, fz.z, fz.w};
    uint32_t hitmask = 0;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const uint32_t meta4 = half ? as_u32(n1.w) : as_u32(n1.z);
        const uint32_t inner4 = (meta4 & (meta4 << 1)) & 0x10101010u;
        const uint32_t imask4 = sext_s8x4(inner4 << 3);
        const uint32_t bitidx4 = (meta4 ^ (octinv4 & imask4)) & 0x1F1F1F1Fu;
        const uint32_t bits4 = (meta4 >> 5) & 0x07070707u;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int c = half * 4 + i, w = c >> 1, hi = c & 1, sh = 8 * i;
            const float tnx = tplane(nxw[w], hi, ax, ox), tfx = tplane(fxw[w], hi, ax, ox);
            const float tny = tplane(nyw[w], hi, ay, oy), tfy = tplane(fyw[w], hi, ay, oy);
            const float tnz = tplane(nzw[w], hi, az, oz), tfz = tplane(fzw[w], hi, az, oz);
            const float cmin = __builtin_fmaxf(fmax3(tnx, tny, tnz), 0.0f);
            const float cmax = __builtin_fminf(fmin3(tfx, tfy, tfz), tmax);
            if (cmin <= cmax) hitmask |= ((bits4 >> sh) & 255u) << ((bitidx4 >> sh) & 255u);
        }
    }
    NodeOut r;
    r.childBase = as_u32(n1.x); r.triBase = as_u32(n1.y); r.hitmask = hitmask; r.imask = ew >> 24;
    return r;
}

template <bool ANYHIT, int LDS_N, int REFILL_MIN>
__global__ __launch_bounds__(WG) void k_cwbvh_h(const char* __restrict__ nodes, const float4* __restrict__ tris, QueryArgs q,
                                                uint32_t* __restrict__ status) {
    __shared__ uint2 stk[LDS_N][WG];
    const uint32_t lane = threadIdx.x;
    LaneStack<uint2, LDS_N, 64> st;
    st.init(&stk[0][lane], (uint2*)q.spill + (blockIdx.x * WG + lane), (size_t)gridDim.x * WG, q.spillStride);
    RayPool<64> pool;
    const uint64_t nRaysTotal = q.nRaysDev ? *q.nRaysDev : q.nRays;   // batch size may live on the device (wavefront queues)
    pool.init(q.poolParts);

    bool active = false;
    uint64_t ri = 0;
    float3 O = make_float3(0, 0, 0), D = O, rD = O;
    float4 hit = make_float4(0, 0, 0, 0);
    bool found = false;
    uint32_t oct = 0, octinv4 = 0, offNX = 32, offNY = 48, offNZ = 64;
    uint2 ng = make_uint2(0u, 0u), tg = make_uint2(0u, 0u);

    for (;;) {
        // ---- ray replacement ---------------------------------------------------------------
        const uint32_t nIdle = (uint32_t)__popcll(__ballot(!active));
        if (nIdle >= (uint32_t)REFILL_MIN) {
            if (!pool.dry()) {
                uint64_t nri = 0;
                if (pool.acquire(!active, q.counter, nRaysTotal, nri)) {
                    ri = nri;
                    const RayRec* rp = q.rays + ri;
                    O = xyz(rp->O); D = xyz(rp->D); rD = xyz(rp->rD);
                    hit = q.fresh ? make_float4(q.freshTmax, 0.f, 0.f, 0.f) : rp->hit;
                    found = false;
                    oct = 7u - ((D.x < 0 ? 4u : 0u) | (D.y < 0 ? 2u : 0u) | (D.z < 0 ? 1u : 0u));
                    octinv4 = oct * 0x01010101u;
                    offNX = rD.x < 0 ? 80u : 32u; offNY = rD.y < 0 ? 96u : 48u; offNZ = rD.z < 0 ? 112u : 64u;
                    ng = make_uint2(0u, 0x80000000u); tg = make_uint2(0u, 0u);
                    st.reset();
                    active = true;
                }
            }
            if (__ballot(active) == 0) break;
        }
        if (!active) continue;

        bool done = false;
        // ---- triangle phase: one test for lanes with a pending triangle group -------------
        if (tg.y != 0) {
            const uint32_t ti = 31u - (uint32_t)__clz(tg.y);
            tg.y &= ~(1u << ti);
            const uint32_t ta = tg.x + ti * 3u;
            const float4 e2 = tris[ta], e1 = tris[ta + 1], v0 = tris[ta + 2];
            TriHit h;
            if (tri_test(O, D, xyz(v0), xyz(e1), xyz(e2), hit.x, h, q.omm, as_u32(v0.w))) {
                found = true;
                if (ANYHIT) done = true;
                else hit = make_float4(h.t, h.u, h.v, v0.w);
            }
        }
        // ---- node phase: lanes without pending triangles ----------------------------------
        if (!done && tg.y == 0) {
            if (ng.y <= 0x00FFFFFFu) {
                if (st.empty()) done = true;
                else ng = st.pop();
            }
            if (!done) {
                if (ng.y > 0x00FFFFFFu) {
                    const uint32_t imask = ng.y;
                    const uint32_t bit = 31u - (uint32_t)__clz(ng.y);
                    const uint32_t cbase = ng.x;
                    ng.y &= ~(1u << bit);
                    if (ng.y > 0x00FFFFFFu) st.push(ng);
                    const uint32_t slot = (bit - 24u) ^ oct;
                    const uint32_t rel = __popc(imask & ~(0xFFFFFFFFu << slot));
                    const NodeOut r = visit_node_h(nodes, cbase + rel, O, rD, hit.x, octinv4, offNX, offNY, offNZ);
                    ng.x = r.childBase; tg.x = r.triBase;
                    ng.y = (r.hitmask & 0xFF000000u) | r.imask;
                    tg.y = r.hitmask & 0x00FFFFFFu;
                } else {
                    tg = ng;
                    ng = make_uint2(0u, 0u);
                }
            }
        }
        if (done) {
            if (ANYHIT) q.occluded[ri] = found ? 1 : 0;
            else if (found || q.fresh) q.rays[ri].hit = hit;
            active = false;
        }
    }
    if (st.overflow) atomicOr(status, 1u);
}

// 80-byte reference node -> 128-byte H node (one thread per node).
__global__ void k_cwbvh_relayout(const float4* __restrict__ src, char* __restrict__ dst, uint32_t nNodes) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nNodes) return;
    const float4* n = src + (size_t)k * 5;
    char* d = dst + (size_t)k * 128;
    *(float4*)(d) = n[0];
    *(float4*)(d + 16) = n[1];
    const float4 q[3] = {n[2], n[3], n[4]};
    const uint8_t* qb = (const uint8_t*)q;  // qlo_x[8] qlo_y[8] qlo_z[8] qhi_x[8] qhi_y[8] qhi_z[8]
    _Float16* h = (_Float16*)(d + 32);
    for (int i = 0; i < 48; i++) h[i] = (_Float16)(float)qb[i];
}

}  // namespace

void launch_cwbvh_h(bool anyhit, int variant, const char* nodesH, const float4* tris, const QueryArgs& q, uint32_t* status,
                    uint32_t blocks, hipStream_t s) {
#define TBVH_LAUNCH_H(LDSN, RMIN)                                                                                        \
    do {                                                                                                                 \
        if (anyhit) hipLaunchKernelGGL((k_cwbvh_h<true, LDSN, RMIN>), dim3(blocks), dim3(WG), 0, s, nodesH, tris, q, status); \
        else hipLaunchKernelGGL((k_cwbvh_h<false, LDSN, RMIN>), dim3(blocks), dim3(WG), 0, s, nodesH, tris, q, status);   \
    } while (0)
    switch (variant) {
    case 21: TBVH_LAUNCH_H(8, 8); break;
    case 22: TBVH_LAUNCH_H(12, 16); break;
    default: TBVH_LAUNCH_H(8, 16); break;
    }
#undef TBVH_LAUNCH_H
}

void launch_cwbvh_relayout(const float4* src, char* dst, uint32_t nNodes, hipStream_t s) {
    hipLaunchKernelGGL(k_cwbvh_relayout, dim3((nNodes + 255) / 256), dim3(256), 0, s, src, dst, nNodes);
}

}  // namespace tbvh
