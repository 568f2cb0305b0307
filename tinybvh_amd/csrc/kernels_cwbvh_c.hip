// kernels_cwbvh_c.hip — BVH8_CWBVH traversal with the top of the tree resident in LDS.
//
// Measured on MI355X (tools/ubench/gather_rate.hip): a divergent 16-byte-per-lane load costs
// the CU's single texture-address/L1 pipe ~50-75 cycles per wave-instruction even when every
// line hits L1, and ~10 cycles per distinct 128-byte line when it misses (~11 B/clk/CU from
// L2/MALL/HBM).  A node visit is five such loads, so after the VALU work was trimmed the node
// fetch through the vector-memory pipe is the co-limiter.  LDS serves the same five 16-byte
// reads in ~10 cycles each.  So: nodes are renumbered at upload in surface-area priority order
// (the nodes a random ray is most likely to visit get the lowest indices, sibling groups stay
// contiguous — capi.hip: reorder_cwbvh_priority), every 1024-thread workgroup copies the first
// K nodes into LDS once, and a node visit reads LDS when index < K (38-55 % of all visits on
// the Bistro stand-in with K = 1228) and global memory otherwise.
//
// Schedule as in kernels_cwbvh.hip: persistent waves, one lane = one ray, per-lane ray
// replacement from a wave-local pool, one triangle + one node per lane and iteration, traversal
// stack per lane in LDS with a global spill area.
#include "device_common.h"
#include "lane_stack.h"
#include "ray_pool.h"
#include "kernels.h"
#include "cwbvh_node.h"

namespace tbvh {

namespace {

constexpr int WAVES = 16;
constexpr int WGC = WAVES * 64;

typedef CwNodeHits NodeOut;
// the node test is the shared one (cwbvh_node.h); here the five float4s may come from the LDS cache
__device__ __forceinline__ NodeOut test_children(float4 n0, float4 n1, float4 n2, float4 n3, float4 n4, float3 O, float3 rD,
                                                 float tmax, uint32_t octinv4) {
    return cw_test_node(CwNode{n0, n1, n2, n3, n4}, O, rD, tmax, octinv4);
}

template <bool ANYHIT, int LDS_N, int KMAX, int REFILL_MIN>
__global__ __launch_bounds__(WGC) void k_cwbvh_c(const float4* __restrict__ nodes, const float4* __restrict__ tris, uint32_t nCached,
                                                 QueryArgs q, uint32_t* __restrict__ status) {
    __shared__ float4 cache[KMAX * 5];
    __shared__ uint2 stk[WAVES][LDS_N][64];
    // ---- one-time fill of the LDS node cache ------------------------------------------------
    for (uint32_t i = threadIdx.x; i < nCached * 5u; i += WGC) cache[i] = nodes[i];
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    LaneStack<uint2, LDS_N, 64> st;
    st.init(&stk[wave][0][lane], (uint2*)q.spill + ((size_t)blockIdx.x * WGC + threadIdx.x), (size_t)gridDim.x * WGC, q.spillStride);
    RayPool<64> pool;
    const uint64_t nRaysTotal = q.nRaysDev ? *q.nRaysDev : q.nRays;   // batch size may live on the device (wavefront queues)
    pool.init(q.poolParts);

    bool active = false;
    uint64_t ri = 0;
    float3 O = make_float3(0, 0, 0), D = O, rD = O;
    float4 hit = make_float4(0, 0, 0, 0);
    bool found = false;
    uint32_t oct = 0, octinv4 = 0;
    uint2 ng = make_uint2(0u, 0u), tg = make_uint2(0u, 0u);

    for (;;) {
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
                    ng = make_uint2(0u, 0x80000000u); tg = make_uint2(0u, 0u);
                    st.reset();
                    active = true;
                }
            }
            if (__ballot(active) == 0) break;
        }
        if (!active) continue;

        bool done = false;
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
                    const uint32_t ci = cbase + __popc(imask & ~(0xFFFFFFFFu << slot));
                    float4 n0, n1, n2, n3, n4;
                    if (ci < nCached) {
                        const float4* p = cache + ci * 5u;
                        n0 = p[0]; n1 = p[1]; n2 = p[2]; n3 = p[3]; n4 = p[4];
                    } else {
                        const float4* p = nodes + (size_t)ci * 5u;
                        n0 = p[0]; n1 = p[1]; n2 = p[2]; n3 = p[3]; n4 = p[4];
                    }
                    const NodeOut r = test_children(n0, n1, n2, n3, n4, O, rD, hit.x, octinv4);
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

}  // namespace

uint32_t cwbvh_c_threads_per_block() { return WGC; }

void launch_cwbvh_c(bool anyhit, int variant, const float4* nodes, const float4* tris, uint32_t nNodes, const QueryArgs& q,
                    uint32_t* status, uint32_t blocks, hipStream_t s) {
#define TBVH_LAUNCH_C(LDSN, KMAX, RMIN)                                                                                    \
    do {                                                                                                                   \
        const uint32_t nc = nNodes < (uint32_t)(KMAX) ? nNodes : (uint32_t)(KMAX);                                         \
        if (anyhit) hipLaunchKernelGGL((k_cwbvh_c<true, LDSN, KMAX, RMIN>), dim3(blocks), dim3(WGC), 0, s, nodes, tris, nc, q, status); \
        else hipLaunchKernelGGL((k_cwbvh_c<false, LDSN, KMAX, RMIN>), dim3(blocks), dim3(WGC), 0, s, nodes, tris, nc, q, status);  \
    } while (0)
    switch (variant) {
    case 31: TBVH_LAUNCH_C(8, 585, 16); break;    // top four full levels only
    case 32: TBVH_LAUNCH_C(6, 1432, 16); break;   // shorter LDS stack, bigger cache
    case 33: TBVH_LAUNCH_C(8, 1, 16); break;      // cache off (root only): isolates the workgroup shape
    default: TBVH_LAUNCH_C(8, 1228, 16); break;   // 64 KB of stacks + 96 KB of nodes = all 160 KB
    }
#undef TBVH_LAUNCH_C
}

}  // namespace tbvh
