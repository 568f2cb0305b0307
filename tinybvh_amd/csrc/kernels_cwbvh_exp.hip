// kernels_cwbvh_exp.hip — EXPERIMENT build only (make EXPERIMENTS=1): the round-1 BVH8_CWBVH kernel with all the schedule
// variants that were measured against each other (tbvh_set_variant 1..19, 40..48).  The kernel the
// library ships is kernels_cwbvh.hip.
//
// Replaces batch_cwbvh / isoccluded_cwbvh (traverse_cwbvh.cl:124-570) from scratch.
// Blob format: nodes 5 x float4, tris 3 x float4 {e2, e1, v0|prim}, both verbatim as
// BVH8_CWBVH::ConvertFrom writes them (tiny_bvh.h:5884-6018).  Traversal
// state machine after Ylitie et al. 2017 as restated by the CPU mirror tiny_bvh.h:7046-7154:
// ngroup = {child base, hits<<24 | imask}, tgroup = {tri base, tri bits}; highest set bit
// first = front-to-back through octinv.  Hit semantics follow BVH::Intersect (inclusive
// t range, miss leaves the record untouched).
//
// Scheduling: persistent one-wave workgroups, one lane = one ray, per-lane ray replacement
// from a wave-local pool (ray_pool.h), traversal stack top in LDS / bottom in global
// (lane_stack.h).
#include "device_common.h"
#include "lane_stack.h"
#include "ray_pool.h"
#include "kernels.h"
#include "cwbvh_node.h"

namespace tbvh {

namespace {

constexpr int WG = 64;

__device__ __forceinline__ float fmin3(float a, float b, float c) { return cw_fmin3(a, b, c); }
__device__ __forceinline__ float fmax3(float a, float b, float c) { return cw_fmax3(a, b, c); }
__device__ __forceinline__ uint32_t sext_s8x4(uint32_t i) { return cw_sext_s8x4(i); }
typedef CwNodeHits NodeResult;

template <int NSTRIDE = 5>
__device__ __forceinline__ NodeResult visit_node(const float4* __restrict__ nodes, uint32_t nodeIdx, float3 O,
                                                 float3 rD, float tmax, uint32_t octinv4) {
    return cw_test_node(cw_load_node<NSTRIDE>(nodes, nodeIdx), O, rD, tmax, octinv4);
}

// Same test with the 48 plane FMAs issued as 24 v_pk_fma_f32 (two children per instruction,
// the ray-dependent scale and offset broadcast through op_sel).  tools/ubench/valu_rate.hip:
// v_pk_fma_f32 5.6 cycles for two FMAs vs 2 x 4.1 for v_fma_f32.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ NodeResult visit_node_pk(const float4* __restrict__ nodes, uint32_t nodeIdx, float3 O,
                                                    float3 rD, float tmax, uint32_t octinv4) {
    const float4* np = nodes + nodeIdx * 5u;
    const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4];
    const uint32_t ew = as_u32(n0.w);
    const float ax = ldexpf(rD.x, (int)(int8_t)(ew)), ay = ldexpf(rD.y, (int)(int8_t)(ew >> 8)), az = ldexpf(rD.z, (int)(int8_t)(ew >> 16));
    const float ox = (n0.x - O.x) * rD.x, oy = (n0.y - O.y) * rD.y, oz = (n0.z - O.z) * rD.z;
    const v2f ax2 = {ax, ax}, ay2 = {ay, ay}, az2 = {az, az}, ox2 = {ox, ox}, oy2 = {oy, oy}, oz2 = {oz, oz};
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
        const uint32_t lox = rD.x < 0 ? qhx : qlx, hix = rD.x < 0 ? qlx : qhx;
        const uint32_t loy = rD.y < 0 ? qhy : qly, hiy = rD.y < 0 ? qly : qhy;
        const uint32_t loz = rD.z < 0 ? qhz : qlz, hiz = rD.z < 0 ? qlz : qhz;
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
            const int s0 = 8 * i, s1 = 8 * i + 8;
#define TBVH_Q2(w) v2f{(float)(((w) >> s0) & 255), (float)(((w) >> s1) & 255)}
            const v2f tnx = __builtin_elementwise_fma(TBVH_Q2(lox), ax2, ox2), tfx = __builtin_elementwise_fma(TBVH_Q2(hix), ax2, ox2);
            const v2f tny = __builtin_elementwise_fma(TBVH_Q2(loy), ay2, oy2), tfy = __builtin_elementwise_fma(TBVH_Q2(hiy), ay2, oy2);
            const v2f tnz = __builtin_elementwise_fma(TBVH_Q2(loz), az2, oz2), tfz = __builtin_elementwise_fma(TBVH_Q2(hiz), az2, oz2);
#undef TBVH_Q2
            const float cmin0 = __builtin_fmaxf(fmax3(tnx.x, tny.x, tnz.x), 0.0f), cmax0 = __builtin_fminf(fmin3(tfx.x, tfy.x, tfz.x), tmax);
            const float cmin1 = __builtin_fmaxf(fmax3(tnx.y, tny.y, tnz.y), 0.0f), cmax1 = __builtin_fminf(fmin3(tfx.y, tfy.y, tfz.y), tmax);
            if (cmin0 <= cmax0) hitmask |= ((bits4 >> s0) & 255u) << ((bitidx4 >> s0) & 255u);
            if (cmin1 <= cmax1) hitmask |= ((bits4 >> s1) & 255u) << ((bitidx4 >> s1) & 255u);
        }
    }
    NodeResult r;
    r.childBase = as_u32(n1.x); r.triBase = as_u32(n1.y); r.hitmask = hitmask; r.imask = ew >> 24;
    return r;
}

// MODE 0: whole-wave batches (a wave takes 64 consecutive rays and finishes them all).
// MODE 1: per-lane replacement from a wave-local pool; REFILL_MIN idle lanes trigger a refill.
// ADAPT: the wave's LockstepGovernor (ray_pool.h) decides when it takes new rays.
template <bool ANYHIT, int MODE, int LDS_N, int REFILL_MIN, bool TRI1 = false, bool STATS = false, int TRI_MIN = 1, bool PKFMA = false, int CHUNK = 64, bool ADAPT = false,
          int NSTRIDE = 5, bool HAS_OMM = true>
__global__ __launch_bounds__(WG) void k_cwbvh(const float4* __restrict__ nodes, const float4* __restrict__ tris,
                                              QueryArgs q, uint32_t* __restrict__ status) {
    __shared__ uint2 stk[LDS_N][WG];
    const uint32_t glane = blockIdx.x * WG + threadIdx.x;
    LaneStack<uint2, LDS_N, WG> st;
    st.init(&stk[0][threadIdx.x], (uint2*)q.spill + glane, gridDim.x * WG, q.spillStride);
    RayPool<CHUNK> pool;
    const uint64_t nRaysTotal = q.nRaysDev ? *q.nRaysDev : q.nRays;   // batch size may live on the device (wavefront queues)
    pool.init(q.poolParts);

    bool active = false;
    uint64_t ri = 0;
    float3 O = make_float3(0, 0, 0), D = O, rD = O;
    float4 hit = make_float4(0, 0, 0, 0);
    bool found = false;
    uint32_t oct = 0, octinv4 = 0;
    uint2 ng = make_uint2(0u, 0u), tg = make_uint2(0u, 0u);
    bool lockstep = ADAPT;                          // ADAPT only; wave-uniform
    uint32_t genIters = 0, genActive = 0, ema = 0;  // ADAPT only; wave-uniform
    unsigned long long sIter = 0, sActive = 0, sNode = 0, sTriIter = 0, sTri = 0, sRefill = 0, sRefilled = 0, sNodeIter = 0, sNodeUni = 0, sNodeLanes = 0;  // STATS only

    for (;;) {
        // ---- ray replacement -------------------------------------------------------------
        const uint64_t idleMask = __ballot(!active);
        const uint32_t nIdle = (uint32_t)__popcll(idleMask);
        if (ADAPT && lockstep) {
            genIters++; genActive += (uint32_t)WG - nIdle;
            if (nIdle == (uint32_t)WG) {   // a generation ended: fold its cohesion into the running estimate
                if (genIters > 1) {
                    const uint32_t e = genActive * 4u / genIters;   // x / 256
                    ema = ema ? (ema + e) >> 1 : e;
                    if (STATS) {   // histogram of per-generation cohesion, 8 bins: <.5 .5-.6 .6-.7 .7-.75 .75-.8 .8-.85 .85-.9 >=.9
                        const uint32_t b = e < 128u ? 0u : e < 154u ? 1u : e < 179u ? 2u : e < 192u ? 3u : e < 205u ? 4u : e < 218u ? 5u : e < 230u ? 6u : 7u;
                        if (threadIdx.x == 0) atomicAdd(q.stats + b, 1ull);
                    } else if (ema < kLockstepKeep) lockstep = false;
                }
                genIters = 0; genActive = 0;
            } else if (!STATS && genIters >= 16u && genActive * 4u < kLockstepBail * genIters) lockstep = false;
        }
        const bool wantRefill = ADAPT ? (lockstep ? nIdle == (uint32_t)WG : nIdle >= (uint32_t)REFILL_MIN)
                                      : (MODE == 0 ? nIdle == (uint32_t)WG : nIdle >= (uint32_t)REFILL_MIN);
        if (wantRefill || (nIdle == (uint32_t)WG)) {
            if (!pool.dry()) {
                uint64_t nri = 0;
                const bool got = pool.acquire(!active, q.counter, nRaysTotal, nri);
                if (STATS) { sRefill++; sRefilled += __popcll(__ballot(got)); }
                if (got) {
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
        if (STATS) { sIter++; sActive += __popcll(__ballot(active)); sNode += __popcll(__ballot(active && ng.y > 0x00FFFFFFu)); }
        if (!active) continue;

        if (TRI1) {
            // ---- interleaved schedule: at most ONE triangle test and ONE node visit per lane and
            // iteration.  A lane with pending triangles sits out the node phase (so the per-ray
            // order of tests is exactly the mirror's: all triangles of a group before the next
            // node), but the rest of the wave does not wait for a lane's whole triangle list.
            bool done = false;
            // triangle phase runs when at least TRI_MIN lanes have a pending triangle, or when no lane could
            // use a node phase instead (so a waiting lane always makes progress eventually)
            bool triPhase = true;
            if (TRI_MIN > 1) {
                const uint32_t nPend = (uint32_t)__popcll(__ballot(tg.y != 0));
                const uint32_t nAct = (uint32_t)__popcll(__ballot(true));
                triPhase = nPend >= (uint32_t)TRI_MIN || nPend == nAct;
            }
            if (triPhase && tg.y != 0) {
                if (STATS) { const unsigned long long m = __ballot(true); if (lane_rank(m) == 0) { sTriIter++; sTri += __popcll(m); } }
                const uint32_t ti = 31u - (uint32_t)__clz(tg.y);
                tg.y &= ~(1u << ti);
                const uint32_t ta = tg.x + ti * 3u;
                const float4 e2 = tris[ta], e1 = tris[ta + 1], v0 = tris[ta + 2];
                TriHit h;
                if (tri_test(O, D, xyz(v0), xyz(e1), xyz(e2), hit.x, h, HAS_OMM ? q.omm : Omm{nullptr, 0}, as_u32(v0.w))) {
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
                        const uint32_t rel = __popc(imask & ~(0xFFFFFFFFu << slot));
                        if (STATS) {   // how many node-visit iterations have all their lanes on ONE node
                            const unsigned long long m = __ballot(true);
                            const uint32_t idx = cbase + rel, f = (uint32_t)__builtin_amdgcn_readfirstlane(idx);
                            const bool uni = __ballot(idx != f) == 0;
                            if (lane_rank(m) == 0) { sNodeIter++; sNodeUni += uni ? 1u : 0u; sNodeLanes += __popcll(m); }
                        }
                        const NodeResult r = PKFMA ? visit_node_pk(nodes, cbase + rel, O, rD, hit.x, octinv4)
                                                   : visit_node<NSTRIDE>(nodes, cbase + rel, O, rD, hit.x, octinv4);
                        ng.x = r.childBase; tg.x = r.triBase;
                        ng.y = (r.hitmask & 0xFF000000u) | r.imask;
                        tg.y = r.hitmask & 0x00FFFFFFu;
                    } else {  // a postponed triangle group came off the stack
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
            continue;
        }

        // ---- one traversal step ----------------------------------------------------------
        if (ng.y > 0x00FFFFFFu) {
            const uint32_t imask = ng.y;
            const uint32_t bit = 31u - (uint32_t)__clz(ng.y);
            const uint32_t cbase = ng.x;
            ng.y &= ~(1u << bit);
            if (ng.y > 0x00FFFFFFu) st.push(ng);
            const uint32_t slot = (bit - 24u) ^ oct;
            const uint32_t rel = __popc(imask & ~(0xFFFFFFFFu << slot));
            const NodeResult r = visit_node(nodes, cbase + rel, O, rD, hit.x, octinv4);
            ng.x = r.childBase; tg.x = r.triBase;
            ng.y = (r.hitmask & 0xFF000000u) | r.imask;
            tg.y = r.hitmask & 0x00FFFFFFu;
        } else {
            tg = ng;
            ng = make_uint2(0u, 0u);
        }
        while (tg.y != 0) {
            if (STATS) { const unsigned long long m = __ballot(true); if (lane_rank(m) == 0) { sTriIter++; sTri += __popcll(m); } }
            const uint32_t ti = 31u - (uint32_t)__clz(tg.y);
            tg.y &= ~(1u << ti);
            const uint32_t ta = tg.x + ti * 3u;
            const float4 e2 = tris[ta], e1 = tris[ta + 1], v0 = tris[ta + 2];
            TriHit h;
            if (tri_test(O, D, xyz(v0), xyz(e1), xyz(e2), hit.x, h, HAS_OMM ? q.omm : Omm{nullptr, 0}, as_u32(v0.w))) {
                found = true;
                if (ANYHIT) break;
                hit = make_float4(h.t, h.u, h.v, v0.w);
            }
        }
        bool done = ANYHIT && found;
        if (!done && ng.y <= 0x00FFFFFFu) {
            if (st.empty()) done = true;
            else ng = st.pop();
        }
        if (done) {
            if (ANYHIT) q.occluded[ri] = found ? 1 : 0;
            else if (found || q.fresh) q.rays[ri].hit = hit;
            active = false;
        }
    }
    if (st.overflow) atomicOr(status, 1u);
    if (STATS && !ADAPT) {
        // sTriIter was counted by the first active lane of each tri iteration: reduce over the wave
        unsigned long long ti = sTriIter;
        for (int o = 32; o > 0; o >>= 1) { ti += __shfl_xor(ti, o); sTri += __shfl_xor(sTri, o); sNodeIter += __shfl_xor(sNodeIter, o); sNodeUni += __shfl_xor(sNodeUni, o); sNodeLanes += __shfl_xor(sNodeLanes, o); }
        if (REFILL_MIN == 64) { sRefill = sNodeIter; sRefilled = sNodeUni; sNode = sNodeLanes; }   // the lockstep statistics variant reports these instead
        if (threadIdx.x == 0) {
            atomicAdd(q.stats + 0, sIter); atomicAdd(q.stats + 1, sActive); atomicAdd(q.stats + 2, sNode);
            atomicAdd(q.stats + 3, ti); atomicAdd(q.stats + 4, sTri); atomicAdd(q.stats + 5, sRefill); atomicAdd(q.stats + 6, sRefilled);
        }
    }
}

// ---------------------------------------------------------------------------------------
// Lean kernel for COHERENT batches (camera rays): one wave per 64 consecutive rays, no
// replacement, no persistent loop, every triangle of a group tested at once — the whole wave
// follows nearly the same path, so the bookkeeping that pays for itself on incoherent rays is
// pure overhead here (measured against the reference's own OpenCL batch_cwbvh on the same GPU:
// tools/vs_reference_opencl.py).  Stack: STACK_N entries per lane in private (scratch) memory,
// like the reference kernel's `uint2 stack[32]` (traverse_cwbvh.cl:133); deeper trees are
// reported through the status word, never silently mis-traversed.
template <bool ANYHIT, int STACK_N>
__global__ __launch_bounds__(WG) void k_cwbvh_lean(const float4* __restrict__ nodes, const float4* __restrict__ tris, QueryArgs q,
                                                   uint32_t* __restrict__ status) {
    const uint64_t nRaysTotal = q.nRaysDev ? *q.nRaysDev : q.nRays;
    // groups are dealt to the grid round-robin (static assignment: camera-like batches cost about the
    // same per group)
    const uint64_t nGroups = (nRaysTotal + WG - 1) / WG;
    bool overflow = false;
  for (uint64_t group = blockIdx.x; group < nGroups; group += gridDim.x) {
    const uint64_t ri = group * WG + threadIdx.x;
    if (ri >= nRaysTotal) continue;
    RayRec* rp = q.rays + ri;
    const float3 O = xyz(rp->O), D = xyz(rp->D), rD = xyz(rp->rD);
    float4 hit = q.fresh ? make_float4(q.freshTmax, 0.f, 0.f, 0.f) : rp->hit;
    bool found = false;
    uint2 stack[STACK_N];
    int sp = 0;
    const uint32_t oct = 7u - ((D.x < 0 ? 4u : 0u) | (D.y < 0 ? 2u : 0u) | (D.z < 0 ? 1u : 0u));
    const uint32_t octinv4 = oct * 0x01010101u;
    uint2 ng = make_uint2(0u, 0x80000000u), tg = make_uint2(0u, 0u);
    for (;;) {
        if (ng.y > 0x00FFFFFFu) {
            const uint32_t imask = ng.y;
            const uint32_t bit = 31u - (uint32_t)__clz(ng.y);
            const uint32_t cbase = ng.x;
            ng.y &= ~(1u << bit);
            if (ng.y > 0x00FFFFFFu) { if (sp < STACK_N) stack[sp++] = ng; else overflow = true; }
            const uint32_t slot = (bit - 24u) ^ oct;
            const uint32_t rel = __popc(imask & ~(0xFFFFFFFFu << slot));
            const NodeResult r = visit_node(nodes, cbase + rel, O, rD, hit.x, octinv4);
            ng.x = r.childBase; tg.x = r.triBase;
            ng.y = (r.hitmask & 0xFF000000u) | r.imask;
            tg.y = r.hitmask & 0x00FFFFFFu;
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
            if (tri_test(O, D, xyz(v0), xyz(e1), xyz(e2), hit.x, h, q.omm, as_u32(v0.w))) {
                found = true;
                if (ANYHIT) break;
                hit = make_float4(h.t, h.u, h.v, v0.w);
            }
        }
        if (ANYHIT && found) break;
        if (ng.y > 0x00FFFFFFu) continue;
        if (sp == 0) break;
        ng = stack[--sp];
    }
    if (ANYHIT) q.occluded[ri] = found ? 1 : 0;
    else if (found || q.fresh) rp->hit = hit;
  }
    if (overflow) atomicOr(status, 1u);
}

}  // namespace

void launch_cwbvh_exp(bool anyhit, int variant, const float4* nodes, const float4* tris, const QueryArgs& q, uint32_t* status,
                  uint32_t blocks, hipStream_t s) {
#define TBVH_LAUNCH(MODE, LDSN, RMIN, ...)                                                                                 \
    do {                                                                                                                   \
        if (anyhit) hipLaunchKernelGGL((k_cwbvh<true, MODE, LDSN, RMIN, ##__VA_ARGS__>), dim3(blocks), dim3(WG), 0, s, nodes, tris, q, status); \
        else hipLaunchKernelGGL((k_cwbvh<false, MODE, LDSN, RMIN, ##__VA_ARGS__>), dim3(blocks), dim3(WG), 0, s, nodes, tris, q, status);  \
    } while (0)
    if (variant == 40 || variant == 41) {   // lean one-wave-per-64-rays kernel (coherent batches)
        const uint64_t ng_ = (q.nRays + WG - 1) / WG;
        const uint32_t nb = (uint32_t)(ng_ < 65536 ? ng_ : 65536);
        if (variant == 40) {
            if (anyhit) hipLaunchKernelGGL((k_cwbvh_lean<true, 32>), dim3(nb), dim3(WG), 0, s, nodes, tris, q, status);
            else hipLaunchKernelGGL((k_cwbvh_lean<false, 32>), dim3(nb), dim3(WG), 0, s, nodes, tris, q, status);
        } else {
            if (anyhit) hipLaunchKernelGGL((k_cwbvh_lean<true, 48>), dim3(nb), dim3(WG), 0, s, nodes, tris, q, status);
            else hipLaunchKernelGGL((k_cwbvh_lean<false, 48>), dim3(nb), dim3(WG), 0, s, nodes, tris, q, status);
        }
        return;
    }
    switch (variant) {
    case 1: TBVH_LAUNCH(0, 16, 64); break;   // whole-wave batches (round-1 v0 behaviour)
    case 2: TBVH_LAUNCH(1, 16, 1); break;    // replace as soon as one lane is idle
    case 3: TBVH_LAUNCH(1, 16, 8); break;
    case 4: TBVH_LAUNCH(1, 16, 32); break;
    case 7:  // instrumented copy of variant 5 (lane-utilisation counters in q.stats)
        hipLaunchKernelGGL((k_cwbvh<false, 1, 8, 16, false, true>), dim3(blocks), dim3(WG), 0, s, nodes, tris, q, status);
        break;
    case 8: TBVH_LAUNCH(1, 8, 16, true); break;   // one triangle + one node per lane and iteration
    case 9:  // instrumented copy of variant 8
        hipLaunchKernelGGL((k_cwbvh<false, 1, 8, 16, true, true>), dim3(blocks), dim3(WG), 0, s, nodes, tris, q, status);
        break;
    case 10: TBVH_LAUNCH(1, 8, 8, true); break;
    case 16: TBVH_LAUNCH(1, 8, 16, true, false, 1, true); break;  // packed plane FMAs
    case 48: TBVH_LAUNCH(1, 8, 64, true, true); break;   // lockstep throughout + statistics (node-visit uniformity)
    case 44: TBVH_LAUNCH(1, 8, 64, true); break;   // lockstep throughout: a wave only takes new rays when all 64 lanes are idle
    case 45: TBVH_LAUNCH(1, 8, 16, true, false, 1, false, 64, true); break;   // adaptive (= default)
    case 47: TBVH_LAUNCH(1, 8, 16, true, false, 1, false, 64, true, 8); break;   // adaptive, nodes padded to 128 bytes (one cache line per node)
    case 46:  // lockstep throughout + histogram of per-generation lane cohesion in q.stats
        hipLaunchKernelGGL((k_cwbvh<false, 1, 8, 16, true, true, 1, false, 64, true>), dim3(blocks), dim3(WG), 0, s, nodes, tris, q, status);
        break;
    case 17: TBVH_LAUNCH(1, 8, 16, true, false, 1, false, 128); break;   // pool chunk sizes: one global atomic per CHUNK rays
    case 18: TBVH_LAUNCH(1, 8, 16, true, false, 1, false, 256); break;
    case 19: TBVH_LAUNCH(0, 16, 64, false, false, 1, false, 256); break;  // whole-wave batches, chunk 256
    case 13: TBVH_LAUNCH(1, 8, 16, true, false, 8); break;    // triangle phase only when >= 8 lanes wait
    case 14: TBVH_LAUNCH(1, 8, 16, true, false, 16); break;
    case 15: TBVH_LAUNCH(1, 8, 16, true, false, 24); break;
    case 11: TBVH_LAUNCH(1, 8, 24, true); break;
    case 5: TBVH_LAUNCH(1, 8, 16); break;    // smaller LDS stack -> more waves per CU
    case 6: TBVH_LAUNCH(1, 12, 16); break;
    case 12: TBVH_LAUNCH(1, 16, 16); break;  // replace when >= 16 lanes are idle, all triangles of a group at once
    default:   // = 45: the adaptive schedule is the default; without opacity micromaps the check is compiled out (+1-2 %)
        if (q.omm.map) TBVH_LAUNCH(1, 8, 16, true, false, 1, false, 64, true);
        else TBVH_LAUNCH(1, 8, 16, true, false, 1, false, 64, true, 5, false);
        break;
    }
#undef TBVH_LAUNCH
}

}  // namespace tbvh
