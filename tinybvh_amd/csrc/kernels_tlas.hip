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
// BLAS layouts: BVH8_CWBVH, BVH4_GPU and BVH_GPU, also mixed within one TLAS (traverse_tlas.cl:50-72 selects the BLAS
// traversal per instance through blasDesc[].blasType).  The point and vector transforms follow the reference build's FMA
// contraction (oracle/tbvh_oracle.c: orc_xform_point / orc_xform_vec, pinned bit for bit against the real
// BVH::IntersectTLAS), and this file is built with -ffp-contract=off, so the transformed ray — and therefore t,u,v —
// match bit for bit.
//
// Three kernels, one result (tests/test_tlas.py runs them all against the oracle and against each other):
//   k_tlas / k_tlas_w5   nested loops, whole-wave batches: TLAS walk -> instances of a leaf -> BLAS traversal.  Fastest while
//                        the 64 rays of a wave stay together (camera rays), 2.5-4 x slower than the flat loop otherwise.
//   k_tlas_flat          ONE loop, per-lane ray replacement: every lane is in TLAS / INSTANCE / BLAS mode and does one
//                        step of it per iteration; a mode's code runs when enough lanes are in it (phase gating).
//   k_tlas_adaptive      starts nested, measures the lane cohesion of each 64-ray generation, moves to the flat loop
//                        when the rays turn out incoherent.
// launch_tlas picks: CWBVH BLASes -> flat; BVH4_GPU BLASes -> adaptive; BVH_GPU BLASes or mixed layouts -> flat.
#include "device_common.h"
#include "lane_stack.h"
#include "ray_pool.h"
#include "kernels.h"
#include "cwbvh_node.h"

namespace tbvh {

namespace {

constexpr int WG = 64;

__device__ __forceinline__ float fmin3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
__device__ __forceinline__ float fmax3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
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

// Lane cohesion of the nested loops, for the adaptive kernel: every TLAS node and every instance entry calls tick() — each
// lane counts its own; at the end of a 64-ray generation mean / max of the lanes' counts says how evenly the wave's rays
// worked (the busiest lane's count is the number of trips the wave made at this level).  TBVH_TLAS_TICK_LDS = 1 counts the
// wave's trips exactly in an LDS word instead (first active lane, one ds_add per tick): same separation, 4 % slower.
#ifndef TBVH_TLAS_TICK_LDS
#define TBVH_TLAS_TICK_LDS 0
#endif
struct WaveTicks {
    TBVH_AS_LDS uint32_t* trips;
    uint32_t mine;
    __device__ __forceinline__ void tick() {
        mine++;
#if TBVH_TLAS_TICK_LDS
        if (lane_rank(__ballot(true)) == 0) __hip_atomic_fetch_add((uint32_t*)trips, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
    }
};

// ---- BLAS traversals; each runs until the stack is back at `base` ------------------------------

template <bool ANYHIT, int LDS_N, bool TICK>
__device__ __forceinline__ void blas_cwbvh(const GlobalF4 nodes, const GlobalF4 tris, RayL& r, StackT<LDS_N>& st, const Omm om, WaveTicks& tk) {
    const int base = st.sp;
    const uint32_t oct = 7u - ((r.D.x < 0 ? 4u : 0u) | (r.D.y < 0 ? 2u : 0u) | (r.D.z < 0 ? 1u : 0u));
    const uint32_t octinv4 = oct * 0x01010101u;
    uint2 ng = make_uint2(0u, 0x80000000u), tg = make_uint2(0u, 0u);
    for (;;) {
        if (TICK) tk.tick();
        if (ng.y > 0x00FFFFFFu) {
            const uint32_t imask = ng.y;
            const uint32_t bit = 31u - (uint32_t)__clz(ng.y);
            const uint32_t cbase = ng.x;
            ng.y &= ~(1u << bit);
            if (ng.y > 0x00FFFFFFu) st.push(ng);
            const uint32_t slot = (bit - 24u) ^ oct;
            const CwNodeHits nh = cw_test_node(cw_load_node(nodes, cbase + __popc(imask & ~(0xFFFFFFFFu << slot))), r.O, r.rD, r.hit.x, octinv4);
            ng.x = nh.childBase; tg.x = nh.triBase;
            ng.y = (nh.hitmask & 0xFF000000u) | nh.imask;
            tg.y = nh.hitmask & 0x00FFFFFFu;
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

template <bool ANYHIT, int LDS_N, bool TICK>
__device__ __forceinline__ void blas_bvh4(const GlobalF4 data, RayL& r, StackT<LDS_N>& st, const Omm om, WaveTicks& tk) {
    const int base = st.sp;
    uint32_t offset = 0;
    for (;;) {
        if (TICK) tk.tick();
        const float4 d0 = data[offset], d1 = data[offset + 1], d2 = data[offset + 2], d3 = data[offset + 3];
        const float sx = d1.x * r.rD.x, sy = d1.y * r.rD.y, sz = d1.z * r.rD.z;
        const float bx = (d0.x - r.O.x) * r.rD.x, by = (d0.y - r.O.y) * r.rD.y, bz = (d0.z - r.O.z) * r.rD.z;
        const uint32_t qx0 = as_u32(d0.w), qx1 = as_u32(d1.w);
        const uint32_t qy0 = as_u32(d2.x), qy1 = as_u32(d2.y), qz0 = as_u32(d2.z), qz1 = as_u32(d2.w);
        float dist[4];
        uint32_t info[4] = { as_u32(d3.x), as_u32(d3.y), as_u32(d3.z), as_u32(d3.w) };
        const bool ngx = sx < 0.f, ngy = sy < 0.f, ngz = sz < 0.f;   // near / far plane words by the sign of the direction (as k_bvh4)
        const uint32_t nx = ngx ? qx1 : qx0, fx = ngx ? qx0 : qx1;
        const uint32_t ny = ngy ? qy1 : qy0, fy = ngy ? qy0 : qy1;
        const uint32_t nz = ngz ? qz1 : qz0, fz = ngz ? qz0 : qz1;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int sh = 8 * i;
            const float x1 = __builtin_fmaf((float)((nx >> sh) & 255), sx, bx), x2 = __builtin_fmaf((float)((fx >> sh) & 255), sx, bx);
            const float y1 = __builtin_fmaf((float)((ny >> sh) & 255), sy, by), y2 = __builtin_fmaf((float)((fy >> sh) & 255), sy, by);
            const float z1 = __builtin_fmaf((float)((nz >> sh) & 255), sz, bz), z2 = __builtin_fmaf((float)((fz >> sh) & 255), sz, bz);
            const float tmin = __builtin_fmaxf(fmax3(x1, y1, z1), 0.0f);
            const float tmax = __builtin_fminf(fmin3(x2, y2, z2), r.hit.x);
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
// ADAPT 0: run until the batch is used up.  ADAPT 1: measure the lane cohesion of every 64-ray generation (WaveTicks) and
// return false as soon as one falls below KEEP / 256 — the caller continues with the flat loop (tlas_flat_body); returns
// true when the batch is used up.  ADAPT 2: statistics (histogram of the cohesion in q.stats), never switches.
template <bool ANYHIT, int BLAS_LAYOUT, int LDS_N, int ADAPT, uint32_t KEEP>
__device__ __forceinline__ bool tlas_body(const float4* __restrict__ tlasNodes, const uint32_t* __restrict__ tlasIdx,
                                          const float4* __restrict__ instances, const BlasDesc* __restrict__ blas, const QueryArgs& q,
                                          StackT<LDS_N>& st, RayPool<64>& pool, const uint64_t nRaysTotal, TBVH_AS_LDS uint32_t* ldsTrips) {
    WaveTicks tk; tk.trips = ldsTrips; tk.mine = 0;
    uint32_t ema = 0;   // ADAPT 1: running cohesion estimate, x / 256
    for (;;) {
        // whole-wave batches here: the nested TLAS/BLAS loops keep per-lane state in registers
        uint64_t ri = 0;
        const bool got = pool.acquire(true, q.counter, nRaysTotal, ri);
        if (__ballot(got) == 0) break;
        if (ADAPT) { tk.mine = 0; if ((threadIdx.x & 63u) == 0) *ldsTrips = 0u; }
        if (got) {
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
            if (ADAPT) tk.tick();
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
                    if (ADAPT) tk.tick();   // cohesion is sampled where the lanes part ways: TLAS nodes and instance entries (ticks inside the BLAS loops cost 18 %; instance entries alone separate camera rays from random ones less cleanly)
                    if (BLAS_LAYOUT == kLayoutCwbvh) blas_cwbvh<ANYHIT, LDS_N, false>(GlobalF4(bd.nodes), GlobalF4(bd.tris), rl, st, Omm{bd.opmap, bd.opmapN}, tk);
                    else blas_bvh4<ANYHIT, LDS_N, false>(GlobalF4(bd.nodes), rl, st, Omm{bd.opmap, bd.opmapN}, tk);
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
        if (ADAPT) {
            uint32_t sum = tk.mine;
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
#if TBVH_TLAS_TICK_LDS
            const uint32_t trips = *ldsTrips;
#else
            uint32_t trips = tk.mine;   // the busiest lane's count: every TLAS node and instance entry it made was a trip of the wave
            for (int o = 32; o > 0; o >>= 1) { const uint32_t t2 = __shfl_xor(trips, o); trips = t2 > trips ? t2 : trips; }
#endif
            const uint32_t e = trips ? sum * 4u / trips : 256u;   // x / 256
            if (ADAPT == 2) {
                const uint32_t b = e < 64u ? 0u : e < 96u ? 1u : e < 128u ? 2u : e < 154u ? 3u : e < 179u ? 4u : e < 205u ? 5u : e < 230u ? 6u : 7u;
                if ((threadIdx.x & 63u) == 0) atomicAdd(q.stats + b, 1ull);
            } else {
                // running estimate as in LockstepGovernor: one ragged generation among coherent ones (a wave at the edge of the
                // image) does not send the wave to the flat loop for the rest of the launch, two in a row do; an incoherent batch
                // leaves after its first generation
                ema = ema ? (ema + e) >> 1 : e;
                if (ema < KEEP) return false;
            }
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// The same query as ONE flat loop with per-lane ray replacement (the structure of kernels_cwbvh.hip / kernels_query.hip):
// every lane is in one of three modes and does one step of it per iteration —
//   TLAS      one 2-wide node of the top-level tree (a leaf switches to INSTANCE),
//   INSTANCE  take the next instance of the current TLAS leaf: mask test, ray into instance space (-> BLAS), or, when
//             the leaf is used up, pop the TLAS stack (-> TLAS) or finish the ray,
//   BLAS      one triangle test or one node visit of the instance's BVH; back at the stack base the world ray is read
//             again from the record (-> INSTANCE),
// so a lane never waits for the longest instance list, the deepest BLAS traversal or the slowest ray of its wave, which
// is what the nested loops of tlas_body cost (three levels of "everybody waits for the slowest").  Idle lanes take new
// rays as in the other kernels.  Per ray the order of instances, nodes and triangles is the nested version's.
// ---------------------------------------------------------------------------------------------------------------
template <bool ANYHIT, int BLAS_LAYOUT, int LDS_N, int REFILL_MIN, int PHASE_MIN, bool ADAPT, bool STATS = false>
__device__ __forceinline__ void tlas_flat_body(const float4* __restrict__ tlasNodes, const uint32_t* __restrict__ tlasIdx,
                                               const float4* __restrict__ instances, const BlasDesc* __restrict__ blas, const QueryArgs& q,
                                               StackT<LDS_N>& st, RayPool<64>& pool, const uint64_t nRaysTotal) {
    enum : uint32_t { M_TLAS = 0, M_INST = 1, M_BLAS = 2 };

    bool active = false, found = false;
    uint64_t ri = 0;
    float3 O = make_float3(0, 0, 0), D = O, rD = O;      // the ray in the CURRENT space (world, or the instance's)
    float4 hit = make_float4(0, 0, 0, 0);
    uint32_t hitInst = 0, rayMask = 0, mode = M_TLAS, node = 0, leafNext = 0, leafEnd = 0, curInst = 0, blasIdx = 0;
    uint32_t blay = (uint32_t)BLAS_LAYOUT;   // layout of the BLAS being traversed; BLAS_LAYOUT == 0: per instance (BlasDesc::layout)
    int base = 0;
    GlobalF4 bnodes, btris;
    // BVH4_GPU BLAS state (kernels_query.hip: k_bvh4); a BVH_GPU BLAS reuses offset (node), leafCnt (triangles left) and
    // leafQ0 (next triangle record) as k_bvh2's node / triLeft / triPtr
    uint32_t offset = 0, leafQ0 = 0, leafQ1 = 0, leafQ2 = 0, leafQ3 = 0, leafCnt = 0, leafCntB = 0;
    // CWBVH BLAS state (kernels_cwbvh.hip: k_cwbvh)
    uint32_t oct = 0;
    uint2 ng = make_uint2(0u, 0u), tg = make_uint2(0u, 0u);

    LockstepGovernor gov;   // ADAPT only: lockstep (whole-wave generations) while the rays are coherent, per-lane replacement otherwise
    gov.init();
    unsigned long long sIter = 0, sAct = 0, sA = 0, sLA = 0, sB = 0, sLB = 0, sC = 0, sLC = 0;   // STATS: phases run and the lanes in them
    for (;;) {
        const uint32_t nIdle = (uint32_t)__popcll(__ballot(!active));
        if ((ADAPT ? gov.want_refill(nIdle, (uint32_t)REFILL_MIN) : nIdle >= (uint32_t)REFILL_MIN) || nIdle == (uint32_t)WG) {
            if (!pool.dry()) {
                uint64_t nri = 0;
                if (pool.acquire(!active, q.counter, nRaysTotal, nri)) {
                    ri = nri;
                    const RayRec* rp = q.rays + ri;
                    O = xyz(rp->O); D = xyz(rp->D); rD = xyz(rp->rD);
                    rayMask = as_u32(rp->O.w);
                    hit = q.fresh ? make_float4(q.freshTmax, 0.f, 0.f, 0.f) : rp->hit;
                    hitInst = as_u32(rp->rD.w);
                    found = false; mode = M_TLAS; node = 0; st.sp = 0;
                    active = true;
                }
            }
            if (__ballot(active) == 0) break;
        }
        // Phase gating: a mode's code runs in this iteration only if at least PHASE_MIN lanes are in that mode, or it is the
        // mode most lanes are in (so somebody always makes progress).  Without it nearly every iteration pays for all three
        // code paths with a handful of lanes each; with it lanes regroup (a lane waits a few iterations for company).
        const uint32_t nA = (uint32_t)__popcll(__ballot(active && mode == M_TLAS)), nB = (uint32_t)__popcll(__ballot(active && mode == M_INST)),
                       nC = (uint32_t)__popcll(__ballot(active && mode == M_BLAS));
        const uint32_t nMax = nA > nB ? (nA > nC ? nA : nC) : (nB > nC ? nB : nC);
        const bool runA = PHASE_MIN <= 1 || nA >= (uint32_t)PHASE_MIN || nA == nMax, runB = PHASE_MIN <= 1 || nB >= (uint32_t)PHASE_MIN || nB == nMax,
                   runC = PHASE_MIN <= 1 || nC >= (uint32_t)PHASE_MIN || nC == nMax;
        if (STATS) { sIter++; sAct += nA + nB + nC; if (runA && nA) { sA++; sLA += nA; } if (runB && nB) { sB++; sLB += nB; } if (runC && nC) { sC++; sLC += nC; } }
        if (!active) continue;
        bool done = false;

        if (mode == M_BLAS) { if (runC) {
            bool pop = false;   // this lane's BLAS step ended with nothing pending: take the next stack entry (or leave the BLAS)
            const uint32_t lay = BLAS_LAYOUT ? (uint32_t)BLAS_LAYOUT : blay;
            if (lay == (uint32_t)kLayoutCwbvh) {
                if (tg.y != 0) {   // one triangle
                    const uint32_t ti = 31u - (uint32_t)__clz(tg.y);
                    tg.y &= ~(1u << ti);
                    const uint32_t ta = tg.x + ti * 3u;
                    const float4 e2 = btris[ta], e1 = btris[ta + 1], v0 = btris[ta + 2];
                    TriHit h;
                    if (tri_test(O, D, xyz(v0), xyz(e1), xyz(e2), hit.x, h)) {
                        const BlasDesc bd = blas[blasIdx];   // opacity micromaps are per BLAS: looked up only for a candidate hit
                        if (!bd.opmap || omm_opaque(Omm{bd.opmap, bd.opmapN}, as_u32(v0.w), h.u, h.v)) {
                            found = true; hitInst = curInst;
                            if (ANYHIT) done = true;
                            else hit = make_float4(h.t, h.u, h.v, v0.w);
                        }
                    }
                }
                if (!done && tg.y == 0) {
                    if (ng.y <= 0x00FFFFFFu) pop = true;
                    else {
                        const uint32_t imask = ng.y;
                        const uint32_t bit = 31u - (uint32_t)__clz(ng.y);
                        const uint32_t cbase = ng.x;
                        ng.y &= ~(1u << bit);
                        if (ng.y > 0x00FFFFFFu) st.push(ng);
                        const uint32_t slot = (bit - 24u) ^ oct;
                        const CwNodeHits nh = cw_test_node(cw_load_node(bnodes, cbase + __popc(imask & ~(0xFFFFFFFFu << slot))), O, rD, hit.x, oct * 0x01010101u);
                        ng.x = nh.childBase; tg.x = nh.triBase;
                        ng.y = (nh.hitmask & 0xFF000000u) | nh.imask;
                        tg.y = nh.hitmask & 0x00FFFFFFu;
                        if (tg.y == 0 && ng.y <= 0x00FFFFFFu) pop = true;
                    }
                }
                if (pop) {
                    if (st.sp == base) mode = M_INST;
                    else {
                        ng = st.pop();
                        if (ng.y <= 0x00FFFFFFu) { tg = ng; ng = make_uint2(0u, 0u); }   // a postponed triangle group
                    }
                }
            } else if (lay == (uint32_t)kLayoutBvh4Gpu) {
                if (leafCnt != 0) {   // one triangle of the pending leaves
                    const uint32_t ta = leafQ0;
                    const float4 v0 = bnodes[ta], e1 = bnodes[ta + 1], e2 = bnodes[ta + 2];
                    leafQ0 += 3u; leafCnt -= 1u;
                    if ((leafCnt & 0xffffu) == 0) {
                        leafQ0 = leafQ1; leafQ1 = leafQ2; leafQ2 = leafQ3;
                        leafCnt = __builtin_amdgcn_alignbit(leafCntB, leafCnt, 16); leafCntB >>= 16;
                    }
                    TriHit h;
                    if (tri_test(O, D, xyz(v0), xyz(e1), xyz(e2), hit.x, h)) {
                        const BlasDesc bd = blas[blasIdx];
                        if (!bd.opmap || omm_opaque(Omm{bd.opmap, bd.opmapN}, as_u32(v0.w), h.u, h.v)) {
                            found = true; hitInst = curInst;
                            if (ANYHIT) done = true;
                            else hit = make_float4(h.t, h.u, h.v, v0.w);
                        }
                    }
                    if (!done && leafCnt == 0) pop = true;
                } else {   // one node
                    const float4 d0 = bnodes[offset], d1 = bnodes[offset + 1], d2 = bnodes[offset + 2], d3 = bnodes[offset + 3];
                    const float sx = d1.x * rD.x, sy = d1.y * rD.y, sz = d1.z * rD.z;
                    const float bx = (d0.x - O.x) * rD.x, by = (d0.y - O.y) * rD.y, bz = (d0.z - O.z) * rD.z;
                    const uint32_t qx0 = as_u32(d0.w), qx1 = as_u32(d1.w);
                    const uint32_t qy0 = as_u32(d2.x), qy1 = as_u32(d2.y), qz0 = as_u32(d2.z), qz1 = as_u32(d2.w);
                    const bool ngx = sx < 0.f, ngy = sy < 0.f, ngz = sz < 0.f;   // near / far plane words by the sign of the direction
                    const uint32_t nx = ngx ? qx1 : qx0, fx = ngx ? qx0 : qx1;
                    const uint32_t ny = ngy ? qy1 : qy0, fy = ngy ? qy0 : qy1;
                    const uint32_t nz = ngz ? qz1 : qz0, fz = ngz ? qz0 : qz1;
                    float dist[4];
                    uint32_t info[4] = { as_u32(d3.x), as_u32(d3.y), as_u32(d3.z), as_u32(d3.w) };
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int sh = 8 * i;
                        const float x1 = __builtin_fmaf((float)((nx >> sh) & 255), sx, bx), x2 = __builtin_fmaf((float)((fx >> sh) & 255), sx, bx);
                        const float y1 = __builtin_fmaf((float)((ny >> sh) & 255), sy, by), y2 = __builtin_fmaf((float)((fy >> sh) & 255), sy, by);
                        const float z1 = __builtin_fmaf((float)((nz >> sh) & 255), sz, bz), z2 = __builtin_fmaf((float)((fz >> sh) & 255), sz, bz);
                        const float tmin = __builtin_fmaxf(fmax3(x1, y1, z1), 0.0f);
                        const float tmax = __builtin_fminf(fmin3(x2, y2, z2), hit.x);
                        dist[i] = (tmin > tmax || info[i] == 0) ? kFar : tmin;
                    }
#define TBVH_CSWAP(a, b) if (dist[a] < dist[b]) { const float tf = dist[a]; dist[a] = dist[b]; dist[b] = tf; const uint32_t tu = info[a]; info[a] = info[b]; info[b] = tu; }
                    TBVH_CSWAP(0, 2) TBVH_CSWAP(1, 3) TBVH_CSWAP(0, 1) TBVH_CSWAP(2, 3) TBVH_CSWAP(1, 2)
#undef TBVH_CSWAP
                    uint32_t nq = 0;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        if (!(dist[i] < kFar)) continue;
                        if (!(info[i] & 0x80000000u)) { st.push(make_uint2(info[i], 0u)); continue; }
                        const uint32_t cnt = (info[i] >> 16) & 0x7fffu;
                        if (cnt == 0) continue;
                        const uint32_t ta = offset + (info[i] & 0xffffu);
                        if (nq == 0) leafQ0 = ta; else if (nq == 1) leafQ1 = ta; else if (nq == 2) leafQ2 = ta; else leafQ3 = ta;
                        if (nq < 2) leafCnt |= cnt << (16 * nq); else leafCntB |= cnt << (16 * (nq - 2));
                        nq++;
                    }
                    if (leafCnt == 0) pop = true;
                }
                if (pop) {
                    if (st.sp == base) mode = M_INST;
                    else offset = st.pop().x;
                }
            } else {   // BVH_GPU (Aila-Laine 2-wide) BLAS: nodes 4 x float4, triangles gathered {v0|prim, e1, e2} (kernels_query.hip: k_bvh2)
                if (leafCnt != 0) {   // one triangle of the current leaf
                    const float4 v0 = btris[leafQ0], e1 = btris[leafQ0 + 1], e2 = btris[leafQ0 + 2];
                    leafQ0 += 3u; leafCnt -= 1u;
                    TriHit h;
                    if (tri_test(O, D, xyz(v0), xyz(e1), xyz(e2), hit.x, h)) {
                        const BlasDesc bd = blas[blasIdx];
                        if (!bd.opmap || omm_opaque(Omm{bd.opmap, bd.opmapN}, as_u32(v0.w), h.u, h.v)) {
                            found = true; hitInst = curInst;
                            if (ANYHIT) done = true;
                            else hit = make_float4(h.t, h.u, h.v, v0.w);
                        }
                    }
                    if (!done && leafCnt == 0) pop = true;
                } else {   // one node
                    const float4 n0 = bnodes[offset * 4], n1 = bnodes[offset * 4 + 1], n2 = bnodes[offset * 4 + 2], n3 = bnodes[offset * 4 + 3];
                    const uint32_t triCount = as_u32(n2.w);
                    if (triCount) { leafCnt = triCount; leafQ0 = as_u32(n3.w) * 3u; }
                    else {
                        const float3 ro = make_float3(O.x * rD.x, O.y * rD.y, O.z * rD.z);
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
                            offset = l;
                        } else if (hL) offset = l;
                        else if (hR) offset = r;
                        else pop = true;
                    }
                }
                if (pop) {
                    if (st.sp == base) mode = M_INST;
                    else offset = st.pop().x;
                }
            }
            if (mode == M_INST && !done) {   // back in world space: the ray as the caller gave it
                const RayRec* rp = q.rays + ri;
                O = xyz(rp->O); D = xyz(rp->D); rD = xyz(rp->rD);
            }
        } } else if (mode == M_INST) { if (runB) {
            if (leafNext == leafEnd) {   // TLAS leaf done
                if (st.sp == 0) done = true;
                else { node = st.pop().x; mode = M_TLAS; }
            } else {
                const uint32_t ii = tlasIdx[leafNext++];
                const float4* ip = instances + (size_t)ii * 12;
                const float4 b0 = ip[8], b1 = ip[9];                      // aabbMin|blasIdx, aabbMax|mask
                if (as_u32(b1.w) & rayMask) {                              // tiny_bvh.h:3326
                    const float4 r0 = ip[4], r1 = ip[5], r2 = ip[6], r3 = ip[7];   // invTransform rows
                    // tinybvh_transform_point / _vector with the reference build's contraction (see tlas_body)
                    const float px = __builtin_fmaf(r0.z, O.z, __builtin_fmaf(r0.x, O.x, r0.y * O.y)) + r0.w;
                    const float py = __builtin_fmaf(r1.z, O.z, __builtin_fmaf(r1.x, O.x, r1.y * O.y)) + r1.w;
                    const float pz = __builtin_fmaf(r2.z, O.z, __builtin_fmaf(r2.x, O.x, r2.y * O.y)) + r2.w;
                    const float w = __builtin_fmaf(r3.z, O.z, __builtin_fmaf(r3.x, O.x, r3.y * O.y)) + r3.w;
                    const float3 lD = make_float3(__builtin_fmaf(r0.z, D.z, __builtin_fmaf(r0.x, D.x, r0.y * D.y)), __builtin_fmaf(r1.z, D.z, __builtin_fmaf(r1.x, D.x, r1.y * D.y)),
                                                  __builtin_fmaf(r2.z, D.z, __builtin_fmaf(r2.x, D.x, r2.y * D.y)));
                    if (w == 1) O = make_float3(px, py, pz);
                    else { const float iw = 1.f / w; O = make_float3(px * iw, py * iw, pz * iw); }
                    D = lD;
                    rD = make_float3(safercp(D.x), safercp(D.y), safercp(D.z));
                    blasIdx = as_u32(b0.w);
                    const BlasDesc bd = blas[blasIdx];
                    bnodes = GlobalF4(bd.nodes); btris = GlobalF4(bd.tris);
                    curInst = ii; base = st.sp; mode = M_BLAS;
                    if (BLAS_LAYOUT == 0) blay = bd.layout;
                    if ((BLAS_LAYOUT ? (uint32_t)BLAS_LAYOUT : blay) == (uint32_t)kLayoutCwbvh) {
                        oct = 7u - ((D.x < 0 ? 4u : 0u) | (D.y < 0 ? 2u : 0u) | (D.z < 0 ? 1u : 0u));
                        ng = make_uint2(0u, 0x80000000u); tg = make_uint2(0u, 0u);
                    } else { offset = 0; leafCnt = 0; leafCntB = 0; }
                }
            }
        } } else if (runA) {
            const float4 n0 = tlasNodes[node * 4], n1 = tlasNodes[node * 4 + 1], n2 = tlasNodes[node * 4 + 2], n3 = tlasNodes[node * 4 + 3];
            const uint32_t cnt = as_u32(n2.w);
            if (cnt) { leafNext = as_u32(n3.w); leafEnd = leafNext + cnt; mode = M_INST; }
            else {
                // SLAB_TEST_TWO_NODES form (tiny_bvh.h:3202-3220), as in the BVH_GPU kernel
                const float3 ro = make_float3(O.x * rD.x, O.y * rD.y, O.z * rD.z);
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
                    if (st.sp == 0) done = true;
                    else node = st.pop().x;
                }
            }
        }
        if (done) {
            RayRec* rp = q.rays + ri;
            if (ANYHIT) q.occluded[ri] = found ? 1 : 0;
            else if (found) { rp->hit = hit; ((uint32_t*)rp)[11] = hitInst; }   // byte 44 = hit.inst
            else if (q.fresh) rp->hit = hit;
            active = false;
        }
    }
    if (STATS && (threadIdx.x & 63u) == 0) {
        atomicAdd(q.stats + 0, sIter); atomicAdd(q.stats + 1, sAct); atomicAdd(q.stats + 2, sA); atomicAdd(q.stats + 3, sLA);
        atomicAdd(q.stats + 4, sB); atomicAdd(q.stats + 5, sLB); atomicAdd(q.stats + 6, sC); atomicAdd(q.stats + 7, sLC);
    }
}

// kernel prologue shared by all TLAS kernels: LDS stack top, spill area, ray pool
#define TBVH_TLAS_PROLOGUE                                                                                                          \
    __shared__ uint2 stk[LDS_N][WG];                                                                                                \
    __shared__ uint32_t ldsTrips;                                                                                                   \
    StackT<LDS_N> st;                                                                                                               \
    st.init(&stk[0][threadIdx.x], (uint2*)q.spill + (blockIdx.x * WG + threadIdx.x), (size_t)gridDim.x * WG, q.spillStride);       \
    RayPool<64> pool;                                                                                                               \
    const uint64_t nRaysTotal = q.nRaysDev ? *q.nRaysDev : q.nRays; /* batch size may live on the device (wavefront queues) */      \
    pool.init(q.poolParts);

template <bool ANYHIT, int BLAS_LAYOUT, int LDS_N = 12, int REFILL_MIN = 16, int PHASE_MIN = 16, bool ADAPT = false>
__global__ __launch_bounds__(WG) void k_tlas_flat_stats(const float4* __restrict__ tlasNodes, const uint32_t* __restrict__ tlasIdx,
                                                  const float4* __restrict__ instances, const BlasDesc* __restrict__ blas, QueryArgs q,
                                                  uint32_t* __restrict__ status) {
    TBVH_TLAS_PROLOGUE
    (void)ldsTrips;
    tlas_flat_body<ANYHIT, BLAS_LAYOUT, LDS_N, REFILL_MIN, PHASE_MIN, ADAPT, true>(tlasNodes, tlasIdx, instances, blas, q, st, pool, nRaysTotal);
    if (st.overflow) atomicOr(status, 1u);
}
template <bool ANYHIT, int BLAS_LAYOUT, int LDS_N = 12, int REFILL_MIN = 16, int PHASE_MIN = 16, bool ADAPT = false>
__global__ __launch_bounds__(WG) void k_tlas_flat(const float4* __restrict__ tlasNodes, const uint32_t* __restrict__ tlasIdx,
                                                  const float4* __restrict__ instances, const BlasDesc* __restrict__ blas, QueryArgs q,
                                                  uint32_t* __restrict__ status) {
    TBVH_TLAS_PROLOGUE
    (void)ldsTrips;
    tlas_flat_body<ANYHIT, BLAS_LAYOUT, LDS_N, REFILL_MIN, PHASE_MIN, ADAPT>(tlasNodes, tlasIdx, instances, blas, q, st, pool, nRaysTotal);
    if (st.overflow) atomicOr(status, 1u);
}
template <bool ANYHIT, int BLAS_LAYOUT, int LDS_N = 12, int REFILL_MIN = 16, int PHASE_MIN = 16, bool ADAPT = false>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_tlas_flat_w5(const float4* __restrict__ tlasNodes, const uint32_t* __restrict__ tlasIdx,
                                                  const float4* __restrict__ instances, const BlasDesc* __restrict__ blas, QueryArgs q,
                                                  uint32_t* __restrict__ status) {
    TBVH_TLAS_PROLOGUE
    (void)ldsTrips;
    tlas_flat_body<ANYHIT, BLAS_LAYOUT, LDS_N, REFILL_MIN, PHASE_MIN, ADAPT>(tlasNodes, tlasIdx, instances, blas, q, st, pool, nRaysTotal);
    if (st.overflow) atomicOr(status, 1u);
}
template <bool ANYHIT, int BLAS_LAYOUT, int LDS_N = 12, int REFILL_MIN = 16, int PHASE_MIN = 16, bool ADAPT = false>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_tlas_flat_w6(const float4* __restrict__ tlasNodes, const uint32_t* __restrict__ tlasIdx,
                                                  const float4* __restrict__ instances, const BlasDesc* __restrict__ blas, QueryArgs q,
                                                  uint32_t* __restrict__ status) {
    TBVH_TLAS_PROLOGUE
    (void)ldsTrips;
    tlas_flat_body<ANYHIT, BLAS_LAYOUT, LDS_N, REFILL_MIN, PHASE_MIN, ADAPT>(tlasNodes, tlasIdx, instances, blas, q, st, pool, nRaysTotal);
    if (st.overflow) atomicOr(status, 1u);
}

template <bool ANYHIT, int BLAS_LAYOUT, int LDS_N = 16>
__global__ __launch_bounds__(WG) void k_tlas(const float4* __restrict__ tlasNodes, const uint32_t* __restrict__ tlasIdx,
                                             const float4* __restrict__ instances, const BlasDesc* __restrict__ blas, QueryArgs q,
                                             uint32_t* __restrict__ status) {
    TBVH_TLAS_PROLOGUE
    tlas_body<ANYHIT, BLAS_LAYOUT, LDS_N, 0, 0u>(tlasNodes, tlasIdx, instances, blas, q, st, pool, nRaysTotal, (TBVH_AS_LDS uint32_t*)&ldsTrips);
    if (st.overflow) atomicOr(status, 1u);
}
// the same with the register budget of 5 waves per SIMD (<= 96 VGPRs; left alone the compiler takes 88-115)
template <bool ANYHIT, int BLAS_LAYOUT, int LDS_N = 16>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_tlas_w5(const float4* __restrict__ tlasNodes, const uint32_t* __restrict__ tlasIdx,
                                             const float4* __restrict__ instances, const BlasDesc* __restrict__ blas, QueryArgs q,
                                             uint32_t* __restrict__ status) {
    TBVH_TLAS_PROLOGUE
    tlas_body<ANYHIT, BLAS_LAYOUT, LDS_N, 0, 0u>(tlasNodes, tlasIdx, instances, blas, q, st, pool, nRaysTotal, (TBVH_AS_LDS uint32_t*)&ldsTrips);
    if (st.overflow) atomicOr(status, 1u);
}

// Adaptive: every wave starts with the nested loops (fastest while its 64 rays stay together) and moves to the flat loop
// with per-lane replacement for the rest of the launch once a generation's lane cohesion drops below KEEP / 256.
template <bool ANYHIT, int BLAS_LAYOUT, int LDS_N = 12, uint32_t KEEP = 128, int STATS = 0>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_tlas_adaptive(const float4* __restrict__ tlasNodes, const uint32_t* __restrict__ tlasIdx,
                                             const float4* __restrict__ instances, const BlasDesc* __restrict__ blas, QueryArgs q,
                                             uint32_t* __restrict__ status) {
    TBVH_TLAS_PROLOGUE
    if (!tlas_body<ANYHIT, BLAS_LAYOUT, LDS_N, STATS ? 2 : 1, KEEP>(tlasNodes, tlasIdx, instances, blas, q, st, pool, nRaysTotal, (TBVH_AS_LDS uint32_t*)&ldsTrips))
        tlas_flat_body<ANYHIT, BLAS_LAYOUT, LDS_N, 16, 32, false>(tlasNodes, tlasIdx, instances, blas, q, st, pool, nRaysTotal);
    if (st.overflow) atomicOr(status, 1u);
}

}  // namespace

void launch_tlas(bool anyhit, int blasLayout, int variant, const float4* tlasNodes, const uint32_t* tlasIdx, const float4* instances,
                 const BlasDesc* blas, const QueryArgs& q, uint32_t* status, uint32_t blocks, hipStream_t s) {
#define TBVH_LT(K, ...)                                                                                                                      \
    do {                                                                                                                                \
        if (blasLayout == kLayoutCwbvh) {                                                                                                       \
            if (anyhit) hipLaunchKernelGGL((K<true, kLayoutCwbvh, ##__VA_ARGS__>), dim3(blocks), dim3(WG), 0, s, tlasNodes, tlasIdx, instances, blas, q, status); \
            else hipLaunchKernelGGL((K<false, kLayoutCwbvh, ##__VA_ARGS__>), dim3(blocks), dim3(WG), 0, s, tlasNodes, tlasIdx, instances, blas, q, status);       \
        } else {                                                                                                                        \
            if (anyhit) hipLaunchKernelGGL((K<true, kLayoutBvh4Gpu, ##__VA_ARGS__>), dim3(blocks), dim3(WG), 0, s, tlasNodes, tlasIdx, instances, blas, q, status); \
            else hipLaunchKernelGGL((K<false, kLayoutBvh4Gpu, ##__VA_ARGS__>), dim3(blocks), dim3(WG), 0, s, tlasNodes, tlasIdx, instances, blas, q, status);       \
        }                                                                                                                               \
    } while (0)
    // Defaults (1000 instances of a 100 k-triangle BLAS; 8.3 M camera rays / 8.4 M random rays, Intersect, MRays/s):
    //                                   BVH4_GPU BLASes        CWBVH BLASes
    //   nested loops (k_tlas_w5)        4350 /  565            3140 /  385
    //   flat loop (k_tlas_flat_w6)      3280 / 1315            3320 / 1690
    //   nested, then flat (adaptive)    4300 / 1370            3000 /  920
    // CWBVH BLASes: the flat loop wins on both; BVH4_GPU BLASes: the nested loops are 30 % faster on camera rays and 2.4 x
    // slower on incoherent ones, so every wave starts nested and moves to the flat loop once its running estimate of the
    // lane cohesion of a 64-ray generation (mean / max of the lanes' TLAS-node and instance-entry counts) falls below 0.55:
    // camera rays: 99 % of the generations above 0.5; random rays: 99.7 % below.
    // BVH_GPU BLASes and TLASes that mix BLAS layouts (as traverse_tlas.cl:50-72 allows: blasDesc[].blasType) exist in the flat loop only
    if (blasLayout == kLayoutBvhGpu || blasLayout == 0) {
        if (blasLayout == kLayoutBvhGpu) {
            if (anyhit) hipLaunchKernelGGL((k_tlas_flat_w6<true, kLayoutBvhGpu, 12, 16, 32>), dim3(blocks), dim3(WG), 0, s, tlasNodes, tlasIdx, instances, blas, q, status);
            else hipLaunchKernelGGL((k_tlas_flat_w6<false, kLayoutBvhGpu, 12, 16, 32>), dim3(blocks), dim3(WG), 0, s, tlasNodes, tlasIdx, instances, blas, q, status);
        } else {
            if (anyhit) hipLaunchKernelGGL((k_tlas_flat_w5<true, 0, 12, 16, 32>), dim3(blocks), dim3(WG), 0, s, tlasNodes, tlasIdx, instances, blas, q, status);
            else hipLaunchKernelGGL((k_tlas_flat_w5<false, 0, 12, 16, 32>), dim3(blocks), dim3(WG), 0, s, tlasNodes, tlasIdx, instances, blas, q, status);
        }
        return;
    }
#if TBVH_EXPERIMENTS
    if (variant == 15) TBVH_LT(k_tlas_flat_stats, 12, 16, 32);   // statistics of the flat loop: phases run, lanes per phase
    else if (variant == 6) TBVH_LT(k_tlas_flat_w6, 12, 16, 32, true);   // flat loop under the lockstep governor
    else if (variant == 7) TBVH_LT(k_tlas_flat_w6, 12, 16, 32);         // flat loop, per-lane replacement throughout
    else if (variant == 9) TBVH_LT(k_tlas_flat_w6, 12, 64, 16);         // flat loop, lockstep throughout
    // flat-loop parameters swept without effect beyond +-3 %: phase threshold 24 / 40 / 48, refill threshold 8 / 24 / 32 (8: incoherent
    // rays +3 %, camera rays -3 %), register budgets of 5 waves per SIMD or the compiler's own (-8 %, -2 %), 8- / 16-entry LDS stack top (0 %, -12 %)
    else if (variant == 12) TBVH_LT(k_tlas_adaptive, 12, 128, 1);        // statistics: cohesion histogram of the nested loops
    else if (variant == 13) TBVH_LT(k_tlas_adaptive, 12, 128);
    else if (variant == 14) TBVH_LT(k_tlas_adaptive, 12, 154);
    else if (variant == 2) TBVH_LT(k_tlas_w5, 8);
    else if (variant == 4) TBVH_LT(k_tlas, 8);
    else if (variant == 5) TBVH_LT(k_tlas, 16);        // round-1 kernel: nested loops, compiler's register budget (4 waves per SIMD)
    else if (variant == 1) TBVH_LT(k_tlas_w5, 16);
    else if (variant != 0) TBVH_LT(k_tlas_w5, 12);       // 3: nested loops at 5 waves per SIMD, 12-entry LDS stack top
    else
#endif
    // CWBVH BLASes: the flat loop.  BVH4_GPU BLASes are served by the unified 4-wide kernel (kernels_tlas4.hip, launch_tlas4); this
    // path only sees them when that TLAS could not be built (more than 2^31 blocks), and then the flat loop is the safe choice
    TBVH_LT(k_tlas_flat_w6, 12, 16, 32);
#undef TBVH_LT
}

bool tlas_variant_valid(int v) { return TBVH_EXPERIMENTS ? ((v >= 0 && v <= 15) || (v >= 21 && v <= 36)) : v == 0; }

}  // namespace tbvh
