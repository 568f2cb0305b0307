// kernels_tlas4.hip — two-level Intersect / IsOccluded over BVH4_GPU BLASes with the TLAS held in the SAME node format.
//
// Why: in the flat TLAS loop of kernels_tlas.hip a lane is in one of three modes (TLAS node, instance entry, BLAS step)
// and a wave runs one mode's code per iteration; with incoherent rays the 64 lanes split about 45 / 10 / 45 % over the
// modes, so every pass over the loop body serves ~22 lanes (rocprofv3 + the statistics variant, 1000 instances,
// 4.2 M random rays: 94 wave-iterations per ray for 37.5 steps, 23 k issued lane-operations per ray of which 16 % are useful,
// VALU 80 % busy: profiles/r02_tlas_flat_counters.txt).  The two big modes differ only in the node format they decode.
// So the TLAS the caller uploads (BVH_GPU nodes over BLASInstance records, tiny_bvh.h:4575-4581, or the device-built
// LBVH of kernels_tlasbuild.hip) is collapsed once per upload / rebuild (kernels_tlaswide.hip) into a 4-wide quantised tree in the BVH4_GPU node
// format (4 blocks: bmin | qxmin, ext/255 | qxmax, qymin qymax qzmin qzmax, childInfo[4]; tiny_bvh.h:1248-1266), whose leaf
// children name an instance (childInfo = 1 << 31 | instance index) instead of an inline triangle run.  A TLAS node
// step and a BLAS node step are then THE SAME CODE on different base pointers, all lanes that have a node to visit take it
// together, and half as many TLAS steps are needed (4-wide instead of 2-wide).  Instance boxes are only culling volumes:
// quantising them (conservatively, as the BLAS encoder does) changes no hit record.
//
// Per-ray order: children nearest first (the BVH4_GPU kernel's sorting network), instances and interior children of a
// TLAS node interleaved by entry distance.  That is not the order of the nested reference loop (traverse_tlas.cl:13-107),
// so among hits at exactly equal t the winner may differ (the tie class of tests/oracle_lib.py).
#include "device_common.h"
#include "lane_stack.h"
#include "ray_pool.h"
#include "ray_split.h"
#include "kernels.h"

namespace tbvh {

namespace {

constexpr int WG = 64;

__device__ __forceinline__ float fmin3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
__device__ __forceinline__ float fmax3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ float safercp(float x) {
    if (x > 1e-12f || x < -1e-12f) return 1.0f / x;
    return x >= 0 ? kFar : -kFar;
}

// =====================================================================================================================
// traversal
// =====================================================================================================================
template <int LDS_N> using Stack32 = LaneStack<uint32_t, LDS_N, WG>;

// lane states of an active ray
enum : uint32_t { S_NODE = 0, S_TRI = 1, S_INST = 2 };

// PN / PT / PI: a state's code runs in an iteration if at least that many lanes are in the state, or it holds the most lanes.
// ADAPT: rays are taken under the lockstep governor (ray_pool.h): whole 64-ray generations while the wave's rays stay together
// STEAL > 0 (idle lanes needed): once the ray pool is dry, idle lanes take the top stack entry — a BLAS subtree, a TLAS subtree or an instance —
// off a lane that is still traversing (ray_split.h)
template <bool ANYHIT, int LDS_N, int REFILL_MIN, int PN, int PT, int PI, bool ADAPT, bool STATS, int STEAL = 0, bool FUSE = false>
__device__ __forceinline__ void tlas4_body(const float4* __restrict__ tlas4, const float4* __restrict__ instances, const BlasDesc* __restrict__ blas,
                                           const QueryArgs& q, uint32_t* __restrict__ status) {
    __shared__ uint32_t stk[LDS_N][WG];
    Stack32<LDS_N> st;
    st.init(&stk[0][threadIdx.x], q.spill + (blockIdx.x * WG + threadIdx.x), (size_t)gridDim.x * WG, q.spillStride);
    RayPool<64> pool;
    const uint64_t nRaysTotal = q.nRaysDev ? *q.nRaysDev : q.nRays;
    pool.init(q.poolParts, q.counterNext);

    bool active = false, found = false, inBlas = false;
    uint64_t ri = 0;
    float3 O = make_float3(0, 0, 0), D = O, rD = O;   // the ray in the CURRENT space (world, or the instance's)
    float4 hit = make_float4(0, 0, 0, 0);
    uint32_t hitInst = 0, rayMask = 0, state = S_NODE, offset = 0, curInst = 0, blasIdx = 0;
    int base = 0;                                     // stack height at which the current BLAS traversal began
    GlobalF4 cur(tlas4);                              // the node stream this lane is walking: the TLAS or an instance's BLAS
    // pending leaves of the current BLAS node (kernels_query.hip: bvh4_body)
    uint32_t leafQ0 = 0, leafQ1 = 0, leafQ2 = 0, leafQ3 = 0, leafCnt = 0, leafCntB = 0;
    unsigned long long sIter = 0, sAct = 0, sN = 0, sLN = 0, sT = 0, sLT = 0, sI = 0, sLI = 0;   // STATS
    __shared__ SplitLds<STEAL ? WG : 1> split;
    int grp = -1;
    LockstepGovernor gov;   // ADAPT only
    gov.init();

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
                    found = false; inBlas = false; state = S_NODE; offset = 0; leafCnt = 0; leafCntB = 0; st.sp = 0;
                    cur = GlobalF4(tlas4);
                    active = true;
                }
            }
            if (__ballot(active) == 0) break;
        }
        const bool tail = STEAL && pool.dry();   // wave-uniform: nothing of the split-ray code costs a vector instruction before the pool is dry
        if (tail && nIdle >= (uint32_t)STEAL) {
            SplitMatch m;
            if (split_match(active && st.sp != 0, !active, m)) {
                uint32_t part = 0;
                int lvl = 0;   // the entry belongs to the donor's BLAS traversal (else to the TLAS level, in world space)
                if (m.gives) {
                    lvl = inBlas && st.sp > base;
                    part = st.pop();
                    if (inBlas && !lvl) base = st.sp;   // the donor's BLAS part of the stack was empty: it now begins one entry lower
                    split_give<ANYHIT>(split, m, grp, found, hit, hitInst);
                }
                const int src = split_take_ray(split, m, O, D, rD, hit, ri, grp);
                part = __shfl(part, src); lvl = __shfl(lvl, src);
                const int donorInBlas = __shfl((int)inBlas, src);
                rayMask = __shfl(rayMask, src); curInst = __shfl(curInst, src); blasIdx = __shfl(blasIdx, src);
                if (m.takes) {
                    found = false; leafCnt = 0; leafCntB = 0; st.sp = 0; base = 0;
                    if (lvl) { inBlas = true; cur = GlobalF4(blas[blasIdx].nodes); state = S_NODE; offset = part; }   // (back at its empty stack the lane "returns" to a TLAS level with nothing left: done)
                    else {
                        inBlas = false; cur = GlobalF4(tlas4);
                        if (donorInBlas) { const RayRec* rp = q.rays + ri; O = xyz(rp->O); D = xyz(rp->D); rD = xyz(rp->rD); }   // the donor's registers hold the instance-space ray
                        if (part & 0x80000000u) { state = S_INST; offset = part & 0x7fffffffu; }
                        else { state = S_NODE; offset = part; }
                    }
                    active = true;
                }
            }
        }
        // Phase gating (PN / PT / PI above)
        const uint32_t nN = (uint32_t)__popcll(__ballot(active && state == S_NODE)), nT = (uint32_t)__popcll(__ballot(active && state == S_TRI)),
                       nI = (uint32_t)__popcll(__ballot(active && state == S_INST));
        const uint32_t nMax = nN > nT ? (nN > nI ? nN : nI) : (nT > nI ? nT : nI);
        const bool runN = nN >= (uint32_t)PN || nN == nMax, runT = nT >= (uint32_t)PT || nT == nMax, runI = nI >= (uint32_t)PI || nI == nMax;
        if (STATS) { sIter++; sAct += nN + nT + nI; if (runN && nN) { sN++; sLN += nN; } if (runT && nT) { sT++; sLT += nT; } if (runI && nI) { sI++; sLI += nI; } }
        if (!active) continue;
        bool done = false, advance = false;   // advance: nothing pending at this level, take the next stack entry
        if (tail && grp >= 0) split_poll<ANYHIT>(split, grp, hit, done);   // a split ray: bounded by its group's closest hit

        // FUSE: a lane whose pending leaves are finished with more of the BLAS on its stack, or that has just entered an instance, takes its node step in the same pass
        bool cont = false;
        if (STEAL && ANYHIT && done) {
        } else if (state == S_TRI) { if (runT) {
            // ---- one triangle of the pending leaves of a BLAS node ---------------------------------------------------
            const uint32_t ta = leafQ0;
            const float4 v0 = cur[ta], e1 = cur[ta + 1], e2 = cur[ta + 2];
            leafQ0 += 3u; leafCnt -= 1u;
            if ((leafCnt & 0xffffu) == 0) {
                leafQ0 = leafQ1; leafQ1 = leafQ2; leafQ2 = leafQ3;
                leafCnt = __builtin_amdgcn_alignbit(leafCntB, leafCnt, 16); leafCntB >>= 16;
            }
            TriHit h;
            if (tri_test(O, D, xyz(v0), xyz(e1), xyz(e2), hit.x, h) && (ANYHIT || (tail && grp >= 0) || hit_wins(h.t, as_u32(v0.w), curInst, found, hit, hitInst))) {
                const BlasDesc bd = blas[blasIdx];   // opacity micromaps are per BLAS: looked up only for a candidate hit
                if (!bd.opmap || omm_opaque(Omm{bd.opmap, bd.opmapN}, as_u32(v0.w), h.u, h.v)) {
                    found = true; hitInst = curInst;
                    if (ANYHIT) done = true;
                    else hit = make_float4(h.t, h.u, h.v, v0.w);
                    if (tail && grp >= 0) split_publish<ANYHIT, true>(split, grp, hit, hitInst);
                }
            }
            if (!done && leafCnt == 0) {
                if (FUSE && st.sp > base) { offset = st.pop(); state = S_NODE; cont = true; }   // (inside a BLAS the stack holds node offsets only)
                else advance = true;
            }
        } } else if (state == S_INST) { if (runI) {
            // ---- enter an instance: `offset` holds its index (tiny_bvh.h:3326-3333) ----------------------------------
            const uint32_t ii = offset;
            const float4* ip = instances + (size_t)ii * 12;
            const float4 b0 = ip[8], b1 = ip[9];                      // aabbMin|blasIdx, aabbMax|mask
            if (as_u32(b1.w) & rayMask) {
                const float4 r0 = ip[4], r1 = ip[5], r2 = ip[6], r3 = ip[7];   // invTransform rows
                // tinybvh_transform_point / _vector with the reference build's contraction (kernels_tlas.hip: tlas_body)
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
                cur = GlobalF4(blas[blasIdx].nodes);
                curInst = ii; base = st.sp; inBlas = true;
                state = S_NODE; offset = 0; leafCnt = 0; leafCntB = 0; cont = FUSE;
            } else advance = true;
        } } else if (runN) cont = true;
        if (cont && runN && state == S_NODE && !done) {
            // ---- one node of the TLAS or of the instance's BLAS: same format, same code -----------------------------------
            const float4 d0 = cur[offset], d1 = cur[offset + 1], d2 = cur[offset + 2], d3 = cur[offset + 3];
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
                const float tmax = __builtin_fminf(fmin3(x2, y2, z2), cull_bound(hit.x));
                dist[i] = (tmin > tmax || info[i] == 0) ? kFar : tmin;
            }
#define TBVH_CSWAP(a, b) if (dist[a] < dist[b]) { const float tf = dist[a]; dist[a] = dist[b]; dist[b] = tf; const uint32_t tu = info[a]; info[a] = info[b]; info[b] = tu; }
            TBVH_CSWAP(0, 2) TBVH_CSWAP(1, 3) TBVH_CSWAP(0, 1) TBVH_CSWAP(2, 3) TBVH_CSWAP(1, 2)
#undef TBVH_CSWAP
            uint32_t nq = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {   // farthest first: the nearest child ends up on top of the stack; leaves queue in the order of bvh4_body (kernels_query.hip)
                if (!(dist[i] < kFar)) continue;
                if (!(info[i] & 0x80000000u) || !inBlas) { st.push(info[i]); continue; }   // interior child, or (TLAS) an instance: 1 << 31 | index
                const uint32_t cnt = (info[i] >> 16) & 0x7fffu;
                if (cnt == 0) continue;
                const uint32_t ta = offset + (info[i] & 0xffffu);
                if (nq == 0) leafQ0 = ta; else if (nq == 1) leafQ1 = ta; else if (nq == 2) leafQ2 = ta; else leafQ3 = ta;
                if (nq < 2) leafCnt |= cnt << (16 * nq); else leafCntB |= cnt << (16 * (nq - 2));
                nq++;
            }
            if (leafCnt != 0) state = S_TRI; else advance = true;
        }
        if (advance && !done) {
            // ---- next stack entry; a BLAS traversal that is back at its base returns to the TLAS with the world ray ----------
            if (inBlas && st.sp == base) {
                inBlas = false; cur = GlobalF4(tlas4);
                const RayRec* rp = q.rays + ri;
                O = xyz(rp->O); D = xyz(rp->D); rD = xyz(rp->rD);
            }
            if (st.sp == 0) done = true;
            else {
                const uint32_t e = st.pop();
                if (!inBlas && (e & 0x80000000u)) { state = S_INST; offset = e & 0x7fffffffu; }
                else { state = S_NODE; offset = e; }
            }
        }
        if (done) {
            RayRec* rp = q.rays + ri;
            if (tail && grp >= 0) split_finish<ANYHIT, true>(split, grp, q, ri);
            else if (ANYHIT) q.occluded[ri] = found ? 1 : 0;
            else if (found) { rp->hit = hit; ((uint32_t*)rp)[11] = hitInst; }   // byte 44 = hit.inst
            else if (q.fresh) rp->hit = hit;
            active = false;
        }
    }
    if (st.overflow) atomicOr(status, 1u);
    if (STATS && (threadIdx.x & 63u) == 0) {
        atomicAdd(q.stats + 0, sIter); atomicAdd(q.stats + 1, sAct); atomicAdd(q.stats + 2, sN); atomicAdd(q.stats + 3, sLN);
        atomicAdd(q.stats + 4, sI); atomicAdd(q.stats + 5, sLI); atomicAdd(q.stats + 6, sT); atomicAdd(q.stats + 7, sLT);
    }
}

template <bool ANYHIT, int LDS_N = 12, int REFILL_MIN = 16, int PN = 24, int PT = 8, int PI = 8, bool ADAPT = false, bool STATS = false, int STEAL = 0, int WAVES = 6, bool FUSE = false>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_tlas4(const float4* __restrict__ tlas4, const float4* __restrict__ instances,
                                                                                           const BlasDesc* __restrict__ blas, QueryArgs q, uint32_t* __restrict__ status) {
    tlas4_body<ANYHIT, LDS_N, REFILL_MIN, PN, PT, PI, ADAPT, STATS, STEAL, FUSE>(tlas4, instances, blas, q, status);
}

}  // namespace

void launch_tlas4(bool anyhit, int variant, const float4* tlas4, const float4* instances, const BlasDesc* blas, const QueryArgs& q, uint32_t* status, uint32_t blocks,
                  hipStream_t s, uint32_t blocks7) {
#define TBVH_T4(...)                                                                                                                     \
    do {                                                                                                                                 \
        if (anyhit) hipLaunchKernelGGL((k_tlas4<true, __VA_ARGS__>), dim3(blocks), dim3(WG), 0, s, tlas4, instances, blas, q, status);  \
        else hipLaunchKernelGGL((k_tlas4<false, __VA_ARGS__>), dim3(blocks), dim3(WG), 0, s, tlas4, instances, blas, q, status);        \
    } while (0)
    (void)variant;
    // thresholds measured on 1000 instances of a 100 k-triangle BLAS, 8.3 M camera / 4.2 M random rays, Intersect MRays/s (IsOccluded on the
    // random rays): 32/32/32 3570 / 2100 (2730); 16/16/16 4010 / 2340 (2910); 16/8/8 4350 / 2480 (3060); 24/8/8 4400 / 2480 (3120);
    // 8/8/8 4200 / 2490 (3050); 16/4/4 4130 / 2380 (2930); under the lockstep governor -2 %; 8-entry LDS stack top -6 %.
    // Before (nested-then-flat over the 2-wide TLAS, kernels_tlas.hip): 4170 / 1220 (1490).
    // batches below 12 M rays, and the wavefront stages (ray count known to the device only), split their last rays over idle lanes (ray_split.h)
    // Register budget of 7 waves per SIMD (72 VGPRs, three spilled dwords) on 28 workgroups per CU: the BLASes of an instanced scene live in the L2s,
    // the loop is latency-bound — 8.3 M camera rays +4 %, 33 M +8 %, random rays +2…5 % over 6 waves; 8 waves (64 VGPRs, eight spilled) lose
    blocks = blocks7;
#ifdef TBVH_EXPERIMENTS
    // round 6, late: the phase thresholds once more, now that fused steps and 7 waves per SIMD are in (debug flags bits 16..19 pick a row; camera rays of config 5)
    switch ((q.flags >> 16) & 15u) {
    case 1: TBVH_T4(12, 16, 32, 8, 8, false, false, 16, 7, true); return;
    case 2: TBVH_T4(12, 16, 16, 8, 8, false, false, 16, 7, true); return;
    case 3: TBVH_T4(12, 16, 24, 4, 4, false, false, 16, 7, true); return;
    case 4: TBVH_T4(12, 16, 24, 16, 16, false, false, 16, 7, true); return;
    case 5: TBVH_T4(12, 16, 24, 8, 16, false, false, 16, 7, true); return;
    case 6: TBVH_T4(12, 16, 24, 16, 8, false, false, 16, 7, true); return;
    case 7: TBVH_T4(12, 8, 24, 8, 8, false, false, 16, 7, true); return;
    case 8: TBVH_T4(12, 32, 24, 8, 8, false, false, 16, 7, true); return;
    case 9: TBVH_T4(12, 16, 40, 8, 8, false, false, 16, 7, true); return;
    default: break;
    }
#endif
    // fused steps (a lane done with its leaves, or entering an instance, takes its node step in the same pass): camera rays +4.5 %, random rays +2.5 %
    if (split_rays_wanted(q)) TBVH_T4(12, 16, 24, 8, 8, false, false, 16, 7, true);
    else TBVH_T4(12, 16, 24, 8, 8, false, false, 0, 7, true);
#undef TBVH_T4
}

}  // namespace tbvh
