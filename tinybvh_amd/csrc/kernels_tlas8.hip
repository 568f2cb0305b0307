// kernels_tlas8.hip — two-level Intersect / IsOccluded over BVH8_CWBVH BLASes with the TLAS held in the CWBVH node format: the
// configuration of the reference's instancing demo (tiny_bvh_gpu2.cpp: CWBVH BLASes under a BVH_GPU TLAS, traverse_tlas.cl:13-107).
//
// Same idea as kernels_tlas4.hip (which see for the measurements that motivate it): the caller's TLAS is collapsed at every upload /
// update / device rebuild (kernels_tlaswide.hip) into an 8-wide compressed tree in the BVH8_CWBVH node format (cwbvh_node.h) whose leaf children stand for
// ONE instance each (a "triangle" slot of the node: meta = 0b001 << 5 | slot offset, the node's triangle base indexes a list of
// instance indices), so that a TLAS node step and a BLAS node step are the same code — cw_test_node on different base pointers — and
// the node phase of the loop serves every lane that has a node to visit, whatever its level.  The triangle phase exists twice: a
// BLAS-level lane tests a triangle, a TLAS-level lane enters an instance.
//
// Traversal state is Ylitie's (node group / triangle group, kernels_cwbvh.hip); entering an instance parks the TLAS-level groups on
// the stack below the BLAS traversal (a parked instance group is recognised by hits == 0 in its upper byte, like the postponed triangle
// groups of the reference kernel).  Within a TLAS node the hit instances are entered before its interior children, children in octant
// order: not the order of the nested reference loop, so among hits at exactly equal t the winner may differ (the tie class).
#include "device_common.h"
#include "lane_stack.h"
#include "ray_pool.h"
#include "ray_split.h"
#include "kernels.h"
#include "cwbvh_node.h"

namespace tbvh {

namespace {

constexpr int WG = 64;

__device__ __forceinline__ float safercp(float x) {
    if (x > 1e-12f || x < -1e-12f) return 1.0f / x;
    return x >= 0 ? kFar : -kFar;
}

// =====================================================================================================================
// traversal
// =====================================================================================================================
enum : uint32_t { S_NODE = 0, S_TRI = 1, S_INST = 2, S_NODE2 = 3, S_TRI2 = 4 };   // S_NODE2 / S_TRI2: a BVH_GPU BLAS under the 8-wide TLAS (MIXED)

// STEAL > 0 (idle lanes needed): once the ray pool is dry, idle lanes take the top stack entry — a node group of the BLAS or of the TLAS, or a
// parked instance group — off a lane that is still traversing (ray_split.h)
// MIXED: the BLASes are BVH8_CWBVH or BVH_GPU, instance by instance (BlasDesc::layout) — the two BLAS types of the reference's traverse_tlas
// (traverse_tlas.cl:50-72: blasType 4 = compressed-wide static geometry, 2 / 3 = Aila-Laine nodes for geometry that is refitted or rebuilt).
// The TLAS is the 8-wide one; a BVH_GPU BLAS adds two lane states with the k_bvh2 node step and leaf loop; its stack entries are node indices.
template <bool ANYHIT, int LDS_N, int REFILL_MIN, int PN, int PT, int PI, bool STATS, int STEAL = 0, bool FUSE = false, bool MIXED = false>
__device__ __forceinline__ void tlas8_body(const float4* __restrict__ tlasNodes, const uint32_t* __restrict__ instRef, const float4* __restrict__ instances,
                                           const BlasDesc* __restrict__ blas, const QueryArgs& q, uint32_t* __restrict__ status) {
    __shared__ uint2 stk[LDS_N][WG];
    LaneStack<uint2, LDS_N, WG> st;
    st.init(&stk[0][threadIdx.x], (uint2*)q.spill + (blockIdx.x * WG + threadIdx.x), (size_t)gridDim.x * WG, q.spillStride);
    RayPool<64> pool;
    const uint64_t nRaysTotal = q.nRaysDev ? *q.nRaysDev : q.nRays;
    pool.init(q.poolParts, q.counterNext);

    bool active = false, found = false, inBlas = false;
    uint64_t ri = 0;
    float3 O = make_float3(0, 0, 0), D = O, rD = O;   // the ray in the CURRENT space
    float4 hit = make_float4(0, 0, 0, 0);
    uint32_t hitInst = 0, rayMask = 0, state = S_NODE, curInst = 0, blasIdx = 0, oct = 0;
    int base = 0;
    GlobalF4 cur(tlasNodes), btris;                   // node stream being walked (TLAS or the instance's BLAS); the BLAS's triangle records
    uint2 ng = make_uint2(0u, 0u), tg = make_uint2(0u, 0u);
    float3 ro = make_float3(0, 0, 0);                 // MIXED, inside a BVH_GPU BLAS: O * rD (SLAB_TEST_TWO_NODES form); its node, pending leaf
    uint32_t node2 = 0, triLeft = 0, triPtr = 0;
    bool blas2 = false;                               // the current BLAS is a BVH_GPU one
    unsigned long long sIter = 0, sAct = 0, sN = 0, sLN = 0, sT = 0, sLT = 0, sI = 0, sLI = 0;   // STATS
    __shared__ SplitLds<STEAL ? WG : 1> split;
    int grp = -1;

    for (;;) {
        const uint32_t nIdle = (uint32_t)__popcll(__ballot(!active));
        if (nIdle >= (uint32_t)REFILL_MIN) {
            if (!pool.dry()) {
                uint64_t nri = 0;
                if (pool.acquire(!active, q.counter, nRaysTotal, nri)) {
                    ri = nri;
                    const RayRec* rp = q.rays + ri;
                    O = xyz(rp->O); D = xyz(rp->D); rD = xyz(rp->rD);
                    rayMask = as_u32(rp->O.w);
                    hit = q.fresh ? make_float4(q.freshTmax, 0.f, 0.f, 0.f) : rp->hit;
                    hitInst = as_u32(rp->rD.w);
                    found = false; inBlas = false; blas2 = false; state = S_NODE; st.sp = 0;
                    oct = cw_oct(D);
                    ng = make_uint2(0u, 0x80000000u); tg = make_uint2(0u, 0u);
                    cur = GlobalF4(tlasNodes);
                    active = true;
                }
            }
            if (__ballot(active) == 0) break;
        }
        const bool tail = STEAL && pool.dry();   // wave-uniform: nothing of the split-ray code costs a vector instruction before the pool is dry
        if (tail && nIdle >= (uint32_t)STEAL) {
            SplitMatch m;
            if (split_match(active && st.sp != 0, !active, m)) {
                uint2 part = make_uint2(0u, 0u);
                int lvl = 0;   // the entry belongs to the donor's BLAS traversal (else to the TLAS level, in world space)
                if (m.gives) {
                    lvl = inBlas && st.sp > base;
                    part = st.pop();
                    if (inBlas && !lvl) base = st.sp;   // the donor's BLAS part of the stack was empty: it now begins one entry lower
                    split_give<ANYHIT>(split, m, grp, found, hit, hitInst);
                }
                const int src = split_take_ray(split, m, O, D, rD, hit, ri, grp);
                part.x = __shfl(part.x, src); part.y = __shfl(part.y, src); lvl = __shfl(lvl, src);
                const int donorInBlas = __shfl((int)inBlas, src);
                rayMask = __shfl(rayMask, src); curInst = __shfl(curInst, src); blasIdx = __shfl(blasIdx, src);
                const int donor2 = MIXED ? __shfl((int)blas2, src) : 0;
                if (m.takes) {
                    found = false; st.sp = 0; base = 0; blas2 = false; triLeft = 0;
                    if (lvl) {   // (back at its empty stack the lane "returns" to a TLAS level with nothing left: done)
                        const BlasDesc bd = blas[blasIdx];
                        inBlas = true; cur = GlobalF4(bd.nodes); btris = GlobalF4(bd.tris);
                        blas2 = MIXED && donor2;
                    } else {
                        inBlas = false; cur = GlobalF4(tlasNodes);
                        if (donorInBlas) { const RayRec* rp = q.rays + ri; O = xyz(rp->O); D = xyz(rp->D); rD = xyz(rp->rD); }   // the donor's registers hold the instance-space ray
                    }
                    oct = cw_oct(D);
                    if (MIXED && blas2) { ro = make_float3(O.x * rD.x, O.y * rD.y, O.z * rD.z); node2 = part.x; ng = make_uint2(0u, 0u); tg = make_uint2(0u, 0u); state = S_NODE2; }
                    else if (part.y > 0x00FFFFFFu) { ng = part; tg = make_uint2(0u, 0u); state = S_NODE; }
                    else { tg = part; ng = make_uint2(0u, 0u); state = S_INST; }   // a parked instance group (TLAS level only)
                    active = true;
                }
            }
        }
        const uint32_t nN = (uint32_t)__popcll(__ballot(active && state == S_NODE)), nT = (uint32_t)__popcll(__ballot(active && state == S_TRI)),
                       nI = (uint32_t)__popcll(__ballot(active && state == S_INST));
        const uint32_t nN2 = MIXED ? (uint32_t)__popcll(__ballot(active && state == S_NODE2)) : 0u, nT2 = MIXED ? (uint32_t)__popcll(__ballot(active && state == S_TRI2)) : 0u;
        uint32_t nMax = nN > nT ? (nN > nI ? nN : nI) : (nT > nI ? nT : nI);
        if (MIXED) { nMax = nN2 > nMax ? nN2 : nMax; nMax = nT2 > nMax ? nT2 : nMax; }
        const bool runN = nN >= (uint32_t)PN || nN == nMax, runT = nT >= (uint32_t)PT || nT == nMax, runI = nI >= (uint32_t)PI || nI == nMax;
        const bool runN2 = MIXED && (nN2 >= (uint32_t)PN || nN2 == nMax), runT2 = MIXED && (nT2 >= (uint32_t)PT || nT2 == nMax);
        if (STATS) { sIter++; sAct += nN + nT + nI; if (runN && nN) { sN++; sLN += nN; } if (runT && nT) { sT++; sLT += nT; } if (runI && nI) { sI++; sLI += nI; } }
        if (!active) continue;
        bool done = false, next = false;   // next: this lane's step is over, decide what it does in the following iteration
        if (tail && grp >= 0) split_poll<ANYHIT>(split, grp, hit, done);   // a split ray: bounded by its group's closest hit

        // FUSE: a lane whose triangle group is finished, or that has just entered an instance, takes its node step in the same pass (the
        // single-level kernel's schedule: one triangle AND one node per pass) instead of waiting for the next one
        bool cont = false;
        bool pop2 = false;   // MIXED: this lane's BVH_GPU step ended with nothing pending
        if (STEAL && ANYHIT && done) {
        } else if (MIXED && state == S_TRI2) { if (runT2) {
            // ---- one triangle of the current leaf of a BVH_GPU BLAS (records {v0|prim, e1, e2}, kernels_query.hip: k_bvh2) -------------------
            const float4 v0 = btris[triPtr], e1 = btris[triPtr + 1], e2 = btris[triPtr + 2];
            triPtr += 3u; triLeft--;
            TriHit h;
            if (tri_test(O, D, xyz(v0), xyz(e1), xyz(e2), hit.x, h) && (ANYHIT || (tail && grp >= 0) || hit_wins(h.t, as_u32(v0.w), curInst, found, hit, hitInst))) {
                const BlasDesc bd = blas[blasIdx];
                if (!bd.opmap || omm_opaque(Omm{bd.opmap, bd.opmapN}, as_u32(v0.w), h.u, h.v)) {
                    found = true; hitInst = curInst;
                    if (ANYHIT) done = true;
                    else hit = make_float4(h.t, h.u, h.v, v0.w);
                    if (tail && grp >= 0) split_publish<ANYHIT, true>(split, grp, hit, hitInst);
                }
            }
            if (!done && triLeft == 0) pop2 = true;
        } } else if (MIXED && state == S_NODE2) { if (runN2) {
            // ---- one node of a BVH_GPU BLAS: SLAB_TEST_TWO_NODES (tiny_bvh.h:3202-3220), as k_bvh2 / k_tlas2 -----------------------------------
            const float4 n0 = cur[node2 * 4], n1 = cur[node2 * 4 + 1], n2 = cur[node2 * 4 + 2], n3 = cur[node2 * 4 + 3];
            const uint32_t cnt = as_u32(n2.w);
            if (cnt) { triLeft = cnt; triPtr = as_u32(n3.w) * 3u; state = S_TRI2; }
            else {
                const float lx1 = __builtin_fmaf(n0.x, rD.x, -ro.x), lx2 = __builtin_fmaf(n1.x, rD.x, -ro.x);
                const float ly1 = __builtin_fmaf(n0.y, rD.y, -ro.y), ly2 = __builtin_fmaf(n1.y, rD.y, -ro.y);
                const float lz1 = __builtin_fmaf(n0.z, rD.z, -ro.z), lz2 = __builtin_fmaf(n1.z, rD.z, -ro.z);
                const float rx1 = __builtin_fmaf(n2.x, rD.x, -ro.x), rx2 = __builtin_fmaf(n3.x, rD.x, -ro.x);
                const float ry1 = __builtin_fmaf(n2.y, rD.y, -ro.y), ry2 = __builtin_fmaf(n3.y, rD.y, -ro.y);
                const float rz1 = __builtin_fmaf(n2.z, rD.z, -ro.z), rz2 = __builtin_fmaf(n3.z, rD.z, -ro.z);
                const float tminL = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(lx1, lx2), __builtin_fminf(ly1, ly2)), __builtin_fminf(lz1, lz2)), 0.0f);
                const float tmaxL = __builtin_fminf(__builtin_fminf(__builtin_fminf(__builtin_fmaxf(lx1, lx2), __builtin_fmaxf(ly1, ly2)), __builtin_fmaxf(lz1, lz2)), cull_bound(hit.x));
                const float tminR = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(rx1, rx2), __builtin_fminf(ry1, ry2)), __builtin_fminf(rz1, rz2)), 0.0f);
                const float tmaxR = __builtin_fminf(__builtin_fminf(__builtin_fminf(__builtin_fmaxf(rx1, rx2), __builtin_fmaxf(ry1, ry2)), __builtin_fmaxf(rz1, rz2)), cull_bound(hit.x));
                const bool hL = tmaxL >= tminL, hR = tmaxR >= tminR;
                uint32_t l = as_u32(n0.w), r = as_u32(n1.w);
                if (hL && hR) {
                    if (tminL > tminR) { const uint32_t t = l; l = r; r = t; }
                    st.push(make_uint2(r, 0u));
                    node2 = l;
                } else if (hL) node2 = l;
                else if (hR) node2 = r;
                else pop2 = true;
            }
        } } else if (state == S_TRI) { if (runT) {
            // ---- one triangle of the BLAS node's triangle group ------------------------------------------------------------
            const uint32_t ti = 31u - (uint32_t)__clz(tg.y);
            tg.y &= ~(1u << ti);
            const uint32_t ta = tg.x + ti * 3u;
            const float4 e2 = btris[ta], e1 = btris[ta + 1], v0 = btris[ta + 2];
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
            next = !done;
            if (FUSE && !done && tg.y == 0 && cw_has_child(ng)) { state = S_NODE; next = false; cont = true; }
        } } else if (state == S_INST) { if (runI) {
            // ---- enter one instance of the TLAS node's instance group (tiny_bvh.h:3326-3333) ----------------------------------
            const uint32_t ti = 31u - (uint32_t)__clz(tg.y);
            tg.y &= ~(1u << ti);
            const uint32_t ii = instRef[tg.x + ti];
            const float4* ip = instances + (size_t)ii * 12;
            const float4 b0 = ip[8], b1 = ip[9];                      // aabbMin|blasIdx, aabbMax|mask
            if (as_u32(b1.w) & rayMask) {
                // park what is left at this TLAS node below the BLAS traversal: its other hit instances, then its interior children
                if (cw_has_child(ng)) st.push(ng);
                if (tg.y != 0) st.push(tg);
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
                const BlasDesc bd = blas[blasIdx];
                cur = GlobalF4(bd.nodes); btris = GlobalF4(bd.tris);
                curInst = ii; base = st.sp; inBlas = true;
                oct = cw_oct(D);
                if (MIXED && bd.layout == (uint32_t)kLayoutBvhGpu) {
                    blas2 = true; ro = make_float3(O.x * rD.x, O.y * rD.y, O.z * rD.z);
                    ng = make_uint2(0u, 0u); tg = make_uint2(0u, 0u); node2 = 0; triLeft = 0;
                    state = S_NODE2;
                } else {
                    ng = make_uint2(0u, 0x80000000u); tg = make_uint2(0u, 0u);
                    state = S_NODE; cont = FUSE;
                }
            } else next = true;
        } } else if (runN) cont = true;
        if (cont && runN && state == S_NODE && !done) {
            // ---- one node of the TLAS or of the instance's BLAS: same format, same code ------------------------------------------
            if (cw_has_child(ng)) {
                const uint32_t ci = cw_next_child(ng, oct);
                if (cw_has_child(ng)) st.push(ng);
                const CwNodeHits r = cw_test_node(cw_load_node(cur, ci), O, rD, cull_bound(hit.x), oct * 0x01010101u);
                ng = make_uint2(r.childBase, (r.hitmask & 0xFF000000u) | r.imask);
                tg = make_uint2(r.triBase, r.hitmask & 0x00FFFFFFu);
            }
            next = true;
        }
        if (MIXED && pop2) {
            if (st.sp == base) { blas2 = false; next = true; }   // the BVH_GPU BLAS is done: back to the TLAS below (ng and tg are empty, so `next` goes to the stack)
            else { node2 = st.pop().x; state = S_NODE2; }
        }
        if (next) {
            // ---- what next: the group in hand, else the stack; a BLAS traversal back at its base returns to the TLAS with the world ray ----
            if (tg.y != 0) state = inBlas ? S_TRI : S_INST;
            else if (cw_has_child(ng)) state = S_NODE;
            else {
                if (inBlas && st.sp == base) {
                    inBlas = false; cur = GlobalF4(tlasNodes);
                    const RayRec* rp = q.rays + ri;
                    O = xyz(rp->O); D = xyz(rp->D); rD = xyz(rp->rD);
                    oct = cw_oct(D);
                }
                if (st.sp == 0) done = true;
                else {
                    const uint2 e = st.pop();
                    if (e.y > 0x00FFFFFFu) { ng = e; tg = make_uint2(0u, 0u); state = S_NODE; }
                    else { tg = e; ng = make_uint2(0u, 0u); state = inBlas ? S_TRI : S_INST; }   // a parked instance group (TLAS level)
                }
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

template <bool ANYHIT, int LDS_N = 12, int REFILL_MIN = 16, int PN = 24, int PT = 8, int PI = 8, bool STATS = false, int STEAL = 0, int WAVES = 6, bool FUSE = false, bool MIXED = false>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_tlas8(const float4* __restrict__ tlasNodes, const uint32_t* __restrict__ instRef,
                                                                                           const float4* __restrict__ instances, const BlasDesc* __restrict__ blas,
                                                                                           QueryArgs q, uint32_t* __restrict__ status) {
    tlas8_body<ANYHIT, LDS_N, REFILL_MIN, PN, PT, PI, STATS, STEAL, FUSE, MIXED>(tlasNodes, instRef, instances, blas, q, status);
}

}  // namespace

void launch_tlas8(bool anyhit, int variant, const float4* tlasNodes, const uint32_t* instRef, const float4* instances, const BlasDesc* blas, const QueryArgs& q,
                  uint32_t* status, uint32_t blocks, hipStream_t s, uint32_t blocks7, bool mixed) {
#define TBVH_T8(...)                                                                                                                                \
    do {                                                                                                                                            \
        if (anyhit) hipLaunchKernelGGL((k_tlas8<true, __VA_ARGS__>), dim3(blocks), dim3(WG), 0, s, tlasNodes, instRef, instances, blas, q, status);  \
        else hipLaunchKernelGGL((k_tlas8<false, __VA_ARGS__>), dim3(blocks), dim3(WG), 0, s, tlasNodes, instRef, instances, blas, q, status);        \
    } while (0)
    (void)variant;
    // batches below 12 M rays, and the wavefront stages (ray count known to the device only), split their last rays over idle lanes (ray_split.h)
    // 7 waves per SIMD on 28 workgroups per CU (kernels_tlas4.hip): +4…5 %.  LDS sets the stack entries kept there: 8 next to the split groups
    // (with 12 only 20 waves per CU fit; 8.3 M camera rays +7 % over 10), 10 without them
    // fused steps (a lane done with its triangles, or entering an instance, takes a node step in the same pass): camera rays +4 %, IsOccluded +5 %
    if (mixed) {   // BVH8_CWBVH and BVH_GPU BLASes under one TLAS: five lane states, so a state's code runs from 8 lanes on (1000 instances, half of each
        // kind: thresholds 32/32/32 2100 / 1080, 24/8/8 2540 / 1320, 16/8/8 2630 / 1440, 8/8/8 2790 / 1490 MRays/s camera / random rays; the three-mode
        // loop of kernels_tlas.hip 2350 / 870); the two more states need the registers of 6 waves per SIMD (7: -4 %)
        if (split_rays_wanted(q)) TBVH_T8(8, 16, 8, 8, 8, false, 16, 6, true, true);
        else TBVH_T8(12, 16, 8, 8, 8, false, 0, 6, true, true);
        return;
    }
    blocks = blocks7;
    if (split_rays_wanted(q)) TBVH_T8(8, 16, 24, 8, 8, false, 16, 7, true);
    else TBVH_T8(10, 16, 24, 8, 8, false, 0, 7, true);
#undef TBVH_T8
}

}  // namespace tbvh
