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
// ONE flat loop with per-lane ray replacement (k_tlas_flat): every lane is in TLAS / INSTANCE / BLAS mode and does one step of it per
// iteration; a mode's code runs when enough lanes are in it (phase gating).  The unified two-level kernels of kernels_tlas4 / 8 / 2.hip
// serve the single-layout TLASes; this loop keeps the mixes that include BVH4_GPU BLASes (launch_tlas).
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

// ---------------------------------------------------------------------------------------------------------------
// The same query as ONE flat loop with per-lane ray replacement (the structure of kernels_cwbvh.hip / kernels_query.hip):
// every lane is in one of three modes and does one step of it per iteration —
//   TLAS      one 2-wide node of the top-level tree (a leaf switches to INSTANCE),
//   INSTANCE  take the next instance of the current TLAS leaf: mask test, ray into instance space (-> BLAS), or, when
//             the leaf is used up, pop the TLAS stack (-> TLAS) or finish the ray,
//   BLAS      one triangle test or one node visit of the instance's BVH; back at the stack base the world ray is read
//             again from the record (-> INSTANCE),
// so a lane never waits for the longest instance list, the deepest BLAS traversal or the slowest ray of its wave, which
// is what nested loops cost (three levels of "everybody waits for the slowest": round 1's first kernel).  Idle lanes take new
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
                    if (tri_test(O, D, xyz(v0), xyz(e1), xyz(e2), hit.x, h) && (ANYHIT || hit_wins(h.t, as_u32(v0.w), curInst, found, hit, hitInst))) {
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
                        const CwNodeHits nh = cw_test_node(cw_load_node(bnodes, cbase + __popc(imask & ~(0xFFFFFFFFu << slot))), O, rD, cull_bound(hit.x), oct * 0x01010101u);
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
                    if (tri_test(O, D, xyz(v0), xyz(e1), xyz(e2), hit.x, h) && (ANYHIT || hit_wins(h.t, as_u32(v0.w), curInst, found, hit, hitInst))) {
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
                        const float tmax = __builtin_fminf(fmin3(x2, y2, z2), cull_bound(hit.x));
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
                    if (tri_test(O, D, xyz(v0), xyz(e1), xyz(e2), hit.x, h) && (ANYHIT || hit_wins(h.t, as_u32(v0.w), curInst, found, hit, hitInst))) {
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
                        const float tmaxL = __builtin_fminf(fmin3(__builtin_fmaxf(lx1, lx2), __builtin_fmaxf(ly1, ly2), __builtin_fmaxf(lz1, lz2)), cull_bound(hit.x));
                        const float tminR = __builtin_fmaxf(fmax3(__builtin_fminf(rx1, rx2), __builtin_fminf(ry1, ry2), __builtin_fminf(rz1, rz2)), 0.0f);
                        const float tmaxR = __builtin_fminf(fmin3(__builtin_fmaxf(rx1, rx2), __builtin_fmaxf(ry1, ry2), __builtin_fmaxf(rz1, rz2)), cull_bound(hit.x));
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
                    // tinybvh_transform_point / _vector with the reference build's contraction (oracle/tbvh_oracle.c: orc_xform_point / orc_xform_vec)
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
                const float tmaxL = __builtin_fminf(fmin3(__builtin_fmaxf(lx1, lx2), __builtin_fmaxf(ly1, ly2), __builtin_fmaxf(lz1, lz2)), cull_bound(hit.x));
                const float tminR = __builtin_fmaxf(fmax3(__builtin_fminf(rx1, rx2), __builtin_fminf(ry1, ry2), __builtin_fminf(rz1, rz2)), 0.0f);
                const float tmaxR = __builtin_fminf(fmin3(__builtin_fmaxf(rx1, rx2), __builtin_fmaxf(ry1, ry2), __builtin_fmaxf(rz1, rz2)), cull_bound(hit.x));
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

// kernel prologue: LDS stack top, spill area, ray pool
#define TBVH_TLAS_PROLOGUE                                                                                                          \
    __shared__ uint2 stk[LDS_N][WG];                                                                                                \
    StackT<LDS_N> st;                                                                                                               \
    st.init(&stk[0][threadIdx.x], (uint2*)q.spill + (blockIdx.x * WG + threadIdx.x), (size_t)gridDim.x * WG, q.spillStride);       \
    RayPool<64> pool;                                                                                                               \
    const uint64_t nRaysTotal = q.nRaysDev ? *q.nRaysDev : q.nRays; /* batch size may live on the device (wavefront queues) */      \
    pool.init(q.poolParts, q.counterNext);

// register budgets: 6 waves per SIMD for one BLAS layout, 5 when the layout is picked per instance (three BLAS steps inlined)
template <bool ANYHIT, int BLAS_LAYOUT, int LDS_N = 12, int REFILL_MIN = 16, int PHASE_MIN = 16>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_tlas_flat_w5(const float4* __restrict__ tlasNodes, const uint32_t* __restrict__ tlasIdx,
                                                  const float4* __restrict__ instances, const BlasDesc* __restrict__ blas, QueryArgs q,
                                                  uint32_t* __restrict__ status) {
    TBVH_TLAS_PROLOGUE
    tlas_flat_body<ANYHIT, BLAS_LAYOUT, LDS_N, REFILL_MIN, PHASE_MIN, false>(tlasNodes, tlasIdx, instances, blas, q, st, pool, nRaysTotal);
    if (st.overflow) atomicOr(status, 1u);
}
template <bool ANYHIT, int BLAS_LAYOUT, int LDS_N = 12, int REFILL_MIN = 16, int PHASE_MIN = 16>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_tlas_flat_w6(const float4* __restrict__ tlasNodes, const uint32_t* __restrict__ tlasIdx,
                                                  const float4* __restrict__ instances, const BlasDesc* __restrict__ blas, QueryArgs q,
                                                  uint32_t* __restrict__ status) {
    TBVH_TLAS_PROLOGUE
    tlas_flat_body<ANYHIT, BLAS_LAYOUT, LDS_N, REFILL_MIN, PHASE_MIN, false>(tlasNodes, tlasIdx, instances, blas, q, st, pool, nRaysTotal);
    if (st.overflow) atomicOr(status, 1u);
}

}  // namespace

// The flat loop serves what the unified two-level kernels (kernels_tlas4 / 8 / 2.hip) do not: TLASes that mix BVH4_GPU BLASes with other
// layouts (blasLayout == 0: the layout is picked per instance, BlasDesc::layout), and single-layout TLASes whose wide TLAS could not be
// built (beyond the index range of the wide node formats).  Round 1's nested and adaptive kernels measured against it are in the history
// (profiles/r01d_tlas_probe.txt, r02_tlas_probe_before.txt).
void launch_tlas(bool anyhit, int blasLayout, const float4* tlasNodes, const uint32_t* tlasIdx, const float4* instances,
                 const BlasDesc* blas, const QueryArgs& q, uint32_t* status, uint32_t blocks, hipStream_t s) {
#define TBVH_LT(K, LAYOUT)                                                                                                               \
    do {                                                                                                                                 \
        if (anyhit) hipLaunchKernelGGL((K<true, LAYOUT, 12, 16, 32>), dim3(blocks), dim3(WG), 0, s, tlasNodes, tlasIdx, instances, blas, q, status); \
        else hipLaunchKernelGGL((K<false, LAYOUT, 12, 16, 32>), dim3(blocks), dim3(WG), 0, s, tlasNodes, tlasIdx, instances, blas, q, status);       \
    } while (0)
    if (blasLayout == kLayoutBvhGpu) TBVH_LT(k_tlas_flat_w6, kLayoutBvhGpu);
    else if (blasLayout == kLayoutCwbvh) TBVH_LT(k_tlas_flat_w6, kLayoutCwbvh);
    else if (blasLayout == kLayoutBvh4Gpu) TBVH_LT(k_tlas_flat_w6, kLayoutBvh4Gpu);
    else TBVH_LT(k_tlas_flat_w5, 0);
#undef TBVH_LT
}

}  // namespace tbvh
