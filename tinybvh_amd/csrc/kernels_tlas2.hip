// kernels_tlas2.hip — two-level Intersect / IsOccluded over BVH_GPU (Aila-Laine 2-wide) BLASes.
//
// The TLAS the caller uploads IS in the BVH_GPU node format (tiny_bvh.h:4575-4581; the device-built one of kernels_tlasbuild.hip
// too), so for BVH_GPU BLASes — the reference's choice for geometry that is refitted or rebuilt per frame (traverse_tlas.cl:66-72) —
// a TLAS node step and a BLAS node step are the same code on different base pointers without any conversion: a leaf of the TLAS
// names instances (tlasIdx[firstTri .. firstTri + triCount)), a leaf of a BLAS names triangles.  Same structure as
// kernels_tlas4.hip / kernels_tlas8.hip: one flat loop, lanes in one of three states (node, triangle, instance entry), a state's
// code runs when enough lanes are in it, per-lane ray replacement, split rays at the end of the launch (ray_split.h).
// The mixed-layout case (BLASes of different layouts under one TLAS) stays with the three-mode loop of kernels_tlas.hip.
//
// Arithmetic: the node test is SLAB_TEST_TWO_NODES (tiny_bvh.h:3202-3220) at both levels and the instance entry is
// tinybvh_transform_point / _vector with the reference build's contraction, exactly as in kernels_tlas.hip (pinned there
// against the real IntersectTLAS); per-ray order is the nested reference loop's (traverse_tlas.cl:13-107): nearer child
// first, the instances of a TLAS leaf in index order.
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

enum : uint32_t { S_NODE = 0, S_TRI = 1, S_INST = 2 };

// PN / PT / PI: a state's code runs in a pass if at least that many lanes are in the state, or it holds the most lanes.
// STEAL > 0 (idle lanes needed): once the ray pool is dry, idle lanes take the top stack entry — a BLAS or a TLAS subtree — off a lane
// that is still traversing.
template <bool ANYHIT, int LDS_N, int REFILL_MIN, int PN, int PT, int PI, int STEAL, int WAVES = 6, int NODE_REPS = 1, bool FUSE = false>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_tlas2(const float4* __restrict__ tlasNodes, const uint32_t* __restrict__ tlasIdx,
                                                                                           const float4* __restrict__ instances, const BlasDesc* __restrict__ blas,
                                                                                           QueryArgs q, uint32_t* __restrict__ status) {
    __shared__ uint32_t stk[LDS_N][WG];
    LaneStack<uint32_t, LDS_N, WG> st;
    st.init(&stk[0][threadIdx.x], q.spill + (blockIdx.x * WG + threadIdx.x), (size_t)gridDim.x * WG, q.spillStride);
    RayPool<64> pool;
    const uint64_t nRaysTotal = q.nRaysDev ? *q.nRaysDev : q.nRays;
    pool.init(q.poolParts, q.counterNext);
    __shared__ SplitLds<STEAL ? WG : 1> split;
    int grp = -1;

    bool active = false, found = false, inBlas = false;
    uint64_t ri = 0;
    float3 O = make_float3(0, 0, 0), D = O, rD = O, ro = O;   // the ray in the CURRENT space (world, or the instance's); ro = O * rD
    float4 hit = make_float4(0, 0, 0, 0);
    uint32_t hitInst = 0, rayMask = 0, state = S_NODE, node = 0, triLeft = 0, triPtr = 0, instNext = 0, instEnd = 0, curInst = 0, blasIdx = 0;
    int base = 0;                  // stack height at which the current BLAS traversal began
    GlobalF4 cur(tlasNodes), btris;   // the nodes this lane is walking (TLAS or the instance's BLAS); the BLAS's triangle records {v0|prim, e1, e2}

    for (;;) {
        const uint32_t nIdle = (uint32_t)__popcll(__ballot(!active));
        if (nIdle >= (uint32_t)REFILL_MIN) {
            if (!pool.dry()) {
                uint64_t nri = 0;
                if (pool.acquire(!active, q.counter, nRaysTotal, nri)) {
                    ri = nri;
                    const RayRec* rp = q.rays + ri;
                    O = xyz(rp->O); D = xyz(rp->D); rD = xyz(rp->rD);
                    ro = make_float3(O.x * rD.x, O.y * rD.y, O.z * rD.z);
                    rayMask = as_u32(rp->O.w);
                    hit = q.fresh ? make_float4(q.freshTmax, 0.f, 0.f, 0.f) : rp->hit;
                    hitInst = as_u32(rp->rD.w);
                    found = false; inBlas = false; state = S_NODE; node = 0; triLeft = 0; instNext = instEnd = 0; st.sp = 0;
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
                    found = false; triLeft = 0; instNext = instEnd = 0; st.sp = 0; base = 0;
                    if (lvl) {   // (back at its empty stack the lane "returns" to a TLAS level with nothing left: done)
                        const BlasDesc bd = blas[blasIdx];
                        inBlas = true; cur = GlobalF4(bd.nodes); btris = GlobalF4(bd.tris);
                    } else {
                        inBlas = false; cur = GlobalF4(tlasNodes);
                        if (donorInBlas) { const RayRec* rp = q.rays + ri; O = xyz(rp->O); D = xyz(rp->D); rD = xyz(rp->rD); }   // the donor's registers hold the instance-space ray
                    }
                    ro = make_float3(O.x * rD.x, O.y * rD.y, O.z * rD.z);
                    state = S_NODE; node = part;
                    active = true;
                }
            }
        }
        const uint32_t nN = (uint32_t)__popcll(__ballot(active && state == S_NODE)), nT = (uint32_t)__popcll(__ballot(active && state == S_TRI)),
                       nI = (uint32_t)__popcll(__ballot(active && state == S_INST));
        const uint32_t nMax = nN > nT ? (nN > nI ? nN : nI) : (nT > nI ? nT : nI);
        const bool runN = nN >= (uint32_t)PN || nN == nMax, runT = nT >= (uint32_t)PT || nT == nMax, runI = nI >= (uint32_t)PI || nI == nMax;
        if (!active) continue;
        bool done = false, advance = false;   // advance: nothing pending here, take what comes next at this level
        if (tail && grp >= 0) split_poll<ANYHIT>(split, grp, hit, done);   // a split ray: bounded by its group's closest hit

        // FUSE: a lane whose leaf is finished with more of the BLAS on its stack, or that has just entered an instance, takes its node step(s) in the same pass
        bool cont = false;
        if (STEAL && ANYHIT && done) {
        } else if (state == S_TRI) { if (runT) {
            // ---- one triangle of the current BLAS leaf -----------------------------------------------------------------------
            const float4 v0 = btris[triPtr], e1 = btris[triPtr + 1], e2 = btris[triPtr + 2];
            triPtr += 3u; triLeft--;
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
            if (!done && triLeft == 0) {
                if (FUSE && st.sp > base) { node = st.pop(); state = S_NODE; cont = true; }
                else advance = true;
            }
        } } else if (state == S_INST) { if (runI) {
            // ---- the next instance of the current TLAS leaf (tiny_bvh.h:3326-3333) -------------------------------------------------
            if (instNext == instEnd) advance = true;
            else {
                const uint32_t ii = tlasIdx[instNext++];
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
                    ro = make_float3(O.x * rD.x, O.y * rD.y, O.z * rD.z);
                    blasIdx = as_u32(b0.w);
                    const BlasDesc bd = blas[blasIdx];
                    cur = GlobalF4(bd.nodes); btris = GlobalF4(bd.tris);
                    curInst = ii; base = st.sp; inBlas = true;
                    state = S_NODE; node = 0; cont = FUSE;
                }
            }
        } } else if (runN) cont = true;
        if (cont && runN && state == S_NODE && !done) {
            // ---- NODE_REPS nodes of the TLAS or of the instance's BLAS (the per-pass bookkeeping is a sizeable part of a 2-wide step, as in
            // k_bvh2): same format, same code at both levels; a lane stops early at a leaf or when nothing was hit ----------------------------
#pragma unroll
            for (int rep = 0; rep < NODE_REPS; rep++) if (state == S_NODE && !advance) {
            const float4 n0 = cur[node * 4], n1 = cur[node * 4 + 1], n2 = cur[node * 4 + 2], n3 = cur[node * 4 + 3];
            const uint32_t cnt = as_u32(n2.w);
            if (cnt) {
                if (inBlas) { triLeft = cnt; triPtr = as_u32(n3.w) * 3u; state = S_TRI; }
                else { instNext = as_u32(n3.w); instEnd = instNext + cnt; state = S_INST; }
            } else {
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
                    st.push(r);
                    node = l;
                } else if (hL) node = l;
                else if (hR) node = r;
                else advance = true;
            }
            }
        }
        if (advance && !done) {
            // ---- a BLAS traversal back at its base returns to the TLAS leaf it came from, with the world ray; then the stack --------------
            bool popNext = true;
            if (inBlas && st.sp == base) {
                inBlas = false; cur = GlobalF4(tlasNodes);
                const RayRec* rp = q.rays + ri;
                O = xyz(rp->O); D = xyz(rp->D); rD = xyz(rp->rD);
                ro = make_float3(O.x * rD.x, O.y * rD.y, O.z * rD.z);
                if (instNext != instEnd) { state = S_INST; popNext = false; }
            }
            if (popNext) {
                if (st.sp == 0) done = true;
                else { node = st.pop(); state = S_NODE; }
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
}

}  // namespace

void launch_tlas2(bool anyhit, int variant, const float4* tlasNodes, const uint32_t* tlasIdx, const float4* instances, const BlasDesc* blas, const QueryArgs& q,
                  uint32_t* status, uint32_t blocks, hipStream_t s, uint32_t blocks7) {
#define TBVH_T2(...)                                                                                                                                \
    do {                                                                                                                                            \
        if (anyhit) hipLaunchKernelGGL((k_tlas2<true, __VA_ARGS__>), dim3(blocks), dim3(WG), 0, s, tlasNodes, tlasIdx, instances, blas, q, status);  \
        else hipLaunchKernelGGL((k_tlas2<false, __VA_ARGS__>), dim3(blocks), dim3(WG), 0, s, tlasNodes, tlasIdx, instances, blas, q, status);        \
    } while (0)
    (void)variant;
    // batches below 12 M rays, and the wavefront stages (ray count known to the device only), split their last rays over idle lanes (ray_split.h)
    // without the split code the kernel fits the register budget of 7 waves per SIMD (28 workgroups per CU: +4…6 %); with it five dwords spill and it loses
    // three node visits per pass: camera rays +11 %, random rays +11 %, IsOccluded +16 % over one (8.3 M / 4.2 M rays); 33 M rays +5…8 %
    // fused steps (a lane done with a leaf, or entering an instance, goes on to its node visits in the same pass): camera rays +9 %, random rays +5 %
    if (split_rays_wanted(q)) TBVH_T2(16, 16, 24, 8, 8, 16, 6, 3, true);
    else { blocks = blocks7; TBVH_T2(16, 16, 24, 8, 8, 0, 7, 3, true); }
#undef TBVH_T2
}

}  // namespace tbvh
