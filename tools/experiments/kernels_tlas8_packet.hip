// kernels_tlas8_packet.hip — two-level Intersect / IsOccluded for COHERENT batches: one traversal per WAVE through the TLAS and through every instance
// any of its rays enters (round 6).  BVH8_CWBVH BLASes under the 8-wide TLAS of kernels_tlaswide.hip — the reference's instancing configuration
// (tiny_bvh_gpu2.cpp: BVH8_CWBVH BLASes, traverse_tlas.cl:13-107 walks the TLAS and calls traverse_cwbvh per instance).
//
// kernels_tlas8.hip gives every ray its own walk of both levels: the 64 camera rays of a 16 x 4-pixel chunk enter the same two or three instances and
// visit nearly the same nodes in them, each lane decoding the same quantised planes (kernels_cwbvh_packet.hip has the numbers for one level).  Here a wave of 64
// CONSECUTIVE rays keeps ONE traversal state:
//   * TLAS level: the node test of cwbvh_packet.h on the wave's rays in world space; a child is entered when ANY live lane's ray enters its box; the
//     "triangle" bits of a TLAS node name instances (kernels_tlas8.hip);
//   * entering an instance: what is left of the TLAS node (its other hit instances, its interior children) is parked on the wave's stack, every lane
//     takes ITS ray into the instance's space (tinybvh_transform_point / _vector with the reference build's contraction, tiny_bvh.h:3326-3333; D is not
//     re-normalised, so t means the same in both spaces), the wave's octant is taken anew from the transformed rays, and the BLAS is walked like a
//     single-level scene by kernels_cwbvh_packet.hip's loop; back at the stack height of entry the rays return to world space;
//   * a lane whose instance mask does not match, whose ray is already occluded (any-hit), or that holds no ray, tests with tcull = -1: it enters nothing.
// A ray meets a SUPERSET of what it would meet alone, in another order; the tie rule (device_common.h: hit_wins with the instance index as the last
// key) makes the record the same bytes as kernels_tlas8.hip's.  Served: the first kernel of a two-kernel launch on TLASes over BVH8_CWBVH BLASes
// (capi_query.hip) — every wave samples the batch's coherence (cwbvh_probe.h) and leaves at once unless it is coherent; kernels_tlas8.hip behind it
// takes whatever the pool still holds.
#include "device_common.h"
#include "ray_pool.h"
#include "kernels.h"
#include "cwbvh_node.h"
#include "cwbvh_probe.h"
#include "cwbvh_packet.h"

namespace tbvh {

namespace {

constexpr int WG = 64;

__device__ __forceinline__ float safercp_pk(float x) {
    if (x > 1e-12f || x < -1e-12f) return 1.0f / x;
    return x >= 0 ? kFar : -kFar;
}
__device__ __forceinline__ const float4* uniform_ptr(const float4* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = sgpr((uint32_t)v), hi = sgpr((uint32_t)(v >> 32));
    return (const float4*)(((uint64_t)hi << 32) | lo);
}

template <bool ANYHIT>
__global__ __launch_bounds__(WG, 6) void k_tlas8_packet(const float4* __restrict__ tlasNodes, const uint32_t* __restrict__ instRef, const float4* __restrict__ instances,
                                                       const BlasDesc* __restrict__ blas, QueryArgs q, uint32_t* __restrict__ status) {
    __shared__ float planes[8][8];
    const uint64_t nRaysTotal = q.nRaysDev ? *q.nRaysDev : q.nRays;
    RayPool<64> pool;
    pool.init(q.poolParts, q.counterNext);
    if (q.probe) {
        uint32_t agree, pairs;
        coherence_sample(q.rays, nRaysTotal, q.fresh != 0u, q.freshTmax, agree, pairs);
        if (blockIdx.x == 0 && threadIdx.x == 0) { q.probe[0] = agree; q.probe[1] = pairs; }
        if (!((pairs != 0 && agree * 10u >= pairs * 6u) || (q.flags & 16u) != 0)) return;
    }
    uint2* const spill = (uint2*)q.spill + (size_t)blockIdx.x * WG;
    const size_t spillRow = (size_t)gridDim.x * WG;
    const uint32_t spillCap = q.spillStride * (uint32_t)WG;
    bool overflow = false;
    const uint32_t lane = threadIdx.x;

    for (;;) {
        uint64_t ri = 0;
        const bool have = pool.acquire(true, q.counter, nRaysTotal, ri);
        if (wave_ballot(have) == 0) break;
        float3 Ow = make_float3(0, 0, 0), Dw = make_float3(0, 0, 1), rDw = make_float3(1e30f, 1e30f, 1);
        float4 hit = make_float4(0, 0, 0, 0);
        uint32_t hitInst = 0, rayMask = 0;
        if (have) {
            const RayRec* rp = q.rays + ri;
            Ow = xyz(rp->O); Dw = xyz(rp->D); rDw = xyz(rp->rD);
            rayMask = as_u32(rp->O.w);
            hit = q.fresh ? make_float4(q.freshTmax, 0.f, 0.f, 0.f) : rp->hit;
            hitInst = as_u32(rp->rD.w);
        }
        bool found = false;
        // ---- the space the wave is in: world, or the current instance's ------------------------------------------------------------
        float3 O = Ow, D = Dw, rD = rDw;
        bool maskOk = true;            // this lane's ray may enter the current instance (BLASInstance::mask & ray mask, tiny_bvh.h:3326)
        uint32_t oct0 = 0;
        bool mixed = false;
        float* myPlane = &planes[0][0];
        auto enter_space = [&]() {     // the wave's octant and where lane L's plane byte goes, from the rays as they are in this space (kernels_cwbvh_packet.hip)
            const uint32_t oct = 7u - ((rD.x < 0 ? 4u : 0u) | (rD.y < 0 ? 2u : 0u) | (rD.z < 0 ? 1u : 0u));
            const unsigned long long haveMask = wave_ballot(have);
            const uint32_t first = (uint32_t)__builtin_ctzll(haveMask);
            oct0 = (uint32_t)__builtin_amdgcn_readlane((int)oct, (int)first);
            mixed = wave_ballot(have && oct != oct0) != 0;
            const bool negX0 = ((7u - oct0) & 4u) != 0, negY0 = ((7u - oct0) & 2u) != 0, negZ0 = ((7u - oct0) & 1u) != 0;
            const uint32_t p = lane >> 3, axis = p % 3u, isHi = p / 3u;
            const bool negA = axis == 0 ? negX0 : axis == 1 ? negY0 : negZ0;
            const uint32_t dstPlane = axis + 3u * (isHi ^ (negA ? 1u : 0u));
            myPlane = &planes[lane & 7u][lane < 48u ? dstPlane : 6u + ((lane >> 3) & 1u)];
        };
        enter_space();

        // ---- wave-uniform traversal state ---------------------------------------------------------------------------------------------
        uint32_t sp = 0, base = 0;     // stack height; height at which the current instance was entered
        bool inBlas = false;
        uint32_t curInst = 0;
        const float4* nodes = tlasNodes;
        const float4* tris = nullptr;
        const uint32_t* opmap = nullptr; uint32_t opmapN = 0;
        uint32_t ngx = 0, ngy = 0x80000000u;   // node group: the TLAS root
        uint32_t tgx = 0, tgy = 0;             // TLAS level: instance group in hand
        uint32_t stkx = 0, stky = 0;           // the stack in the lanes of two registers, spill area behind it
        auto push = [&](uint32_t x, uint32_t y) {
            if (sp < 64u) { const bool mine = lane == sp; stkx = mine ? x : stkx; stky = mine ? y : stky; sp++; }
            else {
                const uint32_t j_ = sp - 64u;
                if (j_ < spillCap) { if (lane == 0) spill[(j_ & 63u) + (size_t)(j_ >> 6) * spillRow] = make_uint2(x, y); sp++; }
                else overflow = true;
            }
        };
        auto pop = [&](uint32_t& x, uint32_t& y) {
            sp--;
            if (sp < 64u) { x = (uint32_t)__builtin_amdgcn_readlane((int)stkx, (int)sp); y = (uint32_t)__builtin_amdgcn_readlane((int)stky, (int)sp); }
            else { const uint32_t j_ = sp - 64u; const uint2 e = spill[(j_ & 63u) + (size_t)(j_ >> 6) * spillRow]; x = sgpr(e.x); y = sgpr(e.y); }
        };

        for (;;) {
            const bool live = have && maskOk && !(ANYHIT && found);
            if (ANYHIT && wave_ballot(have && !found) == 0ull) break;        // every ray of the wave is occluded
            // ---- TLAS level: instances of the group in hand, front-most first ------------------------------------------------------------
            if (!inBlas && tgy != 0u) {
                const uint32_t ti = 31u - (uint32_t)__builtin_clz(tgy);
                tgy &= ~(1u << ti);
                const uint32_t ii = sgpr(instRef[tgx + ti]);
                const float4* ip = instances + (size_t)ii * 12;
                const float4 b0 = ip[8], b1 = ip[9];                          // aabbMin | blasIdx, aabbMax | mask (wave-uniform loads)
                const bool ok = (as_u32(b1.w) & rayMask) != 0u;
                if (wave_ballot(have && ok && !(ANYHIT && found)) == 0ull) continue;
                // park what is left at this TLAS node below the BLAS traversal: its interior children, then its other hit instances (on top)
                if (ngy > 0x00FFFFFFu) push(ngx, ngy);
                if (tgy != 0u) push(tgx, tgy);
                const float4 r0 = ip[4], r1 = ip[5], r2 = ip[6], r3 = ip[7];   // invTransform rows
                const float px = __builtin_fmaf(r0.z, Ow.z, __builtin_fmaf(r0.x, Ow.x, r0.y * Ow.y)) + r0.w;
                const float py = __builtin_fmaf(r1.z, Ow.z, __builtin_fmaf(r1.x, Ow.x, r1.y * Ow.y)) + r1.w;
                const float pz = __builtin_fmaf(r2.z, Ow.z, __builtin_fmaf(r2.x, Ow.x, r2.y * Ow.y)) + r2.w;
                const float w = __builtin_fmaf(r3.z, Ow.z, __builtin_fmaf(r3.x, Ow.x, r3.y * Ow.y)) + r3.w;
                D = make_float3(__builtin_fmaf(r0.z, Dw.z, __builtin_fmaf(r0.x, Dw.x, r0.y * Dw.y)), __builtin_fmaf(r1.z, Dw.z, __builtin_fmaf(r1.x, Dw.x, r1.y * Dw.y)),
                                __builtin_fmaf(r2.z, Dw.z, __builtin_fmaf(r2.x, Dw.x, r2.y * Dw.y)));
                if (w == 1) O = make_float3(px, py, pz);
                else { const float iw = 1.f / w; O = make_float3(px * iw, py * iw, pz * iw); }
                rD = make_float3(safercp_pk(D.x), safercp_pk(D.y), safercp_pk(D.z));
                maskOk = ok;
                const uint32_t bi = sgpr(as_u32(b0.w));
                const BlasDesc bd = blas[bi];
                nodes = uniform_ptr(bd.nodes); tris = uniform_ptr(bd.tris);
                opmap = (const uint32_t*)uniform_ptr((const float4*)bd.opmap); opmapN = sgpr(bd.opmapN);
                curInst = ii; base = sp; inBlas = true;
                ngx = 0; ngy = 0x80000000u; tgx = 0; tgy = 0;
                enter_space();
                continue;
            }
            // ---- the next node: off the group in hand, else off the stack; a BLAS walk back at its base returns to the TLAS ------------
            if (!(ngy > 0x00FFFFFFu)) {
                if (inBlas && sp == base) {
                    inBlas = false; nodes = tlasNodes; maskOk = true;
                    O = Ow; D = Dw; rD = rDw;
                    enter_space();
                }
                if (sp == 0) break;
                uint32_t ex, ey;
                pop(ex, ey);
                if (!(ey > 0x00FFFFFFu)) { tgx = ex; tgy = ey; ngx = 0; ngy = 0; continue; }   // a parked instance group (TLAS level only: BLAS walks push node groups only)
                ngx = ex; ngy = ey;
            }
            const uint32_t imaskWord = ngy;
            const uint32_t bit = 31u - (uint32_t)__builtin_clz(ngy);
            ngy &= ~(1u << bit);
            if (ngy > 0x00FFFFFFu) push(ngx, ngy);                               // children of this group still pending: keep it
            const uint32_t slot = (bit - 24u) ^ oct0;
            const uint32_t ci = sgpr(ngx + (uint32_t)__popc(imaskWord & ~(0xFFFFFFFFu << slot)));
            // ---- the node: fetched once (n0, n1 wave-uniform; lane L: plane byte L), decoded once for the wave --------------------------
            const float4* np = nodes + (size_t)ci * 5u;
            const float4 n0 = np[0], n1 = np[1];
            const uint32_t qb = ((const uint8_t*)(np + 2))[lane < 48u ? lane : 47u];
            __builtin_amdgcn_wave_barrier();
            *myPlane = (float)qb;
            __builtin_amdgcn_wave_barrier();
            const uint32_t ew = sgpr(as_u32(n0.w));
            const float ax = ldexpf(rD.x, (int)(int8_t)(ew)), ay = ldexpf(rD.y, (int)(int8_t)(ew >> 8)), az = ldexpf(rD.z, (int)(int8_t)(ew >> 16));
            const float ox = (n0.x - O.x) * rD.x, oy = (n0.y - O.y) * rD.y, oz = (n0.z - O.z) * rD.z;
            const float tcull = live ? cull_bound(hit.x) : -1.0f;
            const uint32_t m0 = sgpr(as_u32(n1.z)), m1 = sgpr(as_u32(n1.w));
            const uint32_t hitmask = mixed ? pk_test_children<true>(planes, ax, ay, az, ox, oy, oz, tcull, m0, m1, oct0)
                                           : pk_test_children<false>(planes, ax, ay, az, ox, oy, oz, tcull, m0, m1, oct0);
            ngx = sgpr(as_u32(n1.x));
            ngy = (hitmask & 0xFF000000u) | (ew >> 24);
            tgx = sgpr(as_u32(n1.y));
            tgy = hitmask & 0x00FFFFFFu;
            if (!inBlas) continue;                                               // TLAS node: its instance bits are handled at the top of the loop
            // ---- BLAS node: the triangles of the leaves any ray entered, tested by all live lanes ---------------------------------------
            while (tgy != 0u) {
                const uint32_t ti = 31u - (uint32_t)__builtin_clz(tgy);
                tgy &= ~(1u << ti);
                const float4* tp = tris + ((size_t)tgx + ti * 3u);
                const float4 e2 = tp[0], e1 = tp[1], v0 = tp[2];
                TriHit h;
                if (live && !(ANYHIT && found) && tri_test(O, D, xyz(v0), xyz(e1), xyz(e2), hit.x, h) &&
                    (ANYHIT || hit_wins(h.t, as_u32(v0.w), curInst, found, hit, hitInst)) &&
                    (!opmap || omm_opaque(Omm{opmap, opmapN}, as_u32(v0.w), h.u, h.v))) {
                    found = true; hitInst = curInst;
                    if (!ANYHIT) hit = make_float4(h.t, h.u, h.v, v0.w);
                }
            }
        }
        // ---- results ----------------------------------------------------------------------------------------------------------
        if (have) {
            RayRec* rp = q.rays + ri;
            if (ANYHIT) q.occluded[ri] = found ? 1 : 0;
            else if (found) { rp->hit = hit; ((uint32_t*)rp)[11] = hitInst; }   // byte 44 = hit.inst
            else if (q.fresh) rp->hit = hit;
        }
    }
    if (overflow) atomicOr(status, 1u);
}

}  // namespace

void launch_tlas8_packet(bool anyhit, const float4* tlasNodes, const uint32_t* instRef, const float4* instances, const BlasDesc* blas, const QueryArgs& q,
                         uint32_t* status, uint32_t blocks, hipStream_t s) {
    if (anyhit) hipLaunchKernelGGL((k_tlas8_packet<true>), dim3(blocks), dim3(WG), 0, s, tlasNodes, instRef, instances, blas, q, status);
    else hipLaunchKernelGGL((k_tlas8_packet<false>), dim3(blocks), dim3(WG), 0, s, tlasNodes, instRef, instances, blas, q, status);
}

}  // namespace tbvh
