// kernels_cwbvh_packet.hip — BVH8_CWBVH Intersect / IsOccluded for COHERENT batches: one traversal per WAVE (round 5).
//
// The per-lane kernels (kernels_cwbvh.hip) give every ray its own traversal: 64 rays of a camera tile walk almost the same nodes, each lane decodes
// the same 48 quantised planes (48 v_cvt_f32_ubyte of the node test's ~210 VALU instructions) and, one triangle test or node visit out of step
// with its neighbours after a few passes, fetches them on its own (only 5 % of the node phases of a 16.7 M-ray camera batch find every lane on one
// node: DESIGN.md par. 5 "Round 2").  Here a wave of 64 CONSECUTIVE rays walks the tree ONCE:
//   * one wave-uniform traversal state (node group, triangle group; the stack in the lanes of two vector registers, the global spill area behind it);
//   * a node is fetched once per wave (n0, n1 through the scalar cache; its 48 plane bytes one per lane), decoded once per wave — lane L converts
//     byte L — and handed to all lanes through LDS in near / far order of the wave's octant; each lane then spends 6 FMAs + 4 min / max + 1 compare per
//     child on ITS ray, and the wave descends into a child when ANY lane's ray enters its box (culled per lane against that lane's own closest hit,
//     with the cull slack of device_common.h);
//   * the triangles of a leaf any lane entered are tested by all lanes, the record fetched once through the scalar cache.
// A ray therefore meets a SUPERSET of the nodes and triangles it would meet alone; with the library's tie rule (hit_wins: the result does not depend
// on the order or the number of candidates tested) its record is the same bytes.  Work per wave: the UNION of its rays' visits — 38 node visits and
// 12 triangle tests per 64 camera rays of the bench scene where one ray alone makes 26 and 4 (CPU count, tools/packet_union.py) — at about half the
// instructions per visit.  Rays of mixed octants in one wave (1 chunk in 80 of a camera batch) take the slab test through min / max instead of the
// pre-ordered planes.  Served: the coherent flavor of a probed two-kernel launch when the scene's tuner (capi_query.hip: CohTuner) finds it fastest.
// Format / reference: the same blobs as kernels_cwbvh.hip (tiny_bvh.h:5884-6018); semantics of BVH::Intersect / IsOccluded (tiny_bvh.h:3222-3453).
#include "device_common.h"
#include "ray_pool.h"
#include "kernels.h"
#include "cwbvh_node.h"
#include "cwbvh_probe.h"
#include "cwbvh_packet.h"

namespace tbvh {

namespace {

constexpr int WG = 64;

template <bool ANYHIT, bool HAS_OMM>
__global__ __launch_bounds__(WG, 8) void k_cwbvh_packet(const float4* __restrict__ nodes, const float4* __restrict__ tris, QueryArgs q, uint32_t* __restrict__ status) {
    __shared__ float planes[8][8];      // the current node's child boxes as floats: [child][near x, near y, near z, far x, far y, far z, -, -] in units of 2^e from the node's origin
    const uint64_t nRaysTotal = q.nRaysDev ? *q.nRaysDev : q.nRays;
    RayPool<64> pool;
    pool.init(q.poolParts, q.counterNext);
    if (q.probe) {   // the coherent flavor of a two-kernel launch: same protocol as k_cwbvh<PROBED == 3> (the incoherent flavor behind it takes what this one leaves in the pool)
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
        // ---- 64 consecutive rays ------------------------------------------------------------------------------------------
        uint64_t ri = 0;
        const bool have = pool.acquire(true, q.counter, nRaysTotal, ri);
        if (wave_ballot(have) == 0) break;
        float3 O = make_float3(0, 0, 0), D = make_float3(0, 0, 1), rD = make_float3(1e30f, 1e30f, 1);
        float4 hit = make_float4(0, 0, 0, 0);
        if (have) {
            const RayRec* rp = q.rays + ri;
            O = xyz(rp->O); D = xyz(rp->D); rD = xyz(rp->rD);
            hit = q.fresh ? make_float4(q.freshTmax, 0.f, 0.f, 0.f) : rp->hit;
        }
        bool found = false;
        bool on = have;                                   // this lane's ray still takes part
        // (by the signs of rD, which is what the slab test multiplies with: a caller's rD need not agree with D in sign for |D| < 1e-12, tinybvh_safercp)
        const uint32_t oct = 7u - ((rD.x < 0 ? 4u : 0u) | (rD.y < 0 ? 2u : 0u) | (rD.z < 0 ? 1u : 0u));
        // the wave's octant = that of its first ray; lanes of another octant make the chunk "mixed" (slab test through min / max)
        const unsigned long long haveMask = wave_ballot(have);
        const uint32_t first = (uint32_t)__builtin_ctzll(haveMask);
        const uint32_t oct0 = (uint32_t)__builtin_amdgcn_readlane((int)oct, (int)first);
        const bool mixed = wave_ballot(have && oct != oct0) != 0;
        const bool negX0 = ((7u - oct0) & 4u) != 0, negY0 = ((7u - oct0) & 2u) != 0, negZ0 = ((7u - oct0) & 1u) != 0;
        // where lane L's plane goes in planes[child][.]: byte L of the node's 48 is plane p = L >> 3 (qlo_x, qlo_y, qlo_z, qhi_x, qhi_y, qhi_z) of child L & 7;
        // near = lo unless the wave's rays travel in -axis
        const uint32_t p = lane >> 3, axis = p % 3u, isHi = p / 3u;
        const bool negA = axis == 0 ? negX0 : axis == 1 ? negY0 : negZ0;
        const uint32_t dstPlane = axis + 3u * (isHi ^ (negA ? 1u : 0u));
        float* const myPlane = &planes[lane & 7u][lane < 48u ? dstPlane : 6u + ((lane >> 3) & 1u)];   // (columns 6, 7 of a row are spare)

        uint32_t sp = 0;
        uint32_t ngx = 0, ngy = 0x80000000u;
        uint32_t stkx = 0, stky = 0;
        // the wave's next node: taken off the current node group, or off the stack; false when the traversal is over.  Only the wave-uniform state decides
        // (never a ray's hit distance), so the NEXT node can be picked — and its loads issued — before the current node's triangles are tested.
        // The wave's stack lives in the LANES of two vector registers: entry k = lane k of (stkx, stky), written by a select on lane == k and read with v_readlane
        // at a wave-uniform index — no LDS round trip, no exec juggling for "lane 0 only".  Entries beyond 64 (no tree seen needs them) go to the wave's
        // slots of the global spill area.
        auto pick = [&](uint32_t& ci) -> bool {
            if (!(ngy > 0x00FFFFFFu)) {
                if (sp == 0) return false;
                sp--;
                if (sp < 64u) { ngx = (uint32_t)__builtin_amdgcn_readlane((int)stkx, (int)sp); ngy = (uint32_t)__builtin_amdgcn_readlane((int)stky, (int)sp); }
                else { const uint32_t j_ = sp - 64u; const uint2 e = spill[(j_ & 63u) + (size_t)(j_ >> 6) * spillRow]; ngx = sgpr(e.x); ngy = sgpr(e.y); }
            }
            const uint32_t imaskWord = ngy;
            const uint32_t bit = 31u - (uint32_t)__builtin_clz(ngy);
            ngy &= ~(1u << bit);
            if (ngy > 0x00FFFFFFu) {   // children of this group still pending: keep it
                if (sp < 64u) { const bool mine = lane == sp; stkx = mine ? ngx : stkx; stky = mine ? ngy : stky; }   // (lane sp of the pair takes the entry: a compare and two selects)
                else {
                    const uint32_t j_ = sp - 64u;
                    if (j_ < spillCap) { if (lane == 0) spill[(j_ & 63u) + (size_t)(j_ >> 6) * spillRow] = make_uint2(ngx, ngy); }
                    else { overflow = true; sp--; }   // dropped (status bit 1 -> TBVH_E_FORMAT): the stack pointer must not run past the wave's rows of the spill area
                }
                sp++;
            }
            const uint32_t slot = (bit - 24u) ^ oct0;
            ci = sgpr(ngx + (uint32_t)__popc(imaskWord & ~(0xFFFFFFFFu << slot)));
            return true;
        };
        uint32_t ci = 0;
        bool more = pick(ci);
        // a node in flight: n0, n1 (one request for the wave: every lane asks for the same 32 bytes) and lane L's plane byte
        float4 n0 = make_float4(0, 0, 0, 0), n1 = n0;
        uint32_t qb = 0;
        auto fetch = [&](uint32_t c_) {
            const float4* np = nodes + (size_t)c_ * 5u;
            n0 = np[0]; n1 = np[1];
            qb = ((const uint8_t*)(np + 2))[lane < 48u ? lane : 47u];   // (lanes 48..63 re-read byte 47 and park it in an unused column: no exec mask to set up)
        };
        if (more) fetch(ci);
        while (more) {
            // ---- the node: decoded once for the wave ---------------------------------------------------------------------
            __builtin_amdgcn_wave_barrier();               // (every lane has read the previous node's planes before they are overwritten)
            *myPlane = (float)qb;
            __builtin_amdgcn_wave_barrier();
            const uint32_t ew = sgpr(as_u32(n0.w));
            const float ax = ldexpf(rD.x, (int)(int8_t)(ew)), ay = ldexpf(rD.y, (int)(int8_t)(ew >> 8)), az = ldexpf(rD.z, (int)(int8_t)(ew >> 16));
            const float ox = (n0.x - O.x) * rD.x, oy = (n0.y - O.y) * rD.y, oz = (n0.z - O.z) * rD.z;
            const bool lives = ANYHIT ? (on && !found) : on;
            const float tcull = lives ? cull_bound(hit.x) : -1.0f;   // (a lane without a live ray enters no box: entry distances are clamped to >= 0)
            const uint32_t m0 = sgpr(as_u32(n1.z)), m1 = sgpr(as_u32(n1.w));
            const uint32_t hitmask = mixed ? pk_test_children<true>(planes, ax, ay, az, ox, oy, oz, tcull, m0, m1, oct0)
                                           : pk_test_children<false>(planes, ax, ay, az, ox, oy, oz, tcull, m0, m1, oct0);
            ngx = sgpr(as_u32(n1.x));
            ngy = (hitmask & 0xFF000000u) | (ew >> 24);
            uint32_t tgy = hitmask & 0x00FFFFFFu;
            const uint32_t tgx = sgpr(as_u32(n1.y));
            // ---- the next node's loads go out now: they fly while this node's triangles are tested ------------------------------
            // (issuing the first triangle's loads ahead of them as well measured 5-9 % slower)
            more = pick(ci);
            if (more) fetch(ci);
            // ---- the triangles of the leaves any ray entered: every lane tests them -----------------------------------------
            while (tgy != 0u) {
                const uint32_t ti = 31u - (uint32_t)__builtin_clz(tgy);
                tgy &= ~(1u << ti);
                const float4* tp = tris + ((size_t)tgx + ti * 3u);
                const float4 e2 = tp[0], e1 = tp[1], v0 = tp[2];
                TriHit h;
                if ((ANYHIT ? (on && !found) : on) &&
                    tri_test(O, D, xyz(v0), xyz(e1), xyz(e2), hit.x, h, HAS_OMM ? q.omm : Omm{nullptr, 0}, as_u32(v0.w)) &&
                    (ANYHIT || hit_wins(h.t, as_u32(v0.w), found, hit))) {
                    found = true;
                    if (!ANYHIT) hit = make_float4(h.t, h.u, h.v, v0.w);
                }
            }
            if (ANYHIT && wave_ballot(on && !found) == 0ull) break;   // every ray of the wave is occluded
        }
        // ---- results ----------------------------------------------------------------------------------------------------------
        if (have) {
            if (ANYHIT) q.occluded[ri] = found ? 1 : 0;
            else if (found || q.fresh) q.rays[ri].hit = hit;
        }
    }
    if (overflow) atomicOr(status, 1u);
}

}  // namespace

void launch_cwbvh_packet(bool anyhit, const float4* nodes, const float4* tris, const QueryArgs& q, uint32_t* status, uint32_t blocks, hipStream_t s) {
    if (anyhit) {
        if (q.omm.map) hipLaunchKernelGGL((k_cwbvh_packet<true, true>), dim3(blocks), dim3(WG), 0, s, nodes, tris, q, status);
        else hipLaunchKernelGGL((k_cwbvh_packet<true, false>), dim3(blocks), dim3(WG), 0, s, nodes, tris, q, status);
    } else {
        if (q.omm.map) hipLaunchKernelGGL((k_cwbvh_packet<false, true>), dim3(blocks), dim3(WG), 0, s, nodes, tris, q, status);
        else hipLaunchKernelGGL((k_cwbvh_packet<false, false>), dim3(blocks), dim3(WG), 0, s, nodes, tris, q, status);
    }
}

}  // namespace tbvh
