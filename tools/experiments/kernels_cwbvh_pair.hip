// kernels_cwbvh_pair.hip — BVH8_CWBVH Intersect / IsOccluded for INCOHERENT batches with TWO lanes per ray (round 4).
//
// Why: the counters of round 4 (DESIGN.md par. 5 "Round 4") put the bounce launch of kernels_cwbvh.hip on the L1's lookup rate — 0.91 lookups
// per cycle and CU, because one lane per ray pays five divergent 16-byte loads per node visit — with VALU issue at 0.75.  Here two neighbouring
// lanes share a ray: the 80-byte node arrives as THREE pair-coalesced loads (n0 | n1, n2 | n3, n4: consecutive 16-byte pieces of one line are
// one lookup), each lane slab-tests four of the eight children (one `half` of cwbvh_node.h: cw_test_node) and the two hit masks are OR-ed
// across the pair; the triangle phase tests up to TWO triangles of a node at once and merges the candidates by the library's tie rule
// (device_common.h: hit_wins — the smaller (t, prim) first).  Everything else of a ray's state (node group, triangle group, stack, closest
// hit) is kept identically in both lanes, so the pair never diverges and nothing but the node pieces, the partial hit mask and a triangle
// candidate crosses lanes (DPP quad_perm, no LDS).
//
// Same blobs as the incoherent flavor of kernels_cwbvh.hip: the hybrid node copy (priority order, first hybridK nodes packed, the others one
// per 128-byte line with one triangle embedded) and the 64-byte triangle records; strict schedule; ray records with the non-temporal hint.
// Launched as the SECOND kernel of a probed query in place of that flavor for batches that do not split their last rays (>= 12 M rays):
// reads the verdict the first kernel published and leaves at once when the batch is coherent.  Records are a function of ray and scene
// alone (tie rule), so they equal the one-lane kernels' byte for byte (tests/test_cwbvh_schedules.py, tests/test_bench_kernels.py).
#include "device_common.h"
#include "lane_stack.h"
#include "ray_pool.h"
#include "kernels.h"
#include "cwbvh_node.h"

namespace tbvh {

namespace {

constexpr int WG = 64;

// the other lane of the pair / the even lane of the pair (DPP quad_perm [1,0,3,2] and [0,0,2,2])
__device__ __forceinline__ uint32_t pair_swap(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t pair_even(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xA0, 0xF, 0xF, true); }
__device__ __forceinline__ float pair_swap(float v) { return as_f32(pair_swap(as_u32(v))); }

template <bool ANYHIT, bool HAS_OMM>
__global__ __launch_bounds__(WG, 8) void k_cwbvh_pair(const float4* __restrict__ nodes, const float4* __restrict__ tris, QueryArgs q, uint32_t* __restrict__ status) {
    __shared__ uint2 stk[8][WG];
    const uint32_t lane = threadIdx.x;
    const bool h = (lane & 1u) != 0;   // which half of a node's children this lane tests
    const uint32_t glane = blockIdx.x * WG + lane;
    LaneStack<uint2, 8, WG> st;        // (both lanes of a pair keep the same stack: no cross-lane traffic on push / pop; the LDS pipe has room)
    st.init(&stk[0][lane], (uint2*)q.spill + glane, gridDim.x * WG, q.spillStride);
    RayPool<64> pool;
    const uint64_t nRaysTotal = q.nRaysDev ? *q.nRaysDev : q.nRays;
    pool.init(q.poolParts, q.counterNext);
    if (q.probe) {   // the verdict of the query's first kernel (kernels_cwbvh.hip: coherence_sample): this kernel serves incoherent batches
        const uint32_t agree = q.probe[0], pairs = q.probe[1];
        if (pairs != 0 && agree * 10u >= pairs * 6u) return;
    }
    const uint32_t hybridK = q.hybridK;

    bool active = false;
    uint64_t ri = 0;
    float3 O = make_float3(0, 0, 0), D = O, rD = O;
    float4 hit = make_float4(0, 0, 0, 0);
    bool found = false;
    uint32_t oct = 0, octinv4 = 0;
    uint2 ng = make_uint2(0u, 0u), tg = make_uint2(0u, 0u);
    uint32_t tgn = 0;

    for (;;) {
        // ---- ray replacement: idle PAIRS take the next rays of the pool (the even lane draws, the odd one copies) ----------------------------
        const uint32_t nIdle = (uint32_t)__popcll(__ballot(!active));
        if (nIdle >= 16u) {
            if (!pool.dry()) {
                uint64_t nri = 0;
                const bool got = pool.acquire(!active && !h, q.counter, nRaysTotal, nri);
                const bool gotPair = pair_even(got ? 1u : 0u) != 0u;
                const uint32_t lo = pair_even((uint32_t)nri), hi = pair_even((uint32_t)(nri >> 32));
                if (gotPair) {
                    ri = ((uint64_t)hi << 32) | lo;
                    const tbvh_f4* r4 = (const tbvh_f4*)(q.rays + ri);   // read once by one CU: non-temporal, as in the one-lane incoherent flavor
                    const tbvh_f4 a = __builtin_nontemporal_load(r4), b = __builtin_nontemporal_load(r4 + 1), c = __builtin_nontemporal_load(r4 + 2);
                    O = make_float3(a.x, a.y, a.z); D = make_float3(b.x, b.y, b.z); rD = make_float3(c.x, c.y, c.z);
                    hit = q.fresh ? make_float4(q.freshTmax, 0.f, 0.f, 0.f) : q.rays[ri].hit;
                    found = false;
                    oct = cw_oct(D);
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
        // ---- triangle phase: the pair tests the two front-most pending triangles of its group at once ---------------------------------------
        if (tg.y != 0) {
            uint32_t bits = tg.y;
            const uint32_t t0 = 31u - (uint32_t)__clz(bits);
            bits &= ~(1u << t0);
            const bool two = bits != 0;
            const uint32_t t1 = two ? 31u - (uint32_t)__clz(bits) : 0u;
            if (two) bits &= ~(1u << t1);
            tg.y = bits;
            const bool mine = !h || two;
            const uint32_t ti = h ? t1 : t0;
            bool hitC = false;
            TriHit th; th.t = 0.f; th.u = 0.f; th.v = 0.f;
            uint32_t prim = 0u;
            if (mine) {
                const float4* tp = ti == (tg.x >> 27) ? nodes + ((size_t)tgn + 5u) : tris + ((size_t)(tg.x & 0x07FFFFFFu) + ti) * 4u;
                const float4 e2 = tp[0], e1 = tp[1], v0 = tp[2];
                prim = as_u32(v0.w);
                hitC = tri_test(O, D, xyz(v0), xyz(e1), xyz(e2), hit.x, th, HAS_OMM ? q.omm : Omm{nullptr, 0}, prim);
            }
            // merge the pair's candidates: the lexicographically smaller (t, prim) — what testing them one after the other under hit_wins leaves
            const bool hitP = pair_swap(hitC ? 1u : 0u) != 0u;
            const float pt = pair_swap(th.t), pu = pair_swap(th.u), pv = pair_swap(th.v);
            const uint32_t pp = pair_swap(prim);
            if (hitP && (!hitC || pt < th.t || (pt == th.t && pp < prim))) { th.t = pt; th.u = pu; th.v = pv; prim = pp; }
            if ((hitC || hitP) && (ANYHIT || hit_wins(th.t, prim, found, hit))) {
                found = true;
                if (ANYHIT) done = true;
                else hit = make_float4(th.t, th.u, th.v, as_f32(prim));
            }
        }
        // ---- node phase (strict schedule: only with no triangle pending) -----------------------------------------------------------------------
        if (!done && tg.y == 0) {
            bool have = cw_has_child(ng);
            if (!have) {
                if (!st.empty()) { ng = st.pop(); have = true; }
                else done = true;
            }
            if (have) {
                const uint32_t ci = cw_next_child(ng, oct);
                if (cw_has_child(ng)) st.push(ng);
                const uint32_t off = cw_hybrid_offset(ci, hybridK);
                const float4* np = nodes + off;
                // three pair-coalesced loads: (n0 | n1), (n2 | n3), n4
                const float4 R0 = np[h ? 1 : 0], R1 = np[h ? 3 : 2], R2 = np[4];
                // this lane's view of the node: n0 whole; of n1 the two bases and ITS meta word; of n2-n4 ITS word of each plane pair.  The choice between
                // a lane's own register and its partner's is a BITWISE select (v_bfi_b32) on purpose: written as `h ? swap(x) : y` the compiler turns it
                // into a branch on h and executes the DPP move with the partner lane masked off — which then reads zero
                const uint32_t hm = h ? 0xFFFFFFFFu : 0u;
                auto sel = [hm](uint32_t ifOdd, uint32_t ifEven) { return (ifOdd & hm) | (ifEven & ~hm); };
                const uint32_t r0x = as_u32(R0.x), r0y = as_u32(R0.y), r0z = as_u32(R0.z), r0w = as_u32(R0.w);
                const uint32_t r1x = as_u32(R1.x), r1y = as_u32(R1.y), r1z = as_u32(R1.z), r1w = as_u32(R1.w);
                const uint32_t p0x = pair_swap(r0x), p0y = pair_swap(r0y), p0z = pair_swap(r0z), p0w = pair_swap(r0w);
                const uint32_t p1x = pair_swap(r1x), p1y = pair_swap(r1y), p1z = pair_swap(r1z), p1w = pair_swap(r1w);
                const float n0x = as_f32(sel(p0x, r0x)), n0y = as_f32(sel(p0y, r0y)), n0z = as_f32(sel(p0z, r0z));
                const uint32_t ew = sel(p0w, r0w);
                const uint32_t childBase = sel(r0x, p0x), triBase = sel(r0y, p0y), meta4 = sel(r0w, p0z);
                const uint32_t qlx = sel(p1y, r1x), qly = sel(p1w, r1z), qlz = sel(r1y, p1x), qhx = sel(r1w, p1z);
                const uint32_t qhy = sel(as_u32(R2.y), as_u32(R2.x)), qhz = sel(as_u32(R2.w), as_u32(R2.z));
                // the slab test of four children: one `half` of cw_test_node (cwbvh_node.h), same arithmetic
                const float tmax = cull_bound(hit.x);
                const float ax = ldexpf(rD.x, (int)(int8_t)(ew)), ay = ldexpf(rD.y, (int)(int8_t)(ew >> 8)), az = ldexpf(rD.z, (int)(int8_t)(ew >> 16));
                const float ox = (n0x - O.x) * rD.x, oy = (n0y - O.y) * rD.y, oz = (n0z - O.z) * rD.z;
                const uint32_t inner4 = (meta4 & (meta4 << 1)) & 0x10101010u;
                const uint32_t imask4 = cw_sext_s8x4(inner4 << 3);
                const uint32_t bitidx4 = (meta4 ^ (octinv4 & imask4)) & 0x1F1F1F1Fu;
                const uint32_t bits4 = (meta4 >> 5) & 0x07070707u;
                const uint32_t lox = rD.x < 0 ? qhx : qlx, hix = rD.x < 0 ? qlx : qhx;
                const uint32_t loy = rD.y < 0 ? qhy : qly, hiy = rD.y < 0 ? qly : qhy;
                const uint32_t loz = rD.z < 0 ? qhz : qlz, hiz = rD.z < 0 ? qlz : qhz;
                uint32_t hitmask = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int sh = 8 * i;
                    const float tnx = __builtin_fmaf((float)((lox >> sh) & 255), ax, ox), tfx = __builtin_fmaf((float)((hix >> sh) & 255), ax, ox);
                    const float tny = __builtin_fmaf((float)((loy >> sh) & 255), ay, oy), tfy = __builtin_fmaf((float)((hiy >> sh) & 255), ay, oy);
                    const float tnz = __builtin_fmaf((float)((loz >> sh) & 255), az, oz), tfz = __builtin_fmaf((float)((hiz >> sh) & 255), az, oz);
                    const float cmin = __builtin_fmaxf(cw_fmax3(tnx, tny, tnz), 0.0f);
                    const float cmax = __builtin_fminf(cw_fmin3(tfx, tfy, tfz), tmax);
                    if (cmin <= cmax) hitmask |= ((bits4 >> sh) & 255u) << ((bitidx4 >> sh) & 255u);
                }
                hitmask |= pair_swap(hitmask);
                ng = make_uint2(childBase, (hitmask & 0xFF000000u) | (ew >> 24));
                tg = make_uint2(triBase, hitmask & 0x00FFFFFFu);
                tgn = off;
            }
        }
        if (done) {
            if (!h) {
                if (ANYHIT) q.occluded[ri] = found ? 1 : 0;
                else if (found || q.fresh) { tbvh_f4 hv; hv.x = hit.x; hv.y = hit.y; hv.z = hit.z; hv.w = hit.w; __builtin_nontemporal_store(hv, (tbvh_f4*)&q.rays[ri].hit); }
            }
            active = false;
        }
    }
    if (st.overflow) atomicOr(status, 1u);
}

}  // namespace

void launch_cwbvh_pair(bool anyhit, const float4* nodesHybrid, const float4* tris64, const QueryArgs& q, uint32_t* status, uint32_t blocks, hipStream_t s) {
    if (anyhit) {
        if (q.omm.map) hipLaunchKernelGGL((k_cwbvh_pair<true, true>), dim3(blocks), dim3(WG), 0, s, nodesHybrid, tris64, q, status);
        else hipLaunchKernelGGL((k_cwbvh_pair<true, false>), dim3(blocks), dim3(WG), 0, s, nodesHybrid, tris64, q, status);
    } else {
        if (q.omm.map) hipLaunchKernelGGL((k_cwbvh_pair<false, true>), dim3(blocks), dim3(WG), 0, s, nodesHybrid, tris64, q, status);
        else hipLaunchKernelGGL((k_cwbvh_pair<false, false>), dim3(blocks), dim3(WG), 0, s, nodesHybrid, tris64, q, status);
    }
}

}  // namespace tbvh
