// kernels_cwbvh_dual.hip — BVH8_CWBVH Intersect / IsOccluded for INCOHERENT batches with TWO rays per lane (round 6).
//
// The incoherent flavor of kernels_cwbvh.hip (PROBED == 2) issues ~10 200 VALU lane-slots per bounce ray of the bench scene at 0.61 lanes busy: a
// node phase (230 instructions) runs in every pass for 75 % of the lanes — the others hold a triangle to test first or wait for a new ray — and a
// triangle phase (65) in nearly every pass for 18 % (profiles/r06_diffuse.txt).  Lanes differ in PHASE, not in locality.  Here every lane holds two
// rays, each with its own traversal state (node group, triangle group, closest hit, stack), and in every pass
//   * ONE triangle phase serves, per lane, a ray that has a triangle pending (ray 0 if it has one, else ray 1);
//   * ONE node phase serves, per lane, a ray that wants a node (ray 0 if it does, else ray 1);
// the chosen ray's state is copied in and out with selects (~25 instructions per phase).  A lane sits a phase out only when NEITHER of its rays wants
// it: node phase ~0.96 of the lanes, triangle phase ~0.3.  The price: ~96 VGPRs (5 waves per SIMD instead of 8), two stacks per lane in LDS (6 entries each).
// Per-ray order of tests is the strict schedule's, so the records are the per-lane kernel's bytes (and the tie rule makes them order-independent anyway).
// Same copies as that flavor: hybrid node array (cwbvh_node.h: kNodeHybrid) with an embedded triangle per deep node, 64-byte triangle records,
// non-temporal ray records.  No split rays: serves batches of 12 M rays and more (capi_query.hip).
// Format / reference: tiny_bvh.h:5884-6018 (blobs), 3222-3453 (semantics); traverse_cwbvh.cl:124-570 is what it replaces.
#include "device_common.h"
#include "ray_pool.h"
#include "kernels.h"
#include "cwbvh_node.h"

namespace tbvh {

namespace {

constexpr int WG = 64;

#define TBVH_AS_LDS2 __attribute__((address_space(3)))
#define TBVH_AS_GLOBAL2 __attribute__((address_space(1)))

// one ray's traversal state (all members live in registers; two of these per lane)
struct DualRay {
    float3 O, D, rD;
    float4 hit;
    uint2 ng, tg;          // Ylitie's node group / triangle group (cwbvh_node.h)
    uint32_t tgn;          // hybrid copy: float4 offset of the node tg came from (its line may hold one of its triangles)
    uint32_t octinv4;      // (7 - sign octant) in four bytes
    uint32_t ri;           // ray index (batches below 2^32 rays)
    int sp;                // stack height
    bool active, found, negX, negY, negZ;
};

template <bool ANYHIT, int LDS_N, int REFILL, bool HAS_OMM>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_cwbvh_dual(const float4* __restrict__ nodes, const float4* __restrict__ tris, QueryArgs q,
                                                                                             uint32_t* __restrict__ status) {
    typedef unsigned long long W64;
    __shared__ W64 stk[2][LDS_N][WG];
    const uint32_t lane = threadIdx.x;
    TBVH_AS_LDS2 W64* const lds0 = (TBVH_AS_LDS2 W64*)&stk[0][0][lane];
    const uint32_t spillCap = q.spillStride / 2u;                       // 8-byte entries per ray (the lane's rows of the spill area are shared by its two rays)
    const size_t spillRow = (size_t)gridDim.x * WG;
    TBVH_AS_GLOBAL2 W64* const spill0 = (TBVH_AS_GLOBAL2 W64*)((uint2*)q.spill + (blockIdx.x * WG + lane));
    bool overflow = false;
    RayPool<64> pool;
    const uint64_t nRaysTotal = q.nRaysDev ? *q.nRaysDev : q.nRays;
    pool.init(q.poolParts, q.counterNext);
    const uint32_t hybridK = q.hybridK;
    const uint32_t triGate = ((q.flags >> 20) & 15u) * 4u;   // lanes with a triangle pending that start a triangle phase (0: any)

    DualRay r0, r1;
    r0.active = r1.active = false; r0.found = r1.found = false;
    r0.O = r0.D = r0.rD = r1.O = r1.D = r1.rD = make_float3(0, 0, 0);
    r0.hit = r1.hit = make_float4(0, 0, 0, 0);
    r0.ng = r0.tg = r1.ng = r1.tg = make_uint2(0u, 0u);
    r0.tgn = r1.tgn = 0; r0.octinv4 = r1.octinv4 = 0; r0.ri = r1.ri = 0; r0.sp = r1.sp = 0;
    r0.negX = r0.negY = r0.negZ = r1.negX = r1.negY = r1.negZ = false;

    auto take = [&](DualRay& r, bool idle) {   // hands the pool's next rays to the lanes whose slot is idle (convergent call)
        uint64_t nri = 0;
        if (pool.acquire(idle, q.counter, nRaysTotal, nri)) {
            const tbvh_f4* r4 = (const tbvh_f4*)(q.rays + nri);
            const tbvh_f4 a = __builtin_nontemporal_load(r4), b = __builtin_nontemporal_load(r4 + 1), c = __builtin_nontemporal_load(r4 + 2);
            r.O = make_float3(a.x, a.y, a.z); r.D = make_float3(b.x, b.y, b.z); r.rD = make_float3(c.x, c.y, c.z);
            r.hit = q.fresh ? make_float4(q.freshTmax, 0.f, 0.f, 0.f) : q.rays[nri].hit;
            r.found = false;
            r.octinv4 = cw_oct(r.D) * 0x01010101u;
            r.negX = r.rD.x < 0; r.negY = r.rD.y < 0; r.negZ = r.rD.z < 0;
            r.ng = make_uint2(0u, 0x80000000u); r.tg = make_uint2(0u, 0u);
            r.sp = 0; r.ri = (uint32_t)nri;
            r.active = true;
        }
    };
    auto finish = [&](const DualRay& r) {      // the ray's record, once its traversal is over
        if (ANYHIT) q.occluded[r.ri] = r.found ? 1 : 0;
        else if (r.found || q.fresh) { tbvh_f4 hv; hv.x = r.hit.x; hv.y = r.hit.y; hv.z = r.hit.z; hv.w = r.hit.w; __builtin_nontemporal_store(hv, (tbvh_f4*)&q.rays[r.ri].hit); }
    };

    for (;;) {
        // ---- ray replacement: 128 slots per wave ---------------------------------------------------------------------
        const uint32_t nIdle = wave_count(!r0.active) + wave_count(!r1.active);
        if (nIdle >= (uint32_t)REFILL) {
            if (!pool.dry()) { take(r0, !r0.active); if (!pool.dry()) take(r1, !r1.active); }
            if (wave_ballot(r0.active || r1.active) == 0) break;
        }
        // ---- triangle phase: per lane ONE triangle of a ray that has one pending --------------------------------------------------------
        {
            const bool t0 = r0.active && r0.tg.y != 0u, t1 = r1.active && r1.tg.y != 0u;
            // gated: a ray that waits for its triangle costs its lane nothing while the lane's other ray takes node phases, so the phase runs once
            // triGate lanes have a triangle pending — or when no lane could take a node phase instead (progress)
            const uint32_t nTri = wave_count(t0 || t1);
            const bool canNode = (r0.active && r0.tg.y == 0u) || (r1.active && r1.tg.y == 0u);
            if (nTri != 0u && (nTri >= triGate || wave_ballot(canNode) == 0)) {
                const bool in = t0 || t1, b = !t0;
                const float3 O = b ? r1.O : r0.O, D = b ? r1.D : r0.D;
                float4 hit = b ? r1.hit : r0.hit;
                uint2 tg = b ? r1.tg : r0.tg;
                const uint32_t tgn = b ? r1.tgn : r0.tgn;
                bool found = b ? r1.found : r0.found;
                if (in) {
                    const uint32_t ti = 31u - (uint32_t)__clz(tg.y);
                    tg.y &= ~(1u << ti);
                    // the hybrid copy's triangle word: embedded << 27 | first 64-byte record; the embedded triangle sits in the node's own line
                    const float4* tp = ti == (tg.x >> 27) ? nodes + ((size_t)tgn + 5u) : tris + ((size_t)(tg.x & 0x07FFFFFFu) + ti) * 4u;
                    const float4 e2 = tp[0], e1 = tp[1], v0 = tp[2];
                    tri_loads_together(v0);
                    TriHit h;
                    if (!(ANYHIT && found) && tri_test(O, D, xyz(v0), xyz(e1), xyz(e2), hit.x, h, HAS_OMM ? q.omm : Omm{nullptr, 0}, as_u32(v0.w)) &&
                        (ANYHIT || hit_wins(h.t, as_u32(v0.w), found, hit))) {
                        found = true;
                        if (!ANYHIT) hit = make_float4(h.t, h.u, h.v, v0.w);
                        else tg.y = 0u;                                    // occluded: nothing of this ray is left to do (the node phase below ends it)
                    }
                }
                const bool w0 = in && !b, w1 = in && b;
                r0.tg.y = w0 ? tg.y : r0.tg.y; r1.tg.y = w1 ? tg.y : r1.tg.y;
                r0.found = w0 ? found : r0.found; r1.found = w1 ? found : r1.found;
                if (!ANYHIT) {
                    r0.hit.x = w0 ? hit.x : r0.hit.x; r0.hit.y = w0 ? hit.y : r0.hit.y; r0.hit.z = w0 ? hit.z : r0.hit.z; r0.hit.w = w0 ? hit.w : r0.hit.w;
                    r1.hit.x = w1 ? hit.x : r1.hit.x; r1.hit.y = w1 ? hit.y : r1.hit.y; r1.hit.z = w1 ? hit.z : r1.hit.z; r1.hit.w = w1 ? hit.w : r1.hit.w;
                }
            }
        }
        // ---- node phase: per lane ONE node of a ray whose triangle group is empty -----------------------------------------------------------
        {
            const bool n0 = r0.active && r0.tg.y == 0u, n1 = r1.active && r1.tg.y == 0u;
            const bool in = n0 || n1, b = !n0;
            const float3 O = b ? r1.O : r0.O, rD = b ? r1.rD : r0.rD;
            const float tmax = b ? r1.hit.x : r0.hit.x;
            const uint32_t octinv4 = b ? r1.octinv4 : r0.octinv4;
            const bool negX = b ? r1.negX : r0.negX, negY = b ? r1.negY : r0.negY, negZ = b ? r1.negZ : r0.negZ;
            uint2 ng = b ? r1.ng : r0.ng;
            uint2 tg = make_uint2(0u, 0u);
            uint32_t tgn = 0;
            int sp = b ? r1.sp : r0.sp;
            const bool occluded = ANYHIT && (b ? r1.found : r0.found);
            bool done = false;
            if (in) {
                bool have = cw_has_child(ng) && !occluded;
                if (!have) {
                    if (sp > 0 && !occluded) {
                        sp--;
                        W64 w;
                        if (sp < LDS_N) w = lds0[((b ? LDS_N : 0) + sp) * WG];
                        else w = spill0[((size_t)(b ? spillCap : 0u) + (size_t)(sp - LDS_N)) * spillRow];
                        ng = make_uint2((uint32_t)w, (uint32_t)(w >> 32));
                        have = true;
                    } else done = true;
                }
                if (have) {
                    const uint32_t ci = cw_next_child(ng, octinv4 & 7u);
                    if (cw_has_child(ng)) {   // only node groups with children pending are ever pushed
                        const W64 w = ((W64)ng.y << 32) | (W64)ng.x;
                        if (sp < LDS_N) { lds0[((b ? LDS_N : 0) + sp) * WG] = w; sp++; }
                        else if ((uint32_t)(sp - LDS_N) < spillCap) { spill0[((size_t)(b ? spillCap : 0u) + (size_t)(sp - LDS_N)) * spillRow] = w; sp++; }
                        else overflow = true;
                    }
                    const CwNodeHits r = cw_test_node(cw_load_node<kNodeHybrid>(nodes, ci, hybridK), O, rD, cull_bound(tmax), octinv4, negX, negY, negZ);
                    ng = make_uint2(r.childBase, (r.hitmask & 0xFF000000u) | r.imask);
                    tg = make_uint2(r.triBase, r.hitmask & 0x00FFFFFFu);
                    tgn = cw_hybrid_offset(ci, hybridK);
                }
            }
            const bool w0 = in && !b, w1 = in && b;
            r0.ng.x = w0 ? ng.x : r0.ng.x; r0.ng.y = w0 ? ng.y : r0.ng.y; r1.ng.x = w1 ? ng.x : r1.ng.x; r1.ng.y = w1 ? ng.y : r1.ng.y;
            r0.tg.x = w0 ? tg.x : r0.tg.x; r0.tg.y = w0 ? tg.y : r0.tg.y; r1.tg.x = w1 ? tg.x : r1.tg.x; r1.tg.y = w1 ? tg.y : r1.tg.y;
            r0.tgn = w0 ? tgn : r0.tgn; r1.tgn = w1 ? tgn : r1.tgn;
            r0.sp = w0 ? sp : r0.sp; r1.sp = w1 ? sp : r1.sp;
            if (done) {
                if (!b) { finish(r0); r0.active = false; }
                else { finish(r1); r1.active = false; }
            }
        }
    }
    if (overflow) atomicOr(status, 1u);
}

}  // namespace

// nodes: the hybrid node copy, tris: the 64-byte triangle records (capi_scene.hip: prepareIncoherentCopies); batches below 2^32 rays
void launch_cwbvh_dual(bool anyhit, const float4* nodes, const float4* tris, const QueryArgs& q, uint32_t* status, uint32_t blocks, hipStream_t s) {
    if (anyhit) {
        if (q.omm.map) hipLaunchKernelGGL((k_cwbvh_dual<true, 6, 24, true>), dim3(blocks), dim3(WG), 0, s, nodes, tris, q, status);
        else hipLaunchKernelGGL((k_cwbvh_dual<true, 6, 24, false>), dim3(blocks), dim3(WG), 0, s, nodes, tris, q, status);
    } else {
        if (q.omm.map) hipLaunchKernelGGL((k_cwbvh_dual<false, 6, 24, true>), dim3(blocks), dim3(WG), 0, s, nodes, tris, q, status);
        else hipLaunchKernelGGL((k_cwbvh_dual<false, 6, 24, false>), dim3(blocks), dim3(WG), 0, s, nodes, tris, q, status);
    }
}

}  // namespace tbvh
