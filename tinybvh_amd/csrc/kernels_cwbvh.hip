// kernels_cwbvh.hip — BVH8_CWBVH Intersect / IsOccluded for gfx950 (MI355X): the kernel the library ships.
//
// Replaces batch_cwbvh / isoccluded_cwbvh (traverse_cwbvh.cl:124-570) from scratch.
// Blob format: nodes 5 x float4 (cwbvh_node.h), tris 3 x float4 {e2, e1, v0|prim}, both verbatim as
// BVH8_CWBVH::ConvertFrom writes them (tiny_bvh.h:5884-6018).  Traversal state machine after
// Ylitie et al. 2017 as restated by the CPU mirror tiny_bvh.h:7046-7154: ngroup = {child base,
// hits << 24 | imask}, tgroup = {tri base, tri bits}; highest set bit first = front-to-back through
// octinv.  Hit semantics follow BVH::Intersect (inclusive t range, miss leaves the record untouched).
//
// Schedule: persistent one-wave workgroups, one lane = one ray, rays drawn from a striped
// pool (ray_pool.h) under the lockstep governor (whole generations while the wave's rays stay together,
// per-lane replacement once they do not), traversal stack top in LDS / bottom in global (lane_stack.h).
// Per loop iteration a lane does at most ONE triangle test and ONE node visit.
//
// Deferred triangles (SPEC): a lane whose triangle group is not finished may still visit its next node — the
// new node's triangles wait in a second group (tg2) — so lanes with multi-triangle groups no longer sit out
// node phases, and the triangle phase itself only runs once TRI_MIN lanes have a triangle pending (or
// nothing else can make progress).  A node visited ahead of pending triangle tests sees a tmax that may not
// be final: it can only report MORE children than the strict order would (they are culled or re-tested on
// arrival), never fewer, so the nearest hit is unchanged; among triangles at exactly equal t the winner
// may differ from the mirror's (the "tie" class of tests/oracle_lib.py, which BVH::Intersect and the
// reference's own wide layouts already disagree on).
#include "device_common.h"
#include "lane_stack.h"
#include "ray_pool.h"
#include "kernels.h"
#include "cwbvh_node.h"
#include "ray_split.h"
#include "cwbvh_probe.h"

namespace tbvh {

namespace {

constexpr int WG = 64;

// STATS (experiment builds): q.stats[0] wave iterations, [1] sum of active lanes, [2] sum of lanes visiting a node,
// [3] triangle-phase iterations, [4] sum of lanes in them, [5] node phases whose lanes all visit ONE node with ONE octant,
// [6] sum of lanes in those, [7] node-phase iterations
template <bool ANYHIT, int LDS_N, int REFILL_MIN, int TRI_MIN, bool SPEC, bool HAS_OMM, int STATS = 0, int NSTRIDE = 5, int PROBED = 0, int STEAL = 0, int MINW = 8, int TRI2 = 0>
__global__ __launch_bounds__(WG, (STEAL || TRI2 || (SPEC && PROBED == 2)) ? MINW : 1) void k_cwbvh(const float4* __restrict__ nodes, const float4* __restrict__ tris,
                                              QueryArgs q, uint32_t* __restrict__ status) {
    __shared__ uint2 stk[LDS_N][WG];
    const uint32_t glane = blockIdx.x * WG + threadIdx.x;
    LaneStack<uint2, LDS_N, WG> st;
    st.init(&stk[0][threadIdx.x], (uint2*)q.spill + glane, gridDim.x * WG, q.spillStride);
    RayPool<64> pool;
    const uint64_t nRaysTotal = q.nRaysDev ? *q.nRaysDev : q.nRays;   // batch size may live on the device (wavefront queues)
    pool.init(q.poolParts, q.counterNext);
    LockstepGovernor gov;
    gov.init();
    if (PROBED == 2) gov.lockstep = 0u;   // the probe has said the batch is incoherent: no lockstep generation to find that out again (4 M bounce rays +1 %)
    // PROBED: the batch's coherence probe (QueryArgs::probe) picks the schedule for the whole launch: coherent batches (camera rays, shadow rays
    // towards one light) are VALU-bound and run deferred triangles + a gated triangle phase on a third more waves; incoherent ones are bound by
    // the cache-miss path and keep the strict schedule.  PROBED == 1: this kernel holds both schedules (waves beyond q.baseBlocks leave at once
    // when the batch is incoherent; with baseBlocks == 0 the kernel serves coherent batches only).  PROBED == 2: the INCOHERENT flavor of a probed
    // launch (capi.hip launches the two back to back, the one the verdict is not for costs ~10 us): strict schedule, finds the ray pool dry when the
    // coherent flavor ahead of it took the batch, and reads what an incoherent batch is bound by in its cheapest form — the priority-ordered node copy whose deep nodes have a
    // line each (NSTRIDE = kNodeHybrid), triangle records padded to 64 bytes (none straddles a line), ray records with the non-temporal hint
    // (read once by one CU: they should not displace tree lines in the L2s).
    bool coh = false;
    if (PROBED && q.probe) {
        // The probe runs IN the traversal kernel (round 4; until then a 16-workgroup launch of its own ahead of the traversal kernels: one more
        // launch latency per query on the host's critical path): every wave compares the directions of the same kProbePairs neighbouring ray
        // pairs spread over the batch — 8 loads per lane from 32 KB that the first waves leave in the L2s — without a counter to wait on.  The
        // coherent flavor (launched first) publishes block 0's two counts for tbvh_debug_last_probe.
        // NOT launch-uniform in one case, and nothing may depend on it being so: a closest-hit launch that is not `fresh` reads the rays' own
        // hit.x as their reach while earlier waves of the same launch are already writing hit distances there — a pair that counted by its
        // direction alone (reach 1e30) can fail the origin test once one of its rays has been hit, so a late wave may vote "incoherent" where an
        // early one voted "coherent" (parallel rays from scattered origins: an orthographic camera, sun shadow rays traced with Intersect).
        // The verdict therefore only ever picks a SCHEDULE; which rays get traced is the ray pool's business alone: a wave of the coherent
        // flavor that votes "incoherent" leaves its stripe's remaining chunks in the pool, and the incoherent flavor behind it in the stream
        // (PROBED == 2) never looks at the verdict — it draws from the same counters and finds them dry (one atomic per wave) exactly when the
        // coherent flavor traced everything.
        uint32_t agree = 0, pairs = 0;
        if (PROBED != 2) {
            coherence_sample(q.rays, nRaysTotal, q.fresh != 0u, q.freshTmax, agree, pairs);
            if (blockIdx.x == 0 && threadIdx.x == 0) { q.probe[0] = agree; q.probe[1] = pairs; }
        }
        coh = (pairs != 0 && agree * 10u >= pairs * 6u) || (q.flags & 16u) != 0;   // (flag 16: tbvh_set_variant 91 forces the coherent verdict — tests)
        if (PROBED == 3 || PROBED == 4) { if (!coh) return; }   // the coherent flavor of a two-kernel launch: 3 = deferred + gated schedule only (SPEC = true); 4 = the strict
                                                                // schedule for coherent batches too (scenes where deferral loses: the per-scene tuner of capi_query.hip picks)
        else if (PROBED == 1 && !coh && blockIdx.x >= q.baseBlocks) return;   // (the surplus waves of a one-kernel launch: the base grid covers every stripe)
    }
    const uint32_t hybridK = q.hybridK;
#ifdef TBVH_EXPERIMENTS
    const bool ntRays = PROBED == 2 || (q.flags & 1u) != 0, tri64 = PROBED == 2 || (q.flags & 2u) != 0;   // (round 3's A/B switches: debug flags 1 and 2)
#else
    constexpr bool ntRays = PROBED == 2, tri64 = PROBED == 2;
#endif

    bool active = false;
    uint64_t ri = 0;
    float3 O = make_float3(0, 0, 0), D = O, rD = O;
    float4 hit = make_float4(0, 0, 0, 0);
    bool found = false;
    bool negX = false, negY = false, negZ = false;   // rD.x < 0 ...: per ray, kept in scalar lane masks (cw_test_node)
    uint32_t oct = 0, octinv4 = 0;
    uint2 ng = make_uint2(0u, 0u), tg = make_uint2(0u, 0u), tg2 = make_uint2(0u, 0u);
    uint32_t tgn = 0, tgn2 = 0;   // hybrid node copy: where tg's (tg2's) node lives (its line may hold one of its triangles: k_derive_hybrid)
    // STEAL > 0 (idle lanes needed): once the ray pool is dry, idle lanes take pending subtrees off the lanes that still traverse (ray_split.h)
    __shared__ SplitLds<STEAL ? WG : 1> split;
    int grp = -1;
    unsigned long long sIter = 0, sActive = 0, sNodeIter = 0, sNode = 0, sTriIter = 0, sTri = 0, sRefill = 0, sRefilled = 0;  // STATS only
    const unsigned long long tStart = STATS >= 2 ? wall_clock64() : 0ull;   // STATS == 2: wave timeline, as in kernels_query.hip (bvh4_body)
    unsigned long long tDry = 0ull;

    for (;;) {
        // ---- ray replacement -------------------------------------------------------------
        const uint32_t nIdle = wave_count(!active);
        if (gov.want_refill(nIdle, (uint32_t)REFILL_MIN) || nIdle == (uint32_t)WG) {
            if (!pool.dry()) {
                uint64_t nri = 0;
                const bool got = pool.acquire(!active, q.counter, nRaysTotal, nri);
                if (got) {
                    ri = nri;
                    const RayRec* rp = q.rays + ri;
                    if (ntRays) {   // (experiment) a ray record is read once by one CU: keep it from displacing tree lines in the L2
                        const tbvh_f4* r4 = (const tbvh_f4*)rp;
                        const tbvh_f4 a = __builtin_nontemporal_load(r4), b = __builtin_nontemporal_load(r4 + 1), c = __builtin_nontemporal_load(r4 + 2);
                        O = make_float3(a.x, a.y, a.z); D = make_float3(b.x, b.y, b.z); rD = make_float3(c.x, c.y, c.z);
                    } else { O = xyz(rp->O); D = xyz(rp->D); rD = xyz(rp->rD); }
                    hit = q.fresh ? make_float4(q.freshTmax, 0.f, 0.f, 0.f) : rp->hit;
                    found = false;
                    oct = cw_oct(D);
                    octinv4 = oct * 0x01010101u;
                    negX = rD.x < 0; negY = rD.y < 0; negZ = rD.z < 0;
                    ng = make_uint2(0u, 0x80000000u); tg = make_uint2(0u, 0u); tg2 = make_uint2(0u, 0u);
                    st.reset();
                    active = true;
                }
            }
            if (STATS >= 2 && !tDry && pool.dry()) tDry = wall_clock64();
            if (wave_ballot(active) == 0) break;
        }
        const bool tail = STEAL && pool.dry();   // wave-uniform: nothing of the split-ray code costs a vector instruction before the pool is dry
        if (tail && nIdle >= (uint32_t)STEAL) {
            const uint32_t nKids = (uint32_t)__popc(ng.y >> 24);
            SplitMatch m;
            if (split_match(active && (!st.empty() || nKids >= 2u), !active, m)) {
                uint2 part = make_uint2(0u, 0u);
                if (m.gives) {
                    // the NEAREST pending subtree — the one the donor would have entered next, so the chain of dependent steps to the closest
                    // hit gets shorter (the farthest is mostly work a closer hit would have culled) —, or, with an empty stack, every second
                    // pending child of the current node, front to back (the front-most is the highest bit, cw_next_child)
                    if (!st.empty()) part = st.pop();
                    else {
                        uint32_t bits = ng.y >> 24, give = 0u, odd = 0u;
                        while (bits) { const uint32_t hb = 0x80000000u >> __clz(bits); give |= hb & (0u - odd); odd ^= 1u; bits ^= hb; }
                        part = make_uint2(ng.x, (give << 24) | (ng.y & 0x00FFFFFFu));
                        ng.y &= ~(give << 24);
                    }
                    split_give<ANYHIT>(split, m, grp, found, hit);
                }
                const int src = split_take_ray(split, m, O, D, rD, hit, ri, grp);
                part.x = __shfl(part.x, src); part.y = __shfl(part.y, src);
                if (STATS == 5) sRefill += __popcll(__ballot(m.takes));
                if (m.takes) {
                    found = false;
                    oct = cw_oct(D); octinv4 = oct * 0x01010101u;
                    negX = rD.x < 0; negY = rD.y < 0; negZ = rD.z < 0;
                    ng = part; tg = make_uint2(0u, 0u); tg2 = make_uint2(0u, 0u);
                    st.reset();
                    active = true;
                }
            }
        }
        if (STATS == 1) { sIter++; sActive += __popcll(__ballot(active)); }
        if (STATS == 5 && pool.dry()) { sIter++; sActive += __popcll(__ballot(active)); }   // the wave's tail: passes after the pool ran dry, lanes still busy in them
        if (!active) continue;

        bool done = false;
        if (tail && grp >= 0) split_poll<ANYHIT>(split, grp, hit, done);   // a split ray: bounded by its group's closest hit
        // PREF (TRI2 == 2, experiment): the node a lane will visit in THIS pass is known before its triangle test — the next pending child, or the top of
        // the stack; a triangle hit changes what the node's children are tested against, not which node is fetched — so its five loads go out ahead of
        // the triangle phase and the two memory round trips of a pass overlap (strict schedule only: a lane takes its node step iff at most one triangle
        // of its group is left).
        constexpr bool PREF = TRI2 == 2 && !SPEC && PROBED != 1;
        bool preHave = false;
        CwNode preNode;
        uint32_t preCi = 0;
        if (PREF && !done && __popc(tg.y) <= 1) {
            bool have = cw_has_child(ng);
            if (!have && !st.empty()) { ng = st.pop(); have = true; }
            if (have) {
                preCi = cw_next_child(ng, oct);
                if (cw_has_child(ng)) st.push(ng);
                preNode = cw_load_node<NSTRIDE>(nodes, preCi, hybridK);
                preHave = true;
            }
        }
        // ---- triangle phase: runs when enough lanes have a triangle pending, or when no lane could use a node
        // phase instead (so a waiting lane always makes progress) --------------------------------------------
        const bool spec = PROBED == 1 ? coh : SPEC;   // (PROBED == 3 is launched with SPEC = true)
        bool triPhase = true;
        if (TRI_MIN > 1 && (PROBED == 1 ? coh : true)) {
            const uint32_t nPend = wave_count(tg.y != 0);
            const bool canNode = spec ? (tg2.y == 0 && (cw_has_child(ng) || !st.empty())) : tg.y == 0;
            triPhase = nPend >= (uint32_t)TRI_MIN || wave_ballot(canNode) == 0;
        }
        if (triPhase && tg.y != 0 && !(STEAL && ANYHIT && done)) {
            if (STATS == 1) { const unsigned long long m = __ballot(true); if (lane_rank(m) == 0) { sTriIter++; sTri += __popcll(m); } }
            const uint32_t ti = 31u - (uint32_t)__clz(tg.y);
            tg.y &= ~(1u << ti);
            auto record = [&](uint32_t k) -> const float4* {
                if (NSTRIDE == kNodeHybrid)   // the hybrid copy's triangle word: embedded << 27 | first 64-byte record; the embedded triangle sits in the node's own line
                    return k == (tg.x >> 27) ? nodes + ((size_t)tgn + 5u) : tris + ((size_t)(tg.x & 0x07FFFFFFu) + k) * 4u;
                return tris + (tri64 ? (size_t)((__umulhi(tg.x, 0xAAAAAAABu) >> 1) + k) * 4u : (size_t)tg.x + k * 3u);   // (experiment) records padded to 64 bytes: tg.x counts float4s of the packed array
            };
            const float4* tp = record(ti);
            const float4 e2 = tp[0], e1 = tp[1], v0 = tp[2];
            // TRI2 (experiment, round 5): a lane whose group holds a second triangle tests it in the SAME pass — all six loads out together, two tests
            // one after the other —, so a 2- or 3-triangle group costs the lane one pass less in which it sits out the node phase
            // (traverse_cwbvh.cl:289-327 tests a group's triangles in one loop).  Front-most bit first, as the mirror does; the tie rule makes the
            // order immaterial for the record.
            // (gate, flags bits 20..23: only when at least that many lanes of the wave hold a second triangle — the second test is issued for the
            // whole wave whenever ONE lane wants it)
            const bool two = TRI2 == 1 && tg.y != 0 && (((q.flags >> 20) & 15u) == 0u || wave_count(tg.y != 0) >= ((q.flags >> 20) & 15u));
            float4 f2 = make_float4(0, 0, 0, 0), f1 = f2, w0 = f2;
            if (two) {
                const uint32_t tj = 31u - (uint32_t)__clz(tg.y);
                tg.y &= ~(1u << tj);
                const float4* tq = record(tj);
                f2 = tq[0]; f1 = tq[1]; w0 = tq[2];
            }
            // The strict schedule: all three loads of the record issue together.  Left alone the compiler sinks v0's load into the branch behind the
            // determinant test — one load fewer for a lane whose triangle is edge-on, a second memory round trip for every other one.  Bounce rays
            // +7 % on the Bistro and Sponza stand-ins, +7-10 % on 12 M triangles, camera and shadow rays +-1 % (profiles/r04_ab_triangle_loads_together.txt).
            // The deferred + gated schedule keeps the lazy load: -2 % with the loads together (its triangle phases are full of L2 hits, the saved
            // registers are worth more).
            if ((!SPEC || PROBED == 2) && PROBED != 1) { tri_loads_together(v0); if (TRI2 == 1) tri_loads_together(w0); }
            TriHit h;
            if (tri_test(O, D, xyz(v0), xyz(e1), xyz(e2), hit.x, h, HAS_OMM ? q.omm : Omm{nullptr, 0}, as_u32(v0.w)) &&
                (ANYHIT || (tail && grp >= 0) || hit_wins(h.t, as_u32(v0.w), found, hit))) {   // (a split ray's group arbitrates: split_publish)
                found = true;
                if (ANYHIT) done = true;
                else hit = make_float4(h.t, h.u, h.v, v0.w);
                if (tail && grp >= 0) split_publish<ANYHIT>(split, grp, hit);
            }
            if (TRI2 == 1 && two && !(ANYHIT && done)) {
                if (tri_test(O, D, xyz(w0), xyz(f1), xyz(f2), hit.x, h, HAS_OMM ? q.omm : Omm{nullptr, 0}, as_u32(w0.w)) &&
                    (ANYHIT || (tail && grp >= 0) || hit_wins(h.t, as_u32(w0.w), found, hit))) {
                    found = true;
                    if (ANYHIT) done = true;
                    else hit = make_float4(h.t, h.u, h.v, w0.w);
                    if (tail && grp >= 0) split_publish<ANYHIT>(split, grp, hit);
                }
            }
            if ((SPEC || PROBED == 1) && tg.y == 0) { tg = tg2; tg2 = make_uint2(0u, 0u); if (NSTRIDE == kNodeHybrid) tgn = tgn2; }
        }
        // ---- node phase ---------------------------------------------------------------------------------------
        if (PREF) {
            if (!done && tg.y == 0) {
                if (!preHave) done = true;
                else {
                    const CwNodeHits r = cw_test_node(preNode, O, rD, cull_bound(hit.x), octinv4, negX, negY, negZ);
                    ng = make_uint2(r.childBase, (r.hitmask & 0xFF000000u) | r.imask);
                    tg = make_uint2(r.triBase, r.hitmask & 0x00FFFFFFu);
                    if (NSTRIDE == kNodeHybrid) tgn = cw_hybrid_offset(preCi, hybridK);
                }
            }
        } else
        if (!done && (spec ? tg2.y == 0 : tg.y == 0)) {
            bool have = cw_has_child(ng);
            if (!have) {
                if (!st.empty()) { ng = st.pop(); have = true; }   // only node groups with children pending are ever pushed
                else if (tg.y == 0) done = true;
            }
            if (have) {
                const uint32_t ci = cw_next_child(ng, oct);
                if (STATS == 1) {
                    const unsigned long long m = __ballot(true);
                    const bool uni = __ballot(ci != (uint32_t)__builtin_amdgcn_readfirstlane(ci) || oct != (uint32_t)__builtin_amdgcn_readfirstlane(oct)) == 0;
                    if (lane_rank(m) == 0) { sNodeIter++; sNode += __popcll(m); if (uni) { sRefill++; sRefilled += __popcll(m); } }   // [5], [6]: uniform node phases, lanes in them
                }
                if (cw_has_child(ng)) st.push(ng);
                const CwNodeHits r = cw_test_node(cw_load_node<NSTRIDE>(nodes, ci, hybridK), O, rD, cull_bound(hit.x), octinv4, negX, negY, negZ);
                ng = make_uint2(r.childBase, (r.hitmask & 0xFF000000u) | r.imask);
                const uint2 nt = make_uint2(r.triBase, r.hitmask & 0x00FFFFFFu);
                if (tg.y == 0) { tg = nt; if (NSTRIDE == kNodeHybrid) tgn = cw_hybrid_offset(ci, hybridK); }
                else { tg2 = nt; if (NSTRIDE == kNodeHybrid) tgn2 = cw_hybrid_offset(ci, hybridK); }
            }
        }
        if (done) {
            if (tail && grp >= 0) split_finish<ANYHIT>(split, grp, q, ri);
            else if (ANYHIT) q.occluded[ri] = found ? 1 : 0;
            else if (found || q.fresh) {
                if (ntRays) { tbvh_f4 hv; hv.x = hit.x; hv.y = hit.y; hv.z = hit.z; hv.w = hit.w; __builtin_nontemporal_store(hv, (tbvh_f4*)&q.rays[ri].hit); }
                else q.rays[ri].hit = hit;
            }
            active = false;
        }
    }
    if (st.overflow) atomicOr(status, 1u);
    if (STATS == 2 && threadIdx.x == 0 && (blockIdx.x % 31u) == 0) {
        const unsigned long long tEnd = wall_clock64();
        atomicMax(q.stats + 0, ~tStart); atomicMax(q.stats + 1, tStart); atomicAdd(q.stats + 2, tStart);
        atomicMax(q.stats + 3, ~tEnd); atomicMax(q.stats + 4, tEnd); atomicAdd(q.stats + 5, tEnd);
        atomicAdd(q.stats + 6, tDry ? tDry : tEnd); atomicAdd(q.stats + 7, 1ull);
    }
    if (STATS == 5 && threadIdx.x == 0) {   // [0] the longest tail: passes << 32 | busy lane-passes, [1] passes, [2] busy lane-passes, [3] waves, [4] subtrees taken over
        atomicMax(q.stats + 0, (sIter << 32) | (sActive & 0xFFFFFFFFull)); atomicAdd(q.stats + 1, sIter); atomicAdd(q.stats + 2, sActive);
        atomicAdd(q.stats + 3, 1ull); atomicAdd(q.stats + 4, sRefill);
    }
    if ((STATS == 3 || STATS == 4) && threadIdx.x == 0 && (blockIdx.x & 3u) == 1u) {   // histograms over every 4th wave, 64 us bins: 3 = wave ends, 4 = pool dry
        const unsigned long long t = (STATS == 3 ? wall_clock64() : (tDry ? tDry : wall_clock64())) - tStart;
        const unsigned long long b = t / 6400ull;
        atomicAdd(q.stats + (b < 7ull ? b : 7ull), 1ull);
    }
    if (STATS == 1) {
        // the per-phase counters were kept by the first lane of each phase: reduce over the wave
        for (int o = 32; o > 0; o >>= 1) {
            sTriIter += __shfl_xor(sTriIter, o); sTri += __shfl_xor(sTri, o); sNodeIter += __shfl_xor(sNodeIter, o); sNode += __shfl_xor(sNode, o);
            sRefill += __shfl_xor(sRefill, o); sRefilled += __shfl_xor(sRefilled, o);
        }
        if (threadIdx.x == 0) {
            atomicAdd(q.stats + 0, sIter); atomicAdd(q.stats + 1, sActive); atomicAdd(q.stats + 2, sNode); atomicAdd(q.stats + 3, sTriIter);
            atomicAdd(q.stats + 4, sTri); atomicAdd(q.stats + 5, sRefill); atomicAdd(q.stats + 6, sRefilled); atomicAdd(q.stats + 7, sNodeIter);
        }
    }
}

template <bool ANYHIT, int LDS_N, int REFILL_MIN, int TRI_MIN, bool SPEC, int STATS = 0, int NSTRIDE = 5, int PROBED = 0, int STEAL = 0, int MINW = 8, int TRI2 = 0>
void launch_k(const float4* nodes, const float4* tris, const QueryArgs& q, uint32_t* status, uint32_t blocks, hipStream_t s) {
    // without opacity micromaps on the scene the check is compiled out (+1-2 %)
    if (q.omm.map) hipLaunchKernelGGL((k_cwbvh<ANYHIT, LDS_N, REFILL_MIN, TRI_MIN, SPEC, true, STATS, NSTRIDE, PROBED, STEAL, MINW, TRI2>), dim3(blocks), dim3(WG), 0, s, nodes, tris, q, status);
    else hipLaunchKernelGGL((k_cwbvh<ANYHIT, LDS_N, REFILL_MIN, TRI_MIN, SPEC, false, STATS, NSTRIDE, PROBED, STEAL, MINW, TRI2>), dim3(blocks), dim3(WG), 0, s, nodes, tris, q, status);
}

__global__ void k_pad_nodes(const float4* __restrict__ src, float4* __restrict__ dst, uint32_t nNodes) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;   // one thread per float4 of the padded array
    if (i >= nNodes * 8u) return;
    const uint32_t n = i >> 3, k = i & 7u;
    dst[i] = k < 5u ? src[n * 5u + k] : make_float4(0.f, 0.f, 0.f, 0.f);
}

// the hybrid node array (cwbvh_node.h: kNodeHybrid) from the packed one: node i goes to position perm[i], its childBaseIndex follows its
// first child (children stay consecutive in slot order under the priority order).  Round 4: a node on a line of its own has 48 spare bytes —
// exactly one triangle record.  The first triangle of its leaf child with the LARGEST box (the child a ray that visits the node is most likely
// to hit) is copied there, and the node's triangle word becomes  embedded << 27 | first 64-byte record  (embedded = the triangle's index
// relative to the node, kNoEmbedded = none): the traversal reads that triangle from the node's own line — fetched a moment ago — instead of a
// line of the triangle array.  tools/line_model.py (the oracle's mirror on a bounce batch): 32 % of all triangle tests, 2.1 of 36.5 line
// fetches per ray; nearly all of them lines from beyond the L2s.
__global__ void k_derive_hybrid(const float4* __restrict__ src, const uint32_t* __restrict__ perm, float4* __restrict__ dst, uint32_t nNodes, uint32_t hybridK,
                                const float4* __restrict__ tris) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nNodes) return;
    const uint32_t ni = perm ? perm[i] : i;   // (no permutation: trees made on the device are in level order already)
    if (ni >= nNodes) return;                 // (a blob whose child ranges are not a tree: capi_scene.hip refuses the copies; belt and braces)
    float4* o = dst + ((size_t)ni * 8u - (size_t)(ni < hybridK ? ni : hybridK) * 3u);
    const float4* p = src + (size_t)i * 5u;
    const float4 n0 = p[0], n2 = p[2], n3 = p[3], n4 = p[4];
    float4 n1 = p[1];
    if (as_u32(n0.w) >> 24) { const uint32_t cb = as_u32(n1.x); n1.x = as_f32(cb < nNodes ? (perm ? perm[cb] : cb) : 0u); }
    uint32_t emb = kNoEmbedded;
    if (ni >= hybridK && tris) {
        const uint32_t ew = as_u32(n0.w);
        const float sx = ldexpf(1.f, (int)(int8_t)ew), sy = ldexpf(1.f, (int)(int8_t)(ew >> 8)), sz = ldexpf(1.f, (int)(int8_t)(ew >> 16));
        float best = -1.f;
        for (uint32_t j = 0; j < 8; j++) {
            const uint32_t sh = 8u * (j & 3u), hi = j >> 2;
            const uint32_t meta = ((hi ? as_u32(n1.w) : as_u32(n1.z)) >> sh) & 255u;
            if (meta == 0u || (meta & 0x18u) == 0x18u) continue;    // empty slot / interior child (0b001sssss, sssss = 24 + slot)
            const float dx = (float)((int)(((hi ? as_u32(n3.w) : as_u32(n3.z)) >> sh) & 255u) - (int)(((hi ? as_u32(n2.y) : as_u32(n2.x)) >> sh) & 255u)) * sx;
            const float dy = (float)((int)(((hi ? as_u32(n4.y) : as_u32(n4.x)) >> sh) & 255u) - (int)(((hi ? as_u32(n2.w) : as_u32(n2.z)) >> sh) & 255u)) * sy;
            const float dz = (float)((int)(((hi ? as_u32(n4.w) : as_u32(n4.z)) >> sh) & 255u) - (int)(((hi ? as_u32(n3.y) : as_u32(n3.x)) >> sh) & 255u)) * sz;
            const float area = dx * dy + dy * dz + dz * dx;
            if (area > best) { best = area; emb = meta & 31u; }
        }
    }
    const uint32_t triBase = as_u32(n1.y);
    if (emb != kNoEmbedded) {
        const float4* t = tris + (size_t)triBase + (size_t)emb * 3u;
        o[5] = t[0]; o[6] = t[1]; o[7] = t[2];
    }
    n1.y = as_f32((emb << 27) | (triBase / 3u));
    o[0] = n0; o[1] = n1; o[2] = n2; o[3] = n3; o[4] = n4;
}

}  // namespace

void launch_cwbvh_derive_hybrid(const float4* src, const uint32_t* perm, float4* dst, uint32_t nNodes, uint32_t hybridK, const float4* tris, hipStream_t s) {
    hipLaunchKernelGGL(k_derive_hybrid, dim3((nNodes + 255u) / 256u), dim3(256), 0, s, src, perm, dst, nNodes, hybridK, tris);
}

// ---- which instantiation serves a launch: a table, first match wins ----------------------------------------------------------------------
// template arguments of launch_k after ANYHIT:  LDS_N, REFILL_MIN, TRI_MIN, SPEC, STATS, NSTRIDE, PROBED, STEAL, MINW, TRI2
namespace {
struct LaunchSel {
    int nodeStride;        // 5 packed as uploaded, 8 one node per line (padCwbvhIfLarge), kNodeHybrid = the incoherent-batch copies
    bool tail;             // split the last rays over idle lanes (batches below 12 M rays, device-side ray counts)
    bool probed;           // the launch carries a coherence probe
    bool firstOfTwo;       // ... and is the coherent flavor of a two-kernel launch (baseBlocks == 0)
    bool strictFirst;      // ... which the scene's tuner wants on the strict schedule (flag 32)
    bool shallow;          // scene under 48 MB
    uint32_t expFlags;     // experiment builds: QueryArgs::flags
};
typedef void (*LaunchFn)(bool anyhit, const float4* nodes, const float4* tris, const QueryArgs& q, uint32_t* status, uint32_t blocks, hipStream_t s);
template <auto... A> void launch_both(bool anyhit, const float4* nodes, const float4* tris, const QueryArgs& q, uint32_t* status, uint32_t blocks, hipStream_t s) {
    if (anyhit) launch_k<true, A...>(nodes, tris, q, status, blocks, s);
    else launch_k<false, A...>(nodes, tris, q, status, blocks, s);
}
struct LaunchRow {
    bool (*when)(const LaunchSel&);
    LaunchFn fn;
    bool cap28;            // built for 7 waves per SIMD (72 VGPRs): at most 28 one-wave workgroups per CU
};
// Stack entries in LDS next to the split groups: 6 let more than 24 waves per CU fit (what scenes under 48 MB and coherent probed batches are
// launched with; Bistro-size trees measure the same with 6 or 8), deep trees want 8 (30 M triangles: 6 costs 7 % on camera rays, 13 % on bounce rays).
// Those 6-entry kernels with split rays are built for 7 waves per SIMD (72 VGPRs) and launched 28 per CU: under the budget of 8 (64 VGPRs) they
// spill 12-24 bytes per lane inside the loop (round 2 shipped that: 4 M bounce rays 2420 -> 2810 MRays/s without the spill).
// Split rays (ray_split.h): Bistro stand-in 0.26 / 1 / 4 / 8 M rays: camera +7 / +20 / +6 / +4 %, bounce +26 / +23 / +8 / +4 %, shadow +20 / +18 / +6 / +4 %;
// at 16.7 M rays the tail is 5 % of the launch and the kernel's register cap costs as much as it gains.
const LaunchRow kLaunchTable[] = {
    // one node per cache line (scenes whose node array is beyond the Infinity Cache; DESIGN.md par. 5: -17 % bytes, +6 % at 60 M triangles)
    {[](const LaunchSel& x) { return x.nodeStride == 8 && x.tail; },  &launch_both<8, 16, 1, false, 0, 8, 0, 16>, false},
    {[](const LaunchSel& x) { return x.nodeStride == 8; },            &launch_both<8, 16, 1, false, 0, 8>, false},
#ifdef TBVH_EXPERIMENTS
    // round 5: up to two triangle tests per pass (debug flag 0x10000): -13 % on bounce rays (profiles/r05_ab_tri2.txt), not shipped
    {[](const LaunchSel& x) { return x.nodeStride == kNodeHybrid && (x.expFlags & 0x10000u) && x.tail; }, &launch_both<8, 16, 1, false, 0, kNodeHybrid, 2, 16, 6, 1>, false},
    {[](const LaunchSel& x) { return x.nodeStride == kNodeHybrid && (x.expFlags & 0x10000u) != 0; },      &launch_both<8, 16, 1, false, 0, kNodeHybrid, 2, 0, 6, 1>, false},
    // round 6: the ray-replacement threshold of the incoherent flavor (debug flags 0x20000: 8 idle lanes, 0x40000: 32; shipped: 16) — profiles/r06_diffuse.txt
    // (round 6, late) node loads ahead of the triangle phase (TRI2 == 2), at the register budgets of 8 / 7 / 6 waves per SIMD
    {[](const LaunchSel& x) { return x.nodeStride == kNodeHybrid && (x.expFlags & 0x80000u) && !x.tail; }, &launch_both<8, 16, 1, false, 0, kNodeHybrid, 2, 0, 8, 2>, false},
    {[](const LaunchSel& x) { return x.nodeStride == kNodeHybrid && (x.expFlags & 0x1000000u) && !x.tail; }, &launch_both<8, 16, 1, false, 0, kNodeHybrid, 2, 0, 7, 2>, false},
    {[](const LaunchSel& x) { return x.nodeStride == kNodeHybrid && (x.expFlags & 0x2000000u) && !x.tail; }, &launch_both<8, 16, 1, false, 0, kNodeHybrid, 2, 0, 6, 2>, false},
    // (round 6, late) deferred triangles in the incoherent flavor, triangle phase gated at 1 / 4 / 8 / 16 lanes
    {[](const LaunchSel& x) { return x.nodeStride == kNodeHybrid && (x.expFlags & 0x100000u) && !x.tail; }, &launch_both<8, 16, 1, true, 0, kNodeHybrid, 2, 0, 7>, false},
    {[](const LaunchSel& x) { return x.nodeStride == kNodeHybrid && (x.expFlags & 0x200000u) && !x.tail; }, &launch_both<8, 16, 4, true, 0, kNodeHybrid, 2, 0, 7>, false},
    {[](const LaunchSel& x) { return x.nodeStride == kNodeHybrid && (x.expFlags & 0x400000u) && !x.tail; }, &launch_both<8, 16, 8, true, 0, kNodeHybrid, 2, 0, 7>, false},
    {[](const LaunchSel& x) { return x.nodeStride == kNodeHybrid && (x.expFlags & 0x800000u) && !x.tail; }, &launch_both<8, 16, 16, true, 0, kNodeHybrid, 2, 0, 7>, false},
    {[](const LaunchSel& x) { return x.nodeStride == kNodeHybrid && (x.expFlags & 0x20000u) && !x.tail; }, &launch_both<8, 8, 1, false, 0, kNodeHybrid, 2>, false},
    {[](const LaunchSel& x) { return x.nodeStride == kNodeHybrid && (x.expFlags & 0x40000u) && !x.tail; }, &launch_both<8, 32, 1, false, 0, kNodeHybrid, 2>, false},
#endif
    // the incoherent flavor of a probed launch: `nodes` = the hybrid copy, `tris` = the 64-byte records (cwbvh_node.h, capi_query.hip);
    // with split rays built for 6 waves per SIMD (80 VGPRs; left alone the compiler takes 83, one wave per SIMD fewer)
    {[](const LaunchSel& x) { return x.nodeStride == kNodeHybrid && x.tail; }, &launch_both<8, 16, 1, false, 0, kNodeHybrid, 2, 16, 6>, false},
    {[](const LaunchSel& x) { return x.nodeStride == kNodeHybrid; },           &launch_both<8, 16, 1, false, 0, kNodeHybrid, 2>, false},
    // the coherent flavor of a two-kernel probed launch on the STRICT schedule (PROBED == 4): scenes on which the deferred schedule measured
    // slower (the online tuner of capi_query.hip)
    // ... the same for scenes under 48 MB (round 6: they are probed from 768 k rays on, their coherent flavor is this or the packet kernel): the shapes
    // of their unprobed kernels below (6 stack entries in LDS next to the split groups, 7 waves per SIMD)
    {[](const LaunchSel& x) { return x.firstOfTwo && x.strictFirst && x.tail && x.shallow; }, &launch_both<6, 16, 1, false, 0, 5, 4, 16, 7>, true},
    {[](const LaunchSel& x) { return x.firstOfTwo && x.strictFirst && x.tail; }, &launch_both<8, 16, 1, false, 0, 5, 4, 16>, false},
    {[](const LaunchSel& x) { return x.firstOfTwo && x.strictFirst; },           &launch_both<8, 16, 1, false, 0, 5, 4>, false},
    // ... on the deferred + gated schedule (PROBED == 3): no strict path compiled in (camera rays +1.5 %)
    {[](const LaunchSel& x) { return x.firstOfTwo && x.tail; }, &launch_both<6, 16, 8, true, 0, 5, 3, 16, 7>, true},
    {[](const LaunchSel& x) { return x.firstOfTwo; },           &launch_both<8, 16, 8, true, 0, 5, 3>, false},
    // one kernel for both verdicts (scenes without the incoherent-batch copies)
    {[](const LaunchSel& x) { return x.probed && x.tail; }, &launch_both<6, 16, 8, true, 0, 5, 1, 16, 7>, true},
    {[](const LaunchSel& x) { return x.probed; },           &launch_both<8, 16, 8, true, 0, 5, 1>, false},
    // no probe (small batches, small or very large scenes): the strict schedule
    {[](const LaunchSel& x) { return x.tail && x.shallow; }, &launch_both<6, 16, 1, false, 0, 5, 0, 16, 7>, true},
    {[](const LaunchSel& x) { return x.tail; },              &launch_both<8, 16, 1, false, 0, 5, 0, 16>, false},
    {[](const LaunchSel&) { return true; },                  &launch_both<8, 16, 1, false>, false},
};
}  // namespace

void launch_cwbvh(bool anyhit, int variant, const float4* nodes, const float4* tris, const QueryArgs& q, uint32_t* status,
                  uint32_t blocks, hipStream_t s, int nodeStride, bool shallow, uint32_t blocks7) {
    const uint32_t capped = blocks > blocks7 ? blocks7 : blocks;
    // forced schedules (tbvh_set_variant; tests/test_cwbvh_schedules.py, tools/ab_probe.py): kernels the library ships anyway, picked whatever the
    // batch size or the probe says
    switch (variant) {
    case 72: launch_both<8, 16, 1, false>(anyhit, nodes, tris, q, status, blocks, s); return;                  // the strict schedule
    case 75: launch_both<8, 16, 1, false, 0, 5, 0, 16>(anyhit, nodes, tris, q, status, blocks, s); return;     // strict + split rays whatever the batch size
    case 88:                                                                                                   // the probed schedule + split rays whatever the batch size
        if (q.probe) launch_both<6, 16, 8, true, 0, 5, 1, 16, 7>(anyhit, nodes, tris, q, status, capped, s);
        else launch_both<6, 16, 1, false, 0, 5, 0, 16, 7>(anyhit, nodes, tris, q, status, capped, s);
        return;
#ifdef TBVH_EXPERIMENTS
    // diagnostic kernels, only in the experiment build (make EXPERIMENTS=1 -> libtinybvh_amd_exp.so; TBVH_LIB_OVERRIDE points the tools at it): a schedule
    // that is not shipped under this template signature, and the instrumented kernels behind the counters of DESIGN.md par. 5
    case 52: launch_both<8, 16, 8, true>(anyhit, nodes, tris, q, status, blocks, s); return;      // the coherent schedule without the probe: deferred triangles, triangle phase once 8 lanes wait
    case 89: if (q.probe) { launch_both<6, 16, 8, true, 0, 5, 1, 16, 8>(anyhit, nodes, tris, q, status, capped, s); return; } break;   // round 2's shipped form of the probed schedule + split rays: 64 VGPRs, 20 bytes of scratch per lane
    case 59: if (anyhit) break; launch_k<false, 8, 16, 1, false, 1>(nodes, tris, q, status, blocks, s); return;    // lane statistics of the strict schedule (q.stats)
    case 61: if (anyhit) break; launch_k<false, 8, 16, 8, true, 1>(nodes, tris, q, status, blocks, s); return;     // ... of the coherent schedule
    case 73: if (anyhit) break; launch_k<false, 8, 16, 1, false, 2>(nodes, tris, q, status, blocks, s); return;    // wave timeline of the strict schedule
    case 78: if (anyhit) break; launch_k<false, 8, 16, 1, false, 2, 5, 0, 16>(nodes, tris, q, status, blocks, s); return;   // ... with split rays
    case 82: if (anyhit) break; launch_k<false, 8, 16, 1, false, 5>(nodes, tris, q, status, blocks, s); return;    // tail statistics, strict schedule
    case 83: if (anyhit) break; launch_k<false, 8, 16, 1, false, 5, 5, 0, 16>(nodes, tris, q, status, blocks, s); return;   // ... with split rays
#endif
    default: break;
    }
    // with a coherence probe of the batch (capi_query.hip: launchQuery) the schedule is chosen per launch; without one, the strict schedule.
    // Batches below 12 M rays (and the wavefront stages, whose ray count only the device knows) also split their last rays over idle lanes.
    LaunchSel sel;
    sel.nodeStride = nodeStride; sel.tail = split_rays_wanted(q); sel.probed = q.probe != nullptr;
    sel.firstOfTwo = q.probe && q.baseBlocks == 0; sel.strictFirst = (q.flags & 32u) != 0; sel.shallow = shallow; sel.expFlags = q.flags;
    for (const LaunchRow& row : kLaunchTable)
        if (row.when(sel)) { row.fn(anyhit, nodes, tris, q, status, row.cap28 ? capped : blocks, s); return; }
}

namespace {
__global__ void k_pad_tris(const float4* __restrict__ src, float4* __restrict__ dst, uint64_t nTris) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per float4 of the padded array
    if (i >= nTris * 4u) return;
    const uint64_t t = i >> 2; const uint32_t k = (uint32_t)i & 3u;
    dst[i] = k < 3u ? src[t * 3u + k] : make_float4(0.f, 0.f, 0.f, 0.f);
}
}  // namespace
// 48-byte triangle records straddle a 128-byte line 3 times out of 8; at 64 bytes none does (+33 % triangle memory)
void launch_cwbvh_pad_tris(const float4* src, float4* dst, uint64_t nTris, hipStream_t s) {
    hipLaunchKernelGGL(k_pad_tris, dim3((uint32_t)((nTris * 4u + 255u) / 256u)), dim3(256), 0, s, src, dst, nTris);
}

// 80-byte nodes straddle 128-byte lines (1.6 lines per node on average); the padded copy costs 60 % more node memory
void launch_cwbvh_pad(const float4* src, float4* dst, uint32_t nNodes, hipStream_t s) {
    hipLaunchKernelGGL(k_pad_nodes, dim3((nNodes * 8u + 255u) / 256u), dim3(256), 0, s, src, dst, nNodes);
}

bool cwbvh_variant_valid(int v) {
#ifdef TBVH_EXPERIMENTS
    if (v == 52 || v == 89 || v == 59 || v == 61 || v == 73 || v == 78 || v == 82 || v == 83) return true;
#endif
    return v == 0 || v == 90 || v == 91 || v == 92 || v == 72 || v == 75 || v == 88;
}

}  // namespace tbvh
