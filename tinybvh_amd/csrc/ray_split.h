// Split rays: the end of a launch.  Once the ray pool is dry a persistent wave only finishes what its lanes hold, and what
// is left in the end are the longest rays — each a chain of dependent steps — on a few lanes while the others idle: a
// 1 M-ray launch on the Bistro stand-in has handed out its last ray after 250 us and ends after 500 us (wave timeline,
// tools/ab_probe.py --timeline / --hist).  From then on idle lanes take pending subtrees — the NEWEST stack entry, i.e.
// the nearest one: the chain of dependent steps to the closest hit is what has to get shorter — off the lanes that are
// still traversing and walk them for the same ray; takers give away in turn.
//
// The lanes working on one ray form a GROUP, named after the lane that gave work away first (a lane owns at most one
// ray after the pool is dry, so the slot is used once).  In LDS per group: the closest hit so far as one 64-bit key
// (t as an integer that orders like the float | prim), (u, v) of that hit, and the number of members still traversing.
// A member publishes a hit when it finds it; every member bounds its traversal by the group's key (closest hit) or
// stops when the key says "found" (any-hit); the last member to finish writes the ray's record.  Among hits at
// exactly equal t the smaller prim wins — the rule every lane applies on its own as well (device_common.h: hit_wins), so
// a split ray reports what one lane would have found.
#pragma once
#include "device_common.h"
#include "ray_pool.h"

namespace tbvh {

template <int N> struct SplitLds {
    static_assert(N == 64 || N == 1, "split groups assume ONE 64-lane wave per workgroup: threadIdx.x is the lane id, __syncthreads a wave barrier");
    unsigned long long best[N];
    float2 uv[N];
    uint32_t aux[N];       // two-level kernels: the instance of that hit
    uint32_t pending[N];
    uint8_t donorOf[N];    // scratch of one pairing step: the lane of the donor of each rank
};

// t as an integer that orders like the float (a hit at t = -0.0 must not lose against positive ones), then prim
template <bool ANYHIT> __device__ __forceinline__ unsigned long long split_key(float4 hit) {
    const uint32_t tb = as_u32(hit.x), tk = (tb & 0x80000000u) ? ~tb : (tb | 0x80000000u);
    return ANYHIT ? 0ull : (((unsigned long long)tk << 32) | as_u32(hit.w));
}
__device__ __forceinline__ float split_key_t(unsigned long long key) {   // (no hit yet: ~0 decodes to a NaN)
    const uint32_t tk = (uint32_t)(key >> 32);
    return as_f32((tk & 0x80000000u) ? (tk & 0x7FFFFFFFu) : ~tk);
}

// One pairing step, whole wave, wave-uniform control flow:
//   split_match     pairs idle lanes with lanes that can give work away, one to one by rank (false: nothing to pair);
//   split_give      (donors) enter or open the ray's group, leave the own lane id for the taker of the same rank;
//   split_take_ray  (all lanes) the donor's ray lands in the taker's registers; returns the lane a taker copies from, so the
//                   kernel can fetch its own state (the stack entry given away, the instance, ...) the same way.
struct SplitMatch {
    bool gives, takes;
    uint32_t dRank, iRank;
};
__device__ __forceinline__ bool split_match(bool canGive, bool idle, SplitMatch& m) {
    const unsigned long long dm = __ballot(canGive);
    if (dm == 0) return false;
    const unsigned long long im = __ballot(idle);
    const uint32_t nd = (uint32_t)__popcll(dm), ni = (uint32_t)__popcll(im);
    const uint32_t nPairs = nd < ni ? nd : ni;
    m.dRank = lane_rank(dm); m.iRank = lane_rank(im);
    m.gives = canGive && m.dRank < nPairs;
    m.takes = idle && m.iRank < nPairs;
    return nPairs != 0;
}

// A donor enters (or opens) its ray's group and announces itself to the taker of its rank.
template <bool ANYHIT, int N> __device__ __forceinline__ void split_give(SplitLds<N>& L, const SplitMatch& m, int& grp, bool found, float4 hit, uint32_t aux = 0u) {
    L.donorOf[m.dRank] = (uint8_t)threadIdx.x;
    if (grp < 0) {   // a new group: the donor's closest hit so far is its first entry
        grp = (int)threadIdx.x;
        L.best[grp] = found ? split_key<ANYHIT>(hit) : ~0ull;
        L.uv[grp] = make_float2(hit.y, hit.z);
        L.aux[grp] = aux;
        L.pending[grp] = 2u;
    } else atomicAdd(&L.pending[grp], 1u);
}

// After the donors' LDS writes (one wave per workgroup: __syncthreads orders them): the ray of the donor lands in the taker's
// registers; every other lane reads its own (src == its lane), so no second copy of a ray is live.  Returns src.
template <int N> __device__ __forceinline__ int split_take_ray(SplitLds<N>& L, const SplitMatch& m, float3& O, float3& D, float3& rD, float4& hit, uint64_t& ri, int& grp) {
    __syncthreads();
    const int src = m.takes ? (int)L.donorOf[m.iRank] : (int)threadIdx.x;
    O.x = __shfl(O.x, src); O.y = __shfl(O.y, src); O.z = __shfl(O.z, src);
    D.x = __shfl(D.x, src); D.y = __shfl(D.y, src); D.z = __shfl(D.z, src);
    rD.x = __shfl(rD.x, src); rD.y = __shfl(rD.y, src); rD.z = __shfl(rD.z, src);
    hit.x = __shfl(hit.x, src);
    ri = ((uint64_t)__shfl((uint32_t)(ri >> 32), src) << 32) | __shfl((uint32_t)ri, src);
    grp = __shfl(grp, src);
    return src;
}

// Once per pass for a member: closest-hit rays are bounded by the group's best t; any-hit rays end when a member found a hit.
template <bool ANYHIT, int N> __device__ __forceinline__ void split_poll(SplitLds<N>& L, int grp, float4& hit, bool& done) {
    const unsigned long long gb = atomicAdd(&L.best[grp], 0ull);
    if (ANYHIT) done = gb == 0ull;
    else hit.x = __builtin_fminf(hit.x, split_key_t(gb));   // (fminf keeps hit.x against the NaN of "no hit yet")
}

// A member found a hit: the others cull against it from their next pass on.  The key orders candidates by (t, prim) — the library's tie
// rule (device_common.h: hit_wins) — so the group's result does not depend on which member found what first.  INST (two-level kernels): the
// same triangle can be reached through several instances at the same t; then the smaller instance wins.  All members of a group are lanes of
// ONE wave, LDS operations of a wave execute in program order, and the atomics of one instruction are serialised: exactly one of the lanes
// that publish a new best key sees an older (larger) key come back and resets the instance word before the holders of the best key take
// its minimum.
template <bool ANYHIT, bool INST = false, int N> __device__ __forceinline__ void split_publish(SplitLds<N>& L, int grp, float4 hit, uint32_t aux = 0u) {
    const unsigned long long key = split_key<ANYHIT>(hit);
    const unsigned long long old = atomicMin(&L.best[grp], key);
    if (ANYHIT) return;
    const bool best = atomicAdd(&L.best[grp], 0ull) == key;   // (read back: of two members finding hits in one pass only the better one writes)
    if (!INST) { if (best) L.uv[grp] = make_float2(hit.y, hit.z); return; }
    if (best && old > key) L.aux[grp] = 0xFFFFFFFFu;
    if (best) atomicMin(&L.aux[grp], aux);
    if (best && atomicAdd(&L.aux[grp], 0u) == aux) L.uv[grp] = make_float2(hit.y, hit.z);
}

// A member is done; the last one writes the record (INST: two-level kernels, hit.inst at byte 44 of the record).
template <bool ANYHIT, bool INST = false, int N> __device__ __forceinline__ void split_finish(SplitLds<N>& L, int& grp, const QueryArgs& q, uint64_t ri) {
    if (atomicSub(&L.pending[grp], 1u) == 1u) {
        const unsigned long long best = atomicAdd(&L.best[grp], 0ull);
        if (ANYHIT) q.occluded[ri] = best != ~0ull ? 1 : 0;
        else if (best != ~0ull) {
            const float2 uv = L.uv[grp];
            q.rays[ri].hit = make_float4(split_key_t(best), uv.x, uv.y, as_f32((uint32_t)best));
            if (INST) ((uint32_t*)(q.rays + ri))[11] = L.aux[grp];
        } else if (q.fresh) q.rays[ri].hit = make_float4(q.freshTmax, 0.f, 0.f, 0.f);
    }
    grp = -1;
}

}  // namespace tbvh
