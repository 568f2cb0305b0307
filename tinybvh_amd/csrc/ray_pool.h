// ray_pool.h — persistent-wave ray scheduling with per-lane replacement.
//
// Every wave keeps a small wave-uniform pool [next, end) of consecutive ray indices taken
// from the global counter with one atomic per CHUNK rays.  Whenever enough lanes are idle
// (their ray finished), the idle lanes are handed the next indices of the pool in lane
// order — no atomic, no LDS.  Lanes therefore never wait for the slowest ray of a batch;
// the price is that a wave's 64 rays are no longer one contiguous tile.
#pragma once
#include "device_common.h"

namespace tbvh {

__device__ __forceinline__ uint32_t lane_rank(uint64_t mask) {
    // number of set bits of `mask` below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

template <int CHUNK> struct RayPool {
    uint64_t next, end;  // wave-uniform
    bool exhausted;      // wave-uniform: the global counter ran past nRays

    __device__ __forceinline__ void init() { next = end = 0; exhausted = false; }

    // Hands out ray indices to the lanes whose `idle` is set.  Returns true for lanes that
    // received one (in `ri`).  Must be called by the whole wave (convergent).
    __device__ __forceinline__ bool acquire(bool idle, unsigned long long* counter, uint64_t nRays, uint64_t& ri) {
        const uint64_t idleMask = __ballot(idle);
        uint32_t need = (uint32_t)__popcll(idleMask);
        uint32_t given = 0;  // indices handed out before this round, per call
        bool got = false;
        const uint32_t rank = lane_rank(idleMask);
        while (need > 0) {
            if (next == end) {
                if (exhausted) break;
                unsigned long long base = 0;
                if ((threadIdx.x & 63u) == 0)  // the whole wave is here (convergent call)
                    base = atomicAdd(counter, (unsigned long long)CHUNK);
                const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base);
                const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
                next = ((uint64_t)hi << 32) | lo;
                end = next + CHUNK;
                if (end >= nRays) { end = nRays; exhausted = true; }
                if (next >= nRays) { next = end = 0; exhausted = true; break; }
            }
            const uint32_t avail = (uint32_t)(end - next);
            const uint32_t take = need < avail ? need : avail;
            if (idle && !got && rank >= given && rank < given + take) { ri = next + (rank - given); got = true; }
            next += take; given += take; need -= take;
        }
        return got;
    }
};

}  // namespace tbvh
