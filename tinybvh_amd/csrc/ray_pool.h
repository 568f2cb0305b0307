// ray_pool.h — persistent-wave ray scheduling with per-lane replacement.
//
// Every wave keeps a small wave-uniform pool [next, end) of consecutive ray indices.  Whenever
// enough lanes are idle (their ray finished), the idle lanes are handed the next indices of the
// pool in lane order — no atomic, no LDS.  Lanes therefore never wait for the slowest ray of a
// batch; the price is that a wave's 64 rays are no longer one contiguous tile.
//
// The pool is refilled CHUNK rays at a time from global counters.  A single counter caps the
// whole GPU at about 80 M chunk fetches per second (measured on MI355X: same-address device-scope
// atomics are serialised memory-side at ~12 ns each; with 64-ray chunks that is a hard 5.1 GRays/s
// ceiling, reached on the Sponza stand-in).  The chunks of a batch are therefore dealt round-robin
// to 2^partsLog2 STRIPES, each with its own counter on its own 256-byte line: a stripe owns
// one chunk of every row of P chunks (acquire).  A wave draws from stripe blockIdx % P for its whole
// life.  The batch is still consumed as ONE front (good for the L2s), and consecutive chunks go to
// different XCDs just as consecutive workgroups of a plain launch would.
#pragma once
#include "device_common.h"

namespace tbvh {

constexpr int kPoolParts = 64;          // most partitions / counters a batch can have (the launch picks q.poolParts <= this)
constexpr int kPoolCounterStride = 64;  // in uint32_t: 256 bytes between counters

__device__ __forceinline__ uint32_t lane_rank(uint64_t mask) {
    // number of set bits of `mask` below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Per-wave schedule governor of the adaptive kernels.  Every wave starts in LOCKSTEP: it takes 64 new
// rays only when all 64 lanes are idle, so the lanes stay in the same phase and touch the same nodes
// (fastest for coherent batches).  It measures each generation's lane cohesion = active
// lane-iterations / (64 x iterations); when the running estimate falls below kLockstepKeep, or a
// single generation has already wasted more than 1 - kLockstepBail, the wave switches for the rest of
// the launch to per-lane replacement (refillMin idle lanes trigger a refill: fastest for incoherent
// batches).  All state is wave-uniform; no probe pass, no host-side decision.
// Measured per-generation cohesion (tools/perf_probe.py --variant 46): camera / shadow rays of the
// Sponza stand-in 0.75-0.95, bounce rays < 0.6 everywhere, Bistro stand-in camera rays spread over 0.3-1.
constexpr uint32_t kLockstepKeep = 184, kLockstepBail = 179;   // x / 256: 0.72, 0.70
template <uint32_t KEEP = kLockstepKeep, uint32_t BAIL = kLockstepBail> struct LockstepGovernorT {
    // (every member is wave-uniform; callers pass nIdle as a 32-bit scalar — kernels_cwbvh.hip takes it from s_bcnt1_i32_b64 directly: as the 64-bit
    // value __popcll returns it is compared with VECTOR instructions, 64-bit ordered compares do not exist on the scalar unit)
    uint32_t lockstep;
    uint32_t genIters, genActive, ema;
    __device__ __forceinline__ void init() { lockstep = 1u; genIters = 0; genActive = 0; ema = 0; }
    // Call once per traversal iteration with the number of idle lanes; returns whether the wave should
    // take new rays now.
    __device__ __forceinline__ bool want_refill(uint32_t nIdle, uint32_t refillMin) {
        if (lockstep) {
            genIters++; genActive += 64u - nIdle;
            if (nIdle == 64u) {   // a generation ended: fold its cohesion into the running estimate
                if (genIters > 1u) {
                    const uint32_t e = genActive * 4u / genIters;   // x / 256
                    ema = ema ? (ema + e) >> 1 : e;
                    if (ema < KEEP) lockstep = 0u;
                }
                genIters = 0; genActive = 0;
            } else if (genIters >= 16u && genActive * 4u < BAIL * genIters) lockstep = 0u;
        }
        return lockstep ? nIdle == 64u : nIdle >= refillMin;
    }
};

typedef LockstepGovernorT<> LockstepGovernor;

template <int CHUNK> struct RayPool {
    static_assert(CHUNK % 64 == 0, "chunks are whole 64-ray groups");
    uint64_t next, end;   // wave-uniform: rays in hand
    uint32_t stripe;      // wave-uniform: the stripe this wave draws from
    uint32_t partsLog2;   // the batch's chunks are dealt to 2^partsLog2 stripes
    bool exhausted;       // wave-uniform: the stripe ran past the end of the batch

    // zeroOther: the counter area of the NEXT launch on this context (two areas alternate, capi.hip: launchQuery) — the launch before this
    // one drew from it and is over, the one after this one will find it zero without a memset of its own in the stream (a fill kernel and
    // the gap around it: ~8 us of a 120 us launch; a 1280 x 720 path-traced frame is 6 such launches)
    __device__ __forceinline__ void init(uint32_t log2Parts, uint32_t* zeroOther = nullptr) {
        next = end = 0; exhausted = false; partsLog2 = log2Parts;
        stripe = blockIdx.x & ((1u << log2Parts) - 1u);
        if (zeroOther && blockIdx.x == 0)
            for (uint32_t i = threadIdx.x; i <= (uint32_t)kPoolParts; i += blockDim.x) {
                zeroOther[(size_t)i * kPoolCounterStride] = 0u;
                if (i == (uint32_t)kPoolParts) zeroOther[(size_t)i * kPoolCounterStride + 1u] = 0u;   // (the coherence probe's two words)
            }
    }
    __device__ __forceinline__ bool dry() const { return exhausted && next == end; }

    // Hands out ray indices to the lanes whose `idle` is set.  Returns true for lanes that
    // received one (in `ri`).  Must be called by the whole wave (convergent).
    // counters: one 32-bit count per stripe (rays of that stripe already handed out), kPoolCounterStride apart.
    // The stripe's k-th chunk is chunk k * P + (stripe + 5 k) mod P of the batch: with a fixed position in every row of P chunks a
    // stripe would be a set of pixel columns for rays in image order, and columns are not equally expensive (the waves of the cheapest
    // stripe left 130 us into a 1 M-ray launch whose last stripe ran dry at 380 us); rotating the position makes every stripe a sample
    // of all columns.
    __device__ __forceinline__ bool acquire(bool idle, uint32_t* counters, uint64_t nRays, uint64_t& ri) {
        const uint64_t idleMask = __ballot(idle);
        uint32_t need = (uint32_t)__popcll(idleMask);
        uint32_t given = 0;  // indices handed out before this round, per call
        bool got = false;
        const uint32_t rank = lane_rank(idleMask);
        while (need > 0) {
            if (next == end) {
                if (exhausted) break;
                uint32_t base = 0;
                if ((threadIdx.x & 63u) == 0)  // the whole wave is here (convergent call)
                    base = atomicAdd(counters + (size_t)stripe * kPoolCounterStride, (uint32_t)CHUNK);
                const uint32_t k = (uint32_t)__builtin_amdgcn_readfirstlane(base) / (uint32_t)CHUNK;
                const uint32_t pos = (stripe + 5u * k) & ((1u << partsLog2) - 1u);
                next = (((uint64_t)k << partsLog2) + pos) * (uint64_t)CHUNK;
                end = next + CHUNK;
                if (end >= nRays) { end = nRays; exhausted = true; }   // the later rows lie beyond the batch altogether
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
