// lane_stack.h — per-lane traversal stack: the hot top lives in LDS ([entry][lane] layout,
// so a wave's access to any mix of entries is bank-conflict free: the bank depends only on
// the lane), the cold bottom spills to a per-lane global area ([entry][globalLane], so a
// wave's same-depth spill is one coalesced row).
//
// The two halves are addressed through pointers that carry their address space (LDS = 3,
// global = 1).  With generic pointers the compiler folds "LDS or spill" into ONE flat_load /
// flat_store on a selected pointer: a flat access goes through the texture-address path even
// when it lands in LDS and counts against vmcnt AND lgkmcnt, so every push/pop would wait for
// the node fetches in flight.  Typed pointers keep them ds_read/ds_write and global_load/store.
#pragma once
#include "device_common.h"

namespace tbvh {

#define TBVH_AS_LDS __attribute__((address_space(3)))
#define TBVH_AS_GLOBAL __attribute__((address_space(1)))

// stack entries travel as plain machine words
template <typename T> struct StackWord;
template <> struct StackWord<uint32_t> {
    typedef uint32_t W;
    static __device__ __forceinline__ W to(uint32_t v) { return v; }
    static __device__ __forceinline__ uint32_t from(W w) { return w; }
};
template <> struct StackWord<uint2> {
    typedef unsigned long long W;
    static __device__ __forceinline__ W to(uint2 v) { return ((W)v.y << 32) | (W)v.x; }
    static __device__ __forceinline__ uint2 from(W w) { return make_uint2((uint32_t)w, (uint32_t)(w >> 32)); }
};

template <typename T, int LDS_N, int WG> struct LaneStack {
    typedef typename StackWord<T>::W W;
    TBVH_AS_LDS W* lds;        // &shared[0][lane]; entry stride = WG
    TBVH_AS_GLOBAL W* spill;   // &spill[0][globalLane]; entry stride = spillStride
    size_t spillStride;
    uint32_t spillCap;         // entries available in the spill area
    int sp;
    bool overflow;

    // ldsBase = &shared[0][lane] of a [LDS_N][WG] array of T; spillBase = this lane's first spill entry
    __device__ __forceinline__ void init(T* ldsBase, T* spillBase, size_t stride, uint32_t cap) {
        lds = (TBVH_AS_LDS W*)ldsBase;
        spill = (TBVH_AS_GLOBAL W*)spillBase;
        spillStride = stride; spillCap = cap; sp = 0; overflow = false;
    }
    __device__ __forceinline__ void reset() { sp = 0; }
    __device__ __forceinline__ bool empty() const { return sp == 0; }
    __device__ __forceinline__ void push(T v) {
        const W w = StackWord<T>::to(v);
        if (sp < LDS_N) lds[sp * WG] = w;
        else if ((uint32_t)(sp - LDS_N) < spillCap) spill[(size_t)(sp - LDS_N) * spillStride] = w;
        else { overflow = true; return; }
        sp++;
    }
    __device__ __forceinline__ T pop() {
        sp--;
        W w;
        if (sp < LDS_N) w = lds[sp * WG];
        else w = spill[(size_t)(sp - LDS_N) * spillStride];
        return StackWord<T>::from(w);
    }
};

}  // namespace tbvh
