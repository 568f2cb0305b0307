// lane_stack.h — per-lane traversal stack: the hot top lives in LDS ([entry][lane] layout,
// so a wave's access to any mix of entries is bank-conflict free: the bank depends only on
// the lane), the cold bottom spills to a per-lane global area ([entry][globalLane], so a
// wave's same-depth spill is one coalesced row).
#pragma once
#include "device_common.h"

namespace tbvh {

template <typename T, int LDS_N, int WG> struct LaneStack {
    T* lds;             // &shared[0][threadIdx.x]; entry stride = WG
    T* spill;           // &spill[0][globalLane]; entry stride = spillStride
    uint32_t spillStride;
    uint32_t spillCap;  // entries available in the spill area
    int sp;
    bool overflow;

    __device__ __forceinline__ void init(T* ldsBase, T* spillBase, uint32_t stride, uint32_t cap) {
        lds = ldsBase; spill = spillBase; spillStride = stride; spillCap = cap; sp = 0; overflow = false;
    }
    __device__ __forceinline__ void reset() { sp = 0; }
    __device__ __forceinline__ bool empty() const { return sp == 0; }
    __device__ __forceinline__ void push(T v) {
        if (sp < LDS_N) lds[sp * WG] = v;
        else if ((uint32_t)(sp - LDS_N) < spillCap) spill[(size_t)(sp - LDS_N) * spillStride] = v;
        else { overflow = true; return; }
        sp++;
    }
    __device__ __forceinline__ T pop() {
        sp--;
        return sp < LDS_N ? lds[sp * WG] : spill[(size_t)(sp - LDS_N) * spillStride];
    }
};

}  // namespace tbvh
