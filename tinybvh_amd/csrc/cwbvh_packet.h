// cwbvh_packet.h — the node test of the wave-packet traversals (kernels_cwbvh_packet.hip; also the two-level experiment tools/experiments/kernels_tlas8_packet.hip): ONE BVH8_CWBVH node, decoded once
// for the wave, against every lane's own ray.
#pragma once
#include "device_common.h"
#include "cwbvh_node.h"

namespace tbvh {

constexpr int kPacketWG = 64;

__device__ __forceinline__ uint32_t sgpr(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// The children of the wave's current node against THIS lane's ray; the planes come from LDS (decoded once for the wave, near / far in the order of the
// wave's octant).  Returns the WAVE's ordered hit mask (a child counts when any lane's ray enters its box).  Empty slots (meta byte 0: half the slots of
// the library's trees, whose nodes hold 1 interior child and 3 leaves on average) cost no vector instruction — a wave-uniform branch —, and the planes of
// slot c + 1 are read from LDS while slot c is tested.  MIXED: some lane's octant differs from the wave's — near and far are sorted per lane through min / max.
// The scalar unit is shared by the CU's four SIMDs and this kernel leans on it as hard as on the vector units (67 scalar + 20 branch against 81 vector
// instructions per ray before this form), so what a slot's bits look like in the hit mask (`placed`: its unary triangle bits at its offset, or bit
// 24 + (slot ^ octant) for an interior child) is worked out by LANES 0..7 in a handful of vector instructions and read back per slot, instead of eight
// times a dozen scalar ones; a lane whose ray is out of the game carries tcull = -1 and needs no mask of its own.
template <bool MIXED>
__device__ __forceinline__ uint32_t pk_test_children(const float (*planes)[8], float ax, float ay, float az, float ox, float oy, float oz, float tcull,
                                                     uint32_t m0, uint32_t m1, uint32_t oct0) {
    const uint32_t lane = threadIdx.x;
    const uint32_t metaL = ((lane & 4u) ? m1 : m0) >> (8u * (lane & 3u)) & 255u;          // lane c (< 8): slot c's meta byte
    const bool innerL = (metaL & 0x18u) == 0x18u;
    const uint32_t placedL = (metaL >> 5) << ((innerL ? (metaL ^ oct0) : metaL) & 31u);
    const uint32_t nonEmpty = (uint32_t)wave_ballot(lane < 8u && metaL != 0u);
    uint32_t hitmask = 0;
    float4 pa = *(const float4*)&planes[0][0];
    float2 pb = *(const float2*)&planes[0][4];
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const float4 qa = pa; const float2 qb = pb;
        if (c < 7) { pa = *(const float4*)&planes[c + 1][0]; pb = *(const float2*)&planes[c + 1][4]; }
        if (!((nonEmpty >> c) & 1u)) continue;
        // (the six FMAs as three v_pk_fma_f32 — the planes do arrive in register pairs — measured 6 % SLOWER on camera rays, same records: profiles/r06_packet_counters.txt)
        float tnx = __builtin_fmaf(qa.x, ax, ox), tny = __builtin_fmaf(qa.y, ay, oy), tnz = __builtin_fmaf(qa.z, az, oz);
        float tfx = __builtin_fmaf(qa.w, ax, ox), tfy = __builtin_fmaf(qb.x, ay, oy), tfz = __builtin_fmaf(qb.y, az, oz);
        if (MIXED) {
            const float a0 = __builtin_fminf(tnx, tfx), a1 = __builtin_fmaxf(tnx, tfx), b0 = __builtin_fminf(tny, tfy), b1 = __builtin_fmaxf(tny, tfy);
            const float c0 = __builtin_fminf(tnz, tfz), c1 = __builtin_fmaxf(tnz, tfz);
            tnx = a0; tfx = a1; tny = b0; tfy = b1; tnz = c0; tfz = c1;
        }
        const float cmin = __builtin_fmaxf(cw_fmax3(tnx, tny, tnz), 0.0f);
        const float cmax = __builtin_fminf(cw_fmin3(tfx, tfy, tfz), tcull);
        const uint32_t placed = (uint32_t)__builtin_amdgcn_readlane((int)placedL, c);
        hitmask |= wave_ballot(cmin <= cmax) != 0ull ? placed : 0u;
    }
    return hitmask;
}

}  // namespace tbvh
