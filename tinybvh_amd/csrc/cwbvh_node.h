// cwbvh_node.h — the ONE decode of a BVH8_CWBVH node shared by every kernel that walks the format
// (kernels_cwbvh.hip, kernels_tlas.hip and the experiment kernels).
//
// Node = 5 x float4 = 80 bytes, verbatim as BVH8_CWBVH::ConvertFrom writes it (tiny_bvh.h:5884-6018;
// Ylitie et al. 2017):
//   n0   origin.xyz | ex, ey, ez (int8 exponents), imask (which child slots are interior nodes)
//   n1   childBaseIndex | triangleBaseIndex | meta[8] (interior child: 0b001sssss with sssss = 24 + slot;
//        leaf: unary triangle count << 5 | offset of its first triangle)
//   n2-4 qlox[8] qloy[8] qloz[8] qhix[8] qhiy[8] qhiz[8] (child boxes, 8 bits per plane)
// The test below mirrors what traverse_cwbvh.cl:159-280 computes (re-derived: ldexpf folds the exponent
// into the ray instead of the reference's exponent bit trick; two 4-child halves instead of 8 unrolled
// blocks) and returns the ORDERED hit mask of the CPU mirror tiny_bvh.h:7046-7154: bits 24..31 interior
// children in front-to-back order through octinv, bits 0..23 triangles.
#pragma once
#include "device_common.h"

namespace tbvh {

__device__ __forceinline__ float cw_fmin3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
__device__ __forceinline__ float cw_fmax3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

__device__ __forceinline__ uint32_t cw_sext_s8x4(uint32_t i) {
    // every byte with its top bit set becomes 0xff, others 0x00  (x * 255 as (x << 8) - x: v_mul_lo_u32 is a quarter-rate instruction)
    const uint32_t b = (i >> 7) & 0x01010101u;
    return (b << 8) - b;
}

struct CwNode { float4 n0, n1, n2, n3, n4; };   // one node as fetched
struct CwNodeHits { uint32_t childBase, triBase, hitmask, imask; };

// NSTRIDE: float4s between consecutive nodes: 5 = the reference's packed array; 8 = one node per 128-byte line; kNodeHybrid = the first
// hybridK nodes (surface-area priority order: the top of the tree, which lives in the L2s — there the packed form moves 40 % more nodes
// per second) packed, all later ones (fetched from beyond the L2s, where a line is the unit and a packed node straddles 1.6 of them) one
// per line.  hybridK is a multiple of 8, so the padded part starts on a line.
constexpr int kNodeHybrid = 13;
// ... and in the hybrid copy the triangle word of a node is  embedded << 27 | first 64-byte triangle record  (kernels_cwbvh.hip: k_derive_hybrid):
// the node's triangle number `embedded` (kNoEmbedded: none) is ALSO stored in float4s 5..7 of the node's own line.
constexpr uint32_t kNoEmbedded = 31u;
__device__ __forceinline__ uint32_t cw_hybrid_offset(uint32_t nodeIdx, uint32_t hybridK) {   // in float4s: 8 i - 3 min(i, K), as shifts and adds (v_mul_lo_u32 is quarter rate)
    const uint32_t m = nodeIdx < hybridK ? nodeIdx : hybridK;
    return (nodeIdx << 3) - ((m << 1) + m);
}
template <int NSTRIDE = 5>
__device__ __forceinline__ CwNode cw_load_node(const float4* __restrict__ nodes, uint32_t nodeIdx, uint32_t hybridK = 0u) {
    // the offset in float4s fits 32 bits (a CWBVH blob is addressed in 32-bit float4 blocks: tbvh_upload_cwbvh; the padded copy is only made below
    // 2^29 nodes, the hybrid one below 2^32 float4s: capi_scene.hip), so the address is a shift-add and one v_lshl_add_u64 — (size_t)nodeIdx * 5
    // is a v_mad_u64_u32, a quarter-rate instruction.  (Pinned in assembly: written in C the compiler turns i * 4 + i back into the multiply.)
    uint32_t off;
    if (NSTRIDE == kNodeHybrid) off = cw_hybrid_offset(nodeIdx, hybridK);
    else if (NSTRIDE == 5) asm("v_lshl_add_u32 %0, %1, 2, %1" : "=v"(off) : "v"(nodeIdx));
    else off = nodeIdx * (uint32_t)NSTRIDE;
    const float4* np = nodes + off;
    return CwNode{np[0], np[1], np[2], np[3], np[4]};
}
__device__ __forceinline__ CwNode cw_load_node(const GlobalF4 nodes, uint32_t nodeIdx) {
    const size_t ci = (size_t)nodeIdx * 5u;
    return CwNode{nodes[ci], nodes[ci + 1], nodes[ci + 2], nodes[ci + 3], nodes[ci + 4]};
}

// octinv4 = (7 - sign octant of the ray) replicated into four bytes
__device__ __forceinline__ uint32_t cw_oct(float3 D) { return 7u - ((D.x < 0 ? 4u : 0u) | (D.y < 0 ? 2u : 0u) | (D.z < 0 ? 1u : 0u)); }

// Slab-test the 8 children against [0, tmax] (inclusive at both ends, like the oracle's box rule).
// negX / negY / negZ: rD.x < 0 etc. — callers that keep them per ray (kernels_cwbvh.hip: set when a lane takes a ray) save the three compares of
// every node test: as loop-carried bools they live in scalar lane masks that the twelve near / far selects read directly.
__device__ __forceinline__ CwNodeHits cw_test_node(const CwNode& nr, float3 O, float3 rD, float tmax, uint32_t octinv4, bool negX, bool negY, bool negZ) {
    const float4 n0 = nr.n0, n1 = nr.n1, n2 = nr.n2, n3 = nr.n3, n4 = nr.n4;
    const uint32_t ew = as_u32(n0.w);
    const float ax = ldexpf(rD.x, (int)(int8_t)(ew)), ay = ldexpf(rD.y, (int)(int8_t)(ew >> 8)), az = ldexpf(rD.z, (int)(int8_t)(ew >> 16));
    const float ox = (n0.x - O.x) * rD.x, oy = (n0.y - O.y) * rD.y, oz = (n0.z - O.z) * rD.z;
    uint32_t hitmask = 0;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const uint32_t meta4 = half ? as_u32(n1.w) : as_u32(n1.z);
        // interior children (meta = 0b001sssss with sssss = 24 + slot: bits 3 and 4 set) get their bit index through the ray's octant: index ^= octinv.
        // octinv4's bytes are 0..7, so the mask only needs 0x07 per interior byte: y = bit 3 of (meta & meta >> 1), 0x07 = y - (y >> 3) — two full-rate
        // instructions where the 0xff-per-byte sign extension took a v_mul_lo_u32 (quarter rate) on top of its shifts
        const uint32_t inner8 = (meta4 & (meta4 >> 1)) & 0x08080808u;
        const uint32_t imask4 = inner8 - (inner8 >> 3);
        const uint32_t bitidx4 = (meta4 ^ (octinv4 & imask4)) & 0x1F1F1F1Fu;
        const uint32_t bits4 = (meta4 >> 5) & 0x07070707u;
        const uint32_t qlx = half ? as_u32(n2.y) : as_u32(n2.x), qhx = half ? as_u32(n3.w) : as_u32(n3.z);
        const uint32_t qly = half ? as_u32(n2.w) : as_u32(n2.z), qhy = half ? as_u32(n4.y) : as_u32(n4.x);
        const uint32_t qlz = half ? as_u32(n3.y) : as_u32(n3.x), qhz = half ? as_u32(n4.w) : as_u32(n4.z);
        // near / far plane words picked once per axis by the sign of the ray direction
        const uint32_t lox = negX ? qhx : qlx, hix = negX ? qlx : qhx;
        const uint32_t loy = negY ? qhy : qly, hiy = negY ? qly : qhy;
        const uint32_t loz = negZ ? qhz : qlz, hiz = negZ ? qlz : qhz;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int sh = 8 * i;
            const float tnx = __builtin_fmaf((float)((lox >> sh) & 255), ax, ox), tfx = __builtin_fmaf((float)((hix >> sh) & 255), ax, ox);
            const float tny = __builtin_fmaf((float)((loy >> sh) & 255), ay, oy), tfy = __builtin_fmaf((float)((hiy >> sh) & 255), ay, oy);
            const float tnz = __builtin_fmaf((float)((loz >> sh) & 255), az, oz), tfz = __builtin_fmaf((float)((hiz >> sh) & 255), az, oz);
            const float cmin = __builtin_fmaxf(cw_fmax3(tnx, tny, tnz), 0.0f);
            const float cmax = __builtin_fminf(cw_fmin3(tfx, tfy, tfz), tmax);
            // (child bits) << (bit index), both bytes picked by SDWA selects in ONE instruction (the compiler spends a v_bfe_u32 and, for the odd
            // bytes, a shift of the index word as well)
            uint32_t placed;
            if (i == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0" : "=v"(placed) : "v"(bitidx4), "v"(bits4));
            else if (i == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1" : "=v"(placed) : "v"(bitidx4), "v"(bits4));
            else if (i == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2" : "=v"(placed) : "v"(bitidx4), "v"(bits4));
            else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3" : "=v"(placed) : "v"(bitidx4), "v"(bits4));
            if (cmin <= cmax) hitmask |= placed;
        }
    }
    CwNodeHits r;
    r.childBase = as_u32(n1.x); r.triBase = as_u32(n1.y); r.hitmask = hitmask; r.imask = ew >> 24;
    return r;
}

__device__ __forceinline__ CwNodeHits cw_test_node(const CwNode& nr, float3 O, float3 rD, float tmax, uint32_t octinv4) {
    return cw_test_node(nr, O, rD, tmax, octinv4, rD.x < 0, rD.y < 0, rD.z < 0);
}

// Traversal state of one ray in a CWBVH (Ylitie's node group / triangle group):
//   ng = {child base index, hits << 24 | imask}   interior children still to visit, highest set bit first
//   tg = {triangle base,    triangle bits}        triangles still to test
// cw_next_child takes the front-most pending child off ng and returns its node index.
__device__ __forceinline__ uint32_t cw_next_child(uint2& ng, uint32_t oct) {
    const uint32_t imask = ng.y;
    const uint32_t bit = 31u - (uint32_t)__builtin_clz(ng.y);   // (callers have checked cw_has_child: ng.y != 0 — __clz would guard the zero case with a v_min_u32)
    ng.y &= ~(1u << bit);
    const uint32_t slot = (bit - 24u) ^ oct;
    return ng.x + __popc(imask & ~(0xFFFFFFFFu << slot));
}
__device__ __forceinline__ bool cw_has_child(uint2 ng) { return ng.y > 0x00FFFFFFu; }

}  // namespace tbvh
