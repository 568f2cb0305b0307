// kernels_tlaswide.hip — the caller's TLAS (BVH_GPU / Aila-Laine nodes over BLASInstance records) collapsed into the wide node format
// of its BLASes, on the device, at every upload / update / device rebuild:
//   W = 4  BVH4_GPU node format, leaf children name an instance (childInfo = 1 << 31 | index)      -> kernels_tlas4.hip
//   W = 8  BVH8_CWBVH node format, leaf slots stand for one instance each (instRef[triBase + slot])  -> kernels_tlas8.hip
// One workgroup walks the wide tree level by level (a TLAS has thousands of nodes; one launch, no host round trip, so
// tbvh_rebuild_tlas_device stays asynchronous).  A work item — a subtree that becomes one wide node — is handled by W LANES, lane j
// holding child j in registers: opening the largest child, the octant-slot assignment, the exponent search and the quantisation are
// group reductions over those lanes (a one-thread-per-item version kept its child arrays in scratch memory and took 0.6 ms for
// 1000 instances; this one 0.1 ms).
#include <cstdlib>
#include "device_common.h"
#include "kernels.h"
#include "bvh4_encode.h"
#include "tlas_collapse.h"

namespace tbvh {

namespace {

constexpr int kBuildThreads = 1024;

template <int W, typename T> __device__ __forceinline__ T grp_get(T v, uint32_t j) {   // value of lane j of this W-lane group
    return __shfl(v, (int)(((threadIdx.x & 63u) & ~(uint32_t)(W - 1)) + j), 64);
}
template <int W> __device__ __forceinline__ float grp_min(float v) { for (int m = 1; m < W; m <<= 1) v = fminf(v, __shfl_xor(v, m, 64)); return v; }
template <int W> __device__ __forceinline__ float grp_max(float v) { for (int m = 1; m < W; m <<= 1) v = fmaxf(v, __shfl_xor(v, m, 64)); return v; }
template <int W> __device__ __forceinline__ uint32_t grp_or(uint32_t v) { for (int m = 1; m < W; m <<= 1) v |= __shfl_xor(v, m, 64); return v; }

__device__ __forceinline__ Kid kid_get(const Kid& k, int srcLane) {
    Kid r;
    r.mn = make_float3(__shfl(k.mn.x, srcLane, 64), __shfl(k.mn.y, srcLane, 64), __shfl(k.mn.z, srcLane, 64));
    r.mx = make_float3(__shfl(k.mx.x, srcLane, 64), __shfl(k.mx.y, srcLane, 64), __shfl(k.mx.z, srcLane, 64));
    r.ref = __shfl(k.ref, srcLane, 64); r.cnt = __shfl(k.cnt, srcLane, 64);
    return r;
}
// the two halves a kid opens into
__device__ __forceinline__ void open_kid(const float4* __restrict__ al, uint32_t nAL, const uint32_t* __restrict__ idx, const float4* __restrict__ inst, const Kid& k, Kid& a, Kid& b) {
    if (k.cnt == 0xffffffffu) al_children(al, nAL, k.ref, a, b);
    else { const uint32_t h = k.cnt / 2u; a = range_kid(idx, inst, k.ref, h); b = range_kid(idx, inst, k.ref + h, k.cnt - h); }
}

// One work item — a subtree that becomes one wide node — handled by the W lanes of a group (lane j holds child j).  t: the item's index in `in` (n items on
// this level); cNodes / cRefs / cOut: the allocation counters (shared memory in the one-workgroup kernel, device memory in the per-level kernels).
// items: uint4 {ref, cnt (0xffffffff = AL node), W = 4: word index of the parent's childInfo entry to patch (0xffffffff: root) / W = 8: index of the wide node, -}
template <int W>
__device__ __forceinline__ void tlas_wide_item(const float4* __restrict__ al, uint32_t nAL, const uint32_t* __restrict__ idx, uint32_t nIdx, const float4* __restrict__ inst,
                                               float4* __restrict__ nodes, uint32_t capNodes, uint32_t* __restrict__ instRef, uint32_t capRefs,
                                               const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t t, uint32_t n,
                                               uint32_t* cNodes, uint32_t* cRefs, uint32_t* cOut, uint32_t* __restrict__ status) {
    const uint32_t j = threadIdx.x & (uint32_t)(W - 1);          // my place in the group = the child I hold
    const int g0 = (int)((threadIdx.x & 63u) & ~(uint32_t)(W - 1));   // first lane of my group within the wave
    const bool live = t < n;                              // whole groups are live or not: shuffles below stay inside the group
    const uint4 item = live ? in[t] : make_uint4(0u, 0u, 0u, 0u);
    // ---- the item's first one or two children (lane 0 reads them, lane 1 takes the second) --------------------------
    Kid mine; mine.mn = make_float3(1e30f, 1e30f, 1e30f); mine.mx = make_float3(-1e30f, -1e30f, -1e30f); mine.ref = 0; mine.cnt = 0;
    bool valid = false;
    Kid a = mine, b = mine;
    uint32_t nk = 0;
    if (live && j == 0) {
        if (item.y == 0xffffffffu) { al_children(al, nAL, item.x, a, b); nk = 2; }
        else if (item.y <= 1u) { a = range_kid(idx, inst, item.x, item.y); nk = 1; }
        else { Kid self; self.ref = item.x; self.cnt = item.y; open_kid(al, nAL, idx, inst, self, a, b); nk = 2; }
    }
    nk = grp_get<W>(nk, 0);
    { const Kid b0 = kid_get(b, g0); if (j == 0 && nk >= 1) { mine = a; valid = true; } if (j == 1 && nk == 2) { mine = b0; valid = true; } }
    // ---- open the largest child that can be opened until W children are in hand ---------------------------------------
    while (nk < (uint32_t)W) {
        float bestA = (valid && mine.cnt > 1u) ? kid_area(mine) : -1.f;
        uint32_t best = j;
        for (int m = 1; m < W; m <<= 1) {
            const float oA = __shfl_xor(bestA, m, 64); const uint32_t oL = __shfl_xor(best, m, 64);
            if (oA > bestA || (oA == bestA && oL < best)) { bestA = oA; best = oL; }
        }
        if (!(bestA >= 0.f)) break;
        Kid ca = mine, cb = mine;
        if (j == best) open_kid(al, nAL, idx, inst, mine, ca, cb);
        const Kid cb0 = kid_get(cb, g0 + (int)best);
        if (j == best) mine = ca;
        if (j == nk) { mine = cb0; valid = true; }
        nk++;
    }
    if (valid && mine.cnt == 0u) valid = false;           // an empty child (malformed input)
    const bool inner = valid && mine.cnt > 1u, leaf = valid && mine.cnt == 1u;
    const float3 mn = make_float3(grp_min<W>(valid ? mine.mn.x : 1e30f), grp_min<W>(valid ? mine.mn.y : 1e30f), grp_min<W>(valid ? mine.mn.z : 1e30f));
    const float3 mx = make_float3(grp_max<W>(valid ? mine.mx.x : -1e30f), grp_max<W>(valid ? mine.mx.y : -1e30f), grp_max<W>(valid ? mine.mx.z : -1e30f));
    const uint32_t innerLanes = grp_or<W>(inner ? 1u << j : 0u), leafLanes = grp_or<W>(leaf ? 1u << j : 0u);
    const uint32_t nInner = (uint32_t)__popc(innerLanes), nLeaf = (uint32_t)__popc(leafLanes);
    const uint32_t instIdx = leaf ? (mine.ref < nIdx ? idx[mine.ref] : 0u) : 0u;

    if (W == 4) {
        // ---- BVH4_GPU node: children in the order in hand ----------------------------------------------------------------
        uint32_t base = 0, outFirst = 0;
        if (live && j == 0) { base = atomicAdd(cNodes, 4u); outFirst = nInner ? atomicAdd(cOut, nInner) : 0u; }
        base = grp_get<W>(base, 0); outFirst = grp_get<W>(outFirst, 0);
        const bool fits = live && base + 4u <= capNodes;
        if (live && !fits && j == 0) atomicOr(status, 4u);   // capacity miscount: reported (TBVH_E_FORMAT), never a silently wrong tree
        const Bvh4Frame f = bvh4_frame(mn, mx);
        uint32_t q[6] = {0, 0, 0, 0, 0, 0};
        if (valid) bvh4_quantize_child(f, mine.mn, mine.mx, j, q);
        for (int k = 0; k < 6; k++) q[k] = grp_or<W>(q[k]);
        if (fits) {
            float4* nb = nodes + base;
            if (j == 0) {
                if (item.z != 0xffffffffu) ((uint32_t*)nodes)[item.z] = base;
                nb[0] = make_float4(f.bmn[0], f.bmn[1], f.bmn[2], as_f32(q[0]));
                nb[1] = make_float4(f.e255[0], f.e255[1], f.e255[2], as_f32(q[1]));
                nb[2] = make_float4(as_f32(q[2]), as_f32(q[3]), as_f32(q[4]), as_f32(q[5]));
            }
            ((uint32_t*)(nb + 3))[j] = leaf ? (0x80000000u | instIdx) : 0u;     // an interior child's entry is patched by the child, one level on
            if (inner) out[outFirst + (uint32_t)__popc(innerLanes & ((1u << j) - 1u))] = make_uint4(mine.ref, mine.cnt, (base + 3u) * 4u + j, 0u);
        }
    } else {
        // ---- BVH8_CWBVH node: octant slots by the greedy cost matrix of the encoders (cwbvh_encode.h: cw_assign_slots), one kid per lane ----
        const float3 nc = make_float3(0.5f * (mn.x + mx.x), 0.5f * (mn.y + mx.y), 0.5f * (mn.z + mx.z));
        const float dx = 0.5f * (mine.mn.x + mine.mx.x) - nc.x, dy = 0.5f * (mine.mn.y + mine.mx.y) - nc.y, dz = 0.5f * (mine.mn.z + mine.mx.z) - nc.z;
        uint32_t freeSlots = 0xffu;
        int mySlot = -1;
        for (int round = 0; round < 8; round++) {
            float c = 1e30f; uint32_t cs = 8u;
            if (valid && mySlot < 0) {
#pragma unroll
                for (uint32_t s = 0; s < 8u; s++) {
                    const float cst = ((s & 4u) ? -dx : dx) + ((s & 2u) ? -dy : dy) + ((s & 1u) ? -dz : dz);
                    if (((freeSlots >> s) & 1u) && (cst < c || cs == 8u)) { c = cst; cs = s; }
                }
            }
            uint32_t who = cs < 8u ? j : 8u;
            for (int m = 1; m < W; m <<= 1) {   // the cheapest (cost, slot, lane) of the group
                const float oc = __shfl_xor(c, m, 64); const uint32_t os = __shfl_xor(cs, m, 64), ow = __shfl_xor(who, m, 64);
                const bool take = ow < 8u && (who >= 8u || oc < c || (oc == c && (os < cs || (os == cs && ow < who))));
                if (take) { c = oc; cs = os; who = ow; }
            }
            if (who >= 8u) break;
            if (j == who) mySlot = (int)cs;
            freeSlots &= ~(1u << cs);
        }
        const uint32_t slotBit = mySlot >= 0 ? 1u << mySlot : 0u;
        const uint32_t imask = grp_or<W>(inner ? slotBit : 0u), leafSlots = grp_or<W>(leaf ? slotBit : 0u);
        uint32_t childBase = 0, refBase = 0, outFirst = 0;
        if (live && j == 0) {
            childBase = nInner ? atomicAdd(cNodes, nInner) : 0u;
            refBase = nLeaf ? atomicAdd(cRefs, nLeaf) : 0u;
            outFirst = nInner ? atomicAdd(cOut, nInner) : 0u;
        }
        childBase = grp_get<W>(childBase, 0); refBase = grp_get<W>(refBase, 0); outFirst = grp_get<W>(outFirst, 0);
        const bool fits = live && childBase + nInner <= capNodes && refBase + nLeaf <= capRefs && item.z < capNodes;
        if (live && !fits && j == 0) atomicOr(status, 4u);   // capacity miscount: reported (TBVH_E_FORMAT), never a silently wrong tree
        // per-axis exponent: the smallest e with every child plane within 255 steps of 2^e and the far face reached (cwbvh_encode.h)
        const float lo[3] = {mn.x, mn.y, mn.z}, hi[3] = {mx.x, mx.y, mx.z};
        const float cl[3] = {mine.mn.x, mine.mn.y, mine.mn.z}, ch[3] = {mine.mx.x, mine.mx.y, mine.mx.z};
        int e[3];
        uint32_t qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
#pragma unroll
        for (int ax = 0; ax < 3; ax++) {
            const float ext = hi[ax] - lo[ax];
            int ea = ext > 0 ? ilogbf(ext * (1.0f / 255.0f)) : -126;
            if (ea < -126) ea = -126;
            for (;;) {
                const float sc = ldexpf(1.0f, -ea);
                bool bad = valid && ceilf((ch[ax] - lo[ax]) * sc) > 255.f;
                if (lo[ax] + ldexpf(255.0f, ea) < hi[ax]) bad = true;
                if (grp_or<W>(bad ? 1u : 0u) == 0u || ea >= 127) break;
                ea++;
            }
            e[ax] = ea;
            if (valid) {
                const float inv = ldexpf(1.0f, -ea), sc = ldexpf(1.0f, ea);
                int ql = (int)floorf((cl[ax] - lo[ax]) * inv), qh = (int)ceilf((ch[ax] - lo[ax]) * inv);
                ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql); qh = qh < 0 ? 0 : (qh > 255 ? 255 : qh);
                while (ql > 0 && lo[ax] + sc * (float)ql > cl[ax]) ql--;
                while (qh < 255 && lo[ax] + sc * (float)qh < ch[ax]) qh++;
                qlo[ax] = (uint32_t)ql; qhi[ax] = (uint32_t)qh;
            }
        }
        // the child's bytes travel to the lane of its slot: lane s writes slot s of the node (zeros where no child sits)
        uint32_t w0 = 0, w1 = 0, kidOfSlot = 0;
        if (valid && mySlot >= 0) {
            const uint32_t s = (uint32_t)mySlot, below = (1u << s) - 1u;
            const uint32_t meta = inner ? ((1u << 5) | (24u + s)) : ((1u << 5) | (uint32_t)__popc(leafSlots & below));   // leaf: one "triangle" (= instance) at that offset
            w0 = qlo[0] | (qlo[1] << 8) | (qlo[2] << 16) | (meta << 24);
            w1 = qhi[0] | (qhi[1] << 8) | (qhi[2] << 16);
            kidOfSlot = (j + 1u) << (4u * s);
            if (fits) {
                if (inner) out[outFirst + (uint32_t)__popc(imask & below)] = make_uint4(mine.ref, mine.cnt, childBase + (uint32_t)__popc(imask & below), 0u);
                else instRef[refBase + (uint32_t)__popc(leafSlots & below)] = instIdx;
            }
        }
        kidOfSlot = grp_or<W>(kidOfSlot);
        const uint32_t src = (kidOfSlot >> (4u * j)) & 15u;
        const uint32_t v0 = __shfl(w0, g0 + (int)(src ? src - 1u : 0u), 64), v1 = __shfl(w1, g0 + (int)(src ? src - 1u : 0u), 64);
        if (fits) {
            float4* np = nodes + (size_t)item.z * 5;
            uint8_t* nbytes = (uint8_t*)np;
            const uint32_t b0 = src ? v0 : 0u, b1 = src ? v1 : 0u;
            nbytes[24 + j] = (uint8_t)(b0 >> 24);
            nbytes[32 + j] = (uint8_t)b0; nbytes[40 + j] = (uint8_t)(b0 >> 8); nbytes[48 + j] = (uint8_t)(b0 >> 16);
            nbytes[56 + j] = (uint8_t)b1; nbytes[64 + j] = (uint8_t)(b1 >> 8); nbytes[72 + j] = (uint8_t)(b1 >> 16);
            if (j == 0) {
                const uint32_t eim = ((uint32_t)(uint8_t)(int8_t)e[0]) | ((uint32_t)(uint8_t)(int8_t)e[1] << 8) | ((uint32_t)(uint8_t)(int8_t)e[2] << 16) | (imask << 24);
                np[0] = make_float4(lo[0], lo[1], lo[2], as_f32(eim));
                ((uint32_t*)np)[4] = childBase; ((uint32_t*)np)[5] = refBase;
            }
        }
    }
}

template <int W>
__global__ __launch_bounds__(kBuildThreads) void k_tlas_wide_build(const float4* __restrict__ al, uint32_t nAL, const uint32_t* __restrict__ idx, uint32_t nIdx,
                                                                   const float4* __restrict__ inst, float4* __restrict__ nodes, uint32_t capNodes,
                                                                   uint32_t* __restrict__ instRef, uint32_t capRefs, uint4* __restrict__ itemsA, uint4* __restrict__ itemsB,
                                                                   uint32_t* __restrict__ status, const uint32_t* __restrict__ onlyIf) {
    if (onlyIf && *onlyIf == 0u) return;   // (behind the per-level kernels: only when they ran out of levels)
    __shared__ uint32_t sIn, sOut, sNodes, sRefs;
    if (threadIdx.x == 0) {
        const uint32_t rootCnt = as_u32(al[2].w);
        const uint32_t rootTag = W == 4 ? 0xffffffffu : 0u;
        itemsA[0] = rootCnt ? make_uint4(as_u32(al[3].w), rootCnt, rootTag, 0u) : make_uint4(0u, 0xffffffffu, rootTag, 0u);
        sIn = 1; sOut = 0; sNodes = W == 4 ? 0u : 1u; sRefs = 0;   // W = 4 counts blocks (root allocated like every node), W = 8 counts nodes (node 0 = root)
    }
    __syncthreads();
    const uint32_t group = threadIdx.x / (uint32_t)W, nGroups = kBuildThreads / W;
    uint4 *in = itemsA, *out = itemsB;
    for (uint32_t level = 0; level < 4096u; level++) {
        const uint32_t n = sIn;
        if (n == 0) break;
        for (uint32_t t0 = 0; t0 < n; t0 += nGroups)
            tlas_wide_item<W>(al, nAL, idx, nIdx, inst, nodes, capNodes, instRef, capRefs, in, out, t0 + group, n, &sNodes, &sRefs, &sOut, status);
        __threadfence_block();
        __syncthreads();
        if (threadIdx.x == 0) { sIn = sOut; sOut = 0; }
        __syncthreads();
        uint4* tmp = in; in = out; out = tmp;
    }
}

// TLASes of tens of thousands of instances (round 6): the same walk with ONE LAUNCH PER LEVEL over the whole chip — the level's item count, the allocation
// counters and the next level's count live in device memory (cnt: [0] nodes / blocks, [1] instance refs, [2] "ran out of levels", [4 + level] items of that
// level); kParLevels launches go out back to back without a host round trip (an empty level's workgroups leave at once), and should the tree be deeper
// the one-workgroup kernel above builds it again from scratch (k_tlas_wide_check raises cnt[2]).  64 k instances: 1.97 -> 0.82 ms, 262 k: 5.7 -> 2.3 (the rest is the LBVH and the 64 launches).
constexpr uint32_t kParLevels = 64, kParThreads = 256, kParBlocks = 512;

template <int W>
__global__ void k_tlas_wide_init(const float4* __restrict__ al, uint4* __restrict__ itemsA, uint32_t* __restrict__ cnt) {
    if (threadIdx.x < 4u + kParLevels + 1u) cnt[threadIdx.x] = 0u;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t rootCnt = as_u32(al[2].w);
        const uint32_t rootTag = W == 4 ? 0xffffffffu : 0u;
        itemsA[0] = rootCnt ? make_uint4(as_u32(al[3].w), rootCnt, rootTag, 0u) : make_uint4(0u, 0xffffffffu, rootTag, 0u);
        cnt[0] = W == 4 ? 0u : 1u; cnt[4] = 1u;
    }
}

template <int W>
__global__ __launch_bounds__(kParThreads) void k_tlas_wide_level(const float4* __restrict__ al, uint32_t nAL, const uint32_t* __restrict__ idx, uint32_t nIdx,
                                                                const float4* __restrict__ inst, float4* __restrict__ nodes, uint32_t capNodes,
                                                                uint32_t* __restrict__ instRef, uint32_t capRefs, const uint4* __restrict__ in, uint4* __restrict__ out,
                                                                uint32_t* __restrict__ cnt, uint32_t level, uint32_t* __restrict__ status) {
    const uint32_t n = cnt[4u + level];
    if (n == 0) return;
    const uint32_t nGroups = kParThreads / W, group = threadIdx.x / (uint32_t)W;
    for (uint32_t t0 = blockIdx.x * nGroups; t0 < n; t0 += gridDim.x * nGroups)
        tlas_wide_item<W>(al, nAL, idx, nIdx, inst, nodes, capNodes, instRef, capRefs, in, out, t0 + group, n, cnt + 0, cnt + 1, cnt + 5u + level, status);
}

__global__ void k_tlas_wide_check(uint32_t* __restrict__ cnt) { if (cnt[4u + kParLevels] != 0u) cnt[2] = 1u; }

}  // namespace

// capacities: wide nodes <= AL nodes + instances + 2; the 4-wide format counts 16-byte blocks (4 per node)
size_t tlas_wide_scratch_bytes(uint64_t nAL, uint64_t nInst) { return (size_t)(nAL + nInst + 2) * 16 * 2 + 512; }   // + the per-level kernels' counters
uint64_t tlas4_cap_blocks(uint64_t nAL, uint64_t nInst) { return 4 * (nAL + nInst + 2); }
uint64_t tlas8_cap_nodes(uint64_t nAL, uint64_t nInst) { return nAL + nInst + 2; }

// scratch: two item arrays of nAL + nInst + 2 entries (16 bytes each) + 512 bytes of counters (tlas_wide_scratch_bytes)
static uint32_t tlas_par_min() {
    const char* e = getenv("TBVH_TLAS_PAR_MIN");   // (instances from which the wide TLAS is built level by level over the whole chip; tests set it to 1)
    return e ? (uint32_t)strtoul(e, nullptr, 10) : 16384u;   // (8 k instances: 0.30 ms either way; 64 k: 0.82 against 1.97 ms; 262 k: 2.3 against 5.7)
}

void launch_tlas4_build(const float4* al, uint32_t nAL, const uint32_t* idx, uint32_t nIdx, const float4* inst, uint32_t nInst, float4* blocks, uint32_t capBlocks,
                        void* scratch, uint32_t* status, hipStream_t s) {
    uint4* itemsA = (uint4*)scratch;
    uint4* itemsB = itemsA + (size_t)(nAL + nInst + 2);
    uint32_t* cnt = (uint32_t*)(itemsB + (size_t)(nAL + nInst + 2));
    if (nInst >= tlas_par_min()) {
        hipLaunchKernelGGL(k_tlas_wide_init<4>, dim3(1), dim3(128), 0, s, al, itemsA, cnt);
        for (uint32_t l = 0; l < kParLevels; l++)
            hipLaunchKernelGGL(k_tlas_wide_level<4>, dim3(kParBlocks), dim3(kParThreads), 0, s, al, nAL, idx, nIdx, inst, blocks, capBlocks, (uint32_t*)nullptr, 0u,
                               (l & 1u) ? itemsB : itemsA, (l & 1u) ? itemsA : itemsB, cnt, l, status);
        hipLaunchKernelGGL(k_tlas_wide_check, dim3(1), dim3(1), 0, s, cnt);
        hipLaunchKernelGGL(k_tlas_wide_build<4>, dim3(1), dim3(kBuildThreads), 0, s, al, nAL, idx, nIdx, inst, blocks, capBlocks, (uint32_t*)nullptr, 0u, itemsA, itemsB, status, cnt + 2);
        return;
    }
    hipLaunchKernelGGL(k_tlas_wide_build<4>, dim3(1), dim3(kBuildThreads), 0, s, al, nAL, idx, nIdx, inst, blocks, capBlocks, (uint32_t*)nullptr, 0u, itemsA, itemsB, status, (const uint32_t*)nullptr);
}
void launch_tlas8_build(const float4* al, uint32_t nAL, const uint32_t* idx, uint32_t nIdx, const float4* inst, uint32_t nInst, float4* nodes, uint32_t capNodes,
                        uint32_t* instRef, uint32_t capRefs, void* scratch, uint32_t* status, hipStream_t s) {
    uint4* itemsA = (uint4*)scratch;
    uint4* itemsB = itemsA + (size_t)(nAL + nInst + 2);
    uint32_t* cnt = (uint32_t*)(itemsB + (size_t)(nAL + nInst + 2));
    if (nInst >= tlas_par_min()) {
        hipLaunchKernelGGL(k_tlas_wide_init<8>, dim3(1), dim3(128), 0, s, al, itemsA, cnt);
        for (uint32_t l = 0; l < kParLevels; l++)
            hipLaunchKernelGGL(k_tlas_wide_level<8>, dim3(kParBlocks), dim3(kParThreads), 0, s, al, nAL, idx, nIdx, inst, nodes, capNodes, instRef, capRefs,
                               (l & 1u) ? itemsB : itemsA, (l & 1u) ? itemsA : itemsB, cnt, l, status);
        hipLaunchKernelGGL(k_tlas_wide_check, dim3(1), dim3(1), 0, s, cnt);
        hipLaunchKernelGGL(k_tlas_wide_build<8>, dim3(1), dim3(kBuildThreads), 0, s, al, nAL, idx, nIdx, inst, nodes, capNodes, instRef, capRefs, itemsA, itemsB, status, cnt + 2);
        return;
    }
    hipLaunchKernelGGL(k_tlas_wide_build<8>, dim3(1), dim3(kBuildThreads), 0, s, al, nAL, idx, nIdx, inst, nodes, capNodes, instRef, capRefs, itemsA, itemsB, status, (const uint32_t*)nullptr);
}

}  // namespace tbvh
