// kernels_convert.hip — BVH2 -> BVH8_CWBVH layout conversion on the device.
//
// Replaces BVH8_CWBVH::ConvertFrom (tiny_bvh.h:5884-6018; MBVH<8>::ConvertFrom 4975-5048 for the
// collapse) for callers that have a plain BVH2 — the 32-byte BVHNode array of BVH::Build plus primIdx
// and the vertices — and want the compressed wide layout without the single-threaded host conversion:
// upload the BVH2, convert on the GPU, trace.
//
// One workgroup-free pass per level of the WIDE tree (about log8 of the node count, ~10 launches for
// Bistro): a thread owns one wide node.  It starts from the two children of its BVH2 node and opens the
// interior child with the largest surface area until it has 8 (the collapse of MBVH::ConvertFrom and of
// host_builder.cpp: collapse<8>), assigns the children to octant slots with the same greedy cost
// matrix as the host encoder, reserves consecutive node indices for its interior children and
// consecutive triangle records for its leaves with two atomics, quantises the child boxes
// (cwbvh_encode.h) and writes the node, the triangle records {e2, e1, v0|prim} and one work item per
// interior child for the next level.  Node numbering depends on atomic order, everything else is the
// host encoder's result (same node and triangle counts, same boxes).
//
// Input contract = the reference's: leaves hold at most 3 triangles (BVH::SplitLeafs(3), as
// BVH8_CWBVH::ConvertFrom does at tiny_bvh.h:5893-5899); a larger leaf is reported, not guessed at.
#include "cwbvh_encode.h"
#include "bvh4_encode.h"
#include "device_common.h"
#include "kernels.h"

namespace tbvh {

namespace {

struct N2 { float3 mn; uint32_t leftFirst; float3 mx; uint32_t triCount; };

__device__ __forceinline__ N2 load_n2(const float4* __restrict__ nodes2, uint32_t i) {
    const float4 a = nodes2[2 * (size_t)i], b = nodes2[2 * (size_t)i + 1];
    N2 n; n.mn = make_float3(a.x, a.y, a.z); n.leftFirst = as_u32(a.w); n.mx = make_float3(b.x, b.y, b.z); n.triCount = as_u32(b.w);
    return n;
}
__device__ __forceinline__ float half_area(const N2& n) {
    const float ex = n.mx.x - n.mn.x, ey = n.mx.y - n.mn.y, ez = n.mx.z - n.mn.z;
    return ex * ey + ey * ez + ez * ex;
}

// counters: [0] wide nodes allocated, [1] triangles allocated, [2] work items written for the next level
__global__ void k_convert_level(const float4* __restrict__ nodes2, uint32_t nNodes2, const uint32_t* __restrict__ primIdx, uint64_t nIdx,
                                const float4* __restrict__ verts, uint64_t nTris, const uint2* __restrict__ itemsIn, uint32_t nIn,
                                uint2* __restrict__ itemsOut, uint32_t* __restrict__ counters, float4* __restrict__ cwNodes, uint32_t capNodes,
                                float4* __restrict__ cwTris, uint64_t capTris, uint32_t* __restrict__ status) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nIn) return;
    const uint2 item = itemsIn[t];   // x = BVH2 node, y = wide node index
    const N2 self = load_n2(nodes2, item.x);
    uint32_t kids[8];
    uint32_t nk;
    if (self.triCount) { kids[0] = item.x; nk = 1; }   // single-leaf BVH2: the wide root gets that leaf as its only child
    else {
        kids[0] = self.leftFirst; kids[1] = self.leftFirst + 1; nk = 2;
        if (kids[1] >= nNodes2) { atomicOr(status, 4u); return; }
        while (nk < 8u) {
            int best = -1; float bestSA = -1.f;
            for (uint32_t i = 0; i < nk; i++) {
                const N2 c = load_n2(nodes2, kids[i]);
                if (c.triCount) continue;
                const float sa = half_area(c);
                if (sa > bestSA) { bestSA = sa; best = (int)i; }
            }
            if (best < 0) break;
            const uint32_t l = load_n2(nodes2, kids[best]).leftFirst;
            if (l + 1 >= nNodes2) { atomicOr(status, 4u); return; }
            kids[best] = l; kids[nk++] = l + 1;
        }
    }
    // ---- octant slots: greedy on cost[s][i] = dot(centroid_i - centroid_node, dir_s) (host encoder / tiny_bvh.h:5906-5938)
    N2 kid[8];
    for (uint32_t i = 0; i < nk; i++) kid[i] = load_n2(nodes2, kids[i]);
    int slotOf[8], childIn[8];
    {
        float3 cmnA[8], cmxA[8];
        for (uint32_t i = 0; i < nk; i++) { cmnA[i] = kid[i].mn; cmxA[i] = kid[i].mx; }
        cw_assign_slots(nk, self.mn, self.mx, cmnA, cmxA, slotOf, childIn);
    }
    // ---- allocation
    uint32_t nInner = 0, nT = 0;
    for (uint32_t i = 0; i < nk; i++) {
        if (kid[i].triCount == 0) nInner++;
        else { if (kid[i].triCount > 3u) { atomicOr(status, 8u); return; } nT += kid[i].triCount; }
    }
    const uint32_t childBase = nInner ? atomicAdd(counters + 0, nInner) : 0u;
    const uint32_t triFirst = nT ? atomicAdd(counters + 1, nT) : 0u;
    const uint32_t outFirst = nInner ? atomicAdd(counters + 2, nInner) : 0u;
    if ((uint64_t)childBase + nInner > capNodes || (uint64_t)triFirst + nT > capTris) { atomicOr(status, 4u); return; }
    // ---- children in slot order: interior children get consecutive node indices in that order (the traversal
    // finds child s at childBase + popc(imask below s)), leaves consecutive triangle records
    float3 cmn[8], cmx[8];
    bool used[8];
    uint8_t meta[8];
    uint32_t imask = 0, inner = 0, tris = 0;
    for (int s = 0; s < 8; s++) {
        used[s] = childIn[s] >= 0; meta[s] = 0;
        if (!used[s]) continue;
        const N2& c = kid[childIn[s]];
        cmn[s] = c.mn; cmx[s] = c.mx;
        if (c.triCount == 0) {
            imask |= 1u << s;
            meta[s] = (uint8_t)((1u << 5) | (24 + s));
            itemsOut[outFirst + inner] = make_uint2(kids[childIn[s]], childBase + inner);
            inner++;
        } else {
            const uint32_t unary = c.triCount == 1 ? 1u : c.triCount == 2 ? 3u : 7u;
            meta[s] = (uint8_t)((unary << 5) | tris);
            for (uint32_t j = 0; j < c.triCount; j++) {
                const uint64_t pi = (uint64_t)c.leftFirst + j;
                if (!primIdx) {   // RECORD MODE (the 8-wide copy of a BVH4_GPU blob, capi_scene.hip): `verts` holds finished records {v0|prim, e1, e2}, one per leaf entry — the
                                  // bytes the blob's own inline triangles carry; recomputing e1 = v1 - v0 from v1 = v0 + e1 would not give them back
                    if (pi >= nTris) { atomicOr(status, 4u); continue; }
                    float4* o = cwTris + 3 * (uint64_t)(triFirst + tris + j);
                    o[0] = verts[3 * pi + 2]; o[1] = verts[3 * pi + 1]; o[2] = verts[3 * pi];
                    continue;
                }
                const uint32_t prim = pi < nIdx ? primIdx[pi] : 0xffffffffu;
                if (prim >= nTris) { atomicOr(status, 4u); continue; }
                const float4 v0 = verts[3 * (uint64_t)prim], v1 = verts[3 * (uint64_t)prim + 1], v2 = verts[3 * (uint64_t)prim + 2];
                float4* o = cwTris + 3 * (uint64_t)(triFirst + tris + j);
                o[0] = make_float4(v2.x - v0.x, v2.y - v0.y, v2.z - v0.z, v2.w - v0.w);
                o[1] = make_float4(v1.x - v0.x, v1.y - v0.y, v1.z - v0.z, v1.w - v0.w);
                o[2] = make_float4(v0.x, v0.y, v0.z, as_f32(prim));
            }
            tris += c.triCount;
        }
    }
    const uint32_t m0 = meta[0] | (meta[1] << 8) | (meta[2] << 16) | ((uint32_t)meta[3] << 24);
    const uint32_t m1 = meta[4] | (meta[5] << 8) | (meta[6] << 16) | ((uint32_t)meta[7] << 24);
    // the node's frame = its own box united with its children's: the same box for a well-formed BVH2; for one derived from quantised boxes (the 8-wide
    // copy of a BVH4_GPU stream: every level of the source was rounded on its own grid) a child may reach an ulp beyond its parent's stored box, and the
    // quantiser would clip what lies below the frame's origin — a box a grazing ray then misses
    float3 fmn = self.mn, fmx = self.mx;
    for (int s = 0; s < 8; s++) if (used[s]) { fmn = min3(fmn, cmn[s]); fmx = max3(fmx, cmx[s]); }
    cw_quantize_write(cwNodes + (size_t)item.y * 5, fmn, fmx, cmn, cmx, used, imask, childBase, triFirst * 3u, m0, m1);
}


// ---- BVH4_GPU (tiny_bvh.h:5115-5244; host_builder.cpp: encode_bvh4_gpu) ---------------------------------------
// One float4 stream: a node is 4 blocks {bmin | qxmin x4} {ext/255 | qxmax x4} {qymin, qymax, qzmin, qzmax x4}
// {childInfo x4}; the triangles of its leaf children follow it inline as {v0|prim, e1, e2}.  childInfo: interior =
// block index of the child node; leaf = 1<<31 | triCount<<16 | offset of its first triangle relative to the node.
// A node and its inline triangles are one allocation (one atomic); the block index of an interior child is not
// known when the parent is written, so the child patches it into the parent's childInfo word when it is emitted
// one level later (the host encoder does the same with its patchWord).
// counters: [0] blocks allocated, [2] work items for the next level.  items: x = BVH2 node, y = u32 index (into the
// stream viewed as words) of the childInfo word to patch, 0xffffffff for the root.
__global__ void k_convert4_level(const float4* __restrict__ nodes2, uint32_t nNodes2, const uint32_t* __restrict__ primIdx, uint64_t nIdx,
                                 const float4* __restrict__ verts, uint64_t nTris, const uint2* __restrict__ itemsIn, uint32_t nIn,
                                 uint2* __restrict__ itemsOut, uint32_t* __restrict__ counters, float4* __restrict__ blocks, uint64_t capBlocks,
                                 uint32_t* __restrict__ status) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nIn) return;
    const uint2 item = itemsIn[t];
    const N2 self = load_n2(nodes2, item.x);
    uint32_t kids[4];
    uint32_t nk;
    if (self.triCount) { kids[0] = item.x; nk = 1; }   // single-leaf BVH2: the root gets that leaf as its only child
    else {
        kids[0] = self.leftFirst; kids[1] = self.leftFirst + 1; nk = 2;
        if (kids[1] >= nNodes2) { atomicOr(status, 4u); return; }
        while (nk < 4u) {
            int best = -1; float bestSA = -1.f;
            for (uint32_t i = 0; i < nk; i++) {
                const N2 c = load_n2(nodes2, kids[i]);
                if (c.triCount) continue;
                const float sa = half_area(c);
                if (sa > bestSA) { bestSA = sa; best = (int)i; }
            }
            if (best < 0) break;
            const uint32_t l = load_n2(nodes2, kids[best]).leftFirst;
            if (l + 1 >= nNodes2) { atomicOr(status, 4u); return; }
            kids[best] = l; kids[nk++] = l + 1;
        }
    }
    N2 kid[4];
    uint32_t nT = 0, nInner = 0;
    for (uint32_t i = 0; i < nk; i++) {
        kid[i] = load_n2(nodes2, kids[i]);
        if (kid[i].triCount) { if (kid[i].triCount >= 32768u) { atomicOr(status, 8u); return; } nT += kid[i].triCount; } else nInner++;
    }
    if (4u + 3u * nT >= 65536u) { atomicOr(status, 8u); return; }   // inline triangles must stay within the 16-bit relative offset
    const uint32_t base = atomicAdd(counters + 0, 4u + 3u * nT);
    const uint32_t outFirst = nInner ? atomicAdd(counters + 2, nInner) : 0u;
    if ((uint64_t)base + 4u + 3u * nT > capBlocks) { atomicOr(status, 4u); return; }
    if (item.y != 0xffffffffu) ((uint32_t*)blocks)[item.y] = base;
    // quantisation frame and conservative child planes: bvh4_encode.h (shared with the 4-wide TLAS builder)
    const Bvh4Frame frame = bvh4_frame(self.mn, self.mx);
    uint32_t info[4] = {0, 0, 0, 0}, q[6] = {0, 0, 0, 0, 0, 0};   // q: xmin, xmax, ymin, ymax, zmin, zmax; byte i = child i
    uint32_t rel = 4, inner = 0;
    for (uint32_t i = 0; i < nk; i++) {
        const N2& c = kid[i];
        bvh4_quantize_child(frame, c.mn, c.mx, i, q);
        if (c.triCount) {
            info[i] = 0x80000000u | (c.triCount << 16) | rel;
            for (uint32_t j = 0; j < c.triCount; j++) {
                const uint64_t pi = (uint64_t)c.leftFirst + j;
                float4* o = blocks + (size_t)base + rel + 3 * j;
                if (!primIdx) {   // RECORD MODE (as in k_convert_level): `verts` holds finished records {v0|prim, e1, e2}, one per leaf entry, carried over bit for bit
                    if (pi >= nTris) { atomicOr(status, 4u); o[0] = o[1] = o[2] = make_float4(0, 0, 0, 0); continue; }
                    o[0] = verts[3 * pi]; o[1] = verts[3 * pi + 1]; o[2] = verts[3 * pi + 2];
                    continue;
                }
                const uint32_t prim = pi < nIdx ? primIdx[pi] : 0xffffffffu;
                if (prim >= nTris) { atomicOr(status, 4u); o[0] = o[1] = o[2] = make_float4(0, 0, 0, 0); continue; }
                const float4 v0 = verts[3 * (uint64_t)prim], v1 = verts[3 * (uint64_t)prim + 1], v2 = verts[3 * (uint64_t)prim + 2];
                o[0] = make_float4(v0.x, v0.y, v0.z, as_f32(prim));
                o[1] = make_float4(v1.x - v0.x, v1.y - v0.y, v1.z - v0.z, v1.w - v0.w);
                o[2] = make_float4(v2.x - v0.x, v2.y - v0.y, v2.z - v0.z, v2.w - v0.w);
            }
            rel += 3 * c.triCount;
        } else {
            itemsOut[outFirst + inner] = make_uint2(kids[i], (base + 3u) * 4u + i);
            inner++;
        }
    }
    bvh4_write_node(blocks + (size_t)base, frame, q, info);   // interior entries are patched by the children
}

}  // namespace

// itemsA/itemsB: two work-item arrays of capNodes entries; counters: 4 x u32 in device memory.
// Runs level by level until no interior child is left; returns the node and triangle counts.
hipError_t run_convert_cwbvh(const float4* nodes2, uint32_t nNodes2, const uint32_t* primIdx, uint64_t nIdx, const float4* verts, uint64_t nTris,
                             float4* cwNodes, uint32_t capNodes, float4* cwTris, uint64_t capTris, uint2* itemsA, uint2* itemsB, uint32_t* counters,
                             uint32_t* status, hipStream_t s, uint32_t* nNodesOut, uint64_t* nTrisOut, uint32_t* levelsOut) {
    const uint32_t init[4] = {1u, 0u, 0u, 0u};   // wide node 0 = root
    const uint2 root = make_uint2(0u, 0u);
    hipError_t e = hipMemcpyAsync(counters, init, 16, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(itemsA, &root, 8, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    uint32_t nIn = 1, levels = 0;
    uint2 *in = itemsA, *out = itemsB;
    while (nIn) {
        if ((e = hipMemsetAsync(counters + 2, 0, 4, s)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_convert_level, dim3((nIn + 63) / 64), dim3(64), 0, s, nodes2, nNodes2, primIdx, nIdx, verts, nTris, in, nIn, out, counters,
                           cwNodes, capNodes, cwTris, capTris, status);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        uint32_t c[3];
        if ((e = hipMemcpyAsync(c, counters, 12, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
        nIn = c[2];
        *nNodesOut = c[0]; *nTrisOut = c[1];
        uint2* t = in; in = out; out = t;
        if (++levels > 4096) return hipErrorUnknown;   // cyclic input
    }
    *levelsOut = levels;
    return hipSuccess;
}

}  // namespace tbvh

namespace tbvh {
// BVH4_GPU: same driver loop, one stream.  Returns the number of 16-byte blocks.
hipError_t run_convert_bvh4(const float4* nodes2, uint32_t nNodes2, const uint32_t* primIdx, uint64_t nIdx, const float4* verts, uint64_t nTris,
                            float4* blocks, uint64_t capBlocks, uint2* itemsA, uint2* itemsB, uint32_t* counters, uint32_t* status, hipStream_t s,
                            uint64_t* nBlocksOut, uint32_t* levelsOut) {
    const uint32_t init[4] = {0u, 0u, 0u, 0u};
    const uint2 root = make_uint2(0u, 0xffffffffu);
    hipError_t e = hipMemcpyAsync(counters, init, 16, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(itemsA, &root, 8, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    uint32_t nIn = 1, levels = 0;
    uint2 *in = itemsA, *out = itemsB;
    while (nIn) {
        if ((e = hipMemsetAsync(counters + 2, 0, 4, s)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_convert4_level, dim3((nIn + 63) / 64), dim3(64), 0, s, nodes2, nNodes2, primIdx, nIdx, verts, nTris, in, nIn, out, counters,
                           blocks, capBlocks, status);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        uint32_t c[3];
        if ((e = hipMemcpyAsync(c, counters, 12, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
        nIn = c[2];
        *nBlocksOut = c[0];
        uint2* t = in; in = out; out = t;
        if (++levels > 4096) return hipErrorUnknown;   // cyclic input
    }
    *levelsOut = levels;
    return hipSuccess;
}
}  // namespace tbvh
