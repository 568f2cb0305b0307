// kernels_refit.hip — BLAS refit on the device for animated meshes.
//
// Replaces the host loop "move the vertices, BVH::Refit (tiny_bvh.h:3055-3093) / MBVH::Refit
// (4925-4961), convert to the GPU layout again (BVH_GPU::ConvertFrom 4612-4655, BVH8_CWBVH::
// ConvertFrom 5884-6018), upload" by kernels that work directly on the uploaded blobs: same topology,
// same triangle order, every box recomputed bottom-up from the new vertex positions.
//
//   1. the triangle records are re-gathered from the new vertices; the prim index every record carries
//      in v0.w says which triangle it is (works for reference-built blobs too, including SBVH ones
//      whose prims are referenced from several leaves: the leaf box then is the full triangle's box,
//      which is conservative);
//   2. boxes are rebuilt bottom-up in PASSES, one kernel launch each: a node is finished in the first pass in
//      which all of its interior children were finished by an EARLIER pass (done[] holds pass numbers; kernel
//      boundaries make earlier passes visible everywhere).  No atomics and no device-scope fences: on MI355X an
//      agent-scope release/acquire pair writes back and invalidates L2 (the 8 XCDs' L2s are not coherent with
//      each other), ~1-2 us each, which made the classic "last child to arrive climbs on" scheme 26 ms for the
//      5.6 M-node BVH_GPU of Bistro.  The number of passes is the height of the tree (10-14 for CWBVH, 40-60 for
//      BVH_GPU); no level ordering or parent-before-child index assumption is needed;
//   3. CWBVH nodes are re-quantised exactly like the host encoder (host_builder.cpp: encode_cwbvh):
//      origin = node box minimum, per-axis exponent = smallest e with extent <= 255 * 2^e, child planes
//      floor/ceil in units of 2^e and then verified against the decode lo + q * 2^e the kernel uses.
//      Slot assignment, child order and triangle order are topology and stay.
#include "device_common.h"
#include "cwbvh_encode.h"
#include "kernels.h"

#include <vector>

namespace tbvh {

namespace {

// ---- triangle records -----------------------------------------------------------------------

// CWBVH: {e2, e1, v0|prim} per triangle (tiny_bvh.h:6004-6008); BVH_GPU gathered form: {v0|prim, e1, e2}.
template <bool CWBVH_ORDER>
__global__ void k_regather(float4* __restrict__ tris, const float4* __restrict__ verts, uint64_t nRecords, uint64_t nTris, uint32_t* __restrict__ status) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nRecords) return;
    const uint32_t prim = as_u32(tris[3 * t + (CWBVH_ORDER ? 2 : 0)].w);
    if (prim >= nTris) { atomicOr(status, 2u); return; }
    const float4 v0 = verts[3 * (uint64_t)prim], v1 = verts[3 * (uint64_t)prim + 1], v2 = verts[3 * (uint64_t)prim + 2];
    const float4 e1 = make_float4(v1.x - v0.x, v1.y - v0.y, v1.z - v0.z, v1.w - v0.w);
    const float4 e2 = make_float4(v2.x - v0.x, v2.y - v0.y, v2.z - v0.z, v2.w - v0.w);
    const float4 a = make_float4(v0.x, v0.y, v0.z, as_f32(prim));
    if (CWBVH_ORDER) { tris[3 * t] = e2; tris[3 * t + 1] = e1; tris[3 * t + 2] = a; }
    else { tris[3 * t] = a; tris[3 * t + 1] = e1; tris[3 * t + 2] = e2; }
}

__device__ __forceinline__ void grow_prim(const float4* __restrict__ verts, uint32_t prim, float3& mn, float3& mx) {
    for (int k = 0; k < 3; k++) {
        const float4 v = verts[3 * (uint64_t)prim + k];
        const float3 p = make_float3(v.x, v.y, v.z);
        mn = min3(mn, p); mx = max3(mx, p);
    }
}

// ---- BVH_GPU (Aila-Laine) ---------------------------------------------------------------------
// node = {lmin, left | lmax, right | rmin, triCount | rmax, firstTri} (tiny_bvh.h:1095-1105): an interior node
// stores the boxes of its two children, so a child's own box is the union of the two boxes IT stores.

// Box of child c of an interior node; false when c is an interior node that was not finished before this pass.
// Leaves are marked done = 1 by the first pass (pass 2), so later passes can tell "unfinished interior node"
// from the done word alone, without touching the child's record.
__device__ __forceinline__ bool al_child_box(const float4* __restrict__ nodes, uint32_t nNodes, const float4* __restrict__ tris, const float4* __restrict__ verts,
                                             const uint32_t* __restrict__ done, uint32_t pass, uint32_t c, float3& mn, float3& mx) {
    mn = make_float3(1e30f, 1e30f, 1e30f); mx = make_float3(-1e30f, -1e30f, -1e30f);
    if (c >= nNodes) return true;   // malformed: leave an empty box
    const uint32_t d = done[c];
    if (d >= pass || (d == 0u && pass > 2u)) return false;
    const float4 c2 = nodes[(size_t)c * 4 + 2], c3 = nodes[(size_t)c * 4 + 3];
    const uint32_t cnt = as_u32(c2.w);
    if (cnt) {
        const uint32_t first = as_u32(c3.w);
        for (uint32_t k = 0; k < cnt; k++) grow_prim(verts, as_u32(tris[3 * (uint64_t)(first + k)].w), mn, mx);
        return true;
    }
    if (d == 0u) return false;      // first pass: an interior child cannot be finished yet
    const float4 c0 = nodes[(size_t)c * 4], c1 = nodes[(size_t)c * 4 + 1];
    mn = min3(make_float3(c0.x, c0.y, c0.z), make_float3(c2.x, c2.y, c2.z));
    mx = max3(make_float3(c1.x, c1.y, c1.z), make_float3(c3.x, c3.y, c3.z));
    return true;
}

__global__ void k_al_pass(float4* __restrict__ nodes, uint32_t nNodes, const float4* __restrict__ tris, const float4* __restrict__ verts,
                          uint32_t* __restrict__ done, uint32_t pass) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nNodes || done[i]) return;
    float4* np = nodes + (size_t)i * 4;
    const float* w = (const float*)np;
    const uint32_t left = as_u32(w[3]), right = as_u32(w[7]), cnt = as_u32(w[11]);   // the .w of the first three float4
    if (cnt) { done[i] = 1u; return; }   // leaf: nothing stored in it depends on the vertices
    float3 lmn, lmx, rmn, rmx;
    if (!al_child_box(nodes, nNodes, tris, verts, done, pass, left, lmn, lmx)) return;
    if (!al_child_box(nodes, nNodes, tris, verts, done, pass, right, rmn, rmx)) return;
    np[0] = make_float4(lmn.x, lmn.y, lmn.z, as_f32(left)); np[1] = make_float4(lmx.x, lmx.y, lmx.z, as_f32(right));
    np[2] = make_float4(rmn.x, rmn.y, rmn.z, as_f32(0u)); np[3] = make_float4(rmx.x, rmx.y, rmx.z, np[3].w);
    done[i] = pass;
}

// ---- BVH8_CWBVH -----------------------------------------------------------------------------------

// Re-encode node j from the boxes of its children (leaf children: from the vertices; interior children:
// nodeBox[child], complete by the time this runs).  Returns the node's own box.
__device__ void cw_encode(float4* __restrict__ np, uint32_t nNodes, const float4* __restrict__ tris, const float4* __restrict__ verts,
                          const float4* __restrict__ boxMin, const float4* __restrict__ boxMax, float3& outMn, float3& outMx) {
    const uint32_t ew = as_u32(np[0].w), imask = ew >> 24;
    const uint32_t childBase = as_u32(np[1].x), triBase = as_u32(np[1].y);
    const uint32_t m0 = as_u32(np[1].z), m1 = as_u32(np[1].w);
    float3 cmn[8], cmx[8];
    bool used[8];
    float3 mn = make_float3(1e30f, 1e30f, 1e30f), mx = make_float3(-1e30f, -1e30f, -1e30f);
    for (int s = 0; s < 8; s++) {
        const uint32_t meta = ((s < 4 ? m0 : m1) >> (8 * (s & 3))) & 255u;
        used[s] = meta != 0;
        if (!used[s]) continue;
        float3 a = make_float3(1e30f, 1e30f, 1e30f), b = make_float3(-1e30f, -1e30f, -1e30f);
        if ((imask >> s) & 1u) {
            const uint32_t c = childBase + __popc(imask & ((1u << s) - 1u));
            if (c < nNodes) { const float4 bn = boxMin[c], bx = boxMax[c]; a = make_float3(bn.x, bn.y, bn.z); b = make_float3(bx.x, bx.y, bx.z); }
        } else {
            // triBase counts float4 blocks (3 per triangle: the kernel addresses triBase + 3 * triangle), meta bits 0-4 the triangle offset
            const uint32_t first = triBase / 3u + (meta & 31u), cnt = __popc(meta >> 5);
            for (uint32_t k = 0; k < cnt; k++) grow_prim(verts, as_u32(tris[3 * (uint64_t)(first + k) + 2].w), a, b);
        }
        cmn[s] = a; cmx[s] = b;
        mn = min3(mn, a); mx = max3(mx, b);
    }
    outMn = mn; outMx = mx;
    cw_quantize_write(np, mn, mx, cmn, cmx, used, imask, childBase, triBase, m0, m1);
}

__global__ void k_cw_pass(float4* __restrict__ nodes, uint32_t nNodes, const float4* __restrict__ tris, const float4* __restrict__ verts,
                          uint32_t* __restrict__ done, uint32_t pass, float4* __restrict__ boxMin, float4* __restrict__ boxMax) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nNodes || done[j]) return;
    const uint32_t imask = as_u32(nodes[(size_t)j * 5].w) >> 24, base = as_u32(nodes[(size_t)j * 5 + 1].x);
    const uint32_t nc = __popc(imask);
    for (uint32_t k = 0; k < nc; k++) {
        const uint32_t c = base + k;
        if (c >= nNodes) continue;             // malformed: cw_encode leaves that child's box empty
        const uint32_t d = done[c];
        if (d == 0u || d >= pass) return;      // an interior child is not finished yet (or only in this pass)
    }
    float3 mn, mx;
    cw_encode(nodes + (size_t)j * 5, nNodes, tris, verts, boxMin, boxMax, mn, mx);
    boxMin[j] = make_float4(mn.x, mn.y, mn.z, 0.f); boxMax[j] = make_float4(mx.x, mx.y, mx.z, 0.f);
    done[j] = pass;
}

// ---- BVH4_GPU -------------------------------------------------------------------------------------------
// One float4 stream (tiny_bvh.h:5115-5244): node = {bmin | qxmin x4} {ext/255 | qxmax x4} {qymin qymax qzmin qzmax x4}
// {childInfo x4}, the triangles of its leaf children inline after it.  There is no node table, so the first refit walks
// the tree once, level by level, and keeps the list of node offsets per level (k_b4_collect); a refit then runs the
// levels deepest first: each node re-gathers the triangles of its leaf children, takes the boxes its interior children
// left in its slot of `childBox` one launch earlier, re-quantises itself like the host encoder (encode_bvh4_gpu /
// kernels_convert.hip) and hands its own box up.
struct B4Item { uint32_t offset, parent, slot, pad; };   // parent = ordinal of the parent node in the item array, slot 0..3

__global__ void k_b4_collect(const float4* __restrict__ blocks, uint64_t nBlocks, const B4Item* __restrict__ items, uint32_t first, uint32_t count,
                             B4Item* __restrict__ out, uint32_t* __restrict__ counter, uint32_t cap) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const B4Item it = items[first + t];
    if ((uint64_t)it.offset + 4 > nBlocks) return;
    const float4 info = blocks[(size_t)it.offset + 3];
    const uint32_t ci[4] = {as_u32(info.x), as_u32(info.y), as_u32(info.z), as_u32(info.w)};
    for (uint32_t i = 0; i < 4; i++) {
        if (ci[i] == 0u || (ci[i] >> 31)) continue;   // empty or leaf
        const uint32_t k = atomicAdd(counter, 1u);
        if (k < cap) out[k] = B4Item{ci[i], first + t, i, 0u};
    }
}

__global__ void k_b4_refit_level(float4* __restrict__ blocks, uint64_t nBlocks, const float4* __restrict__ verts, uint64_t nTris,
                                 const B4Item* __restrict__ items, uint32_t first, uint32_t count, float4* __restrict__ childBox, uint32_t* __restrict__ status) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const uint32_t ord = first + t;
    const B4Item it = items[ord];
    float4* nb = blocks + (size_t)it.offset;
    const float4 info = nb[3];
    const uint32_t ci[4] = {as_u32(info.x), as_u32(info.y), as_u32(info.z), as_u32(info.w)};
    float3 cmn[4], cmx[4];
    bool used[4];
    float3 mn = make_float3(1e30f, 1e30f, 1e30f), mx = make_float3(-1e30f, -1e30f, -1e30f);
    for (uint32_t i = 0; i < 4; i++) {
        used[i] = ci[i] != 0u;
        if (!used[i]) continue;
        float3 a = make_float3(1e30f, 1e30f, 1e30f), b = make_float3(-1e30f, -1e30f, -1e30f);
        if (ci[i] >> 31) {   // leaf: triangles inline at offset + rel, {v0|prim, e1, e2}
            const uint32_t cnt = (ci[i] >> 16) & 0x7fffu, rel = ci[i] & 0xffffu;
            for (uint32_t j = 0; j < cnt; j++) {
                float4* tr = nb + rel + 3 * j;
                if ((uint64_t)it.offset + rel + 3 * j + 3 > nBlocks) break;
                const uint32_t prim = as_u32(tr[0].w);
                if (prim >= nTris) { atomicOr(status, 2u); continue; }
                const float4 v0 = verts[3 * (uint64_t)prim], v1 = verts[3 * (uint64_t)prim + 1], v2 = verts[3 * (uint64_t)prim + 2];
                tr[0] = make_float4(v0.x, v0.y, v0.z, as_f32(prim));
                tr[1] = make_float4(v1.x - v0.x, v1.y - v0.y, v1.z - v0.z, v1.w - v0.w);
                tr[2] = make_float4(v2.x - v0.x, v2.y - v0.y, v2.z - v0.z, v2.w - v0.w);
                a = min3(a, min3(make_float3(v0.x, v0.y, v0.z), min3(make_float3(v1.x, v1.y, v1.z), make_float3(v2.x, v2.y, v2.z))));
                b = max3(b, max3(make_float3(v0.x, v0.y, v0.z), max3(make_float3(v1.x, v1.y, v1.z), make_float3(v2.x, v2.y, v2.z))));
            }
        } else {             // interior: its box was left in our slot by the previous (deeper) launch
            const float4 bn = childBox[((size_t)ord * 4 + i) * 2], bx = childBox[((size_t)ord * 4 + i) * 2 + 1];
            a = make_float3(bn.x, bn.y, bn.z); b = make_float3(bx.x, bx.y, bx.z);
        }
        cmn[i] = a; cmx[i] = b;
        mn = min3(mn, a); mx = max3(mx, b);
    }
    // quantisation frame exactly as the host encoder's / the device conversion's
    const float bmn[3] = {mn.x, mn.y, mn.z}, bmx[3] = {mx.x, mx.y, mx.z};
    float ext[3], scale[3], e255[3], guard[3];
    for (int a = 0; a < 3; a++) {
        ext[a] = bmx[a] - bmn[a];
        scale[a] = ext[a] > 1e-10f ? 254.999f / ext[a] : 0.f;
        e255[a] = ext[a] * (1.0f / 255.0f);
        guard[a] = 4e-7f * fmaxf(fmaxf(fabsf(bmn[a]), fabsf(bmx[a])), ext[a]);
        if (ext[a] > 0) {   // the decode step must carry 255 steps past the far face: jump there, then settle ulp by ulp
            const float need = ((bmx[a] + guard[a]) - bmn[a]) * (1.0f / 255.0f);
            if (need > e255[a]) e255[a] = need;
            while (bmn[a] + e255[a] * 255.0f < bmx[a] + guard[a]) e255[a] = nextafterf(e255[a], 1e30f);
        }
    }
    uint32_t q[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t i = 0; i < 4; i++) {
        if (!used[i]) continue;
        const float lo3[3] = {cmn[i].x, cmn[i].y, cmn[i].z}, hi3[3] = {cmx[i].x, cmx[i].y, cmx[i].z};
        for (int a = 0; a < 3; a++) {
            int lo = (int)floorf((lo3[a] - bmn[a]) * scale[a]), hi = (int)ceilf((hi3[a] - bmn[a]) * scale[a]);
            lo = lo < 0 ? 0 : (lo > 255 ? 255 : lo); hi = hi < 0 ? 0 : (hi > 255 ? 255 : hi);
            while (lo > 0 && bmn[a] + e255[a] * (float)lo > lo3[a] - guard[a]) lo--;
            while (hi < 255 && bmn[a] + e255[a] * (float)hi < hi3[a] + guard[a]) hi++;
            q[2 * a] |= (uint32_t)lo << (8 * i); q[2 * a + 1] |= (uint32_t)hi << (8 * i);
        }
    }
    nb[0] = make_float4(bmn[0], bmn[1], bmn[2], as_f32(q[0]));
    nb[1] = make_float4(e255[0], e255[1], e255[2], as_f32(q[1]));
    nb[2] = make_float4(as_f32(q[2]), as_f32(q[3]), as_f32(q[4]), as_f32(q[5]));
    if (ord != 0) {   // hand the box up
        childBox[((size_t)it.parent * 4 + it.slot) * 2] = make_float4(mn.x, mn.y, mn.z, 0.f);
        childBox[((size_t)it.parent * 4 + it.slot) * 2 + 1] = make_float4(mx.x, mx.y, mx.z, 0.f);
    }
}

}  // namespace

size_t refit_scratch_bytes(int layout, uint32_t nNodes) {
    // done (u32) [+ boxMin, boxMax (float4 each) for CWBVH]
    return (size_t)nNodes * 4 + (layout == kLayoutCwbvh ? (size_t)nNodes * 32 : 0) + 1024;
}

// scratch layout: done[nNodes] | boxMin[nNodes] | boxMax[nNodes]
hipError_t launch_refit(int layout, float4* nodes, uint32_t nNodes, float4* tris, uint64_t nTriRecords, const float4* verts, uint64_t nTris,
                        void* scratch, uint32_t* status, hipStream_t s) {
    uint32_t* done = (uint32_t*)scratch;
    float4* boxMin = (float4*)(((uintptr_t)(done + nNodes) + 255) & ~(uintptr_t)255);
    float4* boxMax = boxMin + nNodes;
    const uint32_t bs = 128, nb = (nNodes + bs - 1) / bs;
    const uint32_t tb = (uint32_t)((nTriRecords + 255) / 256);
    hipError_t e = hipMemsetAsync(done, 0, (size_t)nNodes * 4, s);
    if (e != hipSuccess) return e;
    if (nTriRecords) {
        if (layout == kLayoutCwbvh) hipLaunchKernelGGL(k_regather<true>, dim3(tb), dim3(256), 0, s, tris, verts, nTriRecords, nTris, status);
        else hipLaunchKernelGGL(k_regather<false>, dim3(tb), dim3(256), 0, s, tris, verts, nTriRecords, nTris, status);
    }
    // passes in batches; after each batch look at the root's done word (pass numbers start at 2)
    const int batch = layout == kLayoutCwbvh ? 6 : 24;
    uint32_t pass = 2, rootDone = 0;
    while (!rootDone) {
        for (int k = 0; k < batch; k++, pass++) {
            if (layout == kLayoutCwbvh) hipLaunchKernelGGL(k_cw_pass, dim3(nb), dim3(bs), 0, s, nodes, nNodes, tris, verts, done, pass, boxMin, boxMax);
            else hipLaunchKernelGGL(k_al_pass, dim3(nb), dim3(bs), 0, s, nodes, nNodes, tris, verts, done, pass);
        }
        if ((e = hipMemcpyAsync(&rootDone, done, 4, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
        if (pass > 200000u) return hipErrorUnknown;   // cyclic blob
    }
    return hipGetLastError();
}

}  // namespace tbvh

namespace tbvh {

// BVH4_GPU refit.  items: capacity capNodes; levelFirst: host array filled by the first call (node list per level, root
// level first); childBox: capNodes * 8 float4.  Returns the number of nodes found (0 on the calls that reuse the lists).
hipError_t run_refit_bvh4(float4* blocks, uint64_t nBlocks, const float4* verts, uint64_t nTris, void* itemsDev, uint32_t capNodes, uint32_t* counterDev,
                          float4* childBox, std::vector<uint32_t>& levelFirst, uint32_t* status, hipStream_t s) {
    B4Item* items = (B4Item*)itemsDev;
    hipError_t e;
    if (levelFirst.empty()) {   // walk the tree once: items[levelFirst[l] .. levelFirst[l + 1]) = the nodes of level l
        const B4Item root = {0u, 0u, 0u, 0u};
        if ((e = hipMemcpyAsync(items, &root, sizeof root, hipMemcpyHostToDevice, s)) != hipSuccess) return e;
        levelFirst.push_back(0); levelFirst.push_back(1);
        for (;;) {
            const uint32_t first = levelFirst[levelFirst.size() - 2], count = levelFirst.back() - first;
            if ((e = hipMemsetAsync(counterDev, 0, 4, s)) != hipSuccess) return e;
            hipLaunchKernelGGL(k_b4_collect, dim3((count + 127) / 128), dim3(128), 0, s, blocks, nBlocks, items, first, count, items + levelFirst.back(), counterDev,
                               capNodes - levelFirst.back());
            uint32_t found = 0;
            if ((e = hipMemcpyAsync(&found, counterDev, 4, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
            if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
            if (found == 0) break;
            if (levelFirst.back() + (uint64_t)found > capNodes || levelFirst.size() > 4096) { levelFirst.clear(); return hipErrorUnknown; }   // malformed (cyclic) stream
            levelFirst.push_back(levelFirst.back() + found);
        }
    }
    for (size_t l = levelFirst.size() - 1; l-- > 0;) {
        const uint32_t first = levelFirst[l], count = levelFirst[l + 1] - first;
        hipLaunchKernelGGL(k_b4_refit_level, dim3((count + 127) / 128), dim3(128), 0, s, blocks, nBlocks, verts, nTris, items, first, count, childBox, status);
    }
    return hipGetLastError();
}

}  // namespace tbvh
