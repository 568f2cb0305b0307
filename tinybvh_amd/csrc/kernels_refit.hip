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
//   2. leaves compute their boxes from the vertices, interior nodes from their children: one thread
//      starts at every node that has no interior child and climbs; an atomic counter per node lets the
//      last arriving child continue (Karras-style), so no level ordering or parent-before-child index
//      assumption is needed;
//   3. CWBVH nodes are re-quantised exactly like the host encoder (host_builder.cpp: encode_cwbvh):
//      origin = node box minimum, per-axis exponent = smallest e with extent <= 255 * 2^e, child planes
//      floor/ceil in units of 2^e and then verified against the decode lo + q * 2^e the kernel uses.
//      Slot assignment, child order and triangle order are topology and stay.
#include "device_common.h"
#include "cwbvh_encode.h"
#include "kernels.h"

namespace tbvh {

namespace {

__device__ __forceinline__ float ld_agent_f(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float3 ld_agent3(const float4* p) {   // written by another CU during this kernel: bypass the non-coherent L1
    const float* f = (const float*)p;
    return make_float3(ld_agent_f(f), ld_agent_f(f + 1), ld_agent_f(f + 2));
}
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
// node = {lmin, left | lmax, right | rmin, triCount | rmax, firstTri} (tiny_bvh.h:1095-1105)

// parent[c] = parent node index, bit 31 set when c is the right child
__global__ void k_al_parents(const float4* __restrict__ nodes, uint32_t nNodes, uint32_t* __restrict__ parent) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nNodes) return;
    if (as_u32(nodes[i * 4 + 2].w)) return;   // leaf
    const uint32_t l = as_u32(nodes[i * 4].w), r = as_u32(nodes[i * 4 + 1].w);
    if (l < nNodes) parent[l] = i;
    if (r < nNodes) parent[r] = i | 0x80000000u;
}

__global__ void k_al_refit(float4* __restrict__ nodes, uint32_t nNodes, const float4* __restrict__ tris, const float4* __restrict__ verts,
                           const uint32_t* __restrict__ parent, uint32_t* __restrict__ arrived) {
    uint32_t node = blockIdx.x * blockDim.x + threadIdx.x;
    if (node >= nNodes) return;
    const uint32_t cnt = as_u32(nodes[node * 4 + 2].w);
    if (!cnt || node == 0) return;   // interior nodes are finished by their last child; a root leaf has no box to store
    const uint32_t first = as_u32(nodes[node * 4 + 3].w);
    float3 mn = make_float3(1e30f, 1e30f, 1e30f), mx = make_float3(-1e30f, -1e30f, -1e30f);
    for (uint32_t k = 0; k < cnt; k++) grow_prim(verts, as_u32(tris[3 * (uint64_t)(first + k)].w), mn, mx);
    for (;;) {
        const uint32_t pe = parent[node], p = pe & 0x7fffffffu;
        if (pe == 0xffffffffu) return;   // not referenced by any node (hole in the blob)
        float4* pn = nodes + (size_t)p * 4;
        const int o = (pe >> 31) ? 2 : 0;        // this child's {min, max} pair inside the parent record
        pn[o] = make_float4(mn.x, mn.y, mn.z, pn[o].w);
        pn[o + 1] = make_float4(mx.x, mx.y, mx.z, pn[o + 1].w);
        __threadfence();
        if (atomicAdd(arrived + p, 1u) == 0u) return;   // the sibling subtree is not finished yet
        __threadfence();
        if (p == 0) return;                              // the root's own box is not stored anywhere
        const float3 smn = ld_agent3(pn + (2 - o)), smx = ld_agent3(pn + (3 - o));
        mn = min3(mn, smn); mx = max3(mx, smx);
        node = p;
    }
}

// ---- BVH8_CWBVH -----------------------------------------------------------------------------------

__global__ void k_cw_parents(const float4* __restrict__ nodes, uint32_t nNodes, uint32_t* __restrict__ parent) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nNodes) return;
    const uint32_t imask = as_u32(nodes[(size_t)j * 5].w) >> 24, base = as_u32(nodes[(size_t)j * 5 + 1].x);
    const uint32_t n = __popc(imask);
    for (uint32_t k = 0; k < n; k++) if (base + k < nNodes) parent[base + k] = j;
}

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
            if (c < nNodes) { a = ld_agent3(boxMin + c); b = ld_agent3(boxMax + c); }
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

__global__ void k_cw_refit(float4* __restrict__ nodes, uint32_t nNodes, const float4* __restrict__ tris, const float4* __restrict__ verts,
                           const uint32_t* __restrict__ parent, uint32_t* __restrict__ arrived, float4* __restrict__ boxMin, float4* __restrict__ boxMax) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nNodes) return;
    if (as_u32(nodes[(size_t)j * 5].w) >> 24) return;   // has interior children: finished by the last of them
    if (j != 0 && parent[j] == 0xffffffffu) return;      // not referenced by any node (hole in the blob)
    for (;;) {
        float3 mn, mx;
        cw_encode(nodes + (size_t)j * 5, nNodes, tris, verts, boxMin, boxMax, mn, mx);
        if (j == 0) return;
        boxMin[j] = make_float4(mn.x, mn.y, mn.z, 0.f); boxMax[j] = make_float4(mx.x, mx.y, mx.z, 0.f);
        __threadfence();
        const uint32_t p = parent[j];
        const uint32_t need = __popc(as_u32(nodes[(size_t)p * 5].w) >> 24);
        if (atomicAdd(arrived + p, 1u) + 1u < need) return;   // other interior children of p are still open
        __threadfence();
        j = p;
    }
}

}  // namespace

size_t refit_scratch_bytes(int layout, uint32_t nNodes) {
    // parent + arrived (u32 each) [+ boxMin, boxMax (float4 each) for CWBVH]
    return (size_t)nNodes * 8 + (layout == 9 ? (size_t)nNodes * 32 : 0) + 1024;
}

// scratch layout: parent[nNodes] | arrived[nNodes] | boxMin[nNodes] | boxMax[nNodes]; parentsValid says whether
// parent[] was already filled by an earlier refit of this scene (topology never changes)
hipError_t launch_refit(int layout, float4* nodes, uint32_t nNodes, float4* tris, uint64_t nTriRecords, const float4* verts, uint64_t nTris,
                        void* scratch, bool parentsValid, uint32_t* status, hipStream_t s) {
    uint32_t* parent = (uint32_t*)scratch;
    uint32_t* arrived = parent + nNodes;
    float4* boxMin = (float4*)(((uintptr_t)(arrived + nNodes) + 255) & ~(uintptr_t)255);
    float4* boxMax = boxMin + nNodes;
    const uint32_t bs = 128, nb = (nNodes + bs - 1) / bs;
    const uint32_t tb = (uint32_t)((nTriRecords + 255) / 256);
    hipError_t e = hipMemsetAsync(arrived, 0, (size_t)nNodes * 4, s);
    if (e == hipSuccess && !parentsValid) e = hipMemsetAsync(parent, 0xff, (size_t)nNodes * 4, s);
    if (e != hipSuccess) return e;
    if (layout == 9) {
        if (nTriRecords) hipLaunchKernelGGL(k_regather<true>, dim3(tb), dim3(256), 0, s, tris, verts, nTriRecords, nTris, status);
        if (!parentsValid) hipLaunchKernelGGL(k_cw_parents, dim3(nb), dim3(bs), 0, s, nodes, nNodes, parent);
        hipLaunchKernelGGL(k_cw_refit, dim3(nb), dim3(bs), 0, s, nodes, nNodes, tris, verts, parent, arrived, boxMin, boxMax);
    } else {
        if (nTriRecords) hipLaunchKernelGGL(k_regather<false>, dim3(tb), dim3(256), 0, s, tris, verts, nTriRecords, nTris, status);
        if (!parentsValid) hipLaunchKernelGGL(k_al_parents, dim3(nb), dim3(bs), 0, s, nodes, nNodes, parent);
        hipLaunchKernelGGL(k_al_refit, dim3(nb), dim3(bs), 0, s, nodes, nNodes, tris, verts, parent, arrived);
    }
    return hipGetLastError();
}

}  // namespace tbvh
