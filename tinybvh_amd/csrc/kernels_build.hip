// kernels_build.hip — BVH construction on the device: triangles -> LBVH (BVH2 in the reference's 32-byte
// BVHNode format) ("a GPU LBVH builder feeding the conversions").
//
// For content that changes topology every frame, or scenes whose host build (BVH::Build,
// tiny_bvh.h:2124-2461: seconds for Bistro) is the bottleneck: 63-bit Morton codes of the triangle
// centroids, radix sort, Karras 2012 topology, bottom-up boxes.  The result is written directly in the
// layout BVH::Build produces — 32-byte nodes {aabbMin, leftFirst, aabbMax, triCount}, root at 0, node 1
// unused, the two children of a node adjacent (tiny_bvh.h:1050-1062) — plus a primIdx array, so it feeds
// tbvh_convert_bvh2_device (kernels_convert.hip) and every other BVH2 consumer unchanged:
//   * Karras interior node i keeps its children at 2 + 2i and 3 + 2i;
//   * an interior node whose sorted range holds at most 3 triangles is emitted as a LEAF over that
//     range (leftFirst = range start, triCount = range size): the 3-triangle leaves CWBVH wants
//     (BVH::SplitLeafs(3) in reverse) at no cost, because a Karras node covers a contiguous sorted range.
// An LBVH is a lower-quality tree than the binned-SAH build (more nodes visited per ray); it is the
// fast path, not the default.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "cwbvh_encode.h"
#include "device_common.h"
#include "kernels.h"

namespace tbvh {

namespace {

__device__ __forceinline__ uint32_t enc_f32(float f) {   // order-preserving float -> uint
    const uint32_t b = as_u32(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec_f32(uint32_t e) { return as_f32((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e); }

__device__ __forceinline__ unsigned long long spread21(uint32_t v) {   // 21 bits -> every third bit of 63
    unsigned long long x = v & 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

// per-triangle box + centroid bounds of the scene.  Grid-stride over the triangles, wave reduction, then ONE set of six
// atomics per wave at the very end: the six words share a cache line, and same-line atomics are serialised memory-side
// (~12 ns each): one set per 64 triangles cost 3 ms for Bistro, one per wave of a 2048-block grid costs nothing.
__global__ void k_tri_boxes(const float4* __restrict__ verts, uint32_t n, float4* __restrict__ triMin, float4* __restrict__ triMax,
                            uint32_t* __restrict__ centreBounds) {
    float3 cmn = make_float3(1e30f, 1e30f, 1e30f), cmx = make_float3(-1e30f, -1e30f, -1e30f);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float3 mn = make_float3(1e30f, 1e30f, 1e30f), mx = make_float3(-1e30f, -1e30f, -1e30f);
        for (int k = 0; k < 3; k++) { const float4 v = verts[3 * (uint64_t)i + k]; const float3 p = make_float3(v.x, v.y, v.z); mn = min3(mn, p); mx = max3(mx, p); }
        triMin[i] = make_float4(mn.x, mn.y, mn.z, 0.f); triMax[i] = make_float4(mx.x, mx.y, mx.z, 0.f);
        const float3 c = make_float3(0.5f * (mn.x + mx.x), 0.5f * (mn.y + mx.y), 0.5f * (mn.z + mx.z));
        cmn = min3(cmn, c); cmx = max3(cmx, c);
    }
    for (int o = 32; o > 0; o >>= 1) {
        cmn = min3(cmn, make_float3(__shfl_xor(cmn.x, o), __shfl_xor(cmn.y, o), __shfl_xor(cmn.z, o)));
        cmx = max3(cmx, make_float3(__shfl_xor(cmx.x, o), __shfl_xor(cmx.y, o), __shfl_xor(cmx.z, o)));
    }
    if ((threadIdx.x & 63u) == 0 && cmn.x <= cmx.x) {
        atomicMin(centreBounds + 0, enc_f32(cmn.x)); atomicMin(centreBounds + 1, enc_f32(cmn.y)); atomicMin(centreBounds + 2, enc_f32(cmn.z));
        atomicMax(centreBounds + 3, enc_f32(cmx.x)); atomicMax(centreBounds + 4, enc_f32(cmx.y)); atomicMax(centreBounds + 5, enc_f32(cmx.z));
    }
}

__global__ void k_tri_morton(const float4* __restrict__ triMin, const float4* __restrict__ triMax, const uint32_t* __restrict__ centreBounds,
                             uint32_t n, unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 a = triMin[i], b = triMax[i];
    const float c[3] = {0.5f * (a.x + b.x), 0.5f * (a.y + b.y), 0.5f * (a.z + b.z)};
    uint32_t q[3];
    for (int k = 0; k < 3; k++) {
        const float lo = dec_f32(centreBounds[k]), hi = dec_f32(centreBounds[3 + k]);
        const float ext = hi - lo;
        float u = ext > 0 ? (c[k] - lo) / ext : 0.f;
        u = u < 0 ? 0.f : (u > 1 ? 1.f : u);
        const uint32_t v = (uint32_t)(u * 2097151.0f);
        q[k] = v > 2097151u ? 2097151u : v;
    }
    // 63-bit codes (21 bits per axis): with 30-bit codes a 2.8 M-triangle scene puts dozens of triangles into one
    // cell, and everything below that is split by array position instead of by space
    keys[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    vals[i] = i;
}

__device__ __forceinline__ int delta(const unsigned long long* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = keys[i], b = keys[j];
    return a == b ? 64 + __clz((uint32_t)(i ^ j)) : __clzll(a ^ b);
}

// Karras 2012.  Per interior node i: its two children (leaf k is numbered n - 1 + k), its sorted range.
// parent[c] = i, bit 31 set for the right child.
__global__ void k_topology(const unsigned long long* __restrict__ keys, uint32_t n, uint32_t* __restrict__ parent, uint2* __restrict__ children, uint2* __restrict__ range) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = (int)n;
    if (i >= N - 1) return;
    const int d = delta(keys, N, i, i + 1) - delta(keys, N, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(keys, N, i, i - d);
    int lmax = 2;
    while (delta(keys, N, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(keys, N, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, N, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
        if (delta(keys, N, i, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const uint32_t left = lo == gamma ? (uint32_t)(N - 1 + gamma) : (uint32_t)gamma;
    const uint32_t right = hi == gamma + 1 ? (uint32_t)(N - 1 + gamma + 1) : (uint32_t)(gamma + 1);
    children[i] = make_uint2(left, right);
    range[i] = make_uint2((uint32_t)lo, (uint32_t)(hi - lo + 1));
    parent[left] = (uint32_t)i;
    parent[right] = (uint32_t)i | 0x80000000u;
}

__device__ __forceinline__ float3 ld_agent3(const float4* p) {   // written by another CU during this kernel: bypass the non-coherent L1
    const float* f = (const float*)p;
    return make_float3(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(f + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                       __hip_atomic_load(f + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// position of a node in the output array: the root at 0, the children of Karras node p at 2 + 2p and 3 + 2p
__device__ __forceinline__ uint32_t out_pos(uint32_t parentEntry) { return 2u + 2u * (parentEntry & 0x7fffffffu) + (parentEntry >> 31); }

// Bottom-up boxes WITHOUT device-scope fences.  A Karras-style climb (the first child to arrive leaves, the
// second continues) needs a release/acquire pair per node, and on MI355X an agent-scope fence writes back and
// invalidates L2 (the 8 XCDs' L2s are not coherent with each other): ~1-2 us each, 25 ms for Bistro.  Instead:
//   pass 0      one thread per triangle writes its leaf node and box;
//   pass p >= 1 one thread per interior node that is not done yet: if both children were done BEFORE this pass
//               (done[] holds the pass number, kernel boundaries make earlier passes visible), take the union,
//               write the BVHNode record, mark done = p.
// A node at height h completes in pass h, so the number of passes is the tree height (40-60 for scenes, each
// pass ~10 us: it only reads 12 bytes per unfinished node).
__global__ void k_wald_leaves(const uint32_t* __restrict__ sortedTri, const float4* __restrict__ triMin, const float4* __restrict__ triMax,
                              const uint32_t* __restrict__ parent, float4* __restrict__ boxMin, float4* __restrict__ boxMax, uint32_t n,
                              float4* __restrict__ nodes32) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t tri = sortedTri[k];
    const float4 mn = triMin[tri], mx = triMax[tri];
    if (n == 1) {   // a single triangle: the root is the leaf
        nodes32[0] = make_float4(mn.x, mn.y, mn.z, as_f32(0u)); nodes32[1] = make_float4(mx.x, mx.y, mx.z, as_f32(1u));
        return;
    }
    const uint32_t node = n - 1 + k;      // Karras numbering: leaves after the n - 1 interior nodes
    float4* o = nodes32 + 2 * (size_t)out_pos(parent[node]);
    o[0] = make_float4(mn.x, mn.y, mn.z, as_f32(k)); o[1] = make_float4(mx.x, mx.y, mx.z, as_f32(1u));
    boxMin[node] = mn; boxMax[node] = mx;
}

__global__ void k_wald_pass(const uint32_t* __restrict__ parent, const uint2* __restrict__ children, const uint2* __restrict__ range,
                            uint32_t* __restrict__ done, float4* __restrict__ boxMin, float4* __restrict__ boxMax, uint32_t n, uint32_t maxLeaf,
                            uint32_t pass, float4* __restrict__ nodes32) {
    const uint32_t node = blockIdx.x * blockDim.x + threadIdx.x;
    if (node + 1 >= n) return;            // n - 1 interior nodes
    if (done[node]) return;
    const uint2 ch = children[node];
    const uint32_t dl = ch.x >= n - 1 ? 1u : done[ch.x], dr = ch.y >= n - 1 ? 1u : done[ch.y];   // leaves were written in pass 0
    if (dl == 0u || dr == 0u || dl >= pass || dr >= pass) return;
    const float4 lmn = boxMin[ch.x], lmx = boxMax[ch.x], rmn = boxMin[ch.y], rmx = boxMax[ch.y];
    const float3 mn = make_float3(fminf(lmn.x, rmn.x), fminf(lmn.y, rmn.y), fminf(lmn.z, rmn.z));
    const float3 mx = make_float3(fmaxf(lmx.x, rmx.x), fmaxf(lmx.y, rmx.y), fmaxf(lmx.z, rmx.z));
    const uint2 rg = range[node];
    const bool leaf = rg.y <= maxLeaf;   // a whole (contiguous) range of at most maxLeaf triangles: one leaf
    const uint32_t pos = node == 0 ? 0u : out_pos(parent[node]);
    float4* o = nodes32 + 2 * (size_t)pos;
    o[0] = make_float4(mn.x, mn.y, mn.z, as_f32(leaf ? rg.x : 2u + 2u * node));
    o[1] = make_float4(mx.x, mx.y, mx.z, as_f32(leaf ? rg.y : 0u));
    boxMin[node] = make_float4(mn.x, mn.y, mn.z, 0.f); boxMax[node] = make_float4(mx.x, mx.y, mx.z, 0.f);
    done[node] = pass;
}

// ---- PLOC (Meister & Bittner 2018: parallel locally-ordered clustering) ---------------------------------------------------------------
// The LBVH splits by Morton bits alone; PLOC builds the tree bottom-up by AGGLOMERATION: the clusters (first: the triangles, in Morton order)
// each look R positions left and right for the neighbour whose union with them has the smallest surface area; clusters that chose EACH OTHER
// merge into a new node; the array is compacted (order kept) and the step repeats until one cluster is left (~log_1.6 n steps).  Quality is
// that of a greedy agglomerative build restricted to a Morton window — close to binned SAH, well above LBVH — for a few sorts' worth of time.
//
// A cluster IS its future BVHNode record (32 bytes: aabbMin, leftFirst, aabbMax, triCount): a leaf cluster {box, position in primIdx, 1}, a
// merged one {union, 2 + 2p, 0} where p numbers the merges; the two records of a merged pair are written to out positions 2 + 2p and 3 + 2p at
// the moment they merge — the adjacent-children format of tiny_bvh.h:1050-1062 falls out, as with the LBVH above.  p and the compacted index come
// from ONE exclusive scan per step over (keeps, leads) packed into 64 bits: node numbering is deterministic.
// Ties: among equal areas a cluster picks by a symmetric hash of the pair (below), so of all pairs at the minimal area the one with the
// smallest hash is mutual: every step merges at least one pair.
constexpr int kPlocBlock = 256;

// Equal areas are the rule, not the exception, in modelled geometry (rows of identical quads): "the lowest index among equals" makes every
// cluster of such a row choose its LEFT neighbour — one mutual pair per row and step, chains instead of trees.  The tie is broken by a hash of
// the PAIR instead (the same value seen from both sides; lower index first, so it is a total order on pairs): along a row of equal areas a
// pair merges when its hash is below both neighbouring pairs' — a third of them per step.
__device__ __forceinline__ uint32_t pair_hash(uint32_t i, uint32_t j) {
    const uint32_t a = i < j ? i : j, b = i < j ? j : i;
    uint32_t h = a * 0x9E3779B1u ^ (b * 0x85EBCA6Bu + 0x7F4A7C15u);
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    return h;
}

template <int R>
__global__ __launch_bounds__(kPlocBlock) void k_ploc_nearest(const float4* __restrict__ cl, const uint32_t* __restrict__ count, uint32_t* __restrict__ nn) {
    __shared__ float tile[6][kPlocBlock + 2 * R];
    const uint32_t c = *count;
    const int base = (int)(blockIdx.x * kPlocBlock) - R;
    if ((uint32_t)(blockIdx.x * kPlocBlock) >= c) return;
    for (int t = threadIdx.x; t < kPlocBlock + 2 * R; t += kPlocBlock) {
        const int g = base + t;
        float4 mn = make_float4(0.f, 0.f, 0.f, 0.f), mx = mn;
        if (g >= 0 && (uint32_t)g < c) { mn = cl[2 * (size_t)g]; mx = cl[2 * (size_t)g + 1]; }
        tile[0][t] = mn.x; tile[1][t] = mn.y; tile[2][t] = mn.z; tile[3][t] = mx.x; tile[4][t] = mx.y; tile[5][t] = mx.z;
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * kPlocBlock + threadIdx.x;
    if (i >= c) return;
    const int me = (int)threadIdx.x + R;
    const float a0 = tile[0][me], a1 = tile[1][me], a2 = tile[2][me], a3 = tile[3][me], a4 = tile[4][me], a5 = tile[5][me];
    float best = 3.0e38f;
    uint32_t bestJ = 0xffffffffu, bestH = 0xffffffffu;
#pragma unroll 4
    for (int d = -R; d <= R; d++) {
        const int g = (int)i + d;
        if (d == 0 || g < 0 || (uint32_t)g >= c) continue;
        const int t = me + d;
        const float ex = fmaxf(a3, tile[3][t]) - fminf(a0, tile[0][t]), ey = fmaxf(a4, tile[4][t]) - fminf(a1, tile[1][t]), ez = fmaxf(a5, tile[5][t]) - fminf(a2, tile[2][t]);
        const float area = ex * ey + ey * ez + ez * ex;
        const uint32_t h = pair_hash(i, (uint32_t)g);
        if (area < best || (area == best && h < bestH)) { best = area; bestJ = (uint32_t)g; bestH = h; }
    }
    nn[i] = bestJ;
}

// keeps | leads << 32 per cluster: a cluster whose choice chose it back merges; the lower index of the pair leads (and stays), the other goes
__global__ void k_ploc_flags(const uint32_t* __restrict__ nn, const uint32_t* __restrict__ count, unsigned long long* __restrict__ flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t c = *count;
    if (i >= c) return;
    const uint32_t j = nn[i];
    const bool mutual = j < c && nn[j] == i;
    const unsigned long long keeps = (mutual && j < i) ? 0ull : 1ull, leads = (mutual && i < j) ? 1ull : 0ull;
    flags[i] = keeps | (leads << 32);
}

// counters: [0] clusters now, [1] merges so far (= next p), [2] clusters after this step, [3] merges after this step
__global__ void k_ploc_merge(const float4* __restrict__ cl, const uint32_t* __restrict__ nn, const unsigned long long* __restrict__ flags,
                             const unsigned long long* __restrict__ scan, uint32_t* __restrict__ counters, float4* __restrict__ clOut, float4* __restrict__ nodes32) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t c = counters[0], pBase = counters[1];
    if (i >= c) return;
    const unsigned long long f = flags[i], sc = scan[i];
    if (i == c - 1u) { counters[2] = (uint32_t)sc + (uint32_t)(f & 1ull); counters[3] = pBase + (uint32_t)(sc >> 32) + (uint32_t)(f >> 32); }
    if (!(f & 1ull)) return;
    const uint32_t to = (uint32_t)sc;
    const float4 amn = cl[2 * (size_t)i], amx = cl[2 * (size_t)i + 1];
    if (f >> 32) {
        const uint32_t j = nn[i], p = pBase + (uint32_t)(sc >> 32);
        const float4 bmn = cl[2 * (size_t)j], bmx = cl[2 * (size_t)j + 1];
        float4* o = nodes32 + 2 * (size_t)(2u + 2u * p);
        o[0] = amn; o[1] = amx; o[2] = bmn; o[3] = bmx;
        clOut[2 * (size_t)to] = make_float4(fminf(amn.x, bmn.x), fminf(amn.y, bmn.y), fminf(amn.z, bmn.z), as_f32(2u + 2u * p));
        clOut[2 * (size_t)to + 1] = make_float4(fmaxf(amx.x, bmx.x), fmaxf(amx.y, bmx.y), fmaxf(amx.z, bmx.z), as_f32(0u));
    } else { clOut[2 * (size_t)to] = amn; clOut[2 * (size_t)to + 1] = amx; }
}
__global__ void k_ploc_advance(uint32_t* __restrict__ counters) { counters[0] = counters[2]; counters[1] = counters[3]; }

__global__ void k_ploc_leaves(const uint32_t* __restrict__ sortedTri, const float4* __restrict__ triMin, const float4* __restrict__ triMax, uint32_t n, float4* __restrict__ cl) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t tri = sortedTri[k];
    const float4 mn = triMin[tri], mx = triMax[tri];
    cl[2 * (size_t)k] = make_float4(mn.x, mn.y, mn.z, as_f32(k)); cl[2 * (size_t)k + 1] = make_float4(mx.x, mx.y, mx.z, as_f32(1u));
}
__global__ void k_ploc_root(const float4* __restrict__ cl, float4* __restrict__ nodes32) {
    if (threadIdx.x == 0) { nodes32[0] = cl[0]; nodes32[1] = cl[1]; nodes32[2] = make_float4(0.f, 0.f, 0.f, 0.f); nodes32[3] = make_float4(0.f, 0.f, 0.f, 0.f); }
}

struct PlocScratch {
    float4 *triMin, *triMax, *clA, *clB;
    unsigned long long *keysA, *keysB, *flags, *scan;
    uint32_t *valsA, *nn, *bounds, *counters;
    void *sortTemp, *scanTemp;
    size_t total;
};
PlocScratch carve_ploc(void* base, uint32_t n, size_t sortTempBytes, size_t scanTempBytes) {
    char* p = (char*)base;
    auto take = [&](size_t bytes) { char* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
    PlocScratch s;
    s.triMin = (float4*)take((size_t)n * 16); s.triMax = (float4*)take((size_t)n * 16);
    s.clA = (float4*)take((size_t)n * 32); s.clB = (float4*)take((size_t)n * 32);
    s.keysA = (unsigned long long*)take((size_t)n * 8); s.keysB = (unsigned long long*)take((size_t)n * 8);
    s.flags = (unsigned long long*)take((size_t)n * 8); s.scan = (unsigned long long*)take((size_t)n * 8);
    s.valsA = (uint32_t*)take((size_t)n * 4); s.nn = (uint32_t*)take((size_t)n * 4);
    s.bounds = (uint32_t*)take(64); s.counters = (uint32_t*)take(64);
    s.sortTemp = take(sortTempBytes); s.scanTemp = take(scanTempBytes);
    s.total = (size_t)(p - (char*)base);
    return s;
}

struct Scratch {
    float4 *triMin, *triMax, *boxMin, *boxMax;
    unsigned long long *keysA, *keysB;
    uint32_t *valsA, *parent, *flags, *bounds;
    uint2 *children, *range;
    void* sortTemp;
    size_t total;
};
Scratch carve(void* base, uint32_t n, size_t sortTempBytes) {
    char* p = (char*)base;
    auto take = [&](size_t bytes) { char* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
    Scratch s;
    s.triMin = (float4*)take((size_t)n * 16); s.triMax = (float4*)take((size_t)n * 16);
    s.boxMin = (float4*)take((size_t)n * 32); s.boxMax = (float4*)take((size_t)n * 32);   // 2n - 1 Karras nodes
    s.keysA = (unsigned long long*)take((size_t)n * 8); s.keysB = (unsigned long long*)take((size_t)n * 8);
    s.valsA = (uint32_t*)take((size_t)n * 4);
    s.parent = (uint32_t*)take((size_t)n * 8);
    s.children = (uint2*)take((size_t)n * 8); s.range = (uint2*)take((size_t)n * 8);
    s.flags = (uint32_t*)take((size_t)n * 4);
    s.bounds = (uint32_t*)take(64);
    s.sortTemp = take(sortTempBytes);
    s.total = (size_t)(p - (char*)base);
    return s;
}

}  // namespace

size_t lbvh_scratch_bytes(uint32_t n, size_t* sortTempBytes) {
    size_t tmp = 0;
    hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n, 0, 63);
    *sortTempBytes = tmp;
    return carve(nullptr, n, tmp).total;
}

// verts: 3 float4 per triangle (device).  Out: nodes32 (2n BVHNode records; [1] unused), primIdx (n entries = the
// triangles in Morton order).  maxLeaf: 1..3 triangles per leaf.
hipError_t launch_lbvh_build(const float4* verts, uint32_t n, uint32_t maxLeaf, float4* nodes32, uint32_t* primIdx, void* scratch, size_t sortTempBytes,
                             hipStream_t s) {
    const Scratch sc = carve(scratch, n, sortTempBytes);
    hipError_t e;
    if ((e = hipMemsetAsync(sc.bounds, 0xff, 12, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(sc.bounds + 3, 0x00, 12, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(sc.flags, 0, (size_t)n * 4, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(nodes32, 0, 64, s)) != hipSuccess) return e;   // root + the unused node 1
    const uint32_t bs = 256, nb = (n + bs - 1) / bs;
    hipLaunchKernelGGL(k_tri_boxes, dim3(nb < 2048u ? nb : 2048u), dim3(bs), 0, s, verts, n, sc.triMin, sc.triMax, sc.bounds);
    hipLaunchKernelGGL(k_tri_morton, dim3(nb), dim3(bs), 0, s, sc.triMin, sc.triMax, sc.bounds, n, sc.keysA, sc.valsA);
    size_t tmp = sortTempBytes;
    if ((e = hipcub::DeviceRadixSort::SortPairs(sc.sortTemp, tmp, sc.keysA, sc.keysB, sc.valsA, primIdx, (int)n, 0, 63, s)) != hipSuccess) return e;
    if (n > 1) hipLaunchKernelGGL(k_topology, dim3(nb), dim3(bs), 0, s, sc.keysB, n, sc.parent, sc.children, sc.range);
    hipLaunchKernelGGL(k_wald_leaves, dim3(nb), dim3(bs), 0, s, primIdx, sc.triMin, sc.triMax, sc.parent, sc.boxMin, sc.boxMax, n, nodes32);
    // passes in batches; after each batch look at the root's done word (pass numbers start at 2: 1 means "leaf")
    uint32_t pass = 2, rootDone = n > 1 ? 0u : 1u;
    while (!rootDone) {
        for (int k = 0; k < 24; k++, pass++)
            hipLaunchKernelGGL(k_wald_pass, dim3(nb), dim3(bs), 0, s, sc.parent, sc.children, sc.range, sc.flags, sc.boxMin, sc.boxMax, n, maxLeaf, pass, nodes32);
        if ((e = hipMemcpyAsync(&rootDone, sc.flags, 4, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
        if (pass > 2u + 24u * 100000u) return hipErrorUnknown;
    }
    return hipGetLastError();
}

size_t ploc_scratch_bytes(uint32_t n, size_t* sortTempBytes, size_t* scanTempBytes) {
    size_t tmp = 0, tmp2 = 0;
    hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n, 0, 63);
    hipcub::DeviceScan::ExclusiveSum(nullptr, tmp2, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, (int)n);
    *sortTempBytes = tmp; *scanTempBytes = tmp2;
    return carve_ploc(nullptr, n, tmp, tmp2).total;
}

// PLOC build (see above).  Out as launch_lbvh_build: nodes32 (2n BVHNode records, [1] unused), primIdx (the triangles in Morton order); one
// triangle per leaf.  radius: search window to each side (8, 16 or 32).  steps (optional): number of clustering steps taken.
hipError_t launch_ploc_build(const float4* verts, uint32_t n, uint32_t radius, float4* nodes32, uint32_t* primIdx, void* scratch, size_t sortTempBytes,
                             size_t scanTempBytes, hipStream_t s, uint32_t* steps) {
    const PlocScratch sc = carve_ploc(scratch, n, sortTempBytes, scanTempBytes);
    hipError_t e;
    if ((e = hipMemsetAsync(sc.bounds, 0xff, 12, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(sc.bounds + 3, 0x00, 12, s)) != hipSuccess) return e;
    const uint32_t bs = 256, nb = (n + bs - 1) / bs;
    hipLaunchKernelGGL(k_tri_boxes, dim3(nb < 2048u ? nb : 2048u), dim3(bs), 0, s, verts, n, sc.triMin, sc.triMax, sc.bounds);
    hipLaunchKernelGGL(k_tri_morton, dim3(nb), dim3(bs), 0, s, sc.triMin, sc.triMax, sc.bounds, n, sc.keysA, sc.valsA);
    size_t tmp = sortTempBytes;
    if ((e = hipcub::DeviceRadixSort::SortPairs(sc.sortTemp, tmp, sc.keysA, sc.keysB, sc.valsA, primIdx, (int)n, 0, 63, s)) != hipSuccess) return e;
    float4 *cur = sc.clA, *nxt = sc.clB;
    hipLaunchKernelGGL(k_ploc_leaves, dim3(nb), dim3(bs), 0, s, primIdx, sc.triMin, sc.triMax, n, cur);
    uint32_t host[4] = {n, 0u, n, 0u};
    if ((e = hipMemcpyAsync(sc.counters, host, 16, hipMemcpyHostToDevice, s)) != hipSuccess) return e;
    uint32_t c = n, nSteps = 0;
    while (c > 1u) {
        const uint32_t g = (c + kPlocBlock - 1) / kPlocBlock;
        if (radius <= 8u) hipLaunchKernelGGL(k_ploc_nearest<8>, dim3(g), dim3(kPlocBlock), 0, s, cur, sc.counters, sc.nn);
        else if (radius <= 16u) hipLaunchKernelGGL(k_ploc_nearest<16>, dim3(g), dim3(kPlocBlock), 0, s, cur, sc.counters, sc.nn);
        else hipLaunchKernelGGL(k_ploc_nearest<32>, dim3(g), dim3(kPlocBlock), 0, s, cur, sc.counters, sc.nn);
        hipLaunchKernelGGL(k_ploc_flags, dim3(g), dim3(kPlocBlock), 0, s, sc.nn, sc.counters, sc.flags);
        size_t t2 = scanTempBytes;
        if ((e = hipcub::DeviceScan::ExclusiveSum(sc.scanTemp, t2, sc.flags, sc.scan, (int)c, s)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_ploc_merge, dim3(g), dim3(kPlocBlock), 0, s, cur, sc.nn, sc.flags, sc.scan, sc.counters, nxt, nodes32);
        hipLaunchKernelGGL(k_ploc_advance, dim3(1), dim3(1), 0, s, sc.counters);
        if ((e = hipMemcpyAsync(host, sc.counters, 16, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
        if (host[0] >= c || host[0] == 0u) return hipErrorUnknown;   // every step merges at least one pair
        c = host[0];
        float4* t = cur; cur = nxt; nxt = t;
        nSteps++;
    }
    hipLaunchKernelGGL(k_ploc_root, dim3(1), dim3(64), 0, s, cur, nodes32);
    if (steps) *steps = nSteps;
    return hipGetLastError();
}

}  // namespace tbvh
