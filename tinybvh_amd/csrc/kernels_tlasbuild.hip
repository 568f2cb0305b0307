// kernels_tlasbuild.hip — per-frame TLAS rebuild on the device.
//
// Replaces, for a TLAS that already lives on the GPU, the host work of the reference's frame loop
// (tiny_bvh_gpu2.cpp:113-130): BLASInstance::Update for every instance (tiny_bvh.h:8386-8427:
// invert the transform, world box of the 8 transformed BLAS-box corners) and
// BVH::Build(BLASInstance*, ...) (tiny_bvh.h:2221-2259) followed by BVH_GPU::ConvertFrom
// (4612-4655).  The instance records keep the reference's 192-byte format and the result is a
// BVH_GPU (Aila-Laine) node array + instance index list, exactly what k_tlas traverses.
//
// The tree is an LBVH (30-bit Morton codes of the instance-box centres, radix sort, Karras 2012
// topology in one pass, bottom-up boxes with one atomic flag per interior node) instead of the
// reference's binned-SAH build: a different but valid TLAS — hit records do not depend on the TLAS
// shape (up to exact-distance ties), which tests/test_tlas_device_build.py checks against the
// oracle.  One leaf per instance, like the reference (TLAS leaves hold exactly one instance after
// its build: tiny_bvh.h:2250-2257).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdio>

#include "device_common.h"
#include "kernels.h"

namespace tbvh {

namespace {

__device__ __forceinline__ uint32_t enc_f32(float f) {   // order-preserving float -> uint
    const uint32_t b = as_u32(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec_f32(uint32_t e) {
    const uint32_t b = (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e;
    return as_f32(b);
}

// BLASInstance::InvertTransform (tiny_bvh.h:8402-8427): the float cofactor formula, evaluated operation for operation
// like the reference build — the same table and order as host_builder.cpp: invert4x4, where the order was found by
// exhaustive search against the real reference.  Host-built, device-built and tinybvh-built records are bit-identical.
struct CofTerm { signed char s; unsigned char a, b, c; };   // s * T[a] * T[b] * T[c]
__device__ const CofTerm kCof[16][6] = {
    {{+1, 5, 10, 15}, {-1, 5, 11, 14}, {-1, 9, 6, 15}, {+1, 9, 7, 14}, {+1, 13, 6, 11}, {-1, 13, 7, 10}},
    {{-1, 1, 10, 15}, {+1, 1, 11, 14}, {+1, 9, 2, 15}, {-1, 9, 3, 14}, {-1, 13, 2, 11}, {+1, 13, 3, 10}},
    {{+1, 1, 6, 15}, {-1, 1, 7, 14}, {-1, 5, 2, 15}, {+1, 5, 3, 14}, {+1, 13, 2, 7}, {-1, 13, 3, 6}},
    {{-1, 1, 6, 11}, {+1, 1, 7, 10}, {+1, 5, 2, 11}, {-1, 5, 3, 10}, {-1, 9, 2, 7}, {+1, 9, 3, 6}},
    {{-1, 4, 10, 15}, {+1, 4, 11, 14}, {+1, 8, 6, 15}, {-1, 8, 7, 14}, {-1, 12, 6, 11}, {+1, 12, 7, 10}},
    {{+1, 0, 10, 15}, {-1, 0, 11, 14}, {-1, 8, 2, 15}, {+1, 8, 3, 14}, {+1, 12, 2, 11}, {-1, 12, 3, 10}},
    {{-1, 0, 6, 15}, {+1, 0, 7, 14}, {+1, 4, 2, 15}, {-1, 4, 3, 14}, {-1, 12, 2, 7}, {+1, 12, 3, 6}},
    {{+1, 0, 6, 11}, {-1, 0, 7, 10}, {-1, 4, 2, 11}, {+1, 4, 3, 10}, {+1, 8, 2, 7}, {-1, 8, 3, 6}},
    {{+1, 4, 9, 15}, {-1, 4, 11, 13}, {-1, 8, 5, 15}, {+1, 8, 7, 13}, {+1, 12, 5, 11}, {-1, 12, 7, 9}},
    {{-1, 0, 9, 15}, {+1, 0, 11, 13}, {+1, 8, 1, 15}, {-1, 8, 3, 13}, {-1, 12, 1, 11}, {+1, 12, 3, 9}},
    {{+1, 0, 5, 15}, {-1, 0, 7, 13}, {-1, 4, 1, 15}, {+1, 4, 3, 13}, {+1, 12, 1, 7}, {-1, 12, 3, 5}},
    {{-1, 0, 5, 11}, {+1, 0, 7, 9}, {+1, 4, 1, 11}, {-1, 4, 3, 9}, {-1, 8, 1, 7}, {+1, 8, 3, 5}},
    {{-1, 4, 9, 14}, {+1, 4, 10, 13}, {+1, 8, 5, 14}, {-1, 8, 6, 13}, {-1, 12, 5, 10}, {+1, 12, 6, 9}},
    {{+1, 0, 9, 14}, {-1, 0, 10, 13}, {-1, 8, 1, 14}, {+1, 8, 2, 13}, {+1, 12, 1, 10}, {-1, 12, 2, 9}},
    {{-1, 0, 5, 14}, {+1, 0, 6, 13}, {+1, 4, 1, 14}, {-1, 4, 2, 13}, {-1, 12, 1, 6}, {+1, 12, 2, 5}},
    {{+1, 0, 5, 10}, {-1, 0, 6, 9}, {-1, 4, 1, 10}, {+1, 4, 2, 9}, {+1, 8, 1, 6}, {-1, 8, 2, 5}},
};
__device__ bool invert4x4(const float* T, float* iT) {
    for (int k = 0; k < 16; k++) {
        const CofTerm* t = kCof[k];
        float m[6];
        for (int j = 0; j < 6; j++) m[j] = T[t[j].a] * T[t[j].b];
        const int r = k < 8 ? 0 : 1;   // the term that is a plain (rounded) product
        const float tr = m[r] * T[t[r].c];
        float s = t[r].s > 0 ? tr : -tr;
        for (int j = 0; j < 6; j++) if (j != r) s = __builtin_fmaf(t[j].s > 0 ? m[j] : -m[j], T[t[j].c], s);
        iT[k] = s;
    }
    const float p1 = T[1] * iT[4];
    float det = __builtin_fmaf(T[0], iT[0], p1);
    det = __builtin_fmaf(T[2], iT[8], det);
    det = __builtin_fmaf(T[3], iT[12], det);
    if (det == 0) return false;
    const float invdet = 1.0f / det;
    for (int i = 0; i < 16; i++) iT[i] *= invdet;
    return true;
}

// One thread per instance: BLASInstance::Update.  Record layout (12 float4): transform rows 0-3,
// invTransform rows 4-7, {aabbMin, blasIdx}, {aabbMax, mask}, 2 x padding.
__global__ void k_instance_update(float4* __restrict__ instances, const float* __restrict__ transforms, const float* __restrict__ blasBounds,
                                  uint32_t n, uint32_t nBlas, float4* __restrict__ instMin, float4* __restrict__ instMax,
                                  uint32_t* __restrict__ centreBounds) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float cLo[3] = {1e30f, 1e30f, 1e30f}, cHi[3] = {-1e30f, -1e30f, -1e30f};   // this lane's centre, reduced over the wave below
    if (i < n) {
    float4* rec = instances + (size_t)i * 12;
    float T[16], inv[16];
    if (transforms) {
        for (int k = 0; k < 16; k++) T[k] = transforms[(size_t)i * 16 + k];
        for (int r = 0; r < 4; r++) rec[r] = make_float4(T[r * 4], T[r * 4 + 1], T[r * 4 + 2], T[r * 4 + 3]);
    } else {
        for (int r = 0; r < 4; r++) { const float4 v = rec[r]; T[r * 4] = v.x; T[r * 4 + 1] = v.y; T[r * 4 + 2] = v.z; T[r * 4 + 3] = v.w; }
    }
    invert4x4(T, inv);   // a singular transform leaves the unscaled cofactors behind, as in the reference ("invert failed. That's bad.")
    for (int r = 0; r < 4; r++) rec[4 + r] = make_float4(inv[r * 4], inv[r * 4 + 1], inv[r * 4 + 2], inv[r * 4 + 3]);
    const uint32_t blasIdx = as_u32(rec[8].w);
    const float* bb = blasBounds + (size_t)(blasIdx < nBlas ? blasIdx : 0u) * 6;
    float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
    for (int j = 0; j < 8; j++) {
        const float p[3] = {(j & 1) ? bb[3] : bb[0], (j & 2) ? bb[4] : bb[1], (j & 4) ? bb[5] : bb[2]};
        float t[3];
        // tinybvh_transform_point (tiny_bvh.h:512-522) as the reference build contracts it (host_builder.cpp: update_instance)
        for (int r = 0; r < 3; r++) t[r] = __builtin_fmaf(T[r * 4 + 2], p[2], __builtin_fmaf(T[r * 4], p[0], T[r * 4 + 1] * p[1])) + T[r * 4 + 3];
        const float ww = __builtin_fmaf(T[14], p[2], __builtin_fmaf(T[12], p[0], T[13] * p[1])) + T[15];
        if (ww != 1.0f) { const float r = 1.0f / ww; t[0] *= r; t[1] *= r; t[2] *= r; }
        for (int a = 0; a < 3; a++) { mn[a] = t[a] < mn[a] ? t[a] : mn[a]; mx[a] = t[a] > mx[a] ? t[a] : mx[a]; }
    }
    rec[8] = make_float4(mn[0], mn[1], mn[2], rec[8].w);
    rec[9] = make_float4(mx[0], mx[1], mx[2], rec[9].w);
    instMin[i] = make_float4(mn[0], mn[1], mn[2], 0.f);
    instMax[i] = make_float4(mx[0], mx[1], mx[2], 0.f);
    for (int a = 0; a < 3; a++) cLo[a] = cHi[a] = 0.5f * (mn[a] + mx[a]);
    }
    // one set of six atomics per wave: the six words share a cache line and same-line atomics are serialised memory-side
    // (~12 ns each); one set per instance made them 70 of the 80 us of a 1000-instance rebuild
    for (int a = 0; a < 3; a++)
        for (int o = 32; o > 0; o >>= 1) { cLo[a] = fminf(cLo[a], __shfl_xor(cLo[a], o)); cHi[a] = fmaxf(cHi[a], __shfl_xor(cHi[a], o)); }
    if ((threadIdx.x & 63u) == 0 && cLo[0] <= cHi[0])
        for (int a = 0; a < 3; a++) { atomicMin(centreBounds + a, enc_f32(cLo[a])); atomicMax(centreBounds + 3 + a, enc_f32(cHi[a])); }
}

__device__ __forceinline__ uint32_t spread10(uint32_t v) {   // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ void k_morton(const float4* __restrict__ instMin, const float4* __restrict__ instMax, const uint32_t* __restrict__ centreBounds,
                         uint32_t n, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 a = instMin[i], b = instMax[i];
    const float c[3] = {0.5f * (a.x + b.x), 0.5f * (a.y + b.y), 0.5f * (a.z + b.z)};
    uint32_t q[3];
    for (int k = 0; k < 3; k++) {
        const float lo = dec_f32(centreBounds[k]), hi = dec_f32(centreBounds[3 + k]);
        const float ext = hi - lo;
        float u = ext > 0 ? (c[k] - lo) / ext : 0.f;
        u = u < 0 ? 0.f : (u > 1 ? 1.f : u);
        const uint32_t v = (uint32_t)(u * 1023.0f);
        q[k] = v > 1023u ? 1023u : v;
    }
    keys[i] = (spread10(q[0]) << 2) | (spread10(q[1]) << 1) | spread10(q[2]);
    vals[i] = i;
}

// Karras 2012, "Maximizing parallelism in the construction of BVHs, octrees and k-d trees":
// interior node i covers a range of the sorted keys found by binary search on common-prefix
// lengths; equal keys are told apart by their position.
__device__ __forceinline__ int delta(const uint32_t* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint32_t a = keys[i], b = keys[j];
    return a == b ? 32 + __clz((uint32_t)(i ^ j)) : __clz(a ^ b);
}

// Node numbering of the result: interior node i -> node i (root = 0), leaf k -> node (n - 1) + k.
__global__ void k_lbvh_topology(const uint32_t* __restrict__ keys, uint32_t n, uint32_t* __restrict__ parent, uint2* __restrict__ children) {
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
    parent[left] = (uint32_t)i;
    parent[right] = (uint32_t)i;
}

__device__ __forceinline__ float4 ld_agent(const float4* p) {   // bypass this CU's (non-coherent) L1: a sibling on another CU wrote it
    const float* f = (const float*)p;
    return make_float4(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(f + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                       __hip_atomic_load(f + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), 0.f);
}

// One thread per leaf: write the leaf node, then climb; the second thread to reach an interior
// node owns it (both child boxes are complete by then), writes its BVH_GPU record and goes on.
__global__ void k_lbvh_nodes(const uint32_t* __restrict__ sortedIdx, const float4* __restrict__ instMin, const float4* __restrict__ instMax,
                             const uint32_t* __restrict__ parent, const uint2* __restrict__ children, uint32_t* __restrict__ flags,
                             float4* __restrict__ boxMin, float4* __restrict__ boxMax, uint32_t n, float4* __restrict__ nodes,
                             uint32_t* __restrict__ tlasIdx) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t inst = sortedIdx[k];
    tlasIdx[k] = inst;
    const uint32_t leaf = n - 1 + k;
    const float4 mn = instMin[inst], mx = instMax[inst];
    // leaf record: triCount = 1 (n2.w), firstTri = k (n3.w); with a single instance the leaf IS the root (node 0)
    float4* ln = nodes + (size_t)(n == 1 ? 0u : leaf) * 4;
    ln[0] = make_float4(0, 0, 0, 0); ln[1] = make_float4(0, 0, 0, 0);
    ln[2] = make_float4(0, 0, 0, as_f32(1u)); ln[3] = make_float4(0, 0, 0, as_f32(k));
    if (n == 1) return;
    boxMin[leaf] = mn; boxMax[leaf] = mx;
    __threadfence();
    uint32_t node = parent[leaf];
    for (;;) {
        if (atomicAdd(flags + node, 1u) == 0u) return;   // first to arrive: the sibling subtree is not finished yet
        __threadfence();
        const uint2 ch = children[node];
        const float4 lmn = ld_agent(boxMin + ch.x), lmx = ld_agent(boxMax + ch.x), rmn = ld_agent(boxMin + ch.y), rmx = ld_agent(boxMax + ch.y);
        float4* o = nodes + (size_t)node * 4;
        o[0] = make_float4(lmn.x, lmn.y, lmn.z, as_f32(ch.x));
        o[1] = make_float4(lmx.x, lmx.y, lmx.z, as_f32(ch.y));
        o[2] = make_float4(rmn.x, rmn.y, rmn.z, as_f32(0u));
        o[3] = make_float4(rmx.x, rmx.y, rmx.z, as_f32(0u));
        if (node == 0) return;
        boxMin[node] = make_float4(fminf(lmn.x, rmn.x), fminf(lmn.y, rmn.y), fminf(lmn.z, rmn.z), 0.f);
        boxMax[node] = make_float4(fmaxf(lmx.x, rmx.x), fmaxf(lmx.y, rmx.y), fmaxf(lmx.z, rmx.z), 0.f);
        __threadfence();
        node = parent[node];
    }
}

}  // namespace

namespace {
struct Scratch {
    float4 *instMin, *instMax, *boxMin, *boxMax;
    uint32_t *keysA, *keysB, *valsA, *valsB, *parent, *flags, *bounds;
    uint2* children;
    void* sortTemp;
    size_t total;
};
// one allocation, carved up here (every part 256-byte aligned); base may be null to just measure
Scratch carve(void* base, uint32_t n, size_t sortTempBytes) {
    char* p = (char*)base;
    auto take = [&](size_t bytes) { char* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
    Scratch s;
    s.instMin = (float4*)take((size_t)n * 16); s.instMax = (float4*)take((size_t)n * 16);
    s.boxMin = (float4*)take((size_t)n * 32);  s.boxMax = (float4*)take((size_t)n * 32);   // 2n - 1 nodes
    s.keysA = (uint32_t*)take((size_t)n * 4); s.keysB = (uint32_t*)take((size_t)n * 4);
    s.valsA = (uint32_t*)take((size_t)n * 4); s.valsB = (uint32_t*)take((size_t)n * 4);
    s.parent = (uint32_t*)take((size_t)n * 8);
    s.children = (uint2*)take((size_t)n * 8);
    s.flags = (uint32_t*)take((size_t)n * 4);
    s.bounds = (uint32_t*)take(64);
    s.sortTemp = take(sortTempBytes);
    s.total = (size_t)(p - (char*)base);
    return s;
}
}  // namespace

size_t tlas_build_scratch_bytes(uint32_t n, size_t* sortTempBytes) {
    size_t tmp = 0;
    hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n, 0, 30);
    *sortTempBytes = tmp;
    return carve(nullptr, n, tmp).total;
}

hipError_t launch_tlas_rebuild(float4* tlasNodes, uint32_t* tlasIdx, float4* instances, const float* transformsDev, const float* blasBoundsDev,
                               uint32_t n, uint32_t nBlas, void* scratch, size_t sortTempBytes, hipStream_t s) {
    const Scratch sc = carve(scratch, n, sortTempBytes);
    float4 *instMin = sc.instMin, *instMax = sc.instMax, *boxMin = sc.boxMin, *boxMax = sc.boxMax;
    uint32_t *keysA = sc.keysA, *keysB = sc.keysB, *valsA = sc.valsA, *valsB = sc.valsB, *parent = sc.parent, *flags = sc.flags, *bounds = sc.bounds;
    uint2* children = sc.children;
    void* sortTemp = sc.sortTemp;
    hipError_t e;
    if ((e = hipMemsetAsync(bounds, 0xff, 12, s)) != hipSuccess) return e;       // centre minima: +max in the ordered encoding
    if ((e = hipMemsetAsync(bounds + 3, 0x00, 12, s)) != hipSuccess) return e;    // centre maxima
    if ((e = hipMemsetAsync(flags, 0, (size_t)n * 4, s)) != hipSuccess) return e;
    const uint32_t bs = 128, nb = (n + bs - 1) / bs;
#define TBVH_STEP(what) do { if ((e = hipGetLastError()) != hipSuccess) { fprintf(stderr, "[tinybvh_amd] TLAS rebuild: %s: %s\n", what, hipGetErrorString(e)); return e; } } while (0)
    hipLaunchKernelGGL(k_instance_update, dim3(nb), dim3(bs), 0, s, instances, transformsDev, blasBoundsDev, n, nBlas, instMin, instMax, bounds);
    TBVH_STEP("instance update");
    hipLaunchKernelGGL(k_morton, dim3(nb), dim3(bs), 0, s, instMin, instMax, bounds, n, keysA, valsA);
    TBVH_STEP("morton codes");
    size_t tmp = sortTempBytes;
    if ((e = hipcub::DeviceRadixSort::SortPairs(sortTemp, tmp, keysA, keysB, valsA, valsB, (int)n, 0, 30, s)) != hipSuccess) { fprintf(stderr, "[tinybvh_amd] TLAS rebuild: radix sort (%zu temp bytes): %s\n", sortTempBytes, hipGetErrorString(e)); return e; }
    if (n > 1) hipLaunchKernelGGL(k_lbvh_topology, dim3(nb), dim3(bs), 0, s, keysB, n, parent, children);
    TBVH_STEP("topology");
    hipLaunchKernelGGL(k_lbvh_nodes, dim3(nb), dim3(bs), 0, s, valsB, instMin, instMax, parent, children, flags, boxMin, boxMax, n, tlasNodes, tlasIdx);
    TBVH_STEP("nodes");
#undef TBVH_STEP
    return hipGetLastError();
}

}  // namespace tbvh
