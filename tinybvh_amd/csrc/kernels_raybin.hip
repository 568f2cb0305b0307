// kernels_raybin.hip — reorder a resident ray batch into bins of (origin cell, direction octant).
//
// Why: an incoherent batch — the bounce rays of a path tracer from depth 2 on — reaches the traversal kernel in the
// order its parents finished, so the 64 rays of a wave and the thousands of waves in flight touch unrelated parts of the tree: 38
// L1-miss lines and 12.5 L2-miss lines per ray on the Bistro stand-in.  The launch consumes its batch as ONE front (ray_pool.h), so
// when neighbouring rays of the batch start in the same cell of space the whole GPU works on one region at a time: the lines it needs
// are in every XCD's L2, and the lanes of a wave share nodes again.  The reference has no counterpart (its wavefront.cl appends
// extension rays in the order Shade's work items finish, wavefront.cl:236-245).
//
// A counting sort in three launches, no comparison sort and no second ray buffer beyond the destination:
//   k_bin_count    key = Morton code of the origin's cell (2^b cells per axis over the scene bounds) | octant of the direction; one
//                  histogram atomic per ray (the counters are as many lines as there are bins / 32: no same-line serialisation beyond
//                  what the rays' own coherence causes);
//   exclusive scan of the histogram (hipcub);
//   k_bin_scatter  slot = atomicAdd(cursor[key]), the 64-byte record is copied to its slot; `perm` (optional) records where it came
//                  from, so a caller that needs results in the original order can gather them back.
// The order inside a bin is the order of the atomics (not reproducible run to run; the set of rays per bin is).
#include <hipcub/hipcub.hpp>

#include "device_common.h"
#include "kernels.h"

namespace tbvh {

namespace {

__device__ __forceinline__ uint32_t spread3(uint32_t v) {   // 10 bits -> every third bit
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__device__ __forceinline__ uint32_t bin_key(const RayRec& r, const RayBinArgs& a) {
    const float fx = (r.O.x - a.lo[0]) * a.scale[0], fy = (r.O.y - a.lo[1]) * a.scale[1], fz = (r.O.z - a.lo[2]) * a.scale[2];
    const float top = (float)((1u << a.cellBits) - 1u);
    const uint32_t cx = (uint32_t)__builtin_fminf(__builtin_fmaxf(fx, 0.f), top), cy = (uint32_t)__builtin_fminf(__builtin_fmaxf(fy, 0.f), top),
                   cz = (uint32_t)__builtin_fminf(__builtin_fmaxf(fz, 0.f), top);
    const uint32_t cell = a.cellBits ? (spread3(cx) | (spread3(cy) << 1) | (spread3(cz) << 2)) : 0u;
    const uint32_t oct = (r.D.x < 0 ? 4u : 0u) | (r.D.y < 0 ? 2u : 0u) | (r.D.z < 0 ? 1u : 0u);
    if (a.flags & 2u) return (oct << (3u * a.cellBits)) | cell;   // octant major: eight spatially sorted runs
    if (a.flags & 1u) return (cell << 3) | oct;                   // octant minor: eight runs per cell
    return cell;
}

__global__ __launch_bounds__(256) void k_bin_count(const RayRec* __restrict__ rays, uint64_t n, const unsigned long long* __restrict__ nDev, RayBinArgs a,
                                                   uint32_t* __restrict__ keys, uint32_t* __restrict__ hist) {
    const uint64_t total = nDev ? *nDev : n;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t k = bin_key(rays[i], a);
        keys[i] = k;
        atomicAdd(hist + k, 1u);
    }
}

__global__ __launch_bounds__(256) void k_bin_scatter(const RayRec* __restrict__ rays, uint64_t n, const unsigned long long* __restrict__ nDev,
                                                     const uint32_t* __restrict__ keys, uint32_t* __restrict__ cursor, RayRec* __restrict__ out,
                                                     uint32_t* __restrict__ perm) {
    const uint64_t total = nDev ? *nDev : n;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t slot = atomicAdd(cursor + keys[i], 1u);
        const RayRec* rp = rays + i;
        RayRec* op = out + slot;
        op->O = rp->O; op->D = rp->D; op->rD = rp->rD; op->hit = rp->hit;
        if (perm) perm[slot] = (uint32_t)i;
    }
}

}  // namespace

uint32_t ray_bin_count(uint32_t cellBits, uint32_t flags) { return 1u << (3u * cellBits + ((flags & 3u) ? 3u : 0u)); }

size_t ray_bin_scratch_bytes(uint64_t n, uint32_t cellBits, uint32_t flags, size_t* scanTempBytes) {
    const uint32_t bins = ray_bin_count(cellBits, flags);
    size_t tmp = 0;
    hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)bins);
    if (scanTempBytes) *scanTempBytes = tmp;
    return (size_t)n * 4 + (size_t)bins * 8 + ((tmp + 255) & ~(size_t)255) + 512;
}

hipError_t launch_ray_bin(const RayRec* in, RayRec* out, uint32_t* perm, uint64_t n, const unsigned long long* nDev, const RayBinArgs& a, void* scratch,
                          size_t scanTempBytes, uint32_t blocks, hipStream_t s) {
    const uint32_t bins = ray_bin_count(a.cellBits, a.flags);
    uint32_t* keys = (uint32_t*)scratch;
    uint32_t* hist = (uint32_t*)(((uintptr_t)(keys + n) + 255) & ~(uintptr_t)255);
    uint32_t* cursor = hist + bins;
    void* tmp = (void*)(((uintptr_t)(cursor + bins) + 255) & ~(uintptr_t)255);
    hipError_t e;
    if ((e = hipMemsetAsync(hist, 0, (size_t)bins * 4, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_bin_count, dim3(blocks), dim3(256), 0, s, in, n, nDev, a, keys, hist);
    if ((e = hipcub::DeviceScan::ExclusiveSum(tmp, scanTempBytes, hist, cursor, (int)bins, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_bin_scatter, dim3(blocks), dim3(256), 0, s, in, n, nDev, keys, cursor, out, perm);
    return hipGetLastError();
}

}  // namespace tbvh
