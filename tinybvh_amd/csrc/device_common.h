// device_common.h — shared device-side definitions for the gfx950 traversal kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tbvh {

constexpr float kFar = 1e30f;  // BVH_FAR, tiny_bvh.h:144

// 64-byte ray record == first 64 bytes of tinybvh::Ray (tiny_bvh.h:689-709) == device
// struct Ray (traverse.cl:11-17).  Read as four 16-byte loads per lane.
struct __attribute__((aligned(16))) RayRec {
    float4 O;   // xyz, w = mask (u32)
    float4 D;   // xyz, w = instIdx (u32)
    float4 rD;  // xyz, w = hit.inst (u32) / pad
    float4 hit; // t, u, v, prim (u32)
};
static_assert(sizeof(RayRec) == 64, "ray record is 64 bytes");

__device__ __forceinline__ uint32_t as_u32(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float as_f32(uint32_t u) { return __uint_as_float(u); }

// Ray/triangle test with the arithmetic of MOLLER_TRUMBORE_TEST (tiny_bvh.h:1644-1656) and
// the explicit FMA contraction pattern shared with oracle/tbvh_oracle.c (orc_tri), which is
// the pattern g++ 11.4 -O3 -mfma picks for the reference build:
//   h = crossA(D,e2); a = dot_zxy(e1,h); u = f*dot_zxy(s,h); q = crossB(s,e1);
//   v = f*dot_zyx(D,q); t = f*dot_zxy(e2,q)
// The kernels are compiled with -ffp-contract=off so nothing else in here is fused, and
// 1/a is the correctly rounded IEEE division (hipcc default).  t,u,v are therefore
// bit-identical to the oracle, and to BVH::Intersect as built by oracle/Makefile.
struct TriHit { float t, u, v; };

__device__ __forceinline__ float3 crossA(float3 a, float3 b) {  // first product fused
    float3 r;
    r.x = __builtin_fmaf(a.y, b.z, -(a.z * b.y));
    r.y = __builtin_fmaf(a.z, b.x, -(a.x * b.z));
    r.z = __builtin_fmaf(a.x, b.y, -(a.y * b.x));
    return r;
}
__device__ __forceinline__ float3 crossB(float3 a, float3 b) {  // second product fused
    float3 r;
    r.x = __builtin_fmaf(-a.z, b.y, a.y * b.z);
    r.y = __builtin_fmaf(-a.x, b.z, a.z * b.x);
    r.z = __builtin_fmaf(-a.y, b.x, a.x * b.y);
    return r;
}
__device__ __forceinline__ float dot_zxy(float3 a, float3 b) {
    return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.x, b.x, a.y * b.y));
}
__device__ __forceinline__ float dot_zyx(float3 a, float3 b) {
    return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x));
}

// Opacity micromaps (BVHBase::SetOpacityMicroMaps, tiny_bvh.h:823-826): n x n bits per triangle; a hit whose
// barycentrics fall on a clear bit is no hit (IntersectTri / TriOccludes, tiny_bvh.h:8514-8522, 8562-8570).
struct Omm { const uint32_t* map; uint32_t n; };   // map == nullptr: none
__device__ __forceinline__ bool omm_opaque(Omm om, uint32_t prim, float u, float v) {
    const float fN = (float)om.n;
    const int row = (int)((u + v) * fN), diag = (int)((1 - u) * fN);
    const int idx = (row * row) + (int)(v * fN) + (diag - ((int)om.n - 1 - row));
    const uint32_t* m = om.map + (size_t)prim * ((om.n * om.n + 31u) >> 5);
    return (m[idx >> 5] >> (idx & 31)) & 1u;
}

// Returns true when the triangle {v0, e1, e2} is hit within [0, tmax] (both ends
// inclusive, like the reference: rejects only t < 0 || t > tmax) and, with opacity
// micromaps set, the hit point is opaque.
__device__ __forceinline__ bool tri_test(float3 O, float3 D, float3 v0, float3 e1, float3 e2,
                                         float tmax, TriHit& h, Omm om = Omm{nullptr, 0u}, uint32_t prim = 0u) {
    const float3 hh = crossA(D, e2);
    const float a = dot_zxy(e1, hh);
    if (__builtin_fabsf(a) < 0.000001f) return false;
    const float f = 1.0f / a;
    const float3 s = make_float3(O.x - v0.x, O.y - v0.y, O.z - v0.z);
    const float u = f * dot_zxy(s, hh);
    const float3 q = crossB(s, e1);
    const float v = f * dot_zyx(D, q);
    if (u < 0 || v < 0 || u + v > 1) return false;
    const float t = f * dot_zxy(e2, q);
    if (t < 0 || t > tmax) return false;
    if (om.map && !omm_opaque(om, prim, u, v)) return false;
    h.t = t; h.u = u; h.v = v;
    return true;
}

// Put right after the three loads of a triangle record: all three go out together.  Left alone the compiler sinks the third (v0, first needed behind
// the determinant test) into that branch — one load fewer for an edge-on triangle, a second memory round trip for every other one.
__device__ __forceinline__ void tri_loads_together(const float4& v0) { asm volatile("" :: "v"(v0.x), "v"(v0.y), "v"(v0.z), "v"(v0.w)); }

// Ballots of a bool straight from the compare that made it (HIP's __ballot takes an int: bool -> 0 / 1 -> compare again, two vector instructions
// per ballot), and lane counts as 32-bit SCALARS (the 64-bit value __popcll returns is compared with vector instructions: the scalar unit has no
// 64-bit ordered compare).
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ uint32_t wave_count(bool p) {
    const unsigned long long m = wave_ballot(p);
    uint32_t n;
    asm("s_bcnt1_i32_b64 %0, %1" : "=s"(n) : "s"(m) : "scc");
    return n;
}

__device__ __forceinline__ float3 xyz(float4 v) { return make_float3(v.x, v.y, v.z); }

// The library's tie rule.  tri_test has established t <= best.x; the candidate replaces the closest hit so far iff it is the
// first hit of this traversal, strictly closer, or equally far with the smaller primitive index (two-level kernels: then the smaller
// instance).  Every candidate at the final distance is tested whatever the visit order (nodes at tmin == best are visited: inclusive slab
// test), so the reported triangle does not depend on the schedule, on split rays or on the layout's child order.  The reference lets the
// LATER test win (tiny_bvh.h:1656: t <= ray.hit.t is accepted), i.e. its result among exactly equal t depends on the traversal order, and
// its own layouts disagree with each other there; oracle/tbvh_oracle.c restates both rules (orc_set_tie_rule).
// Box tests cull against a bound 2^-20 (eight ulps) beyond the closest hit so far.  A triangle lying IN a face of its leaf box — every
// axis-aligned wall — has its distance computed twice, by the slab test and by the triangle test, and the two round differently: with the
// exact bound, whether such a triangle is still tested once a hit within an ulp or two of it has been found depends on which was found
// first, i.e. on the traversal order (in the reference as well: tests/test_full_size.py).  With the slack every candidate within a few ulps of
// the closest hit is tested and the tie rule below picks the same one whatever the order; tri_test still rejects t > hit.t exactly, so no
// record gets farther, and a record can only get CLOSER (by ulps) than one the exact bound would have produced.
__device__ __forceinline__ float cull_bound(float t) { return t * 1.00000095367431640625f; }

__device__ __forceinline__ bool hit_wins(float t, uint32_t prim, bool found, float4 best) {
    return !found || t < best.x || prim < as_u32(best.w);
}
__device__ __forceinline__ bool hit_wins(float t, uint32_t prim, uint32_t inst, bool found, float4 best, uint32_t bestInst) {
    return !found || t < best.x || prim < as_u32(best.w) || (prim == as_u32(best.w) && inst < bestInst);
}

// Kernel launch parameters common to the query kernels.
struct QueryArgs {
    RayRec* rays;          // device, 64-byte stride
    uint64_t nRays;
    uint8_t* occluded;     // any-hit output (1 byte per ray) or nullptr
    uint32_t* spill;       // global overflow area for traversal stacks
    uint32_t spillStride;  // entries per lane in `spill`
    uint32_t* counter;     // dynamic ray-fetch counters (persistent kernels): poolParts of them, 256 bytes apart
    uint32_t* counterNext; // the counter area of the next launch on this context: zeroed by this one (ray_pool.h: RayPool::init); nullptr: no
    uint32_t poolParts;    // log2 of the number of partitions of the batch, each with its own counter (ray_pool.h)
    unsigned long long* stats;  // instrumented variants: lane-utilisation counters
    const unsigned long long* nRaysDev;  // if non-null the batch size is read from device memory (on-device queues)
    uint64_t splitBelow;   // batches of fewer rays split their last rays over idle lanes (ray_split.h); 0: never (TBVH_SPLIT_RAYS=0)
    Omm omm;               // opacity micromaps of the scene (map == nullptr: none)
    uint32_t fresh;        // 1: ignore the stored hit, start every ray from {freshTmax,0,0,0} and always write the record
    float freshTmax;
    // coherence probe of this batch (kernels_cwbvh.hip: coherence_sample, run by the traversal kernel itself): probe[0] = sampled neighbour pairs whose
    // directions agree, probe[1] = pairs sampled (published by the first kernel of the query); nullptr = this launch does not probe (small batches).
    // baseBlocks: workgroups beyond this index only take part when the batch is coherent.
    uint32_t* probe;
    uint32_t baseBlocks;
    uint32_t flags;        // 1 = non-temporal ray loads / hit stores, 2 = triangle records padded to 64 bytes (experiments); 16 = take the batch for coherent whatever the probe finds (variant 91); 32 = the first kernel of a two-flavor launch runs the strict schedule (PROBED == 4)
    uint32_t hybridK;      // BVH8_CWBVH, hybrid node array (cwbvh_node.h: kNodeHybrid): nodes below this index are packed, the others one per line
};

// Host side: does this launch use the kernels with split rays?  Batches below the threshold, and the wavefront stages (ray count
// known to the device only); at 16.7 M rays the tail is 5 % of a launch and those kernels' register cap costs as much as it gains.
inline bool split_rays_wanted(const QueryArgs& q) { return q.splitBelow != 0 && (q.nRaysDev != nullptr || q.nRays < q.splitBelow); }


// A float4 array known to live in global memory.  Pointers that were themselves loaded from
// memory (per-BLAS node / triangle bases) are generic to the compiler, and loads through them
// become flat_load: same latency, but they also hold lgkmcnt, so LDS stack traffic and node
// fetches serialise.  GlobalF4 carries the address space, so the loads stay global_load.
typedef float tbvh_f4 __attribute__((ext_vector_type(4)));
struct GlobalF4 {
    const __attribute__((address_space(1))) tbvh_f4* p;
    __device__ __forceinline__ GlobalF4() : p(nullptr) {}
    __device__ __forceinline__ explicit GlobalF4(const float4* q) : p((const __attribute__((address_space(1))) tbvh_f4*)q) {}
    __device__ __forceinline__ float4 operator[](size_t i) const { const tbvh_f4 v = p[i]; return make_float4(v.x, v.y, v.z, v.w); }
    __device__ __forceinline__ GlobalF4 operator+(size_t o) const { GlobalF4 r; r.p = p + o; return r; }
};

}  // namespace tbvh
