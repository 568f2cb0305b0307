// kernels_raygen.hip — device-side ray generators for the benchmark configurations, so
// that rays are produced and consumed in HBM without host round trips.
//   primary : pinhole camera, 4x4-pixel tiles x spp samples, the speedtest's order
//             (tiny_bvh_speedtest.cpp:517-551)
//   bounce  : one diffuse bounce from the hit points of a traced batch
//             (tiny_bvh_speedtest.cpp:561-587; RNG = WangHash + xorshift32, tools.cl:9-11)
//   shadow  : rays from hit points toward a point light (tiny_bvh_speedtest.cpp:851-865)
// Every generated record is what the tinybvh::Ray constructor would build
// (tiny_bvh.h:695-703): D normalised, rD = tinybvh_safercp(D), mask 0xFFFF, hit.t = tmax.
#include "device_common.h"
#include "kernels.h"

namespace tbvh {

__device__ __forceinline__ float safercp(float x) {  // tiny_bvh.h:442
    if (x > 1e-12f || x < -1e-12f) return 1.0f / x;
    return x >= 0 ? kFar : -kFar;
}
__device__ __forceinline__ float3 normalize3(float3 a) {  // tiny_bvh.h:506-510
    const float l = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
    const float rl = l == 0 ? 0.f : 1.0f / l;
    return make_float3(a.x * rl, a.y * rl, a.z * rl);
}
__device__ __forceinline__ void write_ray(RayRec* r, float3 O, float3 dir, float tmax) {
    const float3 D = normalize3(dir);
    r->O = make_float4(O.x, O.y, O.z, as_f32(0xFFFFu));
    r->D = make_float4(D.x, D.y, D.z, 0.f);
    r->rD = make_float4(safercp(D.x), safercp(D.y), safercp(D.z), 0.f);
    r->hit = make_float4(tmax, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ uint32_t wang_hash(uint32_t s) {
    s = (s ^ 61u) ^ (s >> 16); s *= 9u; s = s ^ (s >> 4); s *= 0x27d4eb2du; return s ^ (s >> 15);
}
__device__ __forceinline__ float rand_float(uint32_t& seed) {
    seed ^= seed << 13; seed ^= seed >> 17; seed ^= seed << 5;
    return (float)seed * 2.3283064365387e-10f;
}

__global__ void k_gen_primary(CameraArgs cam, RayRec* __restrict__ rays, uint64_t first, uint64_t n) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint64_t i = first + k;
    const uint32_t spp = cam.sppX * cam.sppY;
    const uint32_t s = (uint32_t)(i % spp);
    const uint64_t pix = i / spp;
    const uint32_t inTile = (uint32_t)(pix & 15);
    const uint64_t tile = pix >> 4;
    const uint32_t tilesX = cam.width / 4;
    const uint32_t tx = (uint32_t)(tile % tilesX), ty = (uint32_t)(tile / tilesX);
    const uint32_t px = tx * 4 + (inTile & 3), py = ty * 4 + (inTile >> 2);
    const float u = (float)(px * cam.sppX + (s % cam.sppX)) / (float)(cam.width * cam.sppX);
    const float v = (float)(py * cam.sppY + (s / cam.sppX)) / (float)(cam.height * cam.sppY);
    const float3 eye = make_float3(cam.eye[0], cam.eye[1], cam.eye[2]);
    const float3 P = make_float3(cam.p1[0] + u * (cam.p2[0] - cam.p1[0]) + v * (cam.p3[0] - cam.p1[0]),
                                 cam.p1[1] + u * (cam.p2[1] - cam.p1[1]) + v * (cam.p3[1] - cam.p1[1]),
                                 cam.p1[2] + u * (cam.p2[2] - cam.p1[2]) + v * (cam.p3[2] - cam.p1[2]));
    write_ray(rays + k, eye, make_float3(P.x - eye.x, P.y - eye.y, P.z - eye.z), kFar);
}

__global__ void k_gen_bounce(TriSource src, const RayRec* __restrict__ in, RayRec* __restrict__ out, uint64_t n, uint32_t seed) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float3 O = xyz(in[i].O), D = xyz(in[i].D);
    const float4 hit = in[i].hit;
    uint32_t rs = wang_hash(seed + (uint32_t)i * 747796405u + (uint32_t)(i >> 32));
    if (rs == 0) rs = 1;
    float3 R = normalize3(make_float3(rand_float(rs) - 0.5f, rand_float(rs) - 0.5f, rand_float(rs) - 0.5f));
    float3 I;
    if (hit.x < kFar) {
        I = make_float3(O.x + hit.x * D.x, O.y + hit.x * D.y, O.z + hit.x * D.z);
        const uint32_t prim = as_u32(hit.w);
        const float4 a = src.verts[(uint64_t)prim * 3], b = src.verts[(uint64_t)prim * 3 + 1], c = src.verts[(uint64_t)prim * 3 + 2];
        const float3 e1 = make_float3(b.x - a.x, b.y - a.y, b.z - a.z), e2 = make_float3(c.x - a.x, c.y - a.y, c.z - a.z);
        float3 N = normalize3(make_float3(e1.y * e2.z - e1.z * e2.y, e1.z * e2.x - e1.x * e2.z, e1.x * e2.y - e1.y * e2.x));
        if (N.x * D.x + N.y * D.y + N.z * D.z > 0) N = make_float3(-N.x, -N.y, -N.z);
        if (N.x * R.x + N.y * R.y + N.z * R.z < 0) R = make_float3(-R.x, -R.y, -R.z);
    } else {
        I = make_float3(O.x + 20.0f * D.x, O.y + 20.0f * D.y, O.z + 20.0f * D.z);
    }
    write_ray(out + i, make_float3(I.x + 0.001f * R.x, I.y + 0.001f * R.y, I.z + 0.001f * R.z), R, kFar);
}

__global__ void k_gen_shadow(const RayRec* __restrict__ in, RayRec* __restrict__ out, uint64_t n, float lx, float ly, float lz, float eps) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float3 O = xyz(in[i].O), D = xyz(in[i].D);
    const float t = fminf(1000.0f, in[i].hit.x);
    const float3 I = make_float3(O.x + t * D.x, O.y + t * D.y, O.z + t * D.z);
    const float3 L = make_float3(lx - I.x, ly - I.y, lz - I.z);
    const float dist = sqrtf(L.x * L.x + L.y * L.y + L.z * L.z);
    const float3 Ld = normalize3(L);
    write_ray(out + i, make_float3(I.x + Ld.x * eps, I.y + Ld.y * eps, I.z + Ld.z * eps), Ld, dist - eps);
}

__global__ void k_reset_hits(RayRec* __restrict__ rays, uint64_t n, float tmax) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rays[i].hit = make_float4(tmax, 0.f, 0.f, 0.f);
}
void launch_reset_hits(RayRec* rays, uint64_t n, float tmax, hipStream_t s) {
    const uint32_t bs = 256;
    hipLaunchKernelGGL(k_reset_hits, dim3((uint32_t)((n + bs - 1) / bs)), dim3(bs), 0, s, rays, n, tmax);
}

void launch_gen_primary(const CameraArgs& cam, RayRec* rays, uint64_t first, uint64_t n, hipStream_t s) {
    const uint32_t bs = 256;
    hipLaunchKernelGGL(k_gen_primary, dim3((uint32_t)((n + bs - 1) / bs)), dim3(bs), 0, s, cam, rays, first, n);
}
void launch_gen_bounce(const TriSource& src, const RayRec* in, RayRec* out, uint64_t n, uint32_t seed, hipStream_t s) {
    const uint32_t bs = 256;
    hipLaunchKernelGGL(k_gen_bounce, dim3((uint32_t)((n + bs - 1) / bs)), dim3(bs), 0, s, src, in, out, n, seed);
}
void launch_gen_shadow(const RayRec* in, RayRec* out, uint64_t n, float lx, float ly, float lz, float eps, hipStream_t s) {
    const uint32_t bs = 256;
    hipLaunchKernelGGL(k_gen_shadow, dim3((uint32_t)((n + bs - 1) / bs)), dim3(bs), 0, s, in, out, n, lx, ly, lz, eps);
}

// bytes 44..63 (hit.inst, t, u, v, prim) of every ray record, packed as 5 dwords per ray for the host read-back
namespace {
__global__ void k_pack_hits(const RayRec* __restrict__ rays, uint32_t* __restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* w = (const uint32_t*)(rays + i);
    uint32_t* o = out + i * 5;
    o[0] = w[11]; o[1] = w[12]; o[2] = w[13]; o[3] = w[14]; o[4] = w[15];
}
}  // namespace

// ---- the machine's own ceilings, measured where the kernels run (bench.py: roofline) ----------------------------------------------------
// Streaming copy / read, 16 bytes per lane, ONE float4 per thread with the non-temporal hint: the shape that reaches the hardware guide's
// 6.3 TB/s on MI355X (tools/ubench/copy_rate.hip, profiles/r03_copy_rate.txt: 6.4-6.5 TB/s copy, 6.4-6.6 read-only; round 2's grid-stride
// kernel with four loads in flight per lane stopped at 4.5-5.2).
namespace {
__global__ __launch_bounds__(256) void k_stream_copy(const float4* __restrict__ src4, float4* __restrict__ dst4, uint64_t n16) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) __builtin_nontemporal_store(__builtin_nontemporal_load((const tbvh_f4*)src4 + i), (tbvh_f4*)dst4 + i);
}
__global__ __launch_bounds__(256) void k_stream_read(const float4* __restrict__ src4, float* __restrict__ sink, uint64_t n16) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const tbvh_f4* src = (const tbvh_f4*)src4;
    tbvh_f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (; i + 3 * stride < n16; i += 4 * stride)
        acc += __builtin_nontemporal_load(src + i) + __builtin_nontemporal_load(src + i + stride) + __builtin_nontemporal_load(src + i + 2 * stride) + __builtin_nontemporal_load(src + i + 3 * stride);
    for (; i < n16; i += stride) acc += src[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;   // (never true for the memset pattern: keeps the loads alive)
}
// VALU issue ceiling for the instruction mix of the CWBVH node test (cwbvh_node.h: cw_test_node — of 209 VALU per node 48 v_cvt_f32_ubyte,
// 48 v_fma_f32, 32 max3 / min3 / max / min, 8 v_cmp, 20 v_cndmask, the rest integer): a hand-written block of exactly 32 instructions in those
// proportions on independent registers, run by 8 waves per SIMD.  Wave-instructions per second over the whole chip, clock throttling
// included (tools/ubench/valu_issue.hip: a wave64 VALU instruction occupies a SIMD's issue port for ~2.5-3 nominal cycles under this mix; ONE
// wave alone cannot issue faster than one per ~5.6 cycles).
__global__ __launch_bounds__(64) void k_valu_mix(float* __restrict__ out, int iters, float a, float b) {
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    unsigned u0 = threadIdx.x * 2654435761u, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    for (int i = 0; i < iters; i++) {
        asm volatile(
            "v_cvt_f32_ubyte0 %0, %8\n\tv_cvt_f32_ubyte1 %1, %8\n\tv_cvt_f32_ubyte2 %2, %9\n\tv_cvt_f32_ubyte3 %3, %9\n\t"
            "v_cvt_f32_ubyte0 %4, %10\n\tv_cvt_f32_ubyte1 %5, %10\n\tv_cvt_f32_ubyte2 %6, %11\n\tv_cvt_f32_ubyte3 %7, %11\n\t"
            "v_fma_f32 %0, %0, %12, %13\n\tv_fma_f32 %1, %1, %12, %13\n\tv_fma_f32 %2, %2, %12, %13\n\tv_fma_f32 %3, %3, %12, %13\n\t"
            "v_fma_f32 %4, %4, %12, %13\n\tv_fma_f32 %5, %5, %12, %13\n\tv_fma_f32 %6, %6, %12, %13\n\tv_fma_f32 %7, %7, %12, %13\n\t"
            "v_max3_f32 %0, %0, %1, %2\n\tv_min3_f32 %3, %3, %4, %5\n\tv_max_f32 %0, %0, %12\n\tv_min_f32 %3, %3, %13\n\t"
            "v_max3_f32 %6, %6, %7, %1\n\tv_min3_f32 %4, %4, %5, %2\n\t"
            "v_cmp_le_f32 vcc, %0, %3\n\tv_cndmask_b32 %1, %1, %2, vcc\n\tv_cmp_le_f32 vcc, %6, %4\n\tv_cndmask_b32 %5, %5, %7, vcc\n\t"
            "v_lshlrev_b32 %8, 1, %8\n\tv_or_b32 %9, %9, %8\n\tv_add_u32 %10, %10, %9\n\tv_xor_b32 %11, %11, %10\n\tv_and_b32 %8, %8, %11\n\tv_add_u32 %9, %9, %10"
            : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3)
            : "v"(a), "v"(b)
            : "vcc");
    }
    out[blockIdx.x * 64 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + (float)(u0 ^ u1 ^ u2 ^ u3);
}
}  // namespace
void launch_stream_copy(const float4* src, float4* dst, uint64_t n16, hipStream_t s) {
    hipLaunchKernelGGL(k_stream_copy, dim3((uint32_t)((n16 + 255) / 256)), dim3(256), 0, s, src, dst, n16);
}
void launch_stream_read(const float4* src, float* sink, uint64_t n16, uint32_t blocks, hipStream_t s) {
    hipLaunchKernelGGL(k_stream_read, dim3(blocks), dim3(256), 0, s, src, sink, n16);
}
// 32 VALU instructions per loop iteration and wave, by construction
void launch_valu_mix(float* out, int iters, uint32_t blocks, hipStream_t s) {
    hipLaunchKernelGGL(k_valu_mix, dim3(blocks), dim3(64), 0, s, out, iters, 1.0001f, 0.5f);
}

void launch_pack_hits(const RayRec* rays, uint32_t* out, uint64_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_pack_hits, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, rays, out, n);
}

}  // namespace tbvh
