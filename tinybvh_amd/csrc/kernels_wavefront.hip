// kernels_wavefront.hip — device-resident wavefront path tracer stages around the traversal
// kernels: Generate -> { Extend -> Shade } x depth -> Connect -> (accumulate).
//
// Same staging as the reference's wavefront.cl (:52-287): Generate writes primary path states,
// Extend is a nearest-hit batch over the live paths, Shade consumes the hits and appends at most
// one extension ray and one shadow ray per path to on-device queues (wavefront.cl:218-236 does
// it with atomic_inc per path; here one atomic per wave via ballot + mbcnt), Connect is an
// any-hit batch over the shadow queue, and the unoccluded contributions are accumulated per
// pixel.  Queue sizes never visit the host: the traversal kernels read their batch size from
// device memory (QueryArgs::nRaysDev).
//
// Shading is deliberately small (this repository accelerates traversal, not materials):
// Lambert surfaces with the albedo packed as RGB8 in v0.w of the hit triangle
// (rgb32_to_vec3(as_uint(v0.w)), wavefront.cl:160/202; 0 = 70 % grey), one point light with
// next-event estimation, a two-colour sky on a miss, cosine-weighted bounces
// (tools.cl:31-39), RNG = WangHash + xorshift32 (tools.cl:9-11).
#include "device_common.h"
#include "kernels.h"
#include "ray_pool.h"

namespace tbvh {

namespace {

__device__ __forceinline__ float safercp_w(float x) {
    if (x > 1e-12f || x < -1e-12f) return 1.0f / x;
    return x >= 0 ? kFar : -kFar;
}
__device__ __forceinline__ uint32_t wang(uint32_t s) { s = (s ^ 61u) ^ (s >> 16); s *= 9u; s = s ^ (s >> 4); s *= 0x27d4eb2du; return s ^ (s >> 15); }
__device__ __forceinline__ float rnd(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return (float)(s >> 8) * (1.0f / 16777216.0f); }
__device__ __forceinline__ float3 norm3(float3 a) {
    const float l = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
    const float r = l == 0 ? 0.f : 1.0f / l;
    return make_float3(a.x * r, a.y * r, a.z * r);
}
__device__ __forceinline__ void put_ray(RayRec* r, float3 O, float3 D, float tmax) {
    r->O = make_float4(O.x, O.y, O.z, as_f32(0xFFFFu));
    r->D = make_float4(D.x, D.y, D.z, 0.f);
    r->rD = make_float4(safercp_w(D.x), safercp_w(D.y), safercp_w(D.z), 0.f);
    r->hit = make_float4(tmax, 0.f, 0.f, 0.f);
}

// Block-aggregated append: returns this thread's slot in the queue (or ~0 if !want).  ONE global atomic per
// workgroup and queue: same-address atomics are serialised memory-side at ~12 ns each on MI355X, so a
// per-wave append (262 k atomics for a 16.7 M-path stage) would cost milliseconds by itself.
// Must be reached by every thread of the workgroup.  sh: kShadeWaves + 1 words of LDS.
constexpr int kShadeBlock = 1024, kShadeWaves = kShadeBlock / 64;
__device__ __forceinline__ uint32_t queue_slot(bool want, unsigned long long* counter, uint32_t* sh) {
    const uint64_t m = __ballot(want);
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0) sh[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < kShadeWaves; w++) { const uint32_t t = sh[w]; sh[w] = tot; tot += t; }
        sh[kShadeWaves] = tot ? (uint32_t)atomicAdd(counter, (unsigned long long)tot) : 0u;
    }
    __syncthreads();
    const uint32_t slot = sh[kShadeWaves] + sh[wave] + lane_rank(m);
    __syncthreads();   // sh is reused by the next append
    return want ? slot : 0xffffffffu;
}

__global__ void k_wf_generate(CameraArgs cam, RayRec* __restrict__ rays, PathAux* __restrict__ aux, uint64_t n, uint32_t seed) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // same pixel order as k_gen_primary (4x4 tiles), one jittered sample per pixel
    const uint32_t inTile = (uint32_t)(i & 15);
    const uint64_t tile = i >> 4;
    const uint32_t tilesX = cam.width / 4;
    const uint32_t px = (uint32_t)(tile % tilesX) * 4 + (inTile & 3), py = (uint32_t)(tile / tilesX) * 4 + (inTile >> 2);
    uint32_t s = wang(seed * 9781u + (uint32_t)i * 6271u + 1u);
    if (!s) s = 1;
    const float u = ((float)px + rnd(s)) / (float)cam.width, v = ((float)py + rnd(s)) / (float)cam.height;
    const float3 eye = make_float3(cam.eye[0], cam.eye[1], cam.eye[2]);
    const float3 P = make_float3(cam.p1[0] + u * (cam.p2[0] - cam.p1[0]) + v * (cam.p3[0] - cam.p1[0]),
                                 cam.p1[1] + u * (cam.p2[1] - cam.p1[1]) + v * (cam.p3[1] - cam.p1[1]),
                                 cam.p1[2] + u * (cam.p2[2] - cam.p1[2]) + v * (cam.p3[2] - cam.p1[2]));
    put_ray(rays + i, eye, norm3(make_float3(P.x - eye.x, P.y - eye.y, P.z - eye.z)), kFar);
    aux[i].T[0] = aux[i].T[1] = aux[i].T[2] = 1.0f;
    aux[i].pixel = py * cam.width + px;
}

__global__ __launch_bounds__(kShadeBlock) void k_wf_shade(ShadeArgs a) {
    __shared__ uint32_t qsh[kShadeWaves + 1];
    const uint64_t n = *a.nIn;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    bool wantBounce = false, wantShadow = false;
    float3 I = make_float3(0, 0, 0), N = I, R = I, L = I, T = I, contrib = I;
    float ldist = 0;
    uint32_t pixel = 0;
    if (live) {
        const RayRec r = a.in[i];
        const PathAux ax = a.auxIn[i];
        pixel = ax.pixel;
        T = make_float3(ax.T[0], ax.T[1], ax.T[2]);
        const float3 O = xyz(r.O), D = xyz(r.D);
        if (!(r.hit.x < kFar)) {
            // miss: sky = lerp(horizon, zenith) by D.y
            const float k = 0.5f * (D.y + 1.0f);
            const float3 sky = make_float3(a.skyLo[0] + k * (a.skyHi[0] - a.skyLo[0]), a.skyLo[1] + k * (a.skyHi[1] - a.skyLo[1]), a.skyLo[2] + k * (a.skyHi[2] - a.skyLo[2]));
            atomicAdd(&a.accum[pixel * 4 + 0], T.x * sky.x); atomicAdd(&a.accum[pixel * 4 + 1], T.y * sky.y); atomicAdd(&a.accum[pixel * 4 + 2], T.z * sky.z);
        } else {
            const uint32_t prim = as_u32(r.hit.w);
            const float4 v0 = a.verts[(uint64_t)prim * 3], v1 = a.verts[(uint64_t)prim * 3 + 1], v2 = a.verts[(uint64_t)prim * 3 + 2];
            const float3 e1 = make_float3(v1.x - v0.x, v1.y - v0.y, v1.z - v0.z), e2 = make_float3(v2.x - v0.x, v2.y - v0.y, v2.z - v0.z);
            N = norm3(make_float3(e1.y * e2.z - e1.z * e2.y, e1.z * e2.x - e1.x * e2.z, e1.x * e2.y - e1.y * e2.x));
            if (N.x * D.x + N.y * D.y + N.z * D.z > 0) N = make_float3(-N.x, -N.y, -N.z);
            I = make_float3(O.x + r.hit.x * D.x, O.y + r.hit.x * D.y, O.z + r.hit.x * D.z);
            const uint32_t c = as_u32(v0.w);
            const float3 albedo = c ? make_float3((float)((c >> 16) & 255) * 0.00392f, (float)((c >> 8) & 255) * 0.00392f, (float)(c & 255) * 0.00392f)
                                    : make_float3(0.7f, 0.7f, 0.7f);
            uint32_t s = wang(a.seed * 7919u + pixel * 2699u + a.depth * 104729u + 17u);
            if (!s) s = 1;
            // next-event estimation toward the point light
            L = make_float3(a.lightPos[0] - I.x, a.lightPos[1] - I.y, a.lightPos[2] - I.z);
            ldist = sqrtf(L.x * L.x + L.y * L.y + L.z * L.z);
            const float il = ldist > 0 ? 1.0f / ldist : 0.f;
            L = make_float3(L.x * il, L.y * il, L.z * il);
            const float ndl = N.x * L.x + N.y * L.y + N.z * L.z;
            if (ndl > 0 && ldist > 2.0f * a.eps) {
                const float g = ndl * il * il * 0.31830988f;   // albedo/pi * cos / d^2
                contrib = make_float3(T.x * albedo.x * a.lightColor[0] * g, T.y * albedo.y * a.lightColor[1] * g, T.z * albedo.z * a.lightColor[2] * g);
                wantShadow = true;
            }
            if (a.depth + 1 < a.maxDepth) {
                // cosine-weighted bounce about N (tools.cl:31-39): pdf cancels cos/pi, T *= albedo
                const float r0 = rnd(s), r1 = rnd(s);
                const float rr = sqrtf(1.0f - r1 * r1), phi = 6.2831853f * r0;
                float3 t1 = fabsf(N.x) > 0.9f ? make_float3(0, 1, 0) : make_float3(1, 0, 0);
                float3 bx = norm3(make_float3(t1.y * N.z - t1.z * N.y, t1.z * N.x - t1.x * N.z, t1.x * N.y - t1.y * N.x));
                float3 by = make_float3(N.y * bx.z - N.z * bx.y, N.z * bx.x - N.x * bx.z, N.x * bx.y - N.y * bx.x);
                const float cx = cosf(phi) * rr, cy = sinf(phi) * rr;
                R = norm3(make_float3(N.x + cx * bx.x + cy * by.x + 0.f, N.y + cx * bx.y + cy * by.y, N.z + cx * bx.z + cy * by.z));
                T = make_float3(T.x * albedo.x, T.y * albedo.y, T.z * albedo.z);
                wantBounce = true;
            }
        }
    }
    // the whole workgroup takes part in the two queue appends
    const uint32_t sb = queue_slot(wantBounce, a.nOut, qsh);
    if (wantBounce) {
        put_ray(a.out + sb, make_float3(I.x + R.x * a.eps, I.y + R.y * a.eps, I.z + R.z * a.eps), R, kFar);
        PathAux o; o.T[0] = T.x; o.T[1] = T.y; o.T[2] = T.z; o.pixel = pixel;
        a.auxOut[sb] = o;
    }
    const uint32_t ss = queue_slot(wantShadow, a.nShadow, qsh);
    if (wantShadow) {
        put_ray(a.shadow + ss, make_float3(I.x + L.x * a.eps, I.y + L.y * a.eps, I.z + L.z * a.eps), L, ldist - 2.0f * a.eps);
        PathAux o; o.T[0] = contrib.x; o.T[1] = contrib.y; o.T[2] = contrib.z; o.pixel = pixel;
        a.shadowAux[ss] = o;
    }
}

__global__ void k_wf_connect(const uint8_t* __restrict__ occluded, const PathAux* __restrict__ aux, const unsigned long long* __restrict__ nShadow,
                             float* __restrict__ accum) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *nShadow || occluded[i]) return;
    const PathAux a = aux[i];
    atomicAdd(&accum[a.pixel * 4 + 0], a.T[0]); atomicAdd(&accum[a.pixel * 4 + 1], a.T[1]); atomicAdd(&accum[a.pixel * 4 + 2], a.T[2]);
}

}  // namespace

void launch_wf_generate(const CameraArgs& cam, RayRec* rays, PathAux* aux, uint64_t n, uint32_t seed, hipStream_t s) {
    hipLaunchKernelGGL(k_wf_generate, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, cam, rays, aux, n, seed);
}
void launch_wf_shade(const ShadeArgs& a, uint64_t capacity, hipStream_t s) {
    hipLaunchKernelGGL(k_wf_shade, dim3((uint32_t)((capacity + kShadeBlock - 1) / kShadeBlock)), dim3(kShadeBlock), 0, s, a);
}
void launch_wf_connect(const uint8_t* occ, const PathAux* aux, const unsigned long long* nShadow, float* accum, uint64_t capacity, hipStream_t s) {
    hipLaunchKernelGGL(k_wf_connect, dim3((uint32_t)((capacity + 255) / 256)), dim3(256), 0, s, occ, aux, nShadow, accum);
}

}  // namespace tbvh
