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
// Shading follows wavefront.cl:127-246: the material of a triangle lives in v0.w of its first vertex —
// type << 24 | RGB8, type 0 diffuse (Lambert), 1 = MATERIAL_LIGHT (emits lightColor, ends the path), 2 = MATERIAL_SPECULAR
// (pure mirror) — next-event estimation towards a rectangular light in the xz-plane with the reference's solid-angle pdf
// and MIS against the BRDF pdf, the BRDF pdf of a bounce is "postponed" to the next vertex (carried in D.w of the ray
// record, as PathState::T.w does there), path flags PATH_LAST_SPECULAR / PATH_VIA_DIFFUSE packed with pixel and depth
// exactly like PathState::O.w (pixel << 8 | depth << 4 | flags).  Where the .cl file slips, the intent is followed, not
// the letter: a path that leaves the scene is weighted by 1 / pdf like every other vertex (wavefront.cl:151-156 adds the
// sky before the division), the light pdf of a BSDF-sampled light hit uses the hit distance (it reads D.w, which nothing
// ever writes: 1e30), cosine-weighted bounces use Malley's construction (tools.cl:34-39 feeds a half sphere into
// normalize(N + R)).  Extensions kept from the first version: RGB 0 means 70 % grey (scenes without materials), a light of
// size 0 is a point light (no MIS), a two-colour sky, any number of diffuse bounces unless TBVH_WF_ONE_DIFFUSE_BOUNCE asks
// for the reference's single one.  RNG = WangHash + xorshift32 (tools.cl:9-11).
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
constexpr uint32_t kPathLastSpecular = 1u, kPathViaDiffuse = 2u;   // wavefront.cl:9-10
constexpr uint32_t kMaterialLight = 1u, kMaterialSpecular = 2u;    // wavefront.cl:12-13
// pdf: the postponed BRDF pdf of the bounce that made this ray (PathState::T.w); D.w of the 64-byte record is free
// (tinybvh::Ray::instIdx, only used inside the reference's CPU TLAS traversal)
__device__ __forceinline__ void put_ray(RayRec* r, float3 O, float3 D, float tmax, float pdf = 1.0f) {
    r->O = make_float4(O.x, O.y, O.z, as_f32(0xFFFFu));
    r->D = make_float4(D.x, D.y, D.z, pdf);
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

__global__ void k_wf_generate(CameraArgs cam, RayRec* __restrict__ rays, PathAux* __restrict__ aux, uint64_t n, uint32_t seed, uint32_t firstRow, uint32_t bandRows,
                              unsigned long long* __restrict__ queueCounters, uint32_t nCounterWords) {
    // the frame's queue counters (one per queue and depth, each on its own 256-byte line: capi.hip) start at zero, the first path queue at n — set
    // here instead of by a memset and a copy in the stream (each a launch of its own: ~8 us with the gap around it)
    if (queueCounters && blockIdx.x == 0)
        for (uint32_t k = threadIdx.x; k < nCounterWords; k += blockDim.x) queueCounters[k] = k == 0u ? n : 0ull;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // same pixel order as k_gen_primary (4x4 tiles), one jittered sample per pixel
    const uint32_t inTile = (uint32_t)(i & 15);
    const uint64_t tile = i >> 4;
    const uint32_t tilesX = cam.width / 4;
    const uint32_t px = (uint32_t)(tile % tilesX) * 4 + (inTile & 3), py = (uint32_t)(tile / tilesX) * 4 + (inTile >> 2);
    // a band of a larger image (tbvh_wavefront_set_band): cam is the FULL image's camera, this launch covers its rows firstRow ... and draws the
    // random numbers the full image's launch would draw for these pixels (4 x 4 tiles and firstRow a multiple of 4: global ray index = local +
    // firstRow * width)
    (void)bandRows;
    uint32_t s = wang(seed * 9781u + ((uint32_t)i + firstRow * cam.width) * 6271u + 1u);
    if (!s) s = 1;
    const float u = ((float)px + rnd(s)) / (float)cam.width, v = ((float)(py + firstRow) + rnd(s)) / (float)cam.height;
    const float3 eye = make_float3(cam.eye[0], cam.eye[1], cam.eye[2]);
    const float3 P = make_float3(cam.p1[0] + u * (cam.p2[0] - cam.p1[0]) + v * (cam.p3[0] - cam.p1[0]),
                                 cam.p1[1] + u * (cam.p2[1] - cam.p1[1]) + v * (cam.p3[1] - cam.p1[1]),
                                 cam.p1[2] + u * (cam.p2[2] - cam.p1[2]) + v * (cam.p3[2] - cam.p1[2]));
    put_ray(rays + i, eye, norm3(make_float3(P.x - eye.x, P.y - eye.y, P.z - eye.z)), kFar, 1.0f);
    aux[i].T[0] = aux[i].T[1] = aux[i].T[2] = 1.0f;
    aux[i].pixel = ((py * cam.width + px) << 8) | kPathLastSpecular;   // wavefront.cl:88
}

__global__ __launch_bounds__(kShadeBlock) void k_wf_shade(ShadeArgs a) {
    __shared__ uint32_t qsh[kShadeWaves + 1];
    const uint64_t n = *a.nIn;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    bool wantBounce = false, wantShadow = false;
    float3 I = make_float3(0, 0, 0), N = I, R = I, L = I, T = I, contrib = I;
    float ldist = 0, newPdf = 1.0f;
    uint32_t pixel = 0, newFlags = 0;
    if (live) {
        const RayRec r = a.in[i];
        const PathAux ax = a.auxIn[i];
        pixel = ax.pixel >> 8;
        const uint32_t flags = ax.pixel & 15u;
        T = make_float3(ax.T[0], ax.T[1], ax.T[2]);
        const float brdfPdf = r.D.w;                       // postponed pdf of the bounce that led here (1 for camera / mirror)
        const float ipdf = brdfPdf > 0 ? 1.0f / brdfPdf : 0.f;
        const float3 O = xyz(r.O), D = xyz(r.D);
        const bool areaLight = a.lightSize[0] > 0 && a.lightSize[1] > 0;
        const bool letter = (a.flags & 2u) != 0;
        const float lightArea = a.lightSize[0] * a.lightSize[1];
        if (!(r.hit.x < kFar)) {
            // end path on sky (wavefront.cl:151-156): lerp(horizon, zenith) by D.y
            const float k = 0.5f * (D.y + 1.0f);
            const float3 sky = make_float3(a.skyLo[0] + k * (a.skyHi[0] - a.skyLo[0]), a.skyLo[1] + k * (a.skyHi[1] - a.skyLo[1]), a.skyLo[2] + k * (a.skyHi[2] - a.skyLo[2]));
            const float ws = letter ? 1.0f : ipdf;   // wavefront.cl:151-156 adds the sky before :180 divides the postponed pdf out
            atomicAdd(&a.accum[pixel * 4 + 0], T.x * ws * sky.x); atomicAdd(&a.accum[pixel * 4 + 1], T.y * ws * sky.y); atomicAdd(&a.accum[pixel * 4 + 2], T.z * ws * sky.z);
        } else {
            const uint32_t prim = as_u32(r.hit.w);
            // geometry at the hit: a TLAS hit names the instance in byte 44 of the record (wavefront2.cl:180-186 unpacks
            // prim | inst << 24 and picks the BLAS's vertex array); the triangle is in the instance's space
            const float4* vb = a.verts;
            const float4* ip = nullptr;
            if (a.instances) {
                ip = a.instances + (size_t)as_u32(r.rD.w) * 12;
                vb = a.blasVerts[as_u32(ip[8].w)];
            }
            const float4 v0 = vb[(uint64_t)prim * 3], v1 = vb[(uint64_t)prim * 3 + 1], v2 = vb[(uint64_t)prim * 3 + 2];
            const uint32_t c = as_u32(v0.w), materialType = c >> 24;
            if (materialType == kMaterialLight) {
                // end path on light (wavefront.cl:163-178): alone after a mirror or from the camera, else MIS with the light sampling
                float w = ipdf;
                if (!(flags & kPathLastSpecular) && areaLight) {
                    const float solid = __builtin_fminf(6.2831853f, lightArea / (r.hit.x * r.hit.x) * fabsf(D.y));
                    const float lightPdf = solid > 0 ? 1.0f / solid : kFar;
                    w = 1.0f / (lightPdf + brdfPdf);
                    if (letter) w = 0.f;   // wavefront.cl:174 evaluates LightPDF( D4.w ) with D4.w = 1e30 (nothing writes a distance there): 1 / (inf + pdf)
                }
                atomicAdd(&a.accum[pixel * 4 + 0], T.x * w * a.lightColor[0]); atomicAdd(&a.accum[pixel * 4 + 1], T.y * w * a.lightColor[1]);
                atomicAdd(&a.accum[pixel * 4 + 2], T.z * w * a.lightColor[2]);
            } else {
                T = make_float3(T.x * ipdf, T.y * ipdf, T.z * ipdf);   // apply the postponed pdf (wavefront.cl:180)
                const float3 e1 = make_float3(v1.x - v0.x, v1.y - v0.y, v1.z - v0.z), e2 = make_float3(v2.x - v0.x, v2.y - v0.y, v2.z - v0.z);
                N = make_float3(e1.y * e2.z - e1.z * e2.y, e1.z * e2.x - e1.x * e2.z, e1.x * e2.y - e1.y * e2.x);
                if (ip) {   // normal to world space: transpose of the inverse transform (rows 4..6 of the instance record)
                    const float4 i0 = ip[4], i1 = ip[5], i2 = ip[6];
                    N = make_float3(i0.x * N.x + i1.x * N.y + i2.x * N.z, i0.y * N.x + i1.y * N.y + i2.y * N.z, i0.z * N.x + i1.z * N.y + i2.z * N.z);
                }
                N = norm3(N);
                const float nd = N.x * D.x + N.y * D.y + N.z * D.z;
                if (nd > 0) N = make_float3(-N.x, -N.y, -N.z);
                I = make_float3(O.x + r.hit.x * D.x, O.y + r.hit.x * D.y, O.z + r.hit.x * D.z);
                const uint32_t rgb = c & 0xffffffu;
                const float3 color = rgb ? make_float3((float)((rgb >> 16) & 255) * 0.00392f, (float)((rgb >> 8) & 255) * 0.00392f, (float)(rgb & 255) * 0.00392f)
                                         : make_float3(0.7f, 0.7f, 0.7f);
                const uint32_t gpixel = pixel + a.pixelOffset;   // (a band of a larger image: the pixel's index in the full image)
                uint32_t s = wang(a.seed * 7919u + gpixel * 2699u + a.depth * 104729u + 17u);
                if (!s) s = 1;
                float r0 = rnd(s), r1 = rnd(s), r2 = rnd(s), r3 = rnd(s);   // r0, r1: bounce; r2, r3: light sample
                if (a.blueNoise && a.depth == 0 && a.sampleIdx < 4u) {
                    // blue-noise first samples (wavefront.cl:183-189; Noise() of :24-31, with its x = pixel % height, y = pixel / height)
                    const uint32_t nx = (gpixel % a.height) & 127u, ny = (gpixel / a.height) & 127u;
                    const uint32_t w0 = a.blueNoise[((a.sampleIdx * 2u) << 14) + (ny << 7) + nx], w1 = a.blueNoise[((a.sampleIdx * 2u + 1u) << 14) + (ny << 7) + nx];
                    r2 = (float)(w0 >> 16) * 0.00392f; r3 = (float)((w0 >> 8) & 255u) * 0.00392f;   // noise0 -> the light sample
                    r0 = (float)(w1 >> 16) * 0.00392f; r1 = (float)((w1 >> 8) & 255u) * 0.00392f;   // noise1 -> the bounce
                }
                if (materialType != kMaterialSpecular) {
                    // direct illumination: next event estimation (wavefront.cl:205-222)
                    const float3 Pl = areaLight ? make_float3(a.lightPos[0] + (r2 - 0.5f) * a.lightSize[0], a.lightPos[1], a.lightPos[2] + (r3 - 0.5f) * a.lightSize[1])
                                                : make_float3(a.lightPos[0], a.lightPos[1], a.lightPos[2]);
                    L = make_float3(Pl.x - I.x, Pl.y - I.y, Pl.z - I.z);
                    ldist = sqrtf(L.x * L.x + L.y * L.y + L.z * L.z);
                    const float il = ldist > 0 ? 1.0f / ldist : 0.f;
                    L = make_float3(L.x * il, L.y * il, L.z * il);
                    const float ndl = N.x * L.x + N.y * L.y + N.z * L.z;
                    if (ndl > 0 && ldist > 2.0f * a.eps) {
                        float g;
                        if (areaLight) {   // MIS: light pdf = 1 / solid angle of the rectangle seen from I, BRDF pdf = cos / pi
                            const float solid = __builtin_fminf(6.2831853f, lightArea * il * il * fabsf(L.y));
                            const float lightPdf = solid > 0 ? 1.0f / solid : kFar;
                            g = 0.31830988f * ndl / (lightPdf + ndl * 0.31830988f);
                        } else
                            g = ndl * il * il * 0.31830988f;   // point light: albedo / pi * cos / d^2
                        contrib = make_float3(T.x * color.x * a.lightColor[0] * g, T.y * color.y * a.lightColor[1] * g, T.z * color.z * a.lightColor[2] * g);
                        wantShadow = true;
                    }
                }
                if (a.depth + 1 < a.maxDepth) {
                    if (materialType == kMaterialSpecular) {   // wavefront.cl:225-232
                        const float k2 = 2.0f * (N.x * D.x + N.y * D.y + N.z * D.z);
                        R = make_float3(D.x - k2 * N.x, D.y - k2 * N.y, D.z - k2 * N.z);
                        T = make_float3(T.x * color.x, T.y * color.y, T.z * color.z);
                        newPdf = 1.0f; newFlags = kPathLastSpecular;
                        wantBounce = true;
                    } else if (!((a.flags & 1u) && (flags & kPathViaDiffuse))) {   // wavefront.cl:233-242
                        // cosine-weighted bounce about N (Malley): pdf = cos / pi, postponed to the next vertex
                        if (letter) {   // tools.cl:34-39: a half sphere about the WORLD z axis added to N
                            const float rl = sqrtf(__builtin_fmaxf(1.0f - r1 * r1, 0.f)), pl = 12.566371f * r0;
                            R = norm3(make_float3(N.x + cosf(pl) * rl, N.y + sinf(pl) * rl, N.z + r1));
                        } else {
                        const float rr = sqrtf(r1), cz = sqrtf(1.0f - r1), phi = 6.2831853f * r0;
                        const float3 t1 = fabsf(N.x) > 0.9f ? make_float3(0, 1, 0) : make_float3(1, 0, 0);
                        const float3 bx = norm3(make_float3(t1.y * N.z - t1.z * N.y, t1.z * N.x - t1.x * N.z, t1.x * N.y - t1.y * N.x));
                        const float3 by = make_float3(N.y * bx.z - N.z * bx.y, N.z * bx.x - N.x * bx.z, N.x * bx.y - N.y * bx.x);
                        const float cx = cosf(phi) * rr, cy = sinf(phi) * rr;
                        R = norm3(make_float3(cz * N.x + cx * bx.x + cy * by.x, cz * N.y + cx * bx.y + cy * by.y, cz * N.z + cx * bx.z + cy * by.z));
                        }
                        const float ndr = __builtin_fmaxf(N.x * R.x + N.y * R.y + N.z * R.z, 1e-6f);
                        newPdf = ndr * 0.31830988f;
                        const float k3 = ndr * 0.31830988f;   // T *= dot(N, R) * BRDF, BRDF = color / pi
                        T = make_float3(T.x * color.x * k3, T.y * color.y * k3, T.z * color.z * k3);
                        newFlags = kPathViaDiffuse;
                        wantBounce = true;
                    }
                }
            }
        }
    }
    // the whole workgroup takes part in the two queue appends
    const uint32_t sb = queue_slot(wantBounce, a.nOut, qsh);
    if (wantBounce) {
        put_ray(a.out + sb, make_float3(I.x + R.x * a.eps, I.y + R.y * a.eps, I.z + R.z * a.eps), R, kFar, newPdf);
        PathAux o; o.T[0] = T.x; o.T[1] = T.y; o.T[2] = T.z; o.pixel = (pixel << 8) | (((a.depth + 1u) & 15u) << 4) | newFlags;
        a.auxOut[sb] = o;
    }
    const uint32_t ss = queue_slot(wantShadow, a.nShadow, qsh);
    if (wantShadow) {
        put_ray(a.shadow + ss, make_float3(I.x + L.x * a.eps, I.y + L.y * a.eps, I.z + L.z * a.eps), L, ldist - 2.0f * a.eps);
        PathAux o; o.T[0] = contrib.x; o.T[1] = contrib.y; o.T[2] = contrib.z; o.pixel = pixel;
        a.shadowAux[ss] = o;
    }
}

// Finalize (wavefront.cl:275-286): accumulator * scale -> sqrt -> 8 bits per channel, 0x00RRGGBB
__global__ void k_wf_finalize(const float* __restrict__ accum, float scale, uint32_t* __restrict__ pixels, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float r = __builtin_fminf(sqrtf(__builtin_fmaxf(accum[i * 4] * scale, 0.f)), 1.0f) * 255.0f;
    const float g = __builtin_fminf(sqrtf(__builtin_fmaxf(accum[i * 4 + 1] * scale, 0.f)), 1.0f) * 255.0f;
    const float b = __builtin_fminf(sqrtf(__builtin_fmaxf(accum[i * 4 + 2] * scale, 0.f)), 1.0f) * 255.0f;
    pixels[i] = ((uint32_t)(int)r << 16) + ((uint32_t)(int)g << 8) + (uint32_t)(int)b;
}

__global__ void k_wf_connect(const uint8_t* __restrict__ occluded, const PathAux* __restrict__ aux, const unsigned long long* __restrict__ nShadow,
                             float* __restrict__ accum) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *nShadow || occluded[i]) return;
    const PathAux a = aux[i];
    atomicAdd(&accum[a.pixel * 4 + 0], a.T[0]); atomicAdd(&accum[a.pixel * 4 + 1], a.T[1]); atomicAdd(&accum[a.pixel * 4 + 2], a.T[2]);
}

}  // namespace

void launch_wf_generate(const CameraArgs& cam, RayRec* rays, PathAux* aux, uint64_t n, uint32_t seed, uint32_t firstRow, uint32_t bandRows, unsigned long long* queueCounters,
                        uint32_t nCounterWords, hipStream_t s) {
    hipLaunchKernelGGL(k_wf_generate, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, cam, rays, aux, n, seed, firstRow, bandRows, queueCounters, nCounterWords);
}
void launch_wf_shade(const ShadeArgs& a, uint64_t capacity, hipStream_t s) {
    hipLaunchKernelGGL(k_wf_shade, dim3((uint32_t)((capacity + kShadeBlock - 1) / kShadeBlock)), dim3(kShadeBlock), 0, s, a);
}
void launch_wf_finalize(const float* accum, float scale, uint32_t* pixels, uint64_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_wf_finalize, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, accum, scale, pixels, n);
}
void launch_wf_connect(const uint8_t* occ, const PathAux* aux, const unsigned long long* nShadow, float* accum, uint64_t capacity, hipStream_t s) {
    hipLaunchKernelGGL(k_wf_connect, dim3((uint32_t)((capacity + 255) / 256)), dim3(256), 0, s, occ, aux, nShadow, accum);
}

}  // namespace tbvh
