// link_rate.hip — how a host tinybvh::Ray[] can cross the host link on this box (round 5: bench.py detail.host_rays).  Per ray 64 bytes go up (the first
// half of a 128-byte record, or a packed 64-byte record) and 20 come back (bytes 44..63).  Candidates, each timed with HIP events / wall clock:
//   dma      hipMemcpyAsync, pinned, contiguous, up / down / both at once on two streams (the ceiling)
//   dma2d    hipMemcpy2DAsync, pinned, width 64 of pitch 128 up; width 20 of pitch 64 / 128 down (no CPU work, no kernel)
//   kernel   the GPU reads / writes mapped host memory itself: gather 16 B per lane (4 lanes per ray), 64 B per lane (one ray per lane), stride 64 and 128;
//            scatter 4 + 16 B per ray, 32 B per ray (bytes 32..63), 64 B per ray (whole record)
//   cpu      N host threads pack stride 128 -> pinned 64 (memcpy rate of the staging path)
// build: make -C tools/ubench link_rate ; run: tools/ubench/link_rate [Mrays]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void gather16(const char* __restrict__ src, uint32_t stride, f4* __restrict__ dst, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 4) return;
    dst[i] = __builtin_nontemporal_load((const f4*)(src + (i >> 2) * stride) + (i & 3));
}
__global__ __launch_bounds__(256) void gather64(const char* __restrict__ src, uint32_t stride, f4* __restrict__ dst, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f4* p = (const f4*)(src + i * stride);
    const f4 a = __builtin_nontemporal_load(p), b = __builtin_nontemporal_load(p + 1), c = __builtin_nontemporal_load(p + 2), d = __builtin_nontemporal_load(p + 3);
    dst[i * 4] = a; dst[i * 4 + 1] = b; dst[i * 4 + 2] = c; dst[i * 4 + 3] = d;
}
// persistent form: a fixed grid strides over the rays, `depth` rays in flight per lane group
__global__ __launch_bounds__(256) void gather16p(const char* __restrict__ src, uint32_t stride, f4* __restrict__ dst, uint64_t n) {
    const uint64_t total = n * 4, step = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step * 4) {
        f4 v[4]; 
#pragma unroll
        for (int k = 0; k < 4; k++) { const uint64_t j = i + k * step; if (j < total) v[k] = __builtin_nontemporal_load((const f4*)(src + (j >> 2) * stride) + (j & 3)); }
#pragma unroll
        for (int k = 0; k < 4; k++) { const uint64_t j = i + k * step; if (j < total) dst[j] = v[k]; }
    }
}
__global__ __launch_bounds__(256) void scatter20(const f4* __restrict__ src, char* __restrict__ dst, uint32_t stride, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f4 c = src[i * 4 + 2], h = src[i * 4 + 3];
    char* o = dst + i * stride;
    *(float*)(o + 44) = c.w; *(f4*)(o + 48) = h;
}
__global__ __launch_bounds__(256) void scatter32(const f4* __restrict__ src, char* __restrict__ dst, uint32_t stride, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // two lanes per ray: bytes 32..47 and 48..63
    if (i >= n * 2) return;
    *(f4*)(dst + (i >> 1) * stride + 32 + (i & 1) * 16) = src[(i >> 1) * 4 + 2 + (i & 1)];
}
__global__ __launch_bounds__(256) void scatter64(const f4* __restrict__ src, char* __restrict__ dst, uint32_t stride, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // four lanes per ray: the whole 64-byte prefix
    if (i >= n * 4) return;
    *(f4*)(dst + (i >> 2) * stride + (i & 3) * 16) = src[i];
}

template <class F> static double timeit(hipStream_t s, int reps, F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    double best = 1e30;
    for (int r = 0; r <= reps; r++) {
        hipEventRecord(a, s); f(); hipEventRecord(b, s); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        if (r && ms < best) best = ms;
    }
    hipEventDestroy(a); hipEventDestroy(b);
    return best;
}

int main(int argc, char** argv) {
    const uint64_t n = (uint64_t)(argc > 1 ? atof(argv[1]) : 16.0) * 1048576ull;
    hipStream_t s, s2; CK(hipStreamCreate(&s)); CK(hipStreamCreate(&s2));
    char *h128 = nullptr, *h64 = nullptr, *hpack = nullptr;
    CK(hipHostMalloc((void**)&h128, n * 128, hipHostMallocMapped)); CK(hipHostMalloc((void**)&h64, n * 64, hipHostMallocMapped)); CK(hipHostMalloc((void**)&hpack, n * 20, hipHostMallocDefault));
    memset(h128, 1, n * 128); memset(h64, 1, n * 64);
    f4* d = nullptr; char* dpack = nullptr;
    CK(hipMalloc((void**)&d, n * 64)); CK(hipMalloc((void**)&dpack, n * 20));
    char *m128 = nullptr, *m64 = nullptr;
    CK(hipHostGetDevicePointer((void**)&m128, h128, 0)); CK(hipHostGetDevicePointer((void**)&m64, h64, 0));
    const double up = n * 64 / 1e6, down = n * 20 / 1e6;   // MB: ms -> GB/s = MB / ms
    printf("%llu rays: %.0f MB up, %.0f MB down per call\n", (unsigned long long)n, up, down);
    double t;
    t = timeit(s, 3, [&] { hipMemcpyAsync(d, h64, n * 64, hipMemcpyHostToDevice, s); });                 printf("dma    up   contiguous 64 B/ray          %7.2f ms  %6.1f GB/s\n", t, up / t);
    t = timeit(s, 3, [&] { hipMemcpyAsync(hpack, dpack, n * 20, hipMemcpyDeviceToHost, s); });           printf("dma    down contiguous 20 B/ray          %7.2f ms  %6.1f GB/s\n", t, down / t);
    t = timeit(s, 3, [&] { hipMemcpyAsync(h64, d, n * 64, hipMemcpyDeviceToHost, s); });                 printf("dma    down contiguous 64 B/ray          %7.2f ms  %6.1f GB/s\n", t, up / t);
    {   // both directions at once
        hipEvent_t e; hipEventCreate(&e);
        t = timeit(s, 3, [&] { hipMemcpyAsync(h64, (char*)d, n * 32, hipMemcpyDeviceToHost, s2); hipEventRecord(e, s2); hipMemcpyAsync((char*)d + n * 32, h128, n * 32, hipMemcpyHostToDevice, s); hipStreamWaitEvent(s, e, 0); });
        printf("dma    up + down at once, %4.0f MB each   %7.2f ms  %6.1f GB/s each way\n", n * 32 / 1e6, t, n * 32 / 1e6 / t);
        hipEventDestroy(e);
    }
    t = timeit(s, 3, [&] { hipMemcpy2DAsync(d, 64, h128, 128, 64, n, hipMemcpyHostToDevice, s); });      printf("dma2d  up   64 of pitch 128              %7.2f ms  %6.1f GB/s\n", t, up / t);
    t = timeit(s, 2, [&] { hipMemcpy2DAsync(h64 + 44, 64, (char*)d + 44, 64, 20, n, hipMemcpyDeviceToHost, s); });   printf("dma2d  down 20 of pitch 64               %7.2f ms  %6.1f GB/s\n", t, down / t);
    t = timeit(s, 2, [&] { hipMemcpy2DAsync(h128 + 44, 128, (char*)d + 44, 64, 20, n, hipMemcpyDeviceToHost, s); }); printf("dma2d  down 20 of pitch 128              %7.2f ms  %6.1f GB/s\n", t, down / t);
    for (uint32_t stride : {64u, 128u}) {
        const char* src = stride == 64 ? m64 : m128;
        t = timeit(s, 3, [&] { hipLaunchKernelGGL(gather16, dim3((uint32_t)((n * 4 + 255) / 256)), dim3(256), 0, s, src, stride, d, n); });   printf("kernel gather 16 B/lane, stride %3u       %7.2f ms  %6.1f GB/s\n", stride, t, up / t);
        t = timeit(s, 3, [&] { hipLaunchKernelGGL(gather64, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, src, stride, d, n); });       printf("kernel gather 64 B/lane, stride %3u       %7.2f ms  %6.1f GB/s\n", stride, t, up / t);
        for (uint32_t g : {256u, 1024u, 4096u}) {
            t = timeit(s, 3, [&] { hipLaunchKernelGGL(gather16p, dim3(g), dim3(256), 0, s, src, stride, d, n); });                           printf("kernel gather persistent %4u wg, str %3u %7.2f ms  %6.1f GB/s\n", g, stride, t, up / t);
        }
        char* dst = (char*)src;
        t = timeit(s, 3, [&] { hipLaunchKernelGGL(scatter20, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, d, dst, stride, n); });      printf("kernel scatter 4+16 B/ray, stride %3u     %7.2f ms  %6.1f GB/s (of 20 B/ray)\n", stride, t, down / t);
        t = timeit(s, 3, [&] { hipLaunchKernelGGL(scatter32, dim3((uint32_t)((n * 2 + 255) / 256)), dim3(256), 0, s, d, dst, stride, n); });  printf("kernel scatter 32 B/ray, stride %3u       %7.2f ms  %6.1f GB/s (of 20 B/ray)\n", stride, t, down / t);
        t = timeit(s, 3, [&] { hipLaunchKernelGGL(scatter64, dim3((uint32_t)((n * 4 + 255) / 256)), dim3(256), 0, s, d, dst, stride, n); });  printf("kernel scatter 64 B/ray, stride %3u       %7.2f ms  %6.1f GB/s (of 20 B/ray)\n", stride, t, down / t);
    }
    // registered (not hipHostMalloc'ed) memory: what page-locking CALLER memory would give (not done: DESIGN.md par. 0)
    {
        char* p = (char*)aligned_alloc(4096, n * 64); memset(p, 1, n * 64);
        const auto t0 = std::chrono::steady_clock::now();
        CK(hipHostRegister(p, n * 64, hipHostRegisterMapped));
        const double reg = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        char* mp = nullptr; CK(hipHostGetDevicePointer((void**)&mp, p, 0));
        t = timeit(s, 3, [&] { hipMemcpyAsync(d, p, n * 64, hipMemcpyHostToDevice, s); });                                                     printf("registered: hipHostRegister %.1f ms; dma up %7.2f ms %6.1f GB/s", reg, t, up / t);
        t = timeit(s, 3, [&] { hipLaunchKernelGGL(gather16, dim3((uint32_t)((n * 4 + 255) / 256)), dim3(256), 0, s, mp, 64u, d, n); });       printf("; kernel gather %7.2f ms %6.1f GB/s\n", t, up / t);
        hipHostUnregister(p); free(p);
    }
    // the CPU side of the staged path: threads packing stride 128 -> pinned 64
    {
        char* pageable = (char*)aligned_alloc(4096, n * 128); memset(pageable, 2, n * 128);
        for (int threads : {1, 4, 8, 16, 32}) {
            if ((unsigned)threads > std::thread::hardware_concurrency()) break;
            double best = 1e30;
            for (int r = 0; r < 3; r++) {
                const auto t0 = std::chrono::steady_clock::now();
                std::vector<std::thread> th;
                for (int k = 0; k < threads; k++) th.emplace_back([=] { for (uint64_t i = n * k / threads; i < n * (k + 1) / threads; i++) memcpy(h64 + i * 64, pageable + i * 128, 64); });
                for (auto& x : th) x.join();
                best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            }
            printf("cpu pack stride 128 -> pinned 64, %2d threads %7.2f ms  %6.1f GB/s\n", threads, best, up / best);
        }
        free(pageable);
    }
    return 0;
}
