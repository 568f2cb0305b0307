// gather_rate.hip — throughput of divergent per-lane loads (the traversal's node / triangle
// fetch pattern): every lane reads BYTES contiguous bytes (as dwordx4 / x2 / x1 loads) from a
// pseudo-random record of a table of TABLE_MB megabytes.  Reports cycles per wave-level load
// instruction per CU and the achieved GB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int NLOADS, int STRIDE_B>
__global__ __launch_bounds__(64) void k(const uint4* __restrict__ table, uint32_t nRecords, float* out, int iters) {
    uint32_t s = (blockIdx.x * 64 + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0;
    for (int i = 0; i < iters; i++) {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        const uint32_t rec = s % nRecords;
        const char* p = (const char*)table + (size_t)rec * STRIDE_B;
        uint4 v[NLOADS];
#pragma unroll
        for (int j = 0; j < NLOADS; j++) v[j] = *(const uint4*)(p + 16 * j);
#pragma unroll
        for (int j = 0; j < NLOADS; j++) acc += __uint_as_float(v[j].x ^ v[j].w);
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int NLOADS, int STRIDE_B> void run(const uint4* t, size_t tableBytes, float* out, int wavesPerCU, const char* label) {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * wavesPerCU, iters = 4000;
    const uint32_t nRec = (uint32_t)(tableBytes / STRIDE_B);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NLOADS, STRIDE_B>), dim3(blocks), dim3(64), 0, 0, t, nRec, out, 64);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NLOADS, STRIDE_B>), dim3(blocks), dim3(64), 0, 0, t, nRec, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instrPerCU = (double)wavesPerCU * iters * NLOADS;
    const double cyc = ms * 1e-3 * 2.4e9 / instrPerCU;
    const double gbs = (double)blocks * 64 * iters * NLOADS * 16 / (ms * 1e-3) / 1e9;
    printf("%-28s table %5zu MB  waves/CU %2d  %6.1f cyc(@2.4GHz)/wave-load/CU  %8.1f GB/s  (%.2f ms)\n", label, tableBytes >> 20, wavesPerCU, cyc, gbs, ms);
}

int main() {
    const size_t maxBytes = (size_t)512 << 20;
    uint4* t; hipMalloc(&t, maxBytes); hipMemset(t, 1, maxBytes);
    float* out; hipMalloc(&out, 1 << 24);
    for (size_t mb : {1, 16, 96, 512}) {
        const size_t b = mb << 20;
        for (int w : {8, 24}) {
            run<1, 16>(t, b, out, w, "1 x 16B (16B records)");
            run<5, 80>(t, b, out, w, "5 x 16B (80B records)");
            run<5, 128>(t, b, out, w, "5 x 16B (128B aligned)");
            run<8, 128>(t, b, out, w, "8 x 16B (128B records)");
            run<3, 48>(t, b, out, w, "3 x 16B (48B records)");
        }
    }
    return 0;
}
