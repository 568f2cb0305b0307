// gather_lanes.hip — does a divergent per-lane load cost the CU's vector-memory pipe per INSTRUCTION or per ACTIVE LANE?
// Same pattern as gather_rate.hip (every active lane reads an 80-byte record as 5 x dwordx4 from a pseudo-random place of
// a table), but only the first ACTIVE lanes of each wave take part.  Reports cycles per wave-level load instruction per CU.
// If the cost were per instruction, 16 active lanes would cost what 64 cost; per lane, a quarter.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int NLOADS, int STRIDE_B>
__global__ __launch_bounds__(64) void k(const uint4* __restrict__ table, uint32_t nRecords, float* out, int iters, int active) {
    uint32_t s = (blockIdx.x * 64 + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0;
    if ((int)(threadIdx.x & 63) < active) {
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
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int NLOADS, int STRIDE_B> void run(const uint4* t, size_t tableBytes, float* out, int wavesPerCU, int active, const char* label) {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * wavesPerCU, iters = 2000;
    const uint32_t nRec = (uint32_t)(tableBytes / STRIDE_B);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NLOADS, STRIDE_B>), dim3(blocks), dim3(64), 0, 0, t, nRec, out, 64, active);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NLOADS, STRIDE_B>), dim3(blocks), dim3(64), 0, 0, t, nRec, out, iters, active);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instrPerCU = (double)wavesPerCU * iters * NLOADS;
    const double cyc = ms * 1e-3 * 2.4e9 / instrPerCU;
    const double gbs = (double)blocks * active * iters * NLOADS * 16 / (ms * 1e-3) / 1e9;
    printf("%-26s table %5zu MB  waves/CU %2d  active lanes %2d  %6.1f cyc(@2.4GHz)/wave-load/CU  %6.2f cyc/lane-load  %8.1f GB/s\n", label, tableBytes >> 20, wavesPerCU, active, cyc,
           cyc / active, gbs);
}

int main() {
    const size_t maxBytes = (size_t)1024 << 20;
    uint4* t; hipMalloc(&t, maxBytes); hipMemset(t, 1, maxBytes);
    float* out; hipMalloc(&out, 1 << 24);
    for (size_t mb : {1, 64, 1024}) {
        const size_t b = mb << 20;
        for (int active : {8, 16, 32, 48, 64}) {
            run<5, 80>(t, b, out, 24, active, "5 x 16B (80B records)");
            run<5, 128>(t, b, out, 24, active, "5 x 16B (128B aligned)");
            run<3, 48>(t, b, out, 24, active, "3 x 16B (48B records)");
        }
    }
    return 0;
}
