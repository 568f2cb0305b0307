// valu_rate.hip — issue-rate microbenchmark for the VALU ops the traversal kernels are made of.
// One wave per SIMD x OCC waves, each runs ITER iterations of 32 independent copies of one
// instruction (8 accumulators x 4).  Reports cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP> __global__ __launch_bounds__(64) void k(float* out, int iters, float a, float b) {
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    float2 p0 = {r0, r1}, p1 = {r2, r3}, p2 = {r4, r5}, p3 = {r6, r7}, pa = {a, a}, pb = {b, b};
    unsigned u0 = threadIdx.x * 2654435761u, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (OP == 0) {
#define X(n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r##n) : "v"(a), "v"(b));
                REP8(X)
#undef X
            } else if (OP == 1) {
#define X(n) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(r##n) : "v"(u##n));
                REP8(X)
#undef X
            } else if (OP == 2) {
#define X(n) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r##n) : "v"(a), "v"(b));
                REP8(X)
#undef X
            } else if (OP == 3) {
#define X(n) asm volatile("v_min_f32 %0, %0, %1" : "+v"(r##n) : "v"(a));
                REP8(X)
#undef X
            } else if (OP == 4) {
#define X(n) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r##n) : "v"(a));
                REP8(X)
#undef X
            } else if (OP == 5) {
#define X(n) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(r##n), "v"(a) : "vcc");
                REP8(X)
#undef X
            } else if (OP == 6) {
#define X(n) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(u##n));
                REP8(X)
#undef X
            } else if (OP == 7) {
#define X(n) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(u##n));
                REP8(X)
#undef X
            } else if (OP == 8) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(pa), "v"(pb));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(pa), "v"(pb));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(pa), "v"(pb));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(pa), "v"(pb));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(pa), "v"(pb));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(pa), "v"(pb));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(pa), "v"(pb));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(pa), "v"(pb));
            } else if (OP == 9) {
#define X(n) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(u##n) : "v"(u0), "v"(u1));
                REP8(X)
#undef X
            } else if (OP == 10) {
#define X(n) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r##n) : "v"(a));
                REP8(X)
#undef X
            } else if (OP == 11) {
#define X(n) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u##n) : "v"(u1));
                REP8(X)
#undef X
            } else if (OP == 12) {
#define X(n) asm volatile("v_rcp_f32 %0, %0" : "+v"(r##n));
                REP8(X)
#undef X
            } else if (OP == 13) {
#define X(n) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(r##n) : "v"(u1));
                REP8(X)
#undef X
            } else if (OP == 14) {
#define X(n) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r##n) : "v"(a), "v"(b));
                REP8(X)
#undef X
            } else if (OP == 16) {
#define X(n) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(r##n) : "v"(u##n), "v"(a));
                REP8(X)
#undef X
            } else if (OP == 17) {
#define X(n) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(r##n) : "v"(u##n));
                REP8(X)
#undef X
            } else if (OP == 18) {
#define X(n) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p0) : "v"(pa)); asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p1) : "v"(pa));
                X(0) X(1) X(2) X(3)
#undef X
            } else if (OP == 19) {
#define X(n) asm volatile("v_cvt_f32_ubyte0_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(r##n) : "v"(u##n));
                REP8(X)
#undef X
            } else if (OP == 15) {
#define X(n) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u##n) : "v"(u0), "v"(u1));
                REP8(X)
#undef X
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y +
                                         (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7);
}

template <int OP> double run(float* d, int occ, int iters) {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 4 * occ;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, 16, 1.0001f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // instructions per SIMD = occ waves * iters * 32
    const double instr = (double)occ * iters * 32;
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    return ms * 1e-3 * (double)clk * 1e3 / instr;  // cycles (at max clock) per wave-instruction per SIMD
}

int main() {
    float* d; hipMalloc(&d, 1 << 26);
    const char* names[] = {"v_fma_f32", "v_cvt_f32_ubyte1", "v_max3_f32", "v_min_f32", "v_cndmask_b32", "v_cmp_le_f32", "v_lshlrev_b32", "v_bfe_u32",
                           "v_pk_fma_f32", "v_or3_b32", "v_mul_f32", "v_add_u32", "v_rcp_f32", "v_ldexp_f32", "v_med3_f32", "v_perm_b32", "v_fma_mix_f32", "v_cvt_f32_f16", "v_pk_mul_f32", "v_cvt_f32_ubyte_sdwa"};
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("clock attr %d kHz\n", clk);
    for (int occ : {4}) {
        const int it = 40000;
        double r[20] = {run<0>(d, occ, it), run<1>(d, occ, it), run<2>(d, occ, it), run<3>(d, occ, it), run<4>(d, occ, it), run<5>(d, occ, it), run<6>(d, occ, it), run<7>(d, occ, it),
                        run<8>(d, occ, it), run<9>(d, occ, it), run<10>(d, occ, it), run<11>(d, occ, it), run<12>(d, occ, it), run<13>(d, occ, it), run<14>(d, occ, it), run<15>(d, occ, it), run<16>(d, occ, it), run<17>(d, occ, it), run<18>(d, occ, it), run<19>(d, occ, it)};
        for (int i = 0; i < 20; i++) printf("occ %d  %-18s %.2f cyc/instr/SIMD\n", occ, names[i], r[i]);
    }
    return 0;
}
