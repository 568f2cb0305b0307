// valu_issue.hip — how many cycles does one wave64 VALU instruction occupy a SIMD's issue port on gfx950?
//
// Settles the peak behind roofline.valu_issue (round-2 review): the hardware guide says SIMD-32, v_fma_f32 (wave64) 2 cycles, 157.3 TFLOP/s
// of FP32 vector math; round 2's valu_rate.hip measured 4.1 cycles with wall-clock time x the NOMINAL clock attribute (blind to DVFS), at one
// occupancy, with asm-volatile chains only.  Here:
//   * cycles are read in the kernel (s_memtime: the shader clock itself — clock64()), per wave, next to the wall-clock time of the launch
//     (HIP events); their ratio is the clock the chip really ran at;
//   * occupancy 1 / 2 / 4 / 8 waves per SIMD, enforced: dynamic LDS sized so that exactly 4 x occ one-wave workgroups fit a CU;
//   * per op: an asm-volatile chain (8 independent accumulators), a COMPILED chain (plain fmaf / fminf / ... that the compiler schedules as it
//     likes), packed v_pk_fma_f32 in both forms, and the instruction mix of one CWBVH child-box test (cvt_ubyte + fma + max3 / min3 + compare).
// Per SIMD: cycles per wave-instruction = mean wave lifetime in cycles / (occ x instructions per wave): all occ waves of a SIMD run for the
// same time, so together they issue occ x N instructions in one wave lifetime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum Op { FMA_ASM, FMA_C, PKFMA_ASM, PKFMA_C, CVT_UB_ASM, MAX3_ASM, MIN_C, CNDMASK_ASM, MUL_ASM, ADDU_ASM, BOXMIX_C, FMA_DEP_ASM, NOPS };
static const char* kNames[] = {"v_fma_f32 (asm, 8 chains)", "fmaf x 8 (compiled: SLP-packed to v_pk_fma_f32)", "v_pk_fma_f32 (asm, 4 chains)", "float2 fma (compiled, 4 chains)",
                               "v_cvt_f32_ubyte1 (asm)", "v_max3_f32 (asm)", "fminf (compiled)", "v_cndmask_b32 (asm)", "v_mul_f32 (asm)", "v_add_u32 (asm)",
                               "CWBVH child-box mix (compiled)", "v_fma_f32 (asm, ONE dependent chain)"};
// VALU instructions per loop iteration (what the cycles are divided by)
static const int kPerIter[] = {32, 32, 16, 16, 32, 32, 32, 32, 32, 32, 0 /* counted from the ISA: see main */, 32};

typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP> __global__ __launch_bounds__(64) void k(float* out, unsigned long long* cyc, int iters, float a, float b, unsigned seed) {
    extern __shared__ char pad[];   // only sizes the workgroup (occupancy)
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    f2 p0 = {r0, r1}, p1 = {r2, r3}, p2 = {r4, r5}, p3 = {r6, r7};
    const f2 pa = {a, a}, pb = {b, b};
    unsigned u0 = threadIdx.x * 2654435761u + seed, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    const unsigned long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (OP == FMA_ASM) {
#define X(n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r##n) : "v"(a), "v"(b));
                REP8(X)
#undef X
            } else if (OP == FMA_DEP_ASM) {
#define X(n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r0) : "v"(a), "v"(b));
                REP8(X)
#undef X
            } else if (OP == FMA_C) {
#define X(n) r##n = __builtin_fmaf(r##n, a, b);
                REP8(X)
#undef X
            } else if (OP == PKFMA_ASM) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(pa), "v"(pb));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(pa), "v"(pb));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(pa), "v"(pb));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(pa), "v"(pb));
            } else if (OP == PKFMA_C) {
                p0 = __builtin_elementwise_fma(p0, pa, pb); p1 = __builtin_elementwise_fma(p1, pa, pb);
                p2 = __builtin_elementwise_fma(p2, pa, pb); p3 = __builtin_elementwise_fma(p3, pa, pb);
            } else if (OP == CVT_UB_ASM) {
#define X(n) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(r##n) : "v"(u##n));
                REP8(X)
#undef X
            } else if (OP == MAX3_ASM) {
#define X(n) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r##n) : "v"(a), "v"(b));
                REP8(X)
#undef X
            } else if (OP == MIN_C) {
#define X(n) r##n = __builtin_fminf(r##n * a, b);   /* (mul + min: 2 instructions; kPerIter counts 32 for the mins, the muls are reported with them: see main) */
                REP8(X)
#undef X
            } else if (OP == CNDMASK_ASM) {
#define X(n) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r##n) : "v"(a));
                REP8(X)
#undef X
            } else if (OP == MUL_ASM) {
#define X(n) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r##n) : "v"(a));
                REP8(X)
#undef X
            } else if (OP == ADDU_ASM) {
#define X(n) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u##n) : "v"(u1));
                REP8(X)
#undef X
            } else if (OP == BOXMIX_C) {
                // one child of cw_test_node (cwbvh_node.h), twice per j: 6 cvt_ubyte, 6 fma, max3 + max, min3 + min, compare + mask update; every
                // child's plane words derive from the evolving chain, so nothing is loop invariant
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const unsigned w = u0 + (unsigned)(2 * j + c) * 0x01010101u, w2 = u2 ^ w, w3 = u4 + w;
                    const float tnx = __builtin_fmaf((float)(w & 255), a, r0), tfx = __builtin_fmaf((float)((w >> 8) & 255), a, r0);
                    const float tny = __builtin_fmaf((float)(w2 & 255), b, r1), tfy = __builtin_fmaf((float)((w2 >> 8) & 255), b, r1);
                    const float tnz = __builtin_fmaf((float)(w3 & 255), a, r2), tfz = __builtin_fmaf((float)((w3 >> 8) & 255), a, r2);
                    const float cmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(tnx, tny), tnz), 0.f);
                    const float cmax = __builtin_fminf(__builtin_fminf(__builtin_fminf(tfx, tfy), tfz), r3);
                    if (cmin <= cmax) u6 |= 1u << ((w >> 16) & 31);
                    u0 += u6; u2 ^= u0; u4 += u2;   // keep the chain data dependent on the result
                }
            }
        }
    }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y +
                                         (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7);
}

struct Res { double cycPerInstr, nsPerInstr, ghz; };

template <int OP> Res run(float* d, unsigned long long* dc, int cus, int occ, int iters, int perIter) {
    const int blocks = cus * 4 * occ;
    const size_t lds = (size_t)(160 * 1024) / (4 * occ) - 64;   // exactly 4 x occ one-wave workgroups fit a CU
    hipFuncSetAttribute((const void*)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), lds, 0, d, dc, 64, 1.0001f, 0.5f, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), lds, 0, d, dc, iters, 1.0001f, 0.5f, 1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(blocks);
    hipMemcpy(c.data(), dc, blocks * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto v : c) sum += (double)v;
    const double meanCyc = sum / blocks;
    const double instrPerWave = (double)iters * perIter;
    Res r;
    r.cycPerInstr = meanCyc / (occ * instrPerWave);
    r.nsPerInstr = ms * 1e6 / (occ * instrPerWave);
    r.ghz = meanCyc / (ms * 1e6);   // cycles a wave lived / wall time of the launch (launch overhead makes this a lower bound)
    hipEventDestroy(e0); hipEventDestroy(e1);
    return r;
}

int main(int argc, char** argv) {
    int iters = 20000;
    if (argc > 1) iters = atoi(argv[1]);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("device %s: %d CUs, nominal clock %d kHz; %d loop iterations per wave\n", p.name, cus, clk, iters);
    printf("cycles = s_memtime (shader clock) inside the kernel, mean wave lifetime; ns = HIP-event time of the launch; clock = their ratio\n");
    float* d; hipMalloc(&d, (size_t)cus * 32 * 64 * 4);
    unsigned long long* dc; hipMalloc(&dc, (size_t)cus * 32 * 8);
    printf("%-40s %4s %22s %14s %10s\n", "op", "occ", "cycles/instr/SIMD", "ns/instr/SIMD", "clock GHz");
    for (int occ : {1, 2, 4, 8}) {
        // BOXMIX: 207 VALU instructions per loop iteration in the ISA of this build (hipcc -S: 48 v_fma_f32, 48 v_cvt_f32_ubyte, 16 max3 / max, 16 min3 / min,
        // 8 v_cmp, 8 v_cndmask, 63 integer) = 8 child-box tests, the mix of one cw_test_node (209 VALU, DESIGN.md par. 5)
        Res r[NOPS] = {run<FMA_ASM>(d, dc, cus, occ, iters, 32), run<FMA_C>(d, dc, cus, occ, iters, 16), run<PKFMA_ASM>(d, dc, cus, occ, iters, 16),
                       run<PKFMA_C>(d, dc, cus, occ, iters, 16), run<CVT_UB_ASM>(d, dc, cus, occ, iters, 32), run<MAX3_ASM>(d, dc, cus, occ, iters, 32),
                       run<MIN_C>(d, dc, cus, occ, iters, 64), run<CNDMASK_ASM>(d, dc, cus, occ, iters, 32), run<MUL_ASM>(d, dc, cus, occ, iters, 32),
                       run<ADDU_ASM>(d, dc, cus, occ, iters, 32), run<BOXMIX_C>(d, dc, cus, occ, iters / 4, 207), run<FMA_DEP_ASM>(d, dc, cus, occ, iters, 32)};
        for (int i = 0; i < NOPS; i++)
            printf("%-40s %4d %22.2f %14.3f %10.2f%s\n", kNames[i], occ, r[i].cycPerInstr, r[i].nsPerInstr, r[i].ghz,
                   i == BOXMIX_C ? "   (per VALU instruction of the mix: 207 per 8 child tests)" : i == MIN_C ? "   (per instruction of mul + min pairs)" : i == PKFMA_ASM || i == PKFMA_C || i == FMA_C ? "   (per PACKED instruction = 2 FMAs per lane)" : "");
        fflush(stdout);
    }
    // FP32 FMA rate of the whole chip from the best row
    return 0;
}
