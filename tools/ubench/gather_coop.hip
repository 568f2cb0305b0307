// gather_coop.hip — can a wave fetch its lanes' node records faster TOGETHER than each lane fetching its own?
//
// In the traversal kernels every lane loads its own 80-byte node as 5 x global_load_dwordx4: a fully divergent wave instruction costs
// the CU's vector-memory pipe about one cycle per active lane (tools/ubench/gather_lanes.hip) whatever the cache level — 5 x 64 cycles
// per node phase.  Lanes 4q..4q+3 loading four CONSECUTIVE 16-byte pieces of ONE record are a 64-byte coalesced access for the address
// coalescer, so the same bytes could go through in a quarter of the cycles; the pieces then have to reach the lane that owns the node:
//   A  baseline: every lane loads its own record, 5 x dwordx4;
//   B  cooperative through registers + LDS: slot s = k * 64 + lane fetches piece s % 5 of the record of lane s / 5; ds_write_b128 to
//      slot s; the owner reads its 5 pieces back (ds_read_b128);
//   C  cooperative with global_load_lds_dwordx4 (gfx950): the same slots, the load lands in LDS at base + lane * 16 without passing
//      through VGPRs; the owner reads its 5 pieces.
// All three end with the same 80 bytes in the owner lane's registers (checksummed).  Records: `stride` bytes apart in a table of the given
// size (1 MB: L2 hits; 64 MB: beyond the L2s, inside the Infinity Cache; 1 GB: HBM).  Reported: cycles (s_memtime) per record-fetching
// pass per CU, i.e. per "node phase" of 64 lanes, with 24 one-wave workgroups per CU as the traversal kernels run.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned next_rec(unsigned& s, unsigned nRecords) {
    s ^= s << 13; s ^= s >> 17; s ^= s << 5;
    return s % nRecords;
}

template <int MODE, int STRIDE_B>
__global__ __launch_bounds__(64) void k(const char* __restrict__ table, unsigned nRecords, unsigned* out, unsigned long long* cyc, int iters, int active) {
    __shared__ u4 stage[5 * 64];       // B, C: 5 KB per wave
    __shared__ unsigned recOf[64];
    unsigned s = (blockIdx.x * 64 + threadIdx.x) * 2654435761u + 12345u;
    unsigned acc = 0;
    const unsigned lane = threadIdx.x;
    const bool on = (int)lane < active;
    const unsigned long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        const unsigned rec = next_rec(s, nRecords);
        if (MODE == 0) {
            if (on) {
                const char* p = table + (size_t)rec * STRIDE_B;
                u4 v[5];
#pragma unroll
                for (int j = 0; j < 5; j++) v[j] = *(const u4*)(p + 16 * j);
#pragma unroll
                for (int j = 0; j < 5; j++) acc += v[j].x ^ v[j].w;
            }
        } else {
            // owners announce their records, compacted by rank so that the slots of the first `nOn` owners are dense
            const unsigned long long m = __ballot(on);
            const unsigned nOn = (unsigned)__popcll(m);
            const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (on) recOf[rank] = rec;
            __syncthreads();
            const unsigned nSlots = nOn * 5u;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const unsigned slot = (unsigned)k * 64u + lane;
                if (slot < nSlots) {
                    const unsigned owner = slot / 5u, piece = slot - owner * 5u;
                    const char* p = table + (size_t)recOf[owner] * STRIDE_B + piece * 16u;
                    if (MODE == 1) stage[slot] = *(const u4*)p;
                    else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p, (__attribute__((address_space(3))) void*)&stage[k * 64], 16, 0, 0);
                }
            }
            if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (on) {
#pragma unroll
                for (int j = 0; j < 5; j++) { const u4 v = stage[rank * 5u + j]; acc += v.x ^ v.w; }
            }
            __syncthreads();
        }
    }
    const unsigned long long t1 = clock64();
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + lane] = acc;
}

template <int MODE, int STRIDE_B> void run(const char* t, size_t tableBytes, unsigned* out, unsigned long long* dc, int cus, int wavesPerCU, int active, const char* label, unsigned* checksum) {
    const int blocks = cus * wavesPerCU, iters = 1500;
    const unsigned nRec = (unsigned)(tableBytes / STRIDE_B);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, STRIDE_B>), dim3(blocks), dim3(64), 0, 0, t, nRec, out, dc, 32, active);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, STRIDE_B>), dim3(blocks), dim3(64), 0, 0, t, nRec, out, dc, iters, active);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(blocks);
    (void)hipMemcpy(c.data(), dc, (size_t)blocks * 8, hipMemcpyDeviceToHost);
    std::vector<unsigned> o((size_t)blocks * 64);
    (void)hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost);
    unsigned sum = 0; for (auto v : o) sum += v;
    double cs = 0; for (auto v : c) cs += (double)v;
    const double passCyc = cs / blocks / iters;                 // cycles one wave spends per pass
    const double perCU = passCyc / wavesPerCU;                  // the CU completes wavesPerCU passes in that time
    printf("%-46s table %5zu MB  active %2d  %7.1f cyc/pass/wave  %6.1f cyc/pass/CU  %6.2f cyc/record/CU  %8.1f G records/s  checksum %08x%s\n", label, tableBytes >> 20, active,
           passCyc, perCU, perCU / active, (double)blocks * active * iters / (ms * 1e-3) / 1e9, sum, (*checksum && *checksum != sum) ? "  MISMATCH" : "");
    if (!*checksum) *checksum = sum;
    fflush(stdout);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const size_t maxBytes = (size_t)1024 << 20;
    char* t; (void)hipMalloc(&t, maxBytes);
    {   // distinct words, so the checksum notices a piece that went to the wrong lane
        std::vector<unsigned> h(maxBytes / 4);
        for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned)i * 2654435761u;
        (void)hipMemcpy(t, h.data(), maxBytes, hipMemcpyHostToDevice);
    }
    unsigned* out; (void)hipMalloc(&out, (size_t)cus * 32 * 64 * 4);
    unsigned long long* dc; (void)hipMalloc(&dc, (size_t)cus * 32 * 8);
    for (size_t mb : {1, 64, 1024}) {
        const size_t b = mb << 20;
        for (int active : {16, 48, 64}) {
            unsigned c80 = 0, c128 = 0;
            run<0, 80>(t, b, out, dc, cus, 24, active, "A own record, 5 x dwordx4 (80B records)", &c80);
            run<1, 80>(t, b, out, dc, cus, 24, active, "B cooperative via VGPR + LDS (80B records)", &c80);
            run<2, 80>(t, b, out, dc, cus, 24, active, "C cooperative via global_load_lds (80B)", &c80);
            run<0, 128>(t, b, out, dc, cus, 24, active, "A own record (128B aligned)", &c128);
            run<1, 128>(t, b, out, dc, cus, 24, active, "B cooperative via VGPR + LDS (128B aligned)", &c128);
            run<2, 128>(t, b, out, dc, cus, 24, active, "C cooperative via global_load_lds (128B)", &c128);
        }
    }
    return 0;
}
