// ref_ocl.cpp — TEST / MEASUREMENT INFRASTRUCTURE ONLY.
//
// Runs the REFERENCE's own OpenCL traversal kernels (batch_ailalaine / batch_gpu4way /
// batch_cwbvh, traverse_bvh2.cl:209, traverse_bvh4.cl:277, traverse_cwbvh.cl:554) on the GPU
// through ROCm's OpenCL runtime, so that the HIP engine can be timed next to the thing it
// replaces on the same MI355X (BASELINE config 2: "... vs reference traverse_bvh2.cl").
// tiny_ocl.h itself cannot be used headless (InitCL demands cl_khr_gl_sharing,
// tiny_ocl.h:971-978), so this is a minimal host that does what tinyocl::Kernel does:
// "#define ISAMD" + traverse.cl with its three #includes expanded (tiny_ocl.h:758-805), built
// with the reference's options (tiny_ocl.h:816-825), launched 1-D with local size 64 and timed
// with CL profiling events (tiny_bvh_speedtest.cpp:1117-1131).
// The kernel SOURCE TEXT is read from the reference checkout at BUILD time by oracle/Makefile
// (generated header oracle/_ref/ref_cl_source.inc, git-ignored); no reference source is stored
// in this repository.  Output: oracle/_ref/libtinybvh_refocl.so.
#define CL_TARGET_OPENCL_VERSION 200
#include <CL/cl.h>

#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

static const char* kSource =
#include "_ref/ref_cl_source.inc"
    ;
// wavefront.cl with all its #includes expanded (tools.cl, traverse.cl and the per-layout files): the path tracer of
// tiny_bvh_gpu.cpp, run here frame by frame exactly as that demo's Tick() does (tiny_bvh_gpu.cpp:128-158)
static const char* kWavefrontSource =
#include "_ref/ref_wavefront_source.inc"
    ;

// wavefront2.cl (tiny_bvh_gpu2.cpp: TLAS over instanced BVH8_CWBVH BLASes) with its #includes expanded: only its SetRenderData and
// Extend kernels are run here, to time traverse_tlas (traverse_tlas.cl:13-107) next to this library's two-level kernels
static const char* kWavefront2Source =
#include "_ref/ref_wavefront2_source.inc"
    ;

namespace {
cl_context g_ctx = nullptr;
cl_device_id g_dev = nullptr;
cl_command_queue g_q = nullptr;
cl_program g_prog = nullptr;
char g_err[4096] = "";
char g_devname[256] = "";
}  // namespace

extern "C" {

const char* refocl_error() { return g_err; }
const char* refocl_device() { return g_devname; }

int refocl_init() {
    if (g_prog) return 0;
    cl_uint np = 0;
    cl_platform_id plats[8];
    if (clGetPlatformIDs(8, plats, &np) != CL_SUCCESS || np == 0) { snprintf(g_err, sizeof g_err, "no OpenCL platform"); return -1; }
    for (cl_uint p = 0; p < np && !g_dev; p++) {
        cl_uint nd = 0;
        cl_device_id devs[16];
        if (clGetDeviceIDs(plats[p], CL_DEVICE_TYPE_GPU, 16, devs, &nd) == CL_SUCCESS && nd > 0) g_dev = devs[0];
    }
    if (!g_dev) { snprintf(g_err, sizeof g_err, "no OpenCL GPU device"); return -2; }
    clGetDeviceInfo(g_dev, CL_DEVICE_NAME, sizeof g_devname, g_devname, nullptr);
    cl_int e;
    g_ctx = clCreateContext(nullptr, 1, &g_dev, nullptr, nullptr, &e);
    if (e != CL_SUCCESS) { snprintf(g_err, sizeof g_err, "clCreateContext %d", e); return -3; }
    cl_queue_properties props[] = {CL_QUEUE_PROPERTIES, CL_QUEUE_PROFILING_ENABLE, 0};
    g_q = clCreateCommandQueueWithProperties(g_ctx, g_dev, props, &e);
    if (e != CL_SUCCESS) { snprintf(g_err, sizeof g_err, "clCreateCommandQueue %d", e); return -4; }
    const std::string src = std::string("#define ISAMD\n") + kSource;
    const char* s = src.c_str();
    size_t len = src.size();
    g_prog = clCreateProgramWithSource(g_ctx, 1, &s, &len, &e);
    if (e != CL_SUCCESS) { snprintf(g_err, sizeof g_err, "clCreateProgramWithSource %d", e); return -5; }
    e = clBuildProgram(g_prog, 0, nullptr, "-cl-std=CL2.0 -cl-strict-aliasing -cl-fast-relaxed-math -cl-single-precision-constant ", nullptr, nullptr);
    if (e != CL_SUCCESS) {
        size_t n = 0;
        clGetProgramBuildInfo(g_prog, g_dev, CL_PROGRAM_BUILD_LOG, sizeof g_err - 64, g_err + 32, &n);
        memcpy(g_err, "clBuildProgram failed:          ", 32);
        g_prog = nullptr;
        return -6;
    }
    return 0;
}

// Debug aid: write the compiled device binary (an HSA code object) to `path` for llvm-objdump.
int refocl_dump_binary(const char* path) {
    if (refocl_init()) return -1;
    size_t sz = 0;
    clGetProgramInfo(g_prog, CL_PROGRAM_BINARY_SIZES, sizeof sz, &sz, nullptr);
    std::vector<unsigned char> bin(sz);
    unsigned char* ptr = bin.data();
    clGetProgramInfo(g_prog, CL_PROGRAM_BINARIES, sizeof ptr, &ptr, nullptr);
    FILE* f = fopen(path, "wb");
    if (!f) return -2;
    fwrite(bin.data(), 1, sz, f);
    fclose(f);
    return (int)sz;
}

// layout (BVHBase::BVHType, as in include/tinybvh_amd.h): 5 = batch_ailalaine(nodes, idx, verts, rays); 8 = batch_gpu4way(blocks, rays);
//         10 = batch_cwbvh(nodes, tris, rays).  bufN / bytesN are the layout's blobs in that order.
// rays: n packed 64-byte records (in/out).  Runs 1 warm-up + `passes` timed launches with global
// size n and local size 64 (tiny_bvh_speedtest.cpp:1122-1131); returns the mean kernel time in
// milliseconds from the profiling events, or a negative error.
double refocl_run(int layout, const void* buf0, uint64_t bytes0, const void* buf1, uint64_t bytes1, const void* buf2, uint64_t bytes2,
                  void* rays, uint64_t n, int passes) {
    if (refocl_init()) return -1.0;
    const char* name = layout == 5 ? "batch_ailalaine" : layout == 8 ? "batch_gpu4way" : layout == 10 ? "batch_cwbvh" : nullptr;
    if (!name) { snprintf(g_err, sizeof g_err, "bad layout"); return -2.0; }
    cl_int e;
    cl_kernel k = clCreateKernel(g_prog, name, &e);
    if (e != CL_SUCCESS) { snprintf(g_err, sizeof g_err, "clCreateKernel(%s) %d", name, e); return -3.0; }
    const void* bufs[3] = {buf0, buf1, buf2};
    const uint64_t bytes[3] = {bytes0, bytes1, bytes2};
    const int nb = layout == 5 ? 3 : layout == 8 ? 1 : 2;
    cl_mem mem[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < nb; i++) {
        mem[i] = clCreateBuffer(g_ctx, CL_MEM_READ_ONLY | CL_MEM_COPY_HOST_PTR, bytes[i] ? bytes[i] : 16, (void*)bufs[i], &e);
        if (e != CL_SUCCESS) { snprintf(g_err, sizeof g_err, "clCreateBuffer %d (%llu bytes) %d", i, (unsigned long long)bytes[i], e); return -4.0; }
        clSetKernelArg(k, i, sizeof(cl_mem), &mem[i]);
    }
    mem[nb] = clCreateBuffer(g_ctx, CL_MEM_READ_WRITE | CL_MEM_COPY_HOST_PTR, n * 64, rays, &e);
    if (e != CL_SUCCESS) { snprintf(g_err, sizeof g_err, "clCreateBuffer rays %d", e); return -5.0; }
    clSetKernelArg(k, nb, sizeof(cl_mem), &mem[nb]);
    const size_t global = (size_t)n, local = 64;   // n must be a multiple of 64 (the reference's batches are)
    double total = 0;
    for (int p = 0; p <= passes; p++) {
        cl_event ev;
        e = clEnqueueNDRangeKernel(g_q, k, 1, nullptr, &global, &local, 0, nullptr, &ev);
        if (e != CL_SUCCESS) { snprintf(g_err, sizeof g_err, "clEnqueueNDRangeKernel %d", e); return -6.0; }
        clWaitForEvents(1, &ev);
        cl_ulong t0 = 0, t1 = 0;
        clGetEventProfilingInfo(ev, CL_PROFILING_COMMAND_START, sizeof t0, &t0, nullptr);
        clGetEventProfilingInfo(ev, CL_PROFILING_COMMAND_END, sizeof t1, &t1, nullptr);
        if (p) total += (double)(t1 - t0) * 1e-6;
        clReleaseEvent(ev);
    }
    clEnqueueReadBuffer(g_q, mem[nb], CL_TRUE, 0, n * 64, rays, 0, nullptr, nullptr);
    for (int i = 0; i <= nb; i++) clReleaseMemObject(mem[i]);
    clReleaseKernel(k);
    return total / (passes > 0 ? passes : 1);
}

// ---- the reference's wavefront path tracer (wavefront.cl), frame loop of tiny_bvh_gpu.cpp:128-158 --------------------
// nodes / tris: BVH8_CWBVH blobs; verts: the vertex array (3 float4 per triangle, material in v0.w); noise: the 128 x 128 x 8
// blue-noise table; eye, p0, p1, p2: camera (float4 each); renders `frames` frames into one accumulator (spp = 1 .. frames,
// frame seeds as the demo's) and returns accumulator / frames as width * height float4.  Returns 0 or a negative error.
namespace { cl_program g_wprog = nullptr; std::string g_wpatch; }
// patch: "old text=>new text;;old2=>new2" edits applied to the source before it is compiled (NULL or "" = the reference's text as
// it is).  tests/test_wavefront_reference.py uses it for ONE documented purpose: Connect adds a shadow ray's contribution with a
// plain read-modify-write (wavefront.cl:265), which loses contributions when two shadow rays of one pixel are in flight; the
// test's patch makes that addition atomic.  tools/wf_debug.py uses it to switch terms off while bisecting.
int refocl_wavefront(const void* nodes, uint64_t nodeBytes, const void* tris, uint64_t triBytes, const void* verts, uint64_t vertBytes,
                     const uint32_t* noise, const float* eye, const float* p0, const float* p1, const float* p2, uint32_t width, uint32_t height,
                     uint32_t frames, uint32_t iterations, const char* patch, float* out) {
    if (refocl_init()) return -1;
    if (iterations == 0 || iterations > 3) iterations = 3;   // the demo runs 3 (tiny_bvh_gpu.cpp:143); fewer isolates the first vertices for debugging
    cl_int e;
    const std::string wantPatch = patch ? patch : "";
    if (!g_wprog || wantPatch != g_wpatch) {
        if (g_wprog) { clReleaseProgram(g_wprog); g_wprog = nullptr; }
        std::string src = std::string("#define ISAMD\n") + kWavefrontSource;
        for (size_t at = 0; at < wantPatch.size();) {
            size_t end = wantPatch.find(";;", at); if (end == std::string::npos) end = wantPatch.size();
            const std::string one = wantPatch.substr(at, end - at);
            const size_t arrow = one.find("=>");
            if (arrow != std::string::npos) {
                const std::string from = one.substr(0, arrow), to = one.substr(arrow + 2);
                size_t pos = src.find(from), cnt = 0;
                while (pos != std::string::npos && !from.empty()) { src.replace(pos, from.size(), to); pos = src.find(from, pos + to.size()); cnt++; }
                if (cnt == 0) { snprintf(g_err, sizeof g_err, "wavefront: patch text not found: %s", from.c_str()); return -11; }
            }
            at = end + 2;
        }
        const char* s = src.c_str();
        size_t len = src.size();
        g_wprog = clCreateProgramWithSource(g_ctx, 1, &s, &len, &e);
        if (e != CL_SUCCESS) { snprintf(g_err, sizeof g_err, "wavefront: clCreateProgramWithSource %d", e); return -5; }
        e = clBuildProgram(g_wprog, 0, nullptr, "-cl-std=CL2.0 -cl-strict-aliasing -cl-fast-relaxed-math -cl-single-precision-constant ", nullptr, nullptr);
        if (e != CL_SUCCESS) {
            size_t n = 0;
            clGetProgramBuildInfo(g_wprog, g_dev, CL_PROGRAM_BUILD_LOG, sizeof g_err - 64, g_err + 32, &n);
            memcpy(g_err, "wavefront clBuildProgram failed:", 32);
            g_wprog = nullptr;
            return -6;
        }
        g_wpatch = wantPatch;
    }
    const char* names[9] = {"SetRenderData", "Clear", "Generate", "Extend", "Shade", "UpdateCounters1", "UpdateCounters2", "Connect", "Finalize"};
    cl_kernel k[9];
    for (int i = 0; i < 9; i++) {
        k[i] = clCreateKernel(g_wprog, names[i], &e);
        if (e != CL_SUCCESS) { snprintf(g_err, sizeof g_err, "wavefront: clCreateKernel(%s) %d", names[i], e); return -7; }
    }
    cl_uint cus = 0;
    clGetDeviceInfo(g_dev, CL_DEVICE_MAX_COMPUTE_UNITS, sizeof cus, &cus, nullptr);
    const size_t N = (size_t)width * height;
    auto mk = [&](size_t bytes, const void* host, cl_mem_flags fl) {
        cl_int ee;
        cl_mem m = clCreateBuffer(g_ctx, fl | (host ? CL_MEM_COPY_HOST_PTR : 0), bytes ? bytes : 16, (void*)host, &ee);
        if (ee != CL_SUCCESS) { snprintf(g_err, sizeof g_err, "wavefront: clCreateBuffer(%zu) %d", bytes, ee); return (cl_mem) nullptr; }
        return m;
    };
    cl_mem mNodes = mk(nodeBytes, nodes, CL_MEM_READ_ONLY), mTris = mk(triBytes, tris, CL_MEM_READ_ONLY), mVerts = mk(vertBytes, verts, CL_MEM_READ_ONLY);
    cl_mem mNoise = mk(128 * 128 * 8 * 4, noise, CL_MEM_READ_ONLY);
    cl_mem mIn = mk(N * 64, nullptr, CL_MEM_READ_WRITE), mOut = mk(N * 64, nullptr, CL_MEM_READ_WRITE);
    cl_mem mConn = mk(N * 3 * 48, nullptr, CL_MEM_READ_WRITE), mAcc = mk(N * 16, nullptr, CL_MEM_READ_WRITE);
    if (!mNodes || !mTris || !mVerts || !mNoise || !mIn || !mOut || !mConn || !mAcc) return -8;
    auto run1 = [&](cl_kernel kk, size_t global, size_t local) {
        const cl_int ee = clEnqueueNDRangeKernel(g_q, kk, 1, nullptr, &global, local ? &local : nullptr, 0, nullptr, nullptr);
        if (ee != CL_SUCCESS) snprintf(g_err, sizeof g_err, "wavefront: clEnqueueNDRangeKernel %d", ee);
        return ee;
    };
    auto run2 = [&](cl_kernel kk) {
        const size_t g[2] = {width, height};
        const cl_int ee = clEnqueueNDRangeKernel(g_q, kk, 2, nullptr, g, nullptr, 0, nullptr, nullptr);
        if (ee != CL_SUCCESS) snprintf(g_err, sizeof g_err, "wavefront: clEnqueueNDRangeKernel 2D %d", ee);
        return ee;
    };
    // Clear (tiny_bvh_gpu.cpp:133-137)
    clSetKernelArg(k[1], 0, sizeof(cl_mem), &mAcc);
    if (run1(k[1], N, 0)) return -9;
    cl_uint frameIdx = 1;
    const cl_int nPrimary = (cl_int)N;
    const cl_uint W = width, H = height;
    for (cl_uint spp = 1; spp <= frames; spp++, frameIdx++) {
        // SetRenderData( N, eye, p0, p1, p2, frameIdx, W, H, nodes, tris, noise )  (:139-140)
        clSetKernelArg(k[0], 0, sizeof(cl_int), &nPrimary);
        clSetKernelArg(k[0], 1, 16, eye); clSetKernelArg(k[0], 2, 16, p0); clSetKernelArg(k[0], 3, 16, p1); clSetKernelArg(k[0], 4, 16, p2);
        clSetKernelArg(k[0], 5, sizeof(cl_uint), &frameIdx); clSetKernelArg(k[0], 6, sizeof(cl_uint), &W); clSetKernelArg(k[0], 7, sizeof(cl_uint), &H);
        clSetKernelArg(k[0], 8, sizeof(cl_mem), &mNodes); clSetKernelArg(k[0], 9, sizeof(cl_mem), &mTris); clSetKernelArg(k[0], 10, sizeof(cl_mem), &mNoise);
        if (run1(k[0], 1, 0)) return -9;
        // Generate( raysOut, spp * 19191 )  (:141-142)
        const cl_uint seed = spp * 19191u;
        clSetKernelArg(k[2], 0, sizeof(cl_mem), &mOut); clSetKernelArg(k[2], 1, sizeof(cl_uint), &seed);
        if (run2(k[2])) return -9;
        for (uint32_t i = 0; i < iterations; i++) {   // (:143-152)
            std::swap(mIn, mOut);
            clSetKernelArg(k[3], 0, sizeof(cl_mem), &mIn);
            if (run1(k[3], (size_t)cus * 64 * 16, 64)) return -9;
            if (run1(k[5], 1, 0)) return -9;
            const cl_uint sampleIdx = spp - 1;
            clSetKernelArg(k[4], 0, sizeof(cl_mem), &mAcc); clSetKernelArg(k[4], 1, sizeof(cl_mem), &mIn); clSetKernelArg(k[4], 2, sizeof(cl_mem), &mOut);
            clSetKernelArg(k[4], 3, sizeof(cl_mem), &mConn); clSetKernelArg(k[4], 4, sizeof(cl_mem), &mVerts); clSetKernelArg(k[4], 5, sizeof(cl_uint), &sampleIdx);
            if (run1(k[4], (size_t)cus * 64 * 16, 64)) return -9;
            if (run1(k[6], 1, 0)) return -9;
        }
        // Connect( accumulator, connections )  (:153-154)
        clSetKernelArg(k[7], 0, sizeof(cl_mem), &mAcc); clSetKernelArg(k[7], 1, sizeof(cl_mem), &mConn);
        if (run1(k[7], (size_t)cus * 64 * 8, 64)) return -9;
    }
    e = clEnqueueReadBuffer(g_q, mAcc, CL_TRUE, 0, N * 16, out, 0, nullptr, nullptr);
    if (e != CL_SUCCESS) { snprintf(g_err, sizeof g_err, "wavefront: clEnqueueReadBuffer %d", e); return -10; }
    const float inv = 1.0f / (float)frames;
    for (size_t i = 0; i < N * 4; i++) out[i] *= inv;
    for (cl_mem m : {mNodes, mTris, mVerts, mNoise, mIn, mOut, mConn, mAcc}) clReleaseMemObject(m);
    for (int i = 0; i < 9; i++) clReleaseKernel(k[i]);
    return 0;
}

// ---- the reference's TLAS traversal (traverse_tlas.cl) through wavefront2.cl's Extend kernel, as tiny_bvh_gpu2.cpp:184-191 launches it -----
// The file as shipped does not compile on its own (traverse_tlas.cl's general path reads blasDesc / blasCWNodes, which only raytracer.cl
// declares); its DEPRECATED_TLAS_PATH branch — "instance 0 is the Bistro, all others the dragon", the branch tiny_bvh_gpu2.cpp was written
// against — does, and with the same BLAS passed for both it is a TLAS over instances of one BVH8_CWBVH BLAS.
// rays: n 64-byte ray records (O, D read; the kernel derives rD with native_recip as the demo does); out: n float4 hits
// (t, u, v, prim + (instance << 24)).  Extend runs `passes` + 1 times with the demo's launch shape (compute units * 64 * 16 work-items of 64);
// returns the mean kernel time in milliseconds or a negative error.
namespace { cl_program g_w2prog = nullptr; }
double refocl_tlas_extend(const void* tlasNodes, uint64_t tlasNodeBytes, const uint32_t* tlasIdx, uint64_t nIdx, const void* instances, uint64_t nInst,
                          const void* blasNodes, uint64_t blasNodeBytes, const void* blasTris, uint64_t blasTriBytes, const void* rays, uint64_t n, int passes,
                          float* out) {
    if (refocl_init()) return -1.0;
    cl_int e;
    if (!g_w2prog) {
        const std::string src = std::string("#define ISAMD\n#define DEPRECATED_TLAS_PATH\n") + kWavefront2Source;
        const char* s = src.c_str();
        size_t len = src.size();
        g_w2prog = clCreateProgramWithSource(g_ctx, 1, &s, &len, &e);
        if (e != CL_SUCCESS) { snprintf(g_err, sizeof g_err, "tlas: clCreateProgramWithSource %d", e); return -5.0; }
        e = clBuildProgram(g_w2prog, 0, nullptr, "-cl-std=CL2.0 -cl-strict-aliasing -cl-fast-relaxed-math -cl-single-precision-constant ", nullptr, nullptr);
        if (e != CL_SUCCESS) {
            size_t m = 0;
            clGetProgramBuildInfo(g_w2prog, g_dev, CL_PROGRAM_BUILD_LOG, sizeof g_err - 64, g_err + 32, &m);
            memcpy(g_err, "wavefront2 clBuildProgram failed", 32);
            g_w2prog = nullptr;
            return -6.0;
        }
    }
    cl_kernel kSet = clCreateKernel(g_w2prog, "SetRenderData", &e);
    if (e != CL_SUCCESS) { snprintf(g_err, sizeof g_err, "tlas: clCreateKernel(SetRenderData) %d", e); return -7.0; }
    cl_kernel kExt = clCreateKernel(g_w2prog, "Extend", &e);
    if (e != CL_SUCCESS) { snprintf(g_err, sizeof g_err, "tlas: clCreateKernel(Extend) %d", e); return -7.0; }
    auto buf = [&](const void* p, uint64_t bytes, cl_mem_flags f) {
        cl_int ee;
        cl_mem m = clCreateBuffer(g_ctx, f | CL_MEM_COPY_HOST_PTR, bytes ? bytes : 16, (void*)p, &ee);
        if (ee != CL_SUCCESS) snprintf(g_err, sizeof g_err, "tlas: clCreateBuffer(%llu bytes) %d", (unsigned long long)bytes, ee);
        return m;
    };
    // PathState { float4 T, O (w: pixel index << 8 | flags), D (w: t), hit } per ray (wavefront2.cl:53-59, Generate :106-119)
    std::vector<float> ps((size_t)n * 16);
    const float* r = (const float*)rays;
    for (uint64_t i = 0; i < n; i++) {
        float* p = ps.data() + i * 16;
        p[0] = p[1] = p[2] = p[3] = 1.f;
        p[4] = r[i * 16 + 0]; p[5] = r[i * 16 + 1]; p[6] = r[i * 16 + 2];
        const uint32_t w = ((uint32_t)i << 8) + 1u; memcpy(p + 7, &w, 4);
        p[8] = r[i * 16 + 4]; p[9] = r[i * 16 + 5]; p[10] = r[i * 16 + 6]; p[11] = 1e30f;
        p[12] = 1e30f; p[13] = p[14] = p[15] = 0.f;
    }
    const uint32_t dummy[4] = {0, 0, 0, 0};
    cl_mem mTlas = buf(tlasNodes, tlasNodeBytes, CL_MEM_READ_ONLY), mIdx = buf(tlasIdx, nIdx * 4, CL_MEM_READ_ONLY), mInst = buf(instances, nInst * 192, CL_MEM_READ_ONLY);
    cl_mem mNodes = buf(blasNodes, blasNodeBytes, CL_MEM_READ_ONLY), mTris = buf(blasTris, blasTriBytes, CL_MEM_READ_ONLY), mDummy = buf(dummy, 16, CL_MEM_READ_ONLY);
    cl_mem mRays = buf(ps.data(), n * 64, CL_MEM_READ_WRITE);
    if (!mTlas || !mIdx || !mInst || !mNodes || !mTris || !mDummy || !mRays) return -8.0;
    cl_uint cus = 0;
    clGetDeviceInfo(g_dev, CL_DEVICE_MAX_COMPUTE_UNITS, sizeof cus, &cus, nullptr);
    const cl_int count = (cl_int)n;
    const float z4[4] = {0, 0, 0, 0};
    const cl_uint one = 1, w = 1024, h = 1024;
    // SetRenderData( N, eye, p0, p1, p2, frameIdx, W, H, bistroNodes, bistroTris, bistroVerts, dragonNodes, dragonTris, dragonVerts, tlasNodes, tlasIdx, instances, blueNoise )
    clSetKernelArg(kSet, 0, sizeof count, &count);
    for (int i = 1; i <= 4; i++) clSetKernelArg(kSet, i, 16, z4);
    clSetKernelArg(kSet, 5, sizeof one, &one); clSetKernelArg(kSet, 6, sizeof w, &w); clSetKernelArg(kSet, 7, sizeof h, &h);
    cl_mem args[10] = {mNodes, mTris, mDummy, mNodes, mTris, mDummy, mTlas, mIdx, mInst, mDummy};
    for (int i = 0; i < 10; i++) clSetKernelArg(kSet, 8 + i, sizeof(cl_mem), &args[i]);
    clSetKernelArg(kExt, 0, sizeof(cl_mem), &mRays);
    const size_t gOne = 1, gExt = (size_t)cus * 64 * 16, local = 64;
    double total = 0;
    for (int p = 0; p <= passes; p++) {
        e = clEnqueueNDRangeKernel(g_q, kSet, 1, nullptr, &gOne, nullptr, 0, nullptr, nullptr);   // resets extendTasks to N
        if (e != CL_SUCCESS) { snprintf(g_err, sizeof g_err, "tlas: SetRenderData %d", e); return -9.0; }
        cl_event ev;
        e = clEnqueueNDRangeKernel(g_q, kExt, 1, nullptr, &gExt, &local, 0, nullptr, &ev);
        if (e != CL_SUCCESS) { snprintf(g_err, sizeof g_err, "tlas: Extend %d", e); return -9.0; }
        clWaitForEvents(1, &ev);
        cl_ulong t0 = 0, t1 = 0;
        clGetEventProfilingInfo(ev, CL_PROFILING_COMMAND_START, sizeof t0, &t0, nullptr);
        clGetEventProfilingInfo(ev, CL_PROFILING_COMMAND_END, sizeof t1, &t1, nullptr);
        if (p) total += (double)(t1 - t0) * 1e-6;
        clReleaseEvent(ev);
    }
    e = clEnqueueReadBuffer(g_q, mRays, CL_TRUE, 0, n * 64, ps.data(), 0, nullptr, nullptr);
    if (e != CL_SUCCESS) { snprintf(g_err, sizeof g_err, "tlas: clEnqueueReadBuffer %d", e); return -10.0; }
    for (uint64_t i = 0; i < n; i++) memcpy(out + i * 4, ps.data() + i * 16 + 12, 16);
    for (cl_mem m : {mTlas, mIdx, mInst, mNodes, mTris, mDummy, mRays}) clReleaseMemObject(m);
    clReleaseKernel(kSet); clReleaseKernel(kExt);
    return total / (passes > 0 ? passes : 1);
}

}  // extern "C"
