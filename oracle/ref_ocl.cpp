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
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

static const char* kSource =
#include "_ref/ref_cl_source.inc"
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

// layout: 4 = batch_ailalaine(nodes, idx, verts, rays); 6 = batch_gpu4way(blocks, rays);
//         9 = batch_cwbvh(nodes, tris, rays).  bufN / bytesN are the layout's blobs in that order.
// rays: n packed 64-byte records (in/out).  Runs 1 warm-up + `passes` timed launches with global
// size n and local size 64 (tiny_bvh_speedtest.cpp:1122-1131); returns the mean kernel time in
// milliseconds from the profiling events, or a negative error.
double refocl_run(int layout, const void* buf0, uint64_t bytes0, const void* buf1, uint64_t bytes1, const void* buf2, uint64_t bytes2,
                  void* rays, uint64_t n, int passes) {
    if (refocl_init()) return -1.0;
    const char* name = layout == 4 ? "batch_ailalaine" : layout == 6 ? "batch_gpu4way" : layout == 9 ? "batch_cwbvh" : nullptr;
    if (!name) { snprintf(g_err, sizeof g_err, "bad layout"); return -2.0; }
    cl_int e;
    cl_kernel k = clCreateKernel(g_prog, name, &e);
    if (e != CL_SUCCESS) { snprintf(g_err, sizeof g_err, "clCreateKernel(%s) %d", name, e); return -3.0; }
    const void* bufs[3] = {buf0, buf1, buf2};
    const uint64_t bytes[3] = {bytes0, bytes1, bytes2};
    const int nb = layout == 4 ? 3 : layout == 6 ? 1 : 2;
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

}  // extern "C"
