// include/shim/tiny_ocl.h — the subset of tinyocl (reference tiny_ocl.h) that tinybvh's batch-traversal call sites use, over the C ABI of the
// MI355X engine (tinybvh_amd.h; link with -ltinybvh_amd).  Put this directory BEFORE the reference checkout on the include path and the
// reference's own GPU mains compile with ZERO edits and trace on the HIP engine instead of OpenCL:
//
//     g++ -std=c++20 -O2 -Iinclude/shim -Iinclude -I<tinybvh checkout> <tinybvh checkout>/tiny_bvh_minimal_gpu.cpp -Ltinybvh_amd -ltinybvh_amd
//
// What is mirrored (reference file:line -> here):
//   tinyocl::Buffer( bytes, hostPtr, flags )        tiny_ocl.h:130-154, 571-628   the caller's host pointer is kept, never owned; without one, GetHostPtr()
//                                                                                allocates a host mirror — here page-locked memory of the library's
//                                                                                (tbvh_pinned_malloc), so ray arrays go up by DMA straight from it
//   Buffer::CopyToDevice / CopyFromDevice           tiny_ocl.h:632-691            blocking copies of the whole buffer or (offset, size) in bytes
//   Buffer::Clear                                   tiny_ocl.h:700-708
//   tinyocl::Kernel( file, entryPoint )             tiny_ocl.h:712-921            the file name is ignored (nothing is compiled at run time); the entry
//                                                                                point selects the engine's kernel:
//        "batch_ailalaine"  ( altNode, idx, verts, rayData )   traverse_bvh2.cl:209-219   BVH_GPU       -> tbvh_upload_bvh_gpu  + tbvh_intersect_device
//        "batch_gpu4way"    ( alt4Node, rayData )              traverse_bvh4.cl:277-286   BVH4_GPU      -> tbvh_upload_bvh4_gpu + tbvh_intersect_device
//        "batch_cwbvh"      ( cwbvhNodes, cwbvhTris, rayData ) traverse_cwbvh.cl:554-570  BVH8_CWBVH    -> tbvh_upload_cwbvh    + tbvh_intersect_device
//                                                                                any other entry point: FatalError, like a kernel that fails to build
//   Kernel::SetArguments( Buffer*... )              tiny_ocl.h:176-347            the argument ORDER of the .cl entry point above
//   Kernel::Run( count, localSize, wait, &event )   tiny_ocl.h:1180-1205          `count` work items = rays (the reference rounds it up to the work-group
//                                                                                size and the kernels have no bounds check: callers size their ray
//                                                                                buffer accordingly; here `count` is clamped to the rays the buffer holds)
//   cl_event + clWaitForEvents + clGetEventProfilingInfo( CL_PROFILING_COMMAND_START / _END )   tiny_bvh_speedtest.cpp:1117-1131: the timing scheme
//                                                                                of the speedtest, served by the engine's HIP events (tbvh_time_last_ms)
// What is NOT here: textures / GL interop, Run2D, kernels other than the three batch entry points (the wavefront demos use tinyhip::PathTracer,
// include/tiny_hip.h).  Buffer sizes are 64-bit here (`unsigned int size` in the reference, tiny_ocl.h:152, caps a buffer at 4 GiB: 64 M rays wrap to 0);
// the constructor takes a size_t where the reference takes an `unsigned int` (every reference call site converts implicitly).
// Errors: like tinyocl (FatalError -> message + exit), in the APPLICATION; the C ABI underneath returns status codes and never exits.
// Scene data is uploaded when a Kernel first runs with its buffers and again after any of them was CopyToDevice()d since (the reference's
// "Sync all data to the GPU. Repeat if anything changes", tiny_bvh_minimal_gpu.cpp:75).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "tinybvh_amd.h"

// ---- the few OpenCL names the call sites touch (tiny_bvh_speedtest.cpp:1117-1131) --------------------------------------------------------------
#ifndef TINYOCL_SHIM_NO_CL_NAMES
struct tinyocl_shim_event { double ms = 0.0; };
typedef tinyocl_shim_event* cl_event;
typedef uint64_t cl_ulong;
typedef int32_t cl_int;
typedef uint32_t cl_uint;
typedef void* cl_mem;
#define CL_SUCCESS 0
#define CL_PROFILING_COMMAND_START 0x1282
#define CL_PROFILING_COMMAND_END 0x1283
inline cl_int clWaitForEvents(cl_uint, const cl_event*) { return CL_SUCCESS; }   // (Kernel::Run has waited already: see there)
inline cl_int clGetEventProfilingInfo(cl_event e, cl_uint what, size_t size, void* out, size_t* sizeRet) {
    if (!e || !out || size < sizeof(cl_ulong)) return -30 /* CL_INVALID_VALUE */;
    const cl_ulong ns = what == CL_PROFILING_COMMAND_END ? (cl_ulong)(e->ms * 1e6 + 0.5) : 0;   // START = 0: only differences are meaningful
    memcpy(out, &ns, sizeof(ns));
    if (sizeRet) *sizeRet = sizeof(ns);
    return CL_SUCCESS;
}
inline cl_int clReleaseEvent(cl_event) { return CL_SUCCESS; }
#endif

namespace tinyocl {

inline void FatalError(const char* fmt, const char* a = "", const char* b = "") {
    fprintf(stderr, "tinyocl (HIP shim): ");
    fprintf(stderr, fmt, a, b);
    fprintf(stderr, "\n");
    exit(1);
}
inline void Check(int rc, const char* what) {
    if (rc) FatalError("%s failed: %s", what, tbvh_last_error());
}

// the process-global device of tinyocl (tiny_ocl.h:362-364: one device, one context) = device TINYOCL_SHIM_DEVICE (default 0) of the engine
inline tbvh_context* Context() {
    static tbvh_context* ctx = nullptr;
    if (!ctx) {
        const char* e = getenv("TINYOCL_SHIM_DEVICE");
        Check(tbvh_init(e ? atoi(e) : 0, &ctx), "tbvh_init");
    }
    return ctx;
}

class Buffer {
public:
    enum { DEFAULT = 0, TEXTURE = 8, TARGET = 16, READONLY = 1, WRITEONLY = 2 };
    Buffer() {}
    Buffer(size_t N, void* ptr = 0, unsigned int t = DEFAULT) { init(N, ptr, t); }   // (the reference: `unsigned int N`, tiny_ocl.h:136)
    Buffer(const Buffer&) = delete;
    Buffer& operator=(const Buffer&) = delete;
    ~Buffer() {
        if (deviceBuffer) tbvh_device_free(Context(), deviceBuffer);
        if (ownData && hostBuffer) tbvh_pinned_free(Context(), hostBuffer);
    }
    cl_mem* GetDevicePtr() { return &deviceBuffer; }
    unsigned int* GetHostPtr() {
        if (!hostBuffer) {   // (tiny_ocl.h:618-628: a host mirror on first request) — page-locked, from the library
            void* p = nullptr;
            Check(tbvh_pinned_malloc(Context(), size64 ? size64 : 16, &p), "tbvh_pinned_malloc");
            memset(p, 0, size64);
            hostBuffer = (unsigned int*)p; ownData = true;
        }
        return hostBuffer;
    }
    void CopyToDevice(const bool blocking = true) { (void)blocking; copyTo(0, size64); }
    void CopyToDevice(const int offset, const int size, const bool blocking = true) { (void)blocking; copyTo((size_t)offset, (size_t)size); }
    void CopyFromDevice(const bool blocking = true) { (void)blocking; copyFrom(0, size64); }
    void CopyFromDevice(const int offset, const int size, const bool blocking = true) { (void)blocking; copyFrom((size_t)offset, (size_t)size); }
    void CopyTo(Buffer* other) {
        if (!other || other->size64 < size64) FatalError("Buffer::CopyTo: destination missing or smaller");
        GetHostPtr(); copyFrom(0, size64);
        memcpy(other->GetHostPtr(), hostBuffer, size64);
        other->copyTo(0, size64);
    }
    void Clear() {
        memset(GetHostPtr(), 0, size64);
        copyTo(0, size64);
    }
    // ---- data members, public like the reference's (tiny_ocl.h:149-153) ----
    unsigned int* hostBuffer = 0;
    cl_mem deviceBuffer = 0;
    unsigned int type = DEFAULT, size = 0 /* in bytes, saturated at 2^32 - 1: use size64 */, textureID = 0;
    bool ownData = false, aligned = false;
    size_t size64 = 0;
    uint64_t version = 0;      // bumped by every CopyToDevice: a Kernel re-uploads its scene when an argument's version moved
private:
    void init(size_t N, void* ptr, unsigned int t) {
        if (t & (TEXTURE | TARGET)) FatalError("Buffer: texture / render-target buffers are not provided by the HIP shim");
        type = t; size64 = N; size = N > 0xffffffffull ? 0xffffffffu : (unsigned int)N;
        hostBuffer = (unsigned int*)ptr; ownData = false;
    }
    void ensureDevice() {
        if (!deviceBuffer) Check(tbvh_device_malloc(Context(), size64 ? size64 : 16, &deviceBuffer), "tbvh_device_malloc");
    }
    void copyTo(size_t offset, size_t bytes) {
        if (offset + bytes > size64) FatalError("Buffer::CopyToDevice: range beyond the buffer");
        ensureDevice();
        if (bytes) Check(tbvh_copy_to_device(Context(), (char*)deviceBuffer + offset, (const char*)GetHostPtr() + offset, bytes), "tbvh_copy_to_device");
        version++;
    }
    void copyFrom(size_t offset, size_t bytes) {
        if (offset + bytes > size64) FatalError("Buffer::CopyFromDevice: range beyond the buffer");
        ensureDevice();
        if (bytes) Check(tbvh_copy_from_device(Context(), (char*)GetHostPtr() + offset, (const char*)deviceBuffer + offset, bytes), "tbvh_copy_from_device");
    }
    friend class Kernel;
};

class Kernel {
public:
    Kernel(const char* file, const char* entryPoint) : source(file ? file : ""), entry(entryPoint ? entryPoint : "") {
        if (entry == "batch_ailalaine") { kind = 5; nArgs = 4; }
        else if (entry == "batch_gpu4way") { kind = 8; nArgs = 2; }
        else if (entry == "batch_cwbvh") { kind = 10; nArgs = 3; }
        else FatalError("kernel '%s' (%s) is not provided by the HIP shim: batch_ailalaine, batch_gpu4way and batch_cwbvh are", entry.c_str(), source.c_str());
        Context();   // (tiny_ocl.h:726: the first Kernel brings the device up)
    }
    Kernel(const Kernel&) = delete;
    Kernel& operator=(const Kernel&) = delete;
    ~Kernel() { if (scene) tbvh_free_scene(scene); }
    void InitArgs() {}
    template <typename... A> void SetArguments(A... a) {
        if ((int)sizeof...(A) != nArgs) FatalError("kernel '%s': wrong number of arguments", entry.c_str());
        int i = 0;
        (SetArgument(i++, a), ...);
    }
    // `count` work items, one ray each (tiny_ocl.h:1180-1205).  The launch is enqueued and WAITED for here when an event is asked for (the speedtest's
    // next statement is clWaitForEvents anyway); without an event it stays asynchronous until the next CopyFromDevice, like the reference's.
    void Run(const size_t count, const size_t localSize = 0, cl_event* eventToWaitFor = 0, cl_event* eventToSet = 0) {
        (void)localSize; (void)eventToWaitFor;   // (one in-order stream: a kernel this one could wait for has been enqueued before it)
        for (int i = 0; i < nArgs; i++) if (!arg[i]) FatalError("kernel '%s': Run before SetArguments", entry.c_str());
        syncScene();
        Buffer* rays = arg[nArgs - 1];
        rays->ensureDevice();
        const size_t have = rays->size64 / 64;
        const size_t n = count < have ? count : have;
        Check(tbvh_intersect_device(scene, rays->deviceBuffer, n), "tbvh_intersect_device");
        if (eventToSet) {
            Check(tbvh_synchronize(Context()), "tbvh_synchronize");
            static tinyocl_shim_event pool[64];
            static unsigned next = 0;
            tinyocl_shim_event* e = &pool[next++ & 63u];
            e->ms = tbvh_time_last_ms(Context());
            *eventToSet = e;
        }
    }
    // beyond the reference's surface: rebinds the LAST argument (the ray buffer of every batch_* entry point) and leaves the scene arguments as they are
    void SetRayBuffer(Buffer* rays) { if (!rays || nArgs == 0) FatalError("kernel '%s': SetRayBuffer needs a buffer", entry.c_str()); arg[nArgs - 1] = rays; }
    // diagnostics beyond the reference's surface
    tbvh_scene* Scene() { syncScene(); return scene; }
private:
    void SetArgument(int idx, Buffer* b) {
        if (idx < 0 || idx >= nArgs || !b) FatalError("kernel '%s': bad argument", entry.c_str());
        arg[idx] = b;
    }
    template <class T> void SetArgument(int, T) { FatalError("kernel '%s': the batch kernels take tinyocl::Buffer* arguments only", entry.c_str()); }
    void syncScene() {
        bool fresh = scene != nullptr;
        for (int i = 0; i < nArgs - 1; i++) fresh = fresh && arg[i] == seenBuf[i] && arg[i]->version == seenVer[i];
        if (fresh) return;
        for (int i = 0; i < nArgs - 1; i++) if (!arg[i]->hostBuffer) FatalError("kernel '%s': a BVH buffer has no host data", entry.c_str());
        tbvh_context* c = Context();
        if (scene) { tbvh_free_scene(scene); scene = nullptr; }
        if (kind == 5)        // ( altNode 64 B each, idx 4 B each, verts 3 x 16 B per triangle )
            Check(tbvh_upload_bvh_gpu(c, arg[0]->hostBuffer, arg[0]->size64 / 64, arg[1]->hostBuffer, arg[1]->size64 / 4, arg[2]->hostBuffer, arg[2]->size64 / 48, &scene), "tbvh_upload_bvh_gpu");
        else if (kind == 8)   // ( bvh4Data: 16-byte blocks )
            Check(tbvh_upload_bvh4_gpu(c, arg[0]->hostBuffer, arg[0]->size64 / 16, &scene), "tbvh_upload_bvh4_gpu");
        else                  // ( bvh8Data, bvh8Tris: 16-byte blocks )
            Check(tbvh_upload_cwbvh(c, arg[0]->hostBuffer, arg[0]->size64 / 16, arg[1]->hostBuffer, arg[1]->size64 / 16, &scene), "tbvh_upload_cwbvh");
        for (int i = 0; i < nArgs - 1; i++) { seenBuf[i] = arg[i]; seenVer[i] = arg[i]->version; }
    }
    std::string source, entry;
    int kind = 0, nArgs = 0;
    Buffer* arg[4] = {nullptr, nullptr, nullptr, nullptr};
    Buffer* seenBuf[3] = {nullptr, nullptr, nullptr};
    uint64_t seenVer[3] = {0, 0, 0};
    tbvh_scene* scene = nullptr;
};

}  // namespace tinyocl
