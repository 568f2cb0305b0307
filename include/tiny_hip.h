// tiny_hip.h — the binding a tinybvh maintainer adds next to tiny_ocl.h: a header-only C++ layer over the C ABI of the MI355X engine
// (tinybvh_amd.h; link with -ltinybvh_amd) for the GPU call sites of the reference:
//   tiny_bvh_speedtest.cpp:1092-1241   three blocks of { tinyocl::Buffer x 2-3, CopyToDevice, Kernel::SetArguments, Kernel::Run,
//                                      clWaitForEvents, clGetEventProfilingInfo, CopyFromDevice }      ->  tinyhip::Scene
//   tiny_bvh_minimal_gpu.cpp:50-93     the same for one layout                                         ->  tinyhip::Scene
//   tiny_bvh_gpu.cpp:128-158           the wavefront frame loop over wavefront.cl                      ->  tinyhip::PathTracer
//   tiny_ocl.h:362-364                 ONE process-global OpenCL device                                ->  tinyhip::Context(device), any number
// Include AFTER tiny_bvh.h (it uses tinybvh::BVH_GPU / BVH4_GPU / BVH8_CWBVH / Ray / bvhvec4 as they are).  Errors: the C ABI returns status
// codes and never exits; this layer mirrors tinyocl's FatalError for the demos — it prints tbvh_last_error() and exits — IN THE APPLICATION.
// Compiled and run by examples/speedtest_gpu_section.cpp and examples/wavefront_demos.cpp (tests/test_examples.py).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "tinybvh_amd.h"

namespace tinyhip {

inline void Check(int rc, const char* what) {
    if (rc) { fprintf(stderr, "tinyhip: %s -> %d: %s\n", what, rc, tbvh_last_error()); exit(1); }
}

// one shared tbvh_context per HIP device, created on first use (tinyocl::Kernel::InitCL creates its single context the same way, tiny_ocl.h:945-1139).
// Threads: every call of the C ABI takes its context's lock, so host threads may share this context and the Scenes on it the way the
// reference's callers share a const BVH (tiny_bvh_speedtest.cpp:1077-1083: Intersect from 8 threads) — their calls serialise.  A thread whose
// queries should OVERLAP with other threads' on the device takes a context of its own: NewContext() + the Scene constructors' last argument.
inline tbvh_context* Context(int device = 0) {
    static tbvh_context* ctx[64] = {};
    static std::mutex m;
    if (device < 0 || device >= 64) { fprintf(stderr, "tinyhip: device %d out of range\n", device); exit(1); }
    std::lock_guard<std::mutex> lk(m);
    if (!ctx[device]) Check(tbvh_init(device, &ctx[device]), "tbvh_init");
    return ctx[device];
}
// a context of the caller's own on `device` (its own stream, staging buffers and lock); release with tbvh_shutdown after its Scenes are gone
inline tbvh_context* NewContext(int device = 0) {
    tbvh_context* c = nullptr;
    Check(tbvh_init(device, &c), "tbvh_init");
    return c;
}
inline int DeviceCount() { const int n = tbvh_device_count(); return n < 0 ? 0 : n; }
// the CL_PROFILING_COMMAND_* equivalent costs two event records per query: a renderer that never calls LastKernelMs() switches it off
inline void SetTiming(bool enabled, tbvh_context* ctx = nullptr, int device = 0) { Check(tbvh_set_timing(ctx ? ctx : Context(device), enabled ? 1 : 0), "tbvh_set_timing"); }

// One uploaded layout: replaces the Buffer triple + Kernel of a speedtest GPU block.  The blobs are consumed verbatim.
class Scene {
public:
    // own: a context from NewContext() for a thread of its own; nullptr = the shared context of the device
    Scene(const tinybvh::BVH_GPU& b, const tinybvh::bvhvec4* verts, int device = 0, tbvh_context* own = nullptr) : dev(device), ctx(own ? own : Context(device)) {
        Check(tbvh_upload_bvh_gpu(ctx, b.bvhNode, b.usedNodes, b.bvh.primIdx, b.bvh.idxCount, verts, b.triCount, &s), "tbvh_upload_bvh_gpu");
    }
    explicit Scene(const tinybvh::BVH4_GPU& b, int device = 0, tbvh_context* own = nullptr) : dev(device), ctx(own ? own : Context(device)) {
        Check(tbvh_upload_bvh4_gpu(ctx, b.bvh4Data, b.usedBlocks, &s), "tbvh_upload_bvh4_gpu");
    }
    explicit Scene(const tinybvh::BVH8_CWBVH& b, int device = 0, tbvh_context* own = nullptr) : dev(device), ctx(own ? own : Context(device)) {
        Check(tbvh_upload_cwbvh(ctx, b.bvh8Data, b.usedBlocks, b.bvh8Tris, (uint64_t)b.bvh8.idxCount * 3, &s), "tbvh_upload_cwbvh");
    }
    // the reference's flow for animated geometry — bvh.Refit() on the host, X.ConvertFrom( bvh ) again (tiny_bvh.h:3055-3093) — without a new
    // Scene: the refitted blob goes into the same device memory, TLASes over this BLAS keep working (tbvh_update_*)
    void Update(const tinybvh::BVH_GPU& b, const tinybvh::bvhvec4* verts) { Check(tbvh_update_bvh_gpu(s, b.bvhNode, b.usedNodes, b.bvh.primIdx, b.bvh.idxCount, verts, b.triCount), "tbvh_update_bvh_gpu"); }
    void Update(const tinybvh::BVH4_GPU& b) { Check(tbvh_update_bvh4_gpu(s, b.bvh4Data, b.usedBlocks), "tbvh_update_bvh4_gpu"); }
    void Update(const tinybvh::BVH8_CWBVH& b) { Check(tbvh_update_cwbvh(s, b.bvh8Data, b.usedBlocks, b.bvh8Tris, (uint64_t)b.bvh8.idxCount * 3), "tbvh_update_cwbvh"); }
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;
    ~Scene() { tbvh_free_scene(s); }
    // batched equivalents of X::Intersect( Ray& ) / X::IsOccluded( const Ray& ): a host tinybvh::Ray[] traced in place (the first 64 of every
    // 128 bytes go to the device, hit.t / u / v / prim come back: the memcpy loop of tiny_bvh_speedtest.cpp:1110-1115 and CopyFromDevice in one call)
    void Intersect(tinybvh::Ray* rays, size_t n) { Check(tbvh_intersect(s, rays, n, sizeof(tinybvh::Ray)), "tbvh_intersect"); }
    void IsOccluded(const tinybvh::Ray* rays, size_t n, uint8_t* out) { Check(tbvh_occluded(s, rays, n, sizeof(tinybvh::Ray), out), "tbvh_occluded"); }
    // device time of the last launch in ms: the CL_PROFILING_COMMAND_START / END read of tiny_bvh_speedtest.cpp:1126-1131
    float LastKernelMs() const { return tbvh_time_last_ms(ctx); }
    // animated geometry: BVH::Refit + ConvertFrom + upload of the reference flow, on the device
    void Refit(const tinybvh::bvhvec4* verts, size_t triCount) { Check(tbvh_refit(s, verts, triCount, 0), "tbvh_refit"); }
    tbvh_scene* Handle() const { return s; }
    int Device() const { return dev; }
    tbvh_context* Ctx() const { return ctx; }
private:
    tbvh_scene* s = nullptr;
    int dev = 0;
    tbvh_context* ctx = nullptr;
};

// tinyocl::Buffer( bytes ) for a ray array that is traced many times (tiny_bvh_speedtest.cpp:1101-1108 wraps its ray array in one per GPU block): page-locked
// host memory of the library's for the object's lifetime (tbvh_pinned_malloc); a PACKED 64-byte ray array in it goes up by DMA straight from there.
class PinnedBuffer {
public:
    explicit PinnedBuffer(size_t bytes, tbvh_context* own = nullptr, int device = 0) : ctx(own ? own : Context(device)) { Check(tbvh_pinned_malloc(ctx, bytes, &p), "tbvh_pinned_malloc"); }
    PinnedBuffer(const PinnedBuffer&) = delete;
    PinnedBuffer& operator=(const PinnedBuffer&) = delete;
    ~PinnedBuffer() { tbvh_pinned_free(ctx, p); }
    void* GetHostPtr() const { return p; }      // (tinyocl::Buffer::GetHostPtr)
private:
    void* p = nullptr;
    tbvh_context* ctx;
};

// One host Ray[] over several devices (each holds a Scene of the same BVH): contiguous shards, results in place.
inline void IntersectSharded(const std::vector<Scene*>& replicas, tinybvh::Ray* rays, size_t n) {
    std::vector<tbvh_scene*> h;
    for (Scene* r : replicas) h.push_back(r->Handle());
    Check(tbvh_intersect_sharded(h.data(), (uint32_t)h.size(), rays, n, sizeof(tinybvh::Ray)), "tbvh_intersect_sharded");
}

// The frame loop of tiny_bvh_gpu.cpp:128-158 (SetRenderData, Generate, { Extend, Shade } x depth, Connect, Finalize) as one object: the image is
// cut into bands of rows, one per device, every band generated, traced, shaded and accumulated where it lives; Pixels() gathers the image.
// With one device this is tbvh_wavefront_render on the whole image.
class PathTracer {
public:
    // scenes[i]: the same BVH uploaded on device i; verts: the scene's vertex array (copied to every device)
    PathTracer(const std::vector<Scene*>& scenes, const tinybvh::bvhvec4* verts, size_t vertCount, uint32_t width, uint32_t height) : W(width), H(height) {
        const uint32_t n = (uint32_t)scenes.size();
        uint32_t row = 0;
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t next = (uint32_t)((uint64_t)(H / 4) * (i + 1) / n) * 4;   // bands of whole 4-row tiles
            tbvh_context* c = scenes[i]->Ctx();
            tbvh_wavefront* w = nullptr;
            Check(tbvh_wavefront_create(c, W, next - row, &w), "tbvh_wavefront_create");
            Check(tbvh_wavefront_set_band(w, row, H), "tbvh_wavefront_set_band");
            void* dv = nullptr;
            Check(tbvh_device_malloc(c, vertCount * sizeof(tinybvh::bvhvec4), &dv), "tbvh_device_malloc");
            Check(tbvh_copy_to_device(c, dv, verts, vertCount * sizeof(tinybvh::bvhvec4)), "tbvh_copy_to_device");
            wf.push_back(w); dverts.push_back(dv); sc.push_back(scenes[i]->Handle()); ctx.push_back(c);
            row = next;
        }
    }
    PathTracer(const PathTracer&) = delete;
    PathTracer& operator=(const PathTracer&) = delete;
    ~PathTracer() {
        for (size_t i = 0; i < wf.size(); i++) { tbvh_wavefront_destroy(wf[i]); tbvh_device_free(ctx[i], dverts[i]); }
    }
    // one frame (sample) of every band; accumulates unless params.clear
    void Render(const tbvh_camera& cam, const tbvh_wf_params& params, std::vector<tbvh_wf_stats>* stats = nullptr, std::vector<float>* dispatchMs = nullptr) {
        if (stats) stats->resize(wf.size());
        if (dispatchMs) dispatchMs->resize(wf.size());
        Check(tbvh_wavefront_render_sharded(wf.data(), sc.data(), dverts.data(), (uint32_t)wf.size(), &cam, &params, stats ? stats->data() : nullptr,
                                            dispatchMs ? dispatchMs->data() : nullptr), "tbvh_wavefront_render_sharded");
    }
    // the float RGBA accumulator of the whole image (W x H x 4)
    void Read(float* rgba) { Check(tbvh_wavefront_read_sharded(wf.data(), (uint32_t)wf.size(), rgba), "tbvh_wavefront_read_sharded"); }
    uint32_t Bands() const { return (uint32_t)wf.size(); }
private:
    uint32_t W, H;
    std::vector<tbvh_wavefront*> wf;
    std::vector<void*> dverts;
    std::vector<tbvh_scene*> sc;
    std::vector<tbvh_context*> ctx;
};

}  // namespace tinyhip
