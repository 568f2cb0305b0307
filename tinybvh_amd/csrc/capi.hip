// capi.hip — implementation of include/tinybvh_amd.h: contexts, uploads, query launches,
// HIP-event timing, host builder entry points.  No torch, no exit(): every failure is a
// status code + tbvh_last_error().
#include "../../include/tinybvh_amd.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "host_builder.h"
#include "kernels.h"
#include "ray_pool.h"

using namespace tbvh;

static_assert(kLayoutBvhGpu == TBVH_LAYOUT_BVH_GPU && kLayoutBvh4Gpu == TBVH_LAYOUT_BVH4_GPU && kLayoutCwbvh == TBVH_LAYOUT_CWBVH,
              "kernels.h and the public header agree on the layout codes");

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) return fail(TBVH_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

}  // namespace

struct tbvh_context {
    int device = 0;
    hipStream_t ownStream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    int numCUs = 0;
    uint32_t blocks = 0;          // persistent grid size (64-thread workgroups)
    uint32_t* spill = nullptr;    // stack spill area
    uint32_t spillEntries = 0;    // 32-bit entries per lane
    unsigned long long* counter = nullptr;  // status word, instrumentation counters
    uint32_t poolParts = 5;   // log2: 32 partitions
    bool incoherentCopies = true;   // TBVH_INCOHERENT_COPIES=0: no hybrid node copy / 64-byte triangle records (prepareIncoherentCopies)
    uint32_t expFlags = 0;     // tbvh_debug_set_flags: QueryArgs::flags of the next launches (experiments)
    bool lastProbed = false;   // the most recent query launch ran the coherence probe (tbvh_debug_last_probe)
    bool gridOverride = false;     // TBVH_BLOCKS_PER_CU / TBVH_RAYS_PER_BLOCK given: no per-scene adjustment
    uint64_t splitBelow = 12ull << 20;   // batches of fewer rays split their last rays over idle lanes; TBVH_SPLIT_RAYS=0 turns that off (tie order then reproducible run to run)
    uint32_t raysPerBlock = 128;   // small batches: one workgroup per this many rays (with split rays, profiles/r02_grid_sweep.txt: 96-128 best on 1 M-ray batches, +5 % over 192; flat at 4 M)
    unsigned long long* pool = nullptr;     // ray-fetch counters: kPoolParts of them, 256 bytes apart (ray_pool.h), + the coherence probe's line; TWO such areas
    int poolCur = 0;              // the area the next launch draws from; its kernels zero the other one for the launch after (no memset in the stream)
    bool poolClean = false;       // both areas are known to be as that scheme leaves them (false: the next launch clears them itself)
    uint32_t* status = nullptr;
    RayRec* stageRays = nullptr;  // staging for host-array queries
    uint64_t stageCap = 0;
    uint8_t* stageOcc = nullptr;
    uint64_t stageOccCap = 0;
    // host-array queries of more than a few 10 k rays go through pinned staging in chunks: worker threads gather the
    // 64-byte prefixes of the caller's records into a pinned buffer while the previous chunk is in flight (a pageable
    // hipMemcpy2D moves ~9 GB/s because one CPU thread does the staging copy), and the 20 result bytes per ray come back
    // packed (k_pack_hits) and are scattered by the same workers
    struct HostPipe* pipe = nullptr;
    void* binScratch = nullptr;   // tbvh_bin_rays_device
    size_t binScratchBytes = 0;
    std::vector<tbvh_scene*> scenes;
};

struct tbvh_scene {
    tbvh_context* ctx = nullptr;
    int layout = 0;
    int variant = 0;
    float4* nodes = nullptr;   // BVH_GPU nodes / BVH4 stream / CWBVH nodes
    float4* tris = nullptr;    // BVH_GPU gathered tris / CWBVH tris
    float4* nodes128 = nullptr; // CWBVH: the same nodes padded to one 128-byte line each (padCwbvhIfLarge: node arrays beyond the Infinity Cache)
    float4* nodesHy = nullptr;  // CWBVH: the same nodes in surface-area priority order, the first hybridK packed, the others one per line (cwbvh_node.h: kNodeHybrid)
    uint32_t* hyPerm = nullptr; // device: position of node i in nodesHy
    float4* tris64 = nullptr;   // CWBVH (experiment flag 2): triangle records padded to 64 bytes
    uint32_t hybridK = 0;
    uint32_t nNodes = 0;
    uint64_t nNodeBlocks = 0, nTriBlocks = 0;
    uint64_t bytes = 0;
    // TLAS (layout = BVH_GPU nodes in `nodes`)
    bool isTlas = false;
    uint32_t* tlasIdx = nullptr;
    float4* instances = nullptr;
    BlasDesc* blasDesc = nullptr;
    int blasLayout = 0;
    bool blasMixCw2 = false;          // blasLayout == 0 and every BLAS is BVH8_CWBVH or BVH_GPU: the reference's two BLAS types (traverse_tlas.cl:50-72)
    uint64_t capNodes = 0, capIdx = 0, capInst = 0;
    uint64_t nInst = 0, nBlas = 0, nTlasNodes = 0, nTlasIdx = 0;
    // the same TLAS collapsed 4-wide in the BVH4_GPU node format (kernels_tlas4.hip), kept current by every upload / update / device rebuild;
    // only for TLASes whose BLASes are all BVH4_GPU
    float4* tlas4 = nullptr;
    uint64_t tlas4Cap = 0;            // blocks
    void* tlas4Scratch = nullptr;
    size_t tlas4ScratchBytes = 0;
    // ... or 8-wide in the BVH8_CWBVH node format (kernels_tlas8.hip) for TLASes whose BLASes are all BVH8_CWBVH; same scratch
    float4* tlas8 = nullptr;
    uint32_t* tlas8Refs = nullptr;
    uint64_t tlas8Cap = 0;            // nodes; instance references: the same number
    // device-side TLAS rebuild (kernels_tlasbuild.hip)
    float* blasBounds = nullptr;      // 6 floats per BLAS
    float* xformStage = nullptr;      // staged transforms (16 floats per instance) when the caller passes host memory
    uint64_t xformStageCap = 0;       // instances the staging buffer holds
    // BLAS <-> TLAS references: a TLAS snapshots its BLASes' device pointers (BlasDesc), so a BLAS knows the TLASes that
    // use it (their descriptors are refreshed when its opacity maps change) and outlives them (tbvh_free_scene on a BLAS
    // that is still referenced only marks it; the memory goes when the last TLAS over it is freed)
    std::vector<tbvh_scene*> blasList;   // TLAS: its BLASes, in blasIdx order
    std::vector<tbvh_scene*> usedBy;     // BLAS: the TLASes built over it (one entry per reference)
    bool zombie = false;                 // BLAS: freed by the caller while still referenced
    void* buildScratch = nullptr;
    size_t buildScratchBytes = 0, sortTempBytes = 0;
    uint64_t buildScratchFor = 0;     // instance count the scratch was sized for
    // device-side BLAS refit (kernels_refit.hip)
    void* refitScratch = nullptr;
    std::vector<uint32_t> b4Levels;   // BVH4_GPU: first node of every tree level in the item list (filled by the first refit)
    float4* vertStage = nullptr;      // staged vertices when the caller passes host memory
    // opacity micromaps (BVHBase::SetOpacityMicroMaps)
    uint32_t* opmap = nullptr;
    uint32_t opmapN = 0;
    uint64_t opmapBytes = 0;
    uint64_t vertStageTris = 0;
};

struct BLASInstanceCheck { float m[32]; float mn[3]; uint32_t blasIdx; float mx[3]; uint32_t mask; uint32_t pad[8]; };
static_assert(sizeof(BLASInstanceCheck) == 192, "BLASInstance is 192 bytes");

struct tbvh_hostbvh {
    int layout = 0;
    BVH2 bvh2;
    std::vector<NodeAL> al;
    std::vector<Vec4> blocksA, blocksB;
};

namespace {

int setDevice(tbvh_context* c) {
    HIP_TRY(hipSetDevice(c->device));
    return 0;
}

int ensureStage(tbvh_context* c, uint64_t n) {
    if (c->stageCap >= n) return 0;
    if (c->stageRays) hipFree(c->stageRays);
    c->stageRays = nullptr; c->stageCap = 0;
    HIP_TRY(hipMalloc((void**)&c->stageRays, n * sizeof(RayRec)));
    c->stageCap = n;
    return 0;
}
int ensureStageOcc(tbvh_context* c, uint64_t n) {
    if (c->stageOccCap >= n) return 0;
    if (c->stageOcc) hipFree(c->stageOcc);
    c->stageOcc = nullptr; c->stageOccCap = 0;
    HIP_TRY(hipMalloc((void**)&c->stageOcc, n));
    c->stageOccCap = n;
    return 0;
}

// ---- pipelined host <-> device staging -------------------------------------------------------------------------
}  // namespace
struct HostPipe {
    static constexpr uint64_t kChunk = 1ull << 18;   // rays per chunk: 16 MB up, 5 MB down
    void* pinUp[2] = {nullptr, nullptr};
    void* pinDown[2] = {nullptr, nullptr};
    hipEvent_t evUp[2] = {nullptr, nullptr}, evDown[2] = {nullptr, nullptr};
    uint32_t* packed = nullptr;   // device: 5 dwords per ray (bytes 44..63 of the record)
    uint64_t packedCap = 0;
    // a small persistent worker pool: parallel_for(n, fn) runs fn(part, parts) on every worker and the caller
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cvWork, cvDone;
    std::function<void(uint32_t, uint32_t)> job;
    uint64_t generation = 0;
    uint32_t pending = 0;
    bool quit = false;
    void start(uint32_t nThreads) {
        for (uint32_t t = 0; t < nThreads; t++)
            workers.emplace_back([this, t, nThreads] {
                uint64_t seen = 0;
                for (;;) {
                    std::function<void(uint32_t, uint32_t)> f;
                    {
                        std::unique_lock<std::mutex> lk(m);
                        cvWork.wait(lk, [&] { return quit || generation != seen; });
                        if (quit) return;
                        seen = generation; f = job;
                    }
                    f(t + 1, nThreads + 1);
                    {
                        std::lock_guard<std::mutex> lk(m);
                        if (--pending == 0) cvDone.notify_all();
                    }
                }
            });
    }
    void parallel_for(const std::function<void(uint32_t, uint32_t)>& f) {
        {
            std::lock_guard<std::mutex> lk(m);
            job = f; pending = (uint32_t)workers.size(); generation++;
        }
        cvWork.notify_all();
        f(0, (uint32_t)workers.size() + 1);
        std::unique_lock<std::mutex> lk(m);
        cvDone.wait(lk, [&] { return pending == 0; });
    }
    ~HostPipe() {
        { std::lock_guard<std::mutex> lk(m); quit = true; }
        cvWork.notify_all();
        for (auto& w : workers) w.join();
        for (int i = 0; i < 2; i++) {
            if (pinUp[i]) hipHostFree(pinUp[i]);
            if (pinDown[i]) hipHostFree(pinDown[i]);
            if (evUp[i]) hipEventDestroy(evUp[i]);
            if (evDown[i]) hipEventDestroy(evDown[i]);
        }
        if (packed) hipFree(packed);
    }
};
namespace {

int ensurePipe(tbvh_context* c, uint64_t n) {
    if (!c->pipe) {
        HostPipe* p = new (std::nothrow) HostPipe;
        if (!p) return fail(TBVH_E_NOMEM, "out of host memory");
        c->pipe = p;
        for (int i = 0; i < 2; i++) {
            HIP_TRY(hipHostMalloc(&p->pinUp[i], HostPipe::kChunk * 64, hipHostMallocDefault));
            HIP_TRY(hipHostMalloc(&p->pinDown[i], HostPipe::kChunk * 20, hipHostMallocDefault));
            HIP_TRY(hipEventCreateWithFlags(&p->evUp[i], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&p->evDown[i], hipEventDisableTiming));
        }
        uint32_t hw = usable_host_threads();
        uint32_t t = hw >= 32 ? 7 : hw >= 8 ? 3 : hw >= 4 ? 1 : 0;   // + the calling thread
        if (const char* e = getenv("TBVH_HOST_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 64) t = (uint32_t)v - 1; }
        p->start(t);
    }
    HostPipe* p = c->pipe;
    if (p->packedCap < n) {
        if (p->packed) hipFree(p->packed);
        p->packed = nullptr; p->packedCap = 0;
        HIP_TRY(hipMalloc((void**)&p->packed, n * 20));
        p->packedCap = n;
    }
    return 0;
}

// caller records (stride bytes apart) -> device array of 64-byte records
int pipeUpload(tbvh_context* c, const char* rays, uint64_t n, uint32_t stride, RayRec* dst) {
    HostPipe* p = c->pipe;
    for (uint64_t first = 0, k = 0; first < n; first += HostPipe::kChunk, k++) {
        const uint64_t cnt = n - first < HostPipe::kChunk ? n - first : HostPipe::kChunk;
        const int b = (int)(k & 1);
        if (k >= 2) HIP_TRY(hipEventSynchronize(p->evUp[b]));   // the DMA that last read this buffer is done
        char* pin = (char*)p->pinUp[b];
        const char* src = rays + first * stride;
        p->parallel_for([=](uint32_t part, uint32_t parts) {
            const uint64_t lo = cnt * part / parts, hi = cnt * (part + 1) / parts;
            if (stride == 64) std::memcpy(pin + lo * 64, src + lo * 64, (hi - lo) * 64);
            else for (uint64_t i = lo; i < hi; i++) std::memcpy(pin + i * 64, src + i * stride, 64);
        });
        HIP_TRY(hipMemcpyAsync(dst + first, pin, cnt * 64, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipEventRecord(p->evUp[b], c->stream));
    }
    return 0;
}

// device records -> bytes 44..63 of the caller's records
int pipeDownloadHits(tbvh_context* c, char* rays, uint64_t n, uint32_t stride, const RayRec* src) {
    HostPipe* p = c->pipe;
    launch_pack_hits(src, p->packed, n, c->stream);
    HIP_TRY(hipGetLastError());
    const uint64_t chunks = (n + HostPipe::kChunk - 1) / HostPipe::kChunk;
    auto issue = [&](uint64_t k) -> int {
        const uint64_t first = k * HostPipe::kChunk, cnt = n - first < HostPipe::kChunk ? n - first : HostPipe::kChunk;
        HIP_TRY(hipMemcpyAsync(p->pinDown[k & 1], (const char*)p->packed + first * 20, cnt * 20, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipEventRecord(p->evDown[k & 1], c->stream));
        return 0;
    };
    if (int r = issue(0)) return r;
    for (uint64_t k = 0; k < chunks; k++) {
        if (k + 1 < chunks) if (int r = issue(k + 1)) return r;   // next chunk in flight while this one is scattered
        HIP_TRY(hipEventSynchronize(p->evDown[k & 1]));
        const uint64_t first = k * HostPipe::kChunk, cnt = n - first < HostPipe::kChunk ? n - first : HostPipe::kChunk;
        const char* pin = (const char*)p->pinDown[k & 1];
        char* dstRays = rays + first * stride;
        p->parallel_for([=](uint32_t part, uint32_t parts) {
            const uint64_t lo = cnt * part / parts, hi = cnt * (part + 1) / parts;
            for (uint64_t i = lo; i < hi; i++) std::memcpy(dstRays + i * stride + 44, pin + i * 20, 20);
        });
    }
    return 0;
}

constexpr uint64_t kPipeMinRays = 1ull << 15;

int launchQuery(tbvh_scene* s, RayRec* d_rays, uint64_t n, uint8_t* d_occ, bool fresh = false, float freshTmax = 1e30f,
                const unsigned long long* nDev = nullptr) {
    tbvh_context* c = s->ctx;
    if (int r = setDevice(c)) return r;
    if (n == 0) return 0;
    const bool any = d_occ != nullptr;
    // ray-fetch counters: two areas alternate; the kernels of this launch zero the other area for the next one.  After anything that went wrong
    // between two launches (poolClean still false) both are cleared here.
    const size_t poolWords = (size_t)(kPoolParts + 1) * kPoolCounterStride;   // + the coherence-probe counters on their own line
    if (!c->poolClean) HIP_TRY(hipMemsetAsync(c->pool, 0, poolWords * 4 * 2, c->stream));
    c->poolClean = false;
    uint32_t* const poolArea = (uint32_t*)c->pool + (size_t)c->poolCur * poolWords;
    QueryArgs q;
    q.rays = d_rays; q.nRays = n; q.occluded = d_occ;
    q.spill = c->spill; q.counter = poolArea; q.counterNext = (uint32_t*)c->pool + (size_t)(c->poolCur ^ 1) * poolWords; q.poolParts = c->poolParts;
    q.stats = c->counter + 8;
    q.fresh = fresh ? 1u : 0u; q.freshTmax = freshTmax; q.nRaysDev = nDev; q.omm = Omm{s->opmap, s->opmapN};
    q.probe = nullptr; q.baseBlocks = 0; q.hybridK = s->hybridK; q.flags = c->expFlags & 1u;
    c->lastProbed = false;
    q.splitBelow = c->splitBelow;
    // persistent grid: 24 one-wave workgroups per CU for large batches; small batches get fewer
    // (about one workgroup per 128 rays, measured best for 1 M-ray launches) so every wave still
    // has a few ray replacements' worth of work
    // A scene that (nearly) lives in the L2s — the Sponza class: < 48 MB of nodes and triangles against 8 x 4 MB of L2
    // plus the Infinity Cache — is latency-bound, not cache-bound: it runs best with a third more waves (32 per CU) of
    // fewer rays each (measured +1..20 % on the Sponza stand-in from 0.26 M to 16.7 M rays; the same shape costs the
    // 196 MB Bistro stand-in 5-10 % on bounce and shadow rays, which thrash the caches more with more waves).
    const uint64_t blobBytes = (!s->isTlas && s->layout == TBVH_LAYOUT_CWBVH) ? (s->nNodeBlocks + s->nTriBlocks) * 16 : s->bytes;   // (without the library's own re-laid-out copies)
    const bool small = !s->isTlas && !c->gridOverride && blobBytes < (48ull << 20);
    const uint32_t perBlock = c->raysPerBlock;
    const uint32_t cap = small ? c->blocks + c->blocks / 3u : c->blocks;
    uint64_t want = (n + perBlock - 1) / perBlock;
    const uint32_t lo = (uint32_t)c->numCUs * 4u;
    uint32_t blocks = (uint32_t)(want < lo ? lo : (want > cap ? cap : want));
    HIP_TRY(hipEventRecord(c->ev0, c->stream));   // the probe below is part of the query's time
    // BVH8_CWBVH scenes beyond the L2s (the `small` class runs dense triangle phases, where the gated schedule loses 15 %) but within reach
    // of the Infinity Cache (beyond it camera rays are bound by memory too: 30 M / 60 M triangles lose 11 / 19 % under the gate), batches of 2 M
    // rays and more (the probe costs ~10 us, 3-4 % of a 1 M-ray launch): a 16-workgroup probe of the batch's coherence (4096 neighbour pairs)
    // lets the traversal kernel pick its schedule for the launch; a coherent batch also gets a third more waves (the surplus leaves at once
    // otherwise).  Bistro stand-in, 16.7 M rays: camera rays +4.5 %, shadow rays +6 %, bounce rays unchanged.
    uint32_t blocksBase = blocks;
    if (!s->isTlas && !small && blobBytes <= (384ull << 20) && s->layout == TBVH_LAYOUT_CWBVH && n >= (1ull << 21) && (s->variant == 0 || s->variant == 88)) {
        uint32_t* probe = poolArea + (size_t)kPoolParts * kPoolCounterStride;
        launch_coherence_probe(d_rays, n, nDev, probe, c->stream);
        HIP_TRY(hipGetLastError());
        q.probe = probe; q.baseBlocks = blocks;
        c->lastProbed = true;
        if (!c->gridOverride && blocks == c->blocks) blocks = c->blocks + c->blocks / 3u;   // 24 -> 32 one-wave workgroups per CU
    }
    if (s->isTlas) {
        const uint32_t blocks7 = (!c->gridOverride && blocks == c->blocks) ? (uint32_t)c->numCUs * 28u : blocks;   // the full grid of the kernels built for 7 waves per SIMD
        if (s->tlas4 && s->blasLayout == TBVH_LAYOUT_BVH4_GPU) {   // BVH4_GPU BLASes: the unified 4-wide kernel
            q.spillStride = c->spillEntries;   // 32-bit stack entries
            launch_tlas4(any, 0, s->tlas4, s->instances, s->blasDesc, q, c->status, blocks, c->stream, blocks7);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(c->ev1, c->stream));
            c->timed = true; c->poolCur ^= 1; c->poolClean = true;   // (the other counter area has been zeroed by what was just enqueued)
            return 0;
        }
        if (s->tlas8 && (s->blasLayout == TBVH_LAYOUT_CWBVH || s->blasMixCw2)) {   // BVH8_CWBVH BLASes (or those and BVH_GPU ones): the unified 8-wide kernel
            q.spillStride = c->spillEntries / 2;   // 8-byte stack entries
            launch_tlas8(any, 0, s->tlas8, s->tlas8Refs, s->instances, s->blasDesc, q, c->status, blocks, c->stream, blocks7, s->blasMixCw2);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(c->ev1, c->stream));
            c->timed = true; c->poolCur ^= 1; c->poolClean = true;   // (the other counter area has been zeroed by what was just enqueued)
            return 0;
        }
        if (s->blasLayout == TBVH_LAYOUT_BVH_GPU) {   // BVH_GPU BLASes: the TLAS already has their node format (kernels_tlas2.hip)
            q.spillStride = c->spillEntries;   // 32-bit stack entries
            launch_tlas2(any, 0, s->nodes, s->tlasIdx, s->instances, s->blasDesc, q, c->status, blocks, c->stream, blocks7);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(c->ev1, c->stream));
            c->timed = true; c->poolCur ^= 1; c->poolClean = true;   // (the other counter area has been zeroed by what was just enqueued)
            return 0;
        }
        q.spillStride = c->spillEntries / 2;
        launch_tlas(any, s->blasLayout, s->nodes, s->tlasIdx, s->instances, s->blasDesc, q, c->status, blocks, c->stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->ev1, c->stream));
        c->timed = true; c->poolCur ^= 1; c->poolClean = true;   // (the other counter area has been zeroed by what was just enqueued)
        return 0;
    }
    switch (s->layout) {
    case TBVH_LAYOUT_BVH_GPU:
        q.spillStride = c->spillEntries;
        launch_bvh2(any, s->variant, s->nodes, s->tris, q, c->status, blocks, c->stream);
        break;
    case TBVH_LAYOUT_BVH4_GPU:
        q.spillStride = c->spillEntries;
        launch_bvh4(any, s->variant, s->nodes, q, c->status, blocks, c->stream);
        break;
    case TBVH_LAYOUT_CWBVH:
        q.spillStride = c->spillEntries / 2;  // 8-byte entries
        {
            const bool autoPad = s->variant == 0 && s->nodes128 != nullptr;   // nodes beyond the Infinity Cache: the padded copy (padCwbvhIfLarge)
            const uint32_t blocks7 = c->gridOverride ? 0xFFFFFFFFu : (uint32_t)c->numCUs * 28u;
            const float4* tris = s->tris;
            if ((c->expFlags & 2u) && s->tris64) { tris = s->tris64; q.flags |= 2u; }   // experiment: 64-byte triangle records in the ordinary kernels too
            // A probed launch on a scene with the incoherent-batch copies (prepareIncoherentCopies) is TWO kernels back to back, each for one
            // verdict of the probe; the one the verdict is not for leaves at once (~10 us).  The coherent flavor keeps the packed arrays as
            // uploaded (its working set lives in the L2s); the incoherent one walks the hybrid node copy and the 64-byte triangle records.
            // Bistro stand-in, 16.7 M rays, interleaved medians (profiles/r03_ab_16m.txt): bounce rays +10 %, camera and shadow rays unchanged.
            const bool twoFlavors = q.probe && s->variant == 0 && !autoPad && s->nodesHy && s->tris64 && !(c->expFlags & 4u);
            if (s->variant == 90 && s->nodesHy && s->tris64)   // diagnostic: the incoherent flavor whatever the batch (tests, tools/ab_configs.py)
                launch_cwbvh(any, 0, s->nodesHy, s->tris64, q, c->status, blocksBase, c->stream, 13, small, blocks7);
            else if (twoFlavors) {
                QueryArgs qa = q;
                qa.baseBlocks = 0;   // every wave of the coherent flavor leaves unless the batch is coherent
                launch_cwbvh(any, 0, s->nodes, tris, qa, c->status, blocks, c->stream, 5, small, blocks7);
                HIP_TRY(hipGetLastError());
                QueryArgs qb = q;
                // the incoherent flavor on 28 one-wave workgroups per CU when the batch fills the grid (24 is the persistent grid's size: 20 / 26 / 28 / 30 /
                // 32 per CU trace bounce rays at -4.4 / +0.6 / +0.9 / +0.5 / +0.5 %, interleaved medians of 13 rounds)
                const uint32_t wX = (c->expFlags >> 8) & 0xffu;   // experiment: another number of waves per CU
                const uint32_t wB = wX ? wX : (c->gridOverride ? 0u : 28u);
                launch_cwbvh(any, 0, s->nodesHy, s->tris64, qb, c->status, (wB && blocksBase == c->blocks) ? (uint32_t)c->numCUs * wB : blocksBase, c->stream, 13, small, blocks7);
            } else
                launch_cwbvh(any, s->variant, autoPad ? s->nodes128 : s->nodes, tris, q, c->status, blocks, c->stream, autoPad ? 8 : 5, small, blocks7);
        }
        break;
    default:
        return fail(TBVH_E_INVALID, "scene layout %d has no query kernel", s->layout);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    c->timed = true; c->poolCur ^= 1; c->poolClean = true;   // (the other counter area has been zeroed by what was just enqueued)
    return 0;
}

int checkStatus(tbvh_context* c) {
    uint32_t st = 0;
    HIP_TRY(hipMemcpyAsync(&st, c->status, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (st & 1u) {
        hipMemsetAsync(c->status, 0, 4, c->stream);
        return fail(TBVH_E_FORMAT, "traversal stack overflow (tree deeper than the spill area allows)");
    }
    if (st & 2u) {
        hipMemsetAsync(c->status, 0, 4, c->stream);
        return fail(TBVH_E_FORMAT, "refit: a triangle record refers to a primitive beyond the vertex array");
    }
    if (st & 4u) {
        hipMemsetAsync(c->status, 0, 4, c->stream);
        return fail(TBVH_E_FORMAT, "wide TLAS build: the node capacity did not hold the collapsed tree");
    }
    return 0;
}

// A BVH8_CWBVH scene whose node array is larger than twice the 256 MB Infinity Cache is traversed through a copy with one node per
// 128-byte line: an 80-byte node straddles 1.6 lines on average, and once the lines come from HBM that is 17 % more traffic than the
// 60 % larger array costs (tools/size_sweep.py, 60 M triangles: bounce rays +6 %; below that size the smaller footprint wins).
int padCwbvhIfLarge(tbvh_scene* s) {
    if (s->layout != TBVH_LAYOUT_CWBVH || s->isTlas || s->nodes128 || (uint64_t)s->nNodes * 80 < (512ull << 20)) return 0;
    tbvh_context* c = s->ctx;
    if (hipMalloc((void**)&s->nodes128, (size_t)s->nNodes * 128) != hipSuccess) { s->nodes128 = nullptr; (void)hipGetLastError(); return 0; }   // no memory to spare: the packed array serves
    launch_cwbvh_pad(s->nodes, s->nodes128, s->nNodes, c->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    s->bytes += (uint64_t)s->nNodes * 128;
    return 0;
}

size_t hybridBytes(uint32_t nNodes, uint32_t K) { return ((size_t)K * 5 + (size_t)(nNodes - K) * 8) * 16; }

// BVH8_CWBVH scenes of the class that gets the per-launch coherence probe (48 - 384 MB of blobs: beyond the L2s, within reach of the Infinity
// Cache) keep two derived copies for INCOHERENT batches (kernels_cwbvh.hip: PROBED == 2): the nodes in surface-area priority order with the
// first kHybridPacked packed and the others one per 128-byte line, and the triangle records padded to 64 bytes.  hostNodes: the blob as
// uploaded (priority order computed on the host, ~0.1 s for 600 k nodes), or nullptr for trees made on the device (tbvh_convert_bvh2_device,
// tbvh_build_device emit level order, which already is close to priority order: no renumbering).  Failure to allocate is not an error: the
// scene then runs the one-kernel path.  TBVH_INCOHERENT_COPIES=0 turns the copies off.
constexpr uint32_t kHybridPacked = 8192;
int prepareIncoherentCopies(tbvh_scene* s, const Vec4* hostNodes) {
    tbvh_context* c = s->ctx;
    const uint64_t blobBytes = (s->nNodeBlocks + s->nTriBlocks) * 16;
    if (s->layout != TBVH_LAYOUT_CWBVH || s->isTlas || !c->incoherentCopies || blobBytes < (48ull << 20) || blobBytes > (384ull << 20) || s->nNodes <= kHybridPacked || !s->nTriBlocks) return 0;
    const uint32_t K = kHybridPacked;
    const uint64_t nT = s->nTriBlocks / 3;
    if (!s->nodesHy && hipMalloc((void**)&s->nodesHy, hybridBytes(s->nNodes, K)) != hipSuccess) { s->nodesHy = nullptr; (void)hipGetLastError(); return 0; }
    if (!s->tris64 && hipMalloc((void**)&s->tris64, nT * 64) != hipSuccess) { s->tris64 = nullptr; (void)hipGetLastError(); hipFree(s->nodesHy); s->nodesHy = nullptr; return 0; }
    if (hostNodes && !s->hyPerm) {
        std::vector<uint32_t> perm;
        cwbvh_priority_order(hostNodes, s->nNodes, perm);
        if (hipMalloc((void**)&s->hyPerm, (size_t)s->nNodes * 4) == hipSuccess) HIP_TRY(hipMemcpyAsync(s->hyPerm, perm.data(), (size_t)s->nNodes * 4, hipMemcpyHostToDevice, c->stream));
        else { s->hyPerm = nullptr; (void)hipGetLastError(); }
        HIP_TRY(hipStreamSynchronize(c->stream));   // perm goes out of scope
    }
    s->hybridK = K;
    HIP_TRY(hipMemsetAsync(s->nodesHy, 0, hybridBytes(s->nNodes, K), c->stream));
    launch_cwbvh_derive_hybrid(s->nodes, s->hyPerm, s->nodesHy, s->nNodes, K, c->stream);
    launch_cwbvh_pad_tris(s->tris, s->tris64, nT, c->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    s->bytes += hybridBytes(s->nNodes, K) + nT * 64;
    return 0;
}

tbvh_scene* newScene(tbvh_context* c, int layout) {
    tbvh_scene* s = new (std::nothrow) tbvh_scene;
    if (!s) return nullptr;
    s->ctx = c; s->layout = layout;
    c->scenes.push_back(s);
    return s;
}

}  // namespace

extern "C" {

int tbvh_abi_version(void) { return TBVH_ABI_VERSION; }
const char* tbvh_last_error(void) { return g_err; }

int tbvh_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { fail(TBVH_E_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e)); return e == hipErrorNoDevice ? 0 : TBVH_E_HIP; }
    return n;
}

int tbvh_init(int device, tbvh_context** out) {
    if (!out) return fail(TBVH_E_INVALID, "tbvh_init: out is null");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(TBVH_E_NODEVICE, "no HIP device available");
    if (device < 0 || device >= n) return fail(TBVH_E_NODEVICE, "device %d out of range (0..%d)", device, n - 1);
    tbvh_context* c = new (std::nothrow) tbvh_context;
    if (!c) return fail(TBVH_E_NOMEM, "out of host memory");
    c->device = device;
    hipDeviceProp_t prop;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->ownStream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e != hipSuccess) { delete c; return fail(TBVH_E_HIP, "context setup failed: %s", hipGetErrorString(e)); }
    c->stream = c->ownStream;
    c->numCUs = prop.multiProcessorCount;
    // persistent grid: one-wave workgroups, enough to fill every SIMD several times over
    c->blocks = (uint32_t)c->numCUs * 24u;
    if (const char* e = getenv("TBVH_BLOCKS_PER_CU")) {  // experiment knob
        const int b = atoi(e);
        if (b >= 1 && b <= 32) { c->blocks = (uint32_t)c->numCUs * (uint32_t)b; c->gridOverride = true; }
    }
    if (const char* e = getenv("TBVH_RAYS_PER_BLOCK")) {  // experiment knob
        const int b = atoi(e);
        if (b >= 64 && b <= 4096) { c->raysPerBlock = (uint32_t)b; c->gridOverride = true; }
    }
    if (const char* e = getenv("TBVH_SPLIT_RAYS")) { if (atoi(e) == 0) c->splitBelow = 0; }
    if (const char* e = getenv("TBVH_INCOHERENT_COPIES")) { if (atoi(e) == 0) c->incoherentCopies = false; }
    if (const char* e = getenv("TBVH_POOL_PARTS")) {  // experiment knob
        const int b = atoi(e);
        if (b >= 0 && (1 << b) <= kPoolParts) c->poolParts = (uint32_t)b;   // log2 of the partition count
    }
    c->spillEntries = 232;  // 32-bit entries per lane beyond the LDS part of the stack
    const size_t spillBytes = (size_t)(c->blocks + c->blocks / 3u) * 64 * c->spillEntries * 4;   // the largest grid any launch uses
    e = hipMalloc((void**)&c->spill, spillBytes);
    if (e == hipSuccess) e = hipMalloc((void**)&c->counter, 256);
    if (e == hipSuccess) e = hipMalloc((void**)&c->pool, (size_t)(kPoolParts + 1) * kPoolCounterStride * 4 * 2);
    if (e != hipSuccess) { tbvh_shutdown(c); return fail(TBVH_E_NOMEM, "device allocation failed: %s", hipGetErrorString(e)); }
    c->status = (uint32_t*)(c->counter + 4);
    hipMemset(c->counter, 0, 256);
    *out = c;
    return 0;
}

void tbvh_shutdown(tbvh_context* c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->ownStream) hipStreamSynchronize(c->ownStream);
    for (;;) {   // TLASes first: they hold references to their BLASes
        tbvh_scene* t = nullptr;
        for (tbvh_scene* s : c->scenes) if (s->isTlas) { t = s; break; }
        if (!t) break;
        tbvh_free_scene(t);
    }
    while (!c->scenes.empty()) tbvh_free_scene(c->scenes.back());
    if (c->spill) hipFree(c->spill);
    if (c->counter) hipFree(c->counter);
    if (c->pool) hipFree(c->pool);
    if (c->stageRays) hipFree(c->stageRays);
    if (c->stageOcc) hipFree(c->stageOcc);
    if (c->binScratch) hipFree(c->binScratch);
    delete c->pipe;
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->ownStream) hipStreamDestroy(c->ownStream);
    delete c;
}

int tbvh_synchronize(tbvh_context* c) {
    if (!c) return fail(TBVH_E_INVALID, "null context");
    if (int r = setDevice(c)) return r;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int tbvh_set_stream(tbvh_context* c, void* s) {
    if (!c) return fail(TBVH_E_INVALID, "null context");
    c->stream = s ? (hipStream_t)s : c->ownStream;
    return 0;
}

// ---- uploads ---------------------------------------------------------------------------

int tbvh_upload_bvh_gpu(tbvh_context* c, const void* nodes64, uint64_t nNodes, const uint32_t* primIdx, uint64_t nIdx,
                        const void* verts16, uint64_t nTris, tbvh_scene** out) {
    if (!c || !nodes64 || !primIdx || !verts16 || !out || nNodes == 0) return fail(TBVH_E_INVALID, "tbvh_upload_bvh_gpu: null/empty argument");
    if (const char* why = validate_bvh_gpu((const NodeAL*)nodes64, nNodes, nIdx)) return fail(TBVH_E_FORMAT, "%s", why);
    if (int r = setDevice(c)) return r;
    tbvh_scene* s = newScene(c, TBVH_LAYOUT_BVH_GPU);
    if (!s) return fail(TBVH_E_NOMEM, "out of host memory");
    uint32_t* dIdx = nullptr; float4* dVerts = nullptr;
    hipError_t e = hipMalloc((void**)&s->nodes, nNodes * 64);
    if (e == hipSuccess) e = hipMalloc((void**)&s->tris, (nIdx ? nIdx : 1) * 48);
    if (e == hipSuccess) e = hipMalloc((void**)&dIdx, (nIdx ? nIdx : 1) * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&dVerts, (nTris ? nTris : 1) * 48);
    if (e == hipSuccess) e = hipMemcpyAsync(s->nodes, nodes64, nNodes * 64, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dIdx, primIdx, nIdx * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dVerts, verts16, nTris * 48, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && nIdx) { launch_gather_tris(dIdx, dVerts, s->tris, nIdx, nTris, c->stream); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (dIdx) hipFree(dIdx);
    if (dVerts) hipFree(dVerts);
    if (e != hipSuccess) { tbvh_free_scene(s); return fail(TBVH_E_HIP, "BVH_GPU upload failed: %s", hipGetErrorString(e)); }
    s->nNodeBlocks = nNodes * 4; s->nTriBlocks = nIdx * 3;
    s->bytes = nNodes * 64 + nIdx * 48;
    *out = s;
    return 0;
}

int tbvh_upload_bvh4_gpu(tbvh_context* c, const void* blocks16, uint64_t nBlocks, tbvh_scene** out) {
    if (!c || !blocks16 || !out || nBlocks < 4) return fail(TBVH_E_INVALID, "tbvh_upload_bvh4_gpu: null/empty argument");
    if (const char* why = validate_bvh4_gpu((const Vec4*)blocks16, nBlocks)) return fail(TBVH_E_FORMAT, "%s", why);
    if (int r = setDevice(c)) return r;
    tbvh_scene* s = newScene(c, TBVH_LAYOUT_BVH4_GPU);
    if (!s) return fail(TBVH_E_NOMEM, "out of host memory");
    hipError_t e = hipMalloc((void**)&s->nodes, nBlocks * 16);
    if (e == hipSuccess) e = hipMemcpyAsync(s->nodes, blocks16, nBlocks * 16, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { tbvh_free_scene(s); return fail(TBVH_E_HIP, "BVH4_GPU upload failed: %s", hipGetErrorString(e)); }
    s->nNodeBlocks = nBlocks; s->bytes = nBlocks * 16;
    *out = s;
    return 0;
}

int tbvh_upload_cwbvh(tbvh_context* c, const void* nodes16, uint64_t nNodeBlocks, const void* tris16, uint64_t nTriBlocks,
                      tbvh_scene** out) {
    if (!c || !nodes16 || !out || nNodeBlocks < 5 || (nTriBlocks && !tris16)) return fail(TBVH_E_INVALID, "tbvh_upload_cwbvh: null/empty argument");
    if (nNodeBlocks % 5) return fail(TBVH_E_FORMAT, "CWBVH node blocks (%llu) not a multiple of 5", (unsigned long long)nNodeBlocks);
    if (const char* why = validate_cwbvh((const Vec4*)nodes16, nNodeBlocks / 5, nTriBlocks)) return fail(TBVH_E_FORMAT, "%s", why);
    if (int r = setDevice(c)) return r;
    tbvh_scene* s = newScene(c, TBVH_LAYOUT_CWBVH);
    if (!s) return fail(TBVH_E_NOMEM, "out of host memory");
    hipError_t e = hipMalloc((void**)&s->nodes, nNodeBlocks * 16);
    if (e == hipSuccess) e = hipMalloc((void**)&s->tris, (nTriBlocks ? nTriBlocks : 1) * 16);
    if (e == hipSuccess) e = hipMemcpyAsync(s->nodes, nodes16, nNodeBlocks * 16, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && nTriBlocks) e = hipMemcpyAsync(s->tris, tris16, nTriBlocks * 16, hipMemcpyHostToDevice, c->stream);
    s->nNodes = (uint32_t)(nNodeBlocks / 5);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { tbvh_free_scene(s); return fail(TBVH_E_HIP, "CWBVH upload failed: %s", hipGetErrorString(e)); }
    s->nNodeBlocks = nNodeBlocks; s->nTriBlocks = nTriBlocks;
    s->bytes = (nNodeBlocks + nTriBlocks) * 16;
    if (int r = padCwbvhIfLarge(s)) { tbvh_free_scene(s); return r; }
    if (int r = prepareIncoherentCopies(s, (const Vec4*)nodes16)) { tbvh_free_scene(s); return r; }
    *out = s;
    return 0;
}

namespace {
// (re)build the 4-wide TLAS from the BVH_GPU nodes on the device; asynchronous on the context's stream
int buildTlas4(tbvh_scene* s) {
    tbvh_context* c = s->ctx;
    if (s->blasLayout == TBVH_LAYOUT_CWBVH || s->blasMixCw2) {
        const uint64_t cap = tlas8_cap_nodes(s->nTlasNodes, s->nInst);
        if (cap > 0x00ffffffull) {   // wide-node indices share a word with 8 flag bits in places: the flat loop serves larger TLASes — and a wide
            // TLAS left from an earlier, smaller upload must not be traversed in its place (launchQuery keys on the pointer)
            if (s->tlas8) hipFree(s->tlas8);
            if (s->tlas8Refs) hipFree(s->tlas8Refs);
            s->tlas8 = nullptr; s->tlas8Refs = nullptr; s->tlas8Cap = 0;
            return 0;
        }
        if (cap > s->tlas8Cap) {
            if (s->tlas8) hipFree(s->tlas8);
            if (s->tlas8Refs) hipFree(s->tlas8Refs);
            s->tlas8 = nullptr; s->tlas8Refs = nullptr; s->tlas8Cap = 0;
            HIP_TRY(hipMalloc((void**)&s->tlas8, cap * 80));
            HIP_TRY(hipMalloc((void**)&s->tlas8Refs, cap * 4));
            s->tlas8Cap = cap;
            s->bytes += cap * 84;
        }
        const size_t sb = tlas_wide_scratch_bytes(s->nTlasNodes, s->nInst);
        if (sb > s->tlas4ScratchBytes) {
            if (s->tlas4Scratch) hipFree(s->tlas4Scratch);
            s->tlas4Scratch = nullptr; s->tlas4ScratchBytes = 0;
            HIP_TRY(hipMalloc(&s->tlas4Scratch, sb));
            s->tlas4ScratchBytes = sb;
        }
        launch_tlas8_build(s->nodes, (uint32_t)s->nTlasNodes, s->tlasIdx, (uint32_t)s->nTlasIdx, s->instances, (uint32_t)s->nInst, s->tlas8, (uint32_t)s->tlas8Cap, s->tlas8Refs,
                           (uint32_t)s->tlas8Cap, s->tlas4Scratch, c->status, c->stream);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (s->blasLayout != TBVH_LAYOUT_BVH4_GPU) return 0;
    const uint64_t cap = tlas4_cap_blocks(s->nTlasNodes, s->nInst);
    if (cap > 0x7fffffffull) {   // beyond 31-bit block offsets: the flat loop serves this TLAS; drop a 4-wide TLAS of an earlier, smaller upload
        if (s->tlas4) hipFree(s->tlas4);
        s->tlas4 = nullptr; s->tlas4Cap = 0;
        return 0;
    }
    if (cap > s->tlas4Cap) {
        if (s->tlas4) hipFree(s->tlas4);
        s->tlas4 = nullptr; s->tlas4Cap = 0;
        HIP_TRY(hipMalloc((void**)&s->tlas4, cap * 16));
        s->tlas4Cap = cap;
        s->bytes += cap * 16;
    }
    const size_t sb = tlas_wide_scratch_bytes(s->nTlasNodes, s->nInst);
    if (sb > s->tlas4ScratchBytes) {
        if (s->tlas4Scratch) hipFree(s->tlas4Scratch);
        s->tlas4Scratch = nullptr; s->tlas4ScratchBytes = 0;
        HIP_TRY(hipMalloc(&s->tlas4Scratch, sb));
        s->tlas4ScratchBytes = sb;
    }
    launch_tlas4_build(s->nodes, (uint32_t)s->nTlasNodes, s->tlasIdx, (uint32_t)s->nTlasIdx, s->instances, (uint32_t)s->nInst, s->tlas4, (uint32_t)s->tlas4Cap, s->tlas4Scratch, c->status, c->stream);
    HIP_TRY(hipGetLastError());
    return 0;
}

int tlasCopy(tbvh_scene* s, const void* nodes64, uint64_t nNodes, const uint32_t* idx, uint64_t nIdx, const void* inst, uint64_t nInst) {
    tbvh_context* c = s->ctx;
    // same hardening as the BLAS uploads: the TLAS kernels index instances[idx[]] and blas[blasIdx] unguarded
    if (const char* why = validate_bvh_gpu((const NodeAL*)nodes64, nNodes, nIdx)) return fail(TBVH_E_FORMAT, "TLAS: %s", why);
    for (uint64_t i = 0; i < nIdx; i++) if (idx[i] >= nInst) return fail(TBVH_E_FORMAT, "TLAS: primIdx[%llu] = %u is not an instance (%llu instances)", (unsigned long long)i, idx[i], (unsigned long long)nInst);
    const BLASInstanceCheck* ic = (const BLASInstanceCheck*)inst;
    for (uint64_t i = 0; i < nInst; i++) if (ic[i].blasIdx >= s->nBlas) return fail(TBVH_E_FORMAT, "instance %llu: blasIdx %u out of range (%llu BLASes)", (unsigned long long)i, ic[i].blasIdx, (unsigned long long)s->nBlas);
    if (nNodes > s->capNodes) { if (s->nodes) hipFree(s->nodes); s->nodes = nullptr; HIP_TRY(hipMalloc((void**)&s->nodes, nNodes * 64)); s->capNodes = nNodes; }
    if (nIdx > s->capIdx) { if (s->tlasIdx) hipFree(s->tlasIdx); s->tlasIdx = nullptr; HIP_TRY(hipMalloc((void**)&s->tlasIdx, nIdx * 4)); s->capIdx = nIdx; }
    if (nInst > s->capInst) { if (s->instances) hipFree(s->instances); s->instances = nullptr; HIP_TRY(hipMalloc((void**)&s->instances, nInst * 192)); s->capInst = nInst; }
    HIP_TRY(hipMemcpyAsync(s->nodes, nodes64, nNodes * 64, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(s->tlasIdx, idx, nIdx * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(s->instances, inst, nInst * 192, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));  // the caller may reuse its host arrays right away
    s->bytes = nNodes * 64 + nIdx * 4 + nInst * 192;
    s->nInst = nInst; s->nTlasNodes = nNodes; s->nTlasIdx = nIdx;
    return buildTlas4(s);
}
}  // namespace

int tbvh_upload_tlas(tbvh_context* c, const void* nodes64, uint64_t nNodes, const uint32_t* idx, uint64_t nIdx, const void* inst,
                     uint64_t nInst, tbvh_scene* const* blas, uint64_t nBlas, tbvh_scene** out) {
    if (!c || !nodes64 || !idx || !inst || !blas || !out || !nNodes || !nIdx || !nInst || !nBlas) return fail(TBVH_E_INVALID, "tbvh_upload_tlas: null/empty argument");
    int layout = 0;
    std::vector<BlasDesc> desc(nBlas);
    for (uint64_t i = 0; i < nBlas; i++) {
        const tbvh_scene* b = blas[i];
        if (!b || b->ctx != c || b->isTlas || b->zombie) return fail(TBVH_E_INVALID, "BLAS %llu is null, freed, a TLAS, or from another context", (unsigned long long)i);
        if (b->layout != TBVH_LAYOUT_CWBVH && b->layout != TBVH_LAYOUT_BVH4_GPU && b->layout != TBVH_LAYOUT_BVH_GPU)
            return fail(TBVH_E_INVALID, "BLAS %llu: layout %d cannot be a BLAS", (unsigned long long)i, b->layout);
        layout = i == 0 ? b->layout : (layout == b->layout ? layout : 0);   // 0: the BLASes mix layouts (traverse_tlas.cl:50-72)
        desc[i].nodes = b->nodes; desc[i].tris = b->tris; desc[i].opmap = b->opmap; desc[i].opmapN = b->opmapN; desc[i].layout = (uint32_t)b->layout;
    }
    if (int r = setDevice(c)) return r;
    tbvh_scene* s = newScene(c, TBVH_LAYOUT_BVH_GPU);
    if (!s) return fail(TBVH_E_NOMEM, "out of host memory");
    s->isTlas = true; s->blasLayout = layout; s->nBlas = nBlas;
    s->blasMixCw2 = layout == 0;
    for (uint64_t i = 0; i < nBlas; i++) if (blas[i]->layout == TBVH_LAYOUT_BVH4_GPU) s->blasMixCw2 = false;
    for (uint64_t i = 0; i < nBlas; i++) { s->blasList.push_back(blas[i]); blas[i]->usedBy.push_back(s); }
    hipError_t e = hipMalloc((void**)&s->blasDesc, nBlas * sizeof(BlasDesc));
    if (e == hipSuccess) e = hipMemcpy(s->blasDesc, desc.data(), nBlas * sizeof(BlasDesc), hipMemcpyHostToDevice);
    if (e != hipSuccess) { tbvh_free_scene(s); return fail(TBVH_E_HIP, "TLAS upload failed: %s", hipGetErrorString(e)); }
    if (int r = tlasCopy(s, nodes64, nNodes, idx, nIdx, inst, nInst)) { tbvh_free_scene(s); return r; }
    *out = s;
    return 0;
}

int tbvh_update_tlas(tbvh_scene* s, const void* nodes64, uint64_t nNodes, const uint32_t* idx, uint64_t nIdx, const void* inst, uint64_t nInst) {
    if (!s || !s->isTlas || !nodes64 || !idx || !inst || !nNodes || !nIdx || !nInst) return fail(TBVH_E_INVALID, "tbvh_update_tlas: not a TLAS or null/empty argument");
    if (int r = setDevice(s->ctx)) return r;
    return tlasCopy(s, nodes64, nNodes, idx, nIdx, inst, nInst);
}

namespace {
// BVH2 (device arrays) -> CWBVH scene.  msBefore: device time already spent on this request (builder), added to the report.
int convertDeviceImpl4(tbvh_context* c, const float4* dN2, uint64_t nNodes2, const uint32_t* dIdx, uint64_t nIdx, const float4* dV, uint64_t nTris,
                       tbvh_scene** out) {
    struct Tmp {
        void *blocks = nullptr, *itA = nullptr, *itB = nullptr, *cnt = nullptr;
        ~Tmp() { for (void* p : {blocks, itA, itB, cnt}) if (p) hipFree(p); }
    } t;
    const uint64_t capItems = nNodes2 / 2 + 2, capBlocks = capItems * 4 + nIdx * 3;
    if (capBlocks > 0xffffffffull) return fail(TBVH_E_INVALID, "BVH2 -> BVH4_GPU: stream would exceed 32-bit block indices");
    HIP_TRY(hipMalloc(&t.blocks, capBlocks * 16));
    HIP_TRY(hipMalloc(&t.itA, capItems * 8)); HIP_TRY(hipMalloc(&t.itB, capItems * 8)); HIP_TRY(hipMalloc(&t.cnt, 16));
    uint64_t nBlocks = 0; uint32_t levels = 0;
    HIP_TRY(run_convert_bvh4(dN2, (uint32_t)nNodes2, dIdx, nIdx, dV, nTris, (float4*)t.blocks, capBlocks, (uint2*)t.itA, (uint2*)t.itB, (uint32_t*)t.cnt, c->status,
                             c->stream, &nBlocks, &levels));
    uint32_t st = 0;
    HIP_TRY(hipMemcpy(&st, c->status, 4, hipMemcpyDeviceToHost));
    if (st & 12u) {
        hipMemset(c->status, 0, 4);
        return fail(TBVH_E_FORMAT, (st & 8u) ? "BVH2 -> BVH4_GPU: a node's inline triangles exceed the 16-bit relative offset (leaves too large)"
                                             : "BVH2 -> BVH4_GPU: malformed BVH2 (child, primitive or triangle index out of range)");
    }
    tbvh_scene* s = newScene(c, TBVH_LAYOUT_BVH4_GPU);
    if (!s) return fail(TBVH_E_NOMEM, "out of host memory");
    hipError_t e = hipMalloc((void**)&s->nodes, nBlocks * 16);
    if (e == hipSuccess) e = hipMemcpyAsync(s->nodes, t.blocks, nBlocks * 16, hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { tbvh_free_scene(s); return fail(TBVH_E_HIP, "BVH2 -> BVH4_GPU: %s", hipGetErrorString(e)); }
    s->nNodeBlocks = nBlocks; s->bytes = nBlocks * 16;
    *out = s;
    return 0;
}

int convertDeviceImpl(tbvh_context* c, int layout, const float4* dN2, uint64_t nNodes2, const uint32_t* dIdx, uint64_t nIdx, const float4* dV, uint64_t nTris,
                      tbvh_scene** out) {
    if (layout == TBVH_LAYOUT_BVH4_GPU) return convertDeviceImpl4(c, dN2, nNodes2, dIdx, nIdx, dV, nTris, out);
    struct Tmp {
        void *nodes = nullptr, *tris = nullptr, *itA = nullptr, *itB = nullptr, *cnt = nullptr;
        ~Tmp() { for (void* p : {nodes, tris, itA, itB, cnt}) if (p) hipFree(p); }
    } t;
    // worst case: every BVH2 interior node becomes a wide node ((n + 1) / 2 of them in a full binary tree, + the root)
    const uint32_t capNodes = (uint32_t)(nNodes2 / 2 + 2);
    HIP_TRY(hipMalloc(&t.nodes, (size_t)capNodes * 80)); HIP_TRY(hipMalloc(&t.tris, nIdx * 48));
    HIP_TRY(hipMalloc(&t.itA, (size_t)capNodes * 8)); HIP_TRY(hipMalloc(&t.itB, (size_t)capNodes * 8)); HIP_TRY(hipMalloc(&t.cnt, 16));
    uint32_t nWide = 0, levels = 0; uint64_t nWideTris = 0;
    HIP_TRY(run_convert_cwbvh(dN2, (uint32_t)nNodes2, dIdx, nIdx, dV, nTris, (float4*)t.nodes, capNodes, (float4*)t.tris, nIdx, (uint2*)t.itA, (uint2*)t.itB,
                              (uint32_t*)t.cnt, c->status, c->stream, &nWide, &nWideTris, &levels));
    uint32_t st = 0;
    HIP_TRY(hipMemcpy(&st, c->status, 4, hipMemcpyDeviceToHost));
    if (st & 12u) {
        hipMemset(c->status, 0, 4);
        return fail(TBVH_E_FORMAT, (st & 8u) ? "BVH2 -> CWBVH: a BVH2 leaf holds more than 3 triangles (SplitLeafs(3) first, like BVH8_CWBVH::ConvertFrom)"
                                             : "BVH2 -> CWBVH: malformed BVH2 (child, primitive or triangle index out of range)");
    }
    tbvh_scene* s = newScene(c, TBVH_LAYOUT_CWBVH);
    if (!s) return fail(TBVH_E_NOMEM, "out of host memory");
    // keep exactly what was produced
    hipError_t e = hipMalloc((void**)&s->nodes, (size_t)nWide * 80);
    if (e == hipSuccess) e = hipMalloc((void**)&s->tris, (nWideTris ? nWideTris : 1) * 48);
    if (e == hipSuccess) e = hipMemcpyAsync(s->nodes, t.nodes, (size_t)nWide * 80, hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess && nWideTris) e = hipMemcpyAsync(s->tris, t.tris, nWideTris * 48, hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { tbvh_free_scene(s); return fail(TBVH_E_HIP, "BVH2 -> CWBVH: %s", hipGetErrorString(e)); }
    s->nNodes = nWide; s->nNodeBlocks = (uint64_t)nWide * 5; s->nTriBlocks = nWideTris * 3;
    s->bytes = (s->nNodeBlocks + s->nTriBlocks) * 16;
    if (int r = padCwbvhIfLarge(s)) { tbvh_free_scene(s); return r; }
    if (int r = prepareIncoherentCopies(s, nullptr)) { tbvh_free_scene(s); return r; }
    *out = s;
    return 0;
}
}  // namespace

int tbvh_convert_bvh2_device(tbvh_context* c, const void* nodes32, uint64_t nNodes2, const uint32_t* primIdx, uint64_t nIdx, const void* verts16,
                             uint64_t nTris, int onDevice, int layout, tbvh_scene** out) {
    if (!c || !nodes32 || !primIdx || !verts16 || !out || nNodes2 == 0 || nIdx == 0 || nTris == 0) return fail(TBVH_E_INVALID, "tbvh_convert_bvh2_device: null/empty argument");
    if (layout != TBVH_LAYOUT_CWBVH && layout != TBVH_LAYOUT_BVH4_GPU) return fail(TBVH_E_INVALID, "tbvh_convert_bvh2_device: target layout %d not supported (BVH8_CWBVH and BVH4_GPU are)", layout);
    if (nNodes2 > 0x7fffffffull || nIdx > 0x7fffffffull) return fail(TBVH_E_INVALID, "tbvh_convert_bvh2_device: BVH2 too large for 32-bit node / triangle indices");
    if (int r = setDevice(c)) return r;
    struct Tmp {
        void *n2 = nullptr, *idx = nullptr, *v = nullptr;
        ~Tmp() { for (void* p : {n2, idx, v}) if (p) hipFree(p); }
    } t;
    const float4 *dN2 = (const float4*)nodes32, *dV = (const float4*)verts16;
    const uint32_t* dIdx = primIdx;
    if (!onDevice) {
        HIP_TRY(hipMalloc(&t.n2, nNodes2 * 32)); HIP_TRY(hipMalloc(&t.idx, nIdx * 4)); HIP_TRY(hipMalloc(&t.v, nTris * 48));
        HIP_TRY(hipMemcpyAsync(t.n2, nodes32, nNodes2 * 32, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(t.idx, primIdx, nIdx * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(t.v, verts16, nTris * 48, hipMemcpyHostToDevice, c->stream));
        dN2 = (const float4*)t.n2; dIdx = (const uint32_t*)t.idx; dV = (const float4*)t.v;
    }
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    const int r = convertDeviceImpl(c, layout, dN2, nNodes2, dIdx, nIdx, dV, nTris, out);
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    c->timed = true;
    return r;
}

namespace {
// builder: 0 = LBVH (maxLeafTris applies), 1 = PLOC (one triangle per leaf; radius = search window to each side)
int buildDeviceImpl(const char* who, tbvh_context* c, const void* verts16, uint64_t nTris, int onDevice, int layout, uint32_t maxLeafTris, int builder, uint32_t radius,
                    tbvh_scene** out) {
    if (!c || !verts16 || !out || nTris == 0) return fail(TBVH_E_INVALID, "%s: null/empty argument", who);
    if (layout != TBVH_LAYOUT_CWBVH && layout != TBVH_LAYOUT_BVH4_GPU) return fail(TBVH_E_INVALID, "%s: target layout %d not supported (BVH8_CWBVH and BVH4_GPU are)", who, layout);
    if (nTris > 0x3fffffffull) return fail(TBVH_E_INVALID, "%s: too many triangles for 32-bit node indices", who);
    if (int r = setDevice(c)) return r;
    struct Tmp {
        void *v = nullptr, *n2 = nullptr, *idx = nullptr, *scratch = nullptr;
        ~Tmp() { for (void* p : {v, n2, idx, scratch}) if (p) hipFree(p); }
    } t;
    const float4* dV = (const float4*)verts16;
    if (!onDevice) {
        HIP_TRY(hipMalloc(&t.v, nTris * 48));
        HIP_TRY(hipMemcpyAsync(t.v, verts16, nTris * 48, hipMemcpyHostToDevice, c->stream));
        dV = (const float4*)t.v;
    }
    size_t sortTemp = 0, scanTemp = 0;
    const size_t scratchBytes = builder == 1 ? ploc_scratch_bytes((uint32_t)nTris, &sortTemp, &scanTemp) : lbvh_scratch_bytes((uint32_t)nTris, &sortTemp);
    HIP_TRY(hipMalloc(&t.n2, nTris * 2 * 32)); HIP_TRY(hipMalloc(&t.idx, nTris * 4)); HIP_TRY(hipMalloc(&t.scratch, scratchBytes));
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    if (builder == 1) HIP_TRY(launch_ploc_build(dV, (uint32_t)nTris, radius, (float4*)t.n2, (uint32_t*)t.idx, t.scratch, sortTemp, scanTemp, c->stream, nullptr));
    else HIP_TRY(launch_lbvh_build(dV, (uint32_t)nTris, maxLeafTris, (float4*)t.n2, (uint32_t*)t.idx, t.scratch, sortTemp, c->stream));
    const int r = convertDeviceImpl(c, layout, (const float4*)t.n2, nTris * 2, (const uint32_t*)t.idx, nTris, dV, nTris, out);
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    c->timed = true;
    return r;
}
}  // namespace

int tbvh_build_device(tbvh_context* c, const void* verts16, uint64_t nTris, int onDevice, int layout, uint32_t maxLeafTris, tbvh_scene** out) {
    const uint32_t leafCap = layout == TBVH_LAYOUT_CWBVH ? 3u : 4u;
    // default: one triangle per leaf for CWBVH.  Contiguous Morton ranges make poor multi-triangle leaves: measured on the
    // Bistro stand-in, 1 / 2 / 3 triangles per leaf trace camera rays at 3629 / 3354 / 3125 and bounce rays at 2323 / 2150 /
    // 1884 MRays/s (the host SAH tree: 3300 / 2480), for 13 instead of 8 ms of build time and 22 % more memory
    if (maxLeafTris == 0) maxLeafTris = layout == TBVH_LAYOUT_CWBVH ? 1u : leafCap;
    if (maxLeafTris > leafCap) return fail(TBVH_E_INVALID, "tbvh_build_device: at most %u triangles per leaf for this layout", leafCap);
    return buildDeviceImpl("tbvh_build_device", c, verts16, nTris, onDevice, layout, maxLeafTris, 0, 0, out);
}

int tbvh_build_device_ploc(tbvh_context* c, const void* verts16, uint64_t nTris, int onDevice, int layout, uint32_t radius, tbvh_scene** out) {
    if (radius == 0) radius = 16;
    if (radius > 32u) return fail(TBVH_E_INVALID, "tbvh_build_device_ploc: search radius %u (1..32; 0 = the default 16)", radius);
    return buildDeviceImpl("tbvh_build_device_ploc", c, verts16, nTris, onDevice, layout, 1, 1, radius, out);
}

namespace {
// the TLASes over BLAS b hold a snapshot of its device pointers: rewrite their entries for b
int refreshBlasDescs(tbvh_scene* b) {
    for (tbvh_scene* t : b->usedBy)
        for (size_t i = 0; i < t->blasList.size(); i++)
            if (t->blasList[i] == b) {
                BlasDesc d; d.nodes = b->nodes; d.tris = b->tris; d.opmap = b->opmap; d.opmapN = b->opmapN; d.layout = (uint32_t)b->layout;
                HIP_TRY(hipMemcpy(t->blasDesc + i, &d, sizeof d, hipMemcpyHostToDevice));
            }
    return 0;
}
}  // namespace

int tbvh_set_opacity_micromaps(tbvh_scene* s, const uint32_t* mapData, uint32_t N, uint64_t nTris, int onDevice) {
    if (!s || s->isTlas) return fail(TBVH_E_INVALID, "tbvh_set_opacity_micromaps: not a BLAS scene (set the maps on the BLASes before uploading their TLAS)");
    tbvh_context* c = s->ctx;
    if (int r = setDevice(c)) return r;
    // validate first, build the new map next, and only then swap it in: every exit leaves the scene and the TLASes over it (their BlasDesc
    // snapshots) pointing at live memory — the old maps on a failure, the new ones on success
    const bool clear = !mapData || N == 0;
    if (!clear && (N > 1024 || nTris == 0)) return fail(TBVH_E_INVALID, "tbvh_set_opacity_micromaps: N = %u, %llu triangles", N, (unsigned long long)nTris);
    uint32_t* fresh = nullptr;
    uint64_t freshBytes = 0;
    if (!clear) {
        const uint64_t wordsPerTri = ((uint64_t)N * N + 31) >> 5, words = wordsPerTri * nTris;
        // the reference's index can run one row past the map when u + v == 1 exactly (tiny_bvh.h:8518-8519): keep that read inside the allocation
        const uint64_t pad = (((uint64_t)N + 1) * (N + 1) + 63) >> 5;
        freshBytes = (words + pad) * 4;
        if (hipMalloc((void**)&fresh, freshBytes) != hipSuccess) { (void)hipGetLastError(); return fail(TBVH_E_NOMEM, "tbvh_set_opacity_micromaps: %llu bytes of device memory", (unsigned long long)freshBytes); }
        hipError_t e = hipMemsetAsync(fresh + words, 0, pad * 4, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(fresh, mapData, words * 4, onDevice ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { hipFree(fresh); return fail(TBVH_E_HIP, "tbvh_set_opacity_micromaps: copying the maps failed: %s", hipGetErrorString(e)); }
    }
    HIP_TRY(hipStreamSynchronize(c->stream));   // no query may still read the old maps
    uint32_t* old = s->opmap;
    const uint64_t oldBytes = s->opmapBytes;
    s->opmap = fresh; s->opmapN = clear ? 0u : N; s->opmapBytes = freshBytes;
    s->bytes += freshBytes; s->bytes -= oldBytes;
    const int r = refreshBlasDescs(s);   // the descriptors are rewritten before the old maps go
    if (old && r == 0) hipFree(old);   // (a failed refresh may have left a descriptor on the old maps: leak them rather than dangle)
    return r;
}

int tbvh_scene_download(tbvh_scene* s, int which, void* dst, uint64_t capBytes, uint64_t* bytesOut) {
    if (!s || s->isTlas || (which != 0 && which != 1)) return fail(TBVH_E_INVALID, "tbvh_scene_download: not a BLAS scene or bad blob selector");
    tbvh_context* c = s->ctx;
    if (int r = setDevice(c)) return r;
    const void* src = which == 0 ? (const void*)s->nodes : (const void*)s->tris;
    const uint64_t bytes = (which == 0 ? s->nNodeBlocks : s->nTriBlocks) * 16;
    if (bytesOut) *bytesOut = src ? bytes : 0;
    if (!dst) return 0;
    if (!src) return fail(TBVH_E_INVALID, "tbvh_scene_download: this layout has no such blob");
    if (capBytes < bytes) return fail(TBVH_E_INVALID, "tbvh_scene_download: buffer too small (%llu < %llu bytes)", (unsigned long long)capBytes, (unsigned long long)bytes);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int tbvh_refit(tbvh_scene* s, const void* verts16, uint64_t nTris, int onDevice) {
    if (!s || !verts16 || !nTris) return fail(TBVH_E_INVALID, "tbvh_refit: null/empty argument");
    if (s->isTlas) return fail(TBVH_E_INVALID, "tbvh_refit: a TLAS is rebuilt with tbvh_rebuild_tlas_device / tbvh_update_tlas");
    tbvh_context* c = s->ctx;
    if (int r = setDevice(c)) return r;
    if (s->layout == TBVH_LAYOUT_BVH4_GPU) {
        // node list per level, child-box hand-over area: sized for the most nodes the stream can hold (4 blocks each)
        const uint32_t capNodes = (uint32_t)(s->nNodeBlocks / 4 + 1);
        if (!s->refitScratch) HIP_TRY(hipMalloc(&s->refitScratch, (size_t)capNodes * (16 + 128) + 256));
        const float4* dv4 = (const float4*)verts16;
        if (!onDevice) {
            if (s->vertStageTris < nTris) {
                if (s->vertStage) hipFree(s->vertStage);
                s->vertStage = nullptr; s->vertStageTris = 0;
                HIP_TRY(hipMalloc((void**)&s->vertStage, nTris * 48));
                s->vertStageTris = nTris;
            }
            HIP_TRY(hipMemcpyAsync(s->vertStage, verts16, nTris * 48, hipMemcpyHostToDevice, c->stream));
            dv4 = s->vertStage;
        }
        char* base = (char*)s->refitScratch;
        uint32_t* counter = (uint32_t*)base;
        void* items = base + 256;
        float4* childBox = (float4*)(base + 256 + (size_t)capNodes * 16);
        HIP_TRY(hipEventRecord(c->ev0, c->stream));
        HIP_TRY(run_refit_bvh4(s->nodes, s->nNodeBlocks, dv4, nTris, items, capNodes, counter, childBox, s->b4Levels, c->status, c->stream));
        HIP_TRY(hipEventRecord(c->ev1, c->stream));
        c->timed = true;
        return 0;
    }
    if (s->layout != TBVH_LAYOUT_CWBVH && s->layout != TBVH_LAYOUT_BVH_GPU)
        return fail(TBVH_E_INVALID, "tbvh_refit: layout %d is not refittable", s->layout);
    const uint32_t nNodes = (uint32_t)(s->layout == TBVH_LAYOUT_CWBVH ? s->nNodeBlocks / 5 : s->nNodeBlocks / 4);
    const uint64_t nRecords = s->nTriBlocks / 3;
    if (!s->refitScratch) HIP_TRY(hipMalloc(&s->refitScratch, refit_scratch_bytes(s->layout, nNodes)));
    const float4* dv = (const float4*)verts16;
    if (!onDevice) {
        if (s->vertStageTris < nTris) {
            if (s->vertStage) hipFree(s->vertStage);
            s->vertStage = nullptr; s->vertStageTris = 0;
            HIP_TRY(hipMalloc((void**)&s->vertStage, nTris * 48));
            s->vertStageTris = nTris;
        }
        HIP_TRY(hipMemcpyAsync(s->vertStage, verts16, nTris * 48, hipMemcpyHostToDevice, c->stream));
        dv = s->vertStage;
    }
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    HIP_TRY(launch_refit(s->layout, s->nodes, nNodes, s->tris, nRecords, dv, nTris, s->refitScratch, c->status, c->stream));
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    c->timed = true;
    // derived node layouts of the experiment kernels would be stale now
    if (s->nodes128) launch_cwbvh_pad(s->nodes, s->nodes128, nNodes, c->stream);   // keep the padded copy current
    if (s->nodesHy) launch_cwbvh_derive_hybrid(s->nodes, s->hyPerm, s->nodesHy, nNodes, s->hybridK, c->stream);
    if (s->tris64) launch_cwbvh_pad_tris(s->tris, s->tris64, s->nTriBlocks / 3, c->stream);
    return 0;
}

int tbvh_rebuild_tlas_device(tbvh_scene* s, const void* transforms, int onDevice, const float* blasBounds6, uint64_t nBlas) {
    if (!s || !s->isTlas) return fail(TBVH_E_INVALID, "tbvh_rebuild_tlas_device: not a TLAS");
    tbvh_context* c = s->ctx;
    if (int r = setDevice(c)) return r;
    const uint64_t n = s->nInst;
    if (n == 0 || n > 0x7fffffffull) return fail(TBVH_E_INVALID, "tbvh_rebuild_tlas_device: %llu instances", (unsigned long long)n);
    if (blasBounds6) {
        if (nBlas != s->nBlas) return fail(TBVH_E_INVALID, "tbvh_rebuild_tlas_device: %llu BLAS bounds for a TLAS over %llu BLASes", (unsigned long long)nBlas, (unsigned long long)s->nBlas);
        if (!s->blasBounds) HIP_TRY(hipMalloc((void**)&s->blasBounds, s->nBlas * 24));
        HIP_TRY(hipMemcpyAsync(s->blasBounds, blasBounds6, s->nBlas * 24, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));   // the caller's array may go away
    }
    if (!s->blasBounds) return fail(TBVH_E_INVALID, "tbvh_rebuild_tlas_device: the first call needs blas_bounds6");
    // an LBVH over n leaves has 2n - 1 nodes and n index entries
    const uint64_t nNodes = 2 * n - 1;
    if (nNodes > s->capNodes) { if (s->nodes) hipFree(s->nodes); s->nodes = nullptr; s->capNodes = 0; HIP_TRY(hipMalloc((void**)&s->nodes, nNodes * 64)); s->capNodes = nNodes; }
    if (n > s->capIdx) { if (s->tlasIdx) hipFree(s->tlasIdx); s->tlasIdx = nullptr; s->capIdx = 0; HIP_TRY(hipMalloc((void**)&s->tlasIdx, n * 4)); s->capIdx = n; }
    if (s->buildScratchFor != n) {
        if (s->buildScratch) hipFree(s->buildScratch);
        s->buildScratch = nullptr; s->buildScratchFor = 0;
        s->buildScratchBytes = tlas_build_scratch_bytes((uint32_t)n, &s->sortTempBytes);
        HIP_TRY(hipMalloc(&s->buildScratch, s->buildScratchBytes));
        s->buildScratchFor = n;
    }
    const float* xf = nullptr;
    if (transforms) {
        if (onDevice) xf = (const float*)transforms;
        else {
            if (s->xformStageCap < n) {   // tbvh_update_tlas may have grown the instance array since the last rebuild
                if (s->xformStage) hipFree(s->xformStage);
                s->xformStage = nullptr; s->xformStageCap = 0;
                HIP_TRY(hipMalloc((void**)&s->xformStage, n * 64));
                s->xformStageCap = n;
            }
            HIP_TRY(hipMemcpyAsync(s->xformStage, transforms, n * 64, hipMemcpyHostToDevice, c->stream));
            xf = s->xformStage;
        }
    }
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    HIP_TRY(launch_tlas_rebuild(s->nodes, s->tlasIdx, s->instances, xf, s->blasBounds, (uint32_t)n, (uint32_t)s->nBlas, s->buildScratch, s->sortTempBytes, c->stream));
    s->bytes = nNodes * 64 + n * 4 + n * 192;
    s->nTlasNodes = nNodes; s->nTlasIdx = n;
    if (int r = buildTlas4(s)) return r;
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    c->timed = true;
    return 0;
}

int tbvh_tlas_download(tbvh_scene* s, void* nodes64, uint64_t capNodes, uint32_t* idx, uint64_t capIdx, void* instances192, uint64_t capInst,
                       uint64_t* nNodesOut) {
    if (!s || !s->isTlas) return fail(TBVH_E_INVALID, "tbvh_tlas_download: not a TLAS");
    tbvh_context* c = s->ctx;
    if (int r = setDevice(c)) return r;
    HIP_TRY(hipStreamSynchronize(c->stream));
    const uint64_t n = s->nInst, nNodes = s->nTlasNodes;
    if (nNodesOut) *nNodesOut = nNodes;
    if (nodes64) { if (capNodes < nNodes) return fail(TBVH_E_INVALID, "tbvh_tlas_download: node buffer too small"); HIP_TRY(hipMemcpy(nodes64, s->nodes, nNodes * 64, hipMemcpyDeviceToHost)); }
    if (idx) { if (capIdx < n) return fail(TBVH_E_INVALID, "tbvh_tlas_download: index buffer too small"); HIP_TRY(hipMemcpy(idx, s->tlasIdx, n * 4, hipMemcpyDeviceToHost)); }
    if (instances192) { if (capInst < n) return fail(TBVH_E_INVALID, "tbvh_tlas_download: instance buffer too small"); HIP_TRY(hipMemcpy(instances192, s->instances, n * 192, hipMemcpyDeviceToHost)); }
    return 0;
}

void tbvh_free_scene(tbvh_scene* s) {
    if (!s) return;
    tbvh_context* c = s->ctx;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    if (!s->isTlas && !s->usedBy.empty()) { s->zombie = true; return; }   // a TLAS still points at this BLAS's memory: freed with the last such TLAS
    if (s->isTlas) {
        std::vector<tbvh_scene*> mine;
        mine.swap(s->blasList);
        for (tbvh_scene* b : mine) {
            for (size_t i = 0; i < b->usedBy.size(); i++) if (b->usedBy[i] == s) { b->usedBy.erase(b->usedBy.begin() + i); break; }
            if (b->zombie && b->usedBy.empty()) { b->zombie = false; tbvh_free_scene(b); }
        }
    }
    if (s->nodes) hipFree(s->nodes);
    if (s->tris) hipFree(s->tris);
    if (s->nodes128) hipFree(s->nodes128);
    if (s->nodesHy) hipFree(s->nodesHy);
    if (s->tris64) hipFree(s->tris64);
    if (s->hyPerm) hipFree(s->hyPerm);
    if (s->tlasIdx) hipFree(s->tlasIdx);
    if (s->instances) hipFree(s->instances);
    if (s->blasDesc) hipFree(s->blasDesc);
    if (s->blasBounds) hipFree(s->blasBounds);
    if (s->xformStage) hipFree(s->xformStage);
    if (s->buildScratch) hipFree(s->buildScratch);
    if (s->tlas4) hipFree(s->tlas4);
    if (s->tlas4Scratch) hipFree(s->tlas4Scratch);
    if (s->tlas8) hipFree(s->tlas8);
    if (s->tlas8Refs) hipFree(s->tlas8Refs);
    if (s->refitScratch) hipFree(s->refitScratch);
    if (s->opmap) hipFree(s->opmap);
    if (s->vertStage) hipFree(s->vertStage);
    for (size_t i = 0; i < c->scenes.size(); i++)
        if (c->scenes[i] == s) { c->scenes.erase(c->scenes.begin() + i); break; }
    delete s;
}
int tbvh_scene_layout(const tbvh_scene* s) { return s ? s->layout : TBVH_E_INVALID; }
uint64_t tbvh_scene_device_bytes(const tbvh_scene* s) { return s ? s->bytes : 0; }

int tbvh_set_variant(tbvh_scene* s, int v) {
    if (!s) return fail(TBVH_E_INVALID, "null scene");
    // only the BVH8_CWBVH kernel keeps diagnostic variants (kernels_cwbvh.hip: forced schedules, instrumented kernels)
    const bool ok = v == 0 || (!s->isTlas && s->layout == TBVH_LAYOUT_CWBVH && cwbvh_variant_valid(v));
    if (!ok) return fail(TBVH_E_INVALID, "unknown variant %d for layout %d", v, s->layout);
    s->variant = v;
    return 0;
}

// ---- queries ---------------------------------------------------------------------------

int tbvh_intersect_device(tbvh_scene* s, void* dRays, uint64_t n) {
    if (!s || (!dRays && n)) return fail(TBVH_E_INVALID, "tbvh_intersect_device: null argument");
    if (((uintptr_t)dRays) & 15) return fail(TBVH_E_INVALID, "ray array must be 16-byte aligned");
    return launchQuery(s, (RayRec*)dRays, n, nullptr);
}

int tbvh_intersect_device_fresh(tbvh_scene* s, void* dRays, uint64_t n, float tmax) {
    if (!s || (!dRays && n)) return fail(TBVH_E_INVALID, "tbvh_intersect_device_fresh: null argument");
    if (((uintptr_t)dRays) & 15) return fail(TBVH_E_INVALID, "ray array must be 16-byte aligned");
    return launchQuery(s, (RayRec*)dRays, n, nullptr, true, tmax);
}

int tbvh_occluded_device(tbvh_scene* s, const void* dRays, uint64_t n, uint8_t* dOcc) {
    if (!s || ((!dRays || !dOcc) && n)) return fail(TBVH_E_INVALID, "tbvh_occluded_device: null argument");
    if (((uintptr_t)dRays) & 15) return fail(TBVH_E_INVALID, "ray array must be 16-byte aligned");
    return launchQuery(s, (RayRec*)dRays, n, dOcc);
}

int tbvh_intersect(tbvh_scene* s, void* rays, uint64_t n, uint32_t stride) {
    if (!s || (!rays && n)) return fail(TBVH_E_INVALID, "tbvh_intersect: null argument");
    if (stride < 64 || (stride & 3)) return fail(TBVH_E_INVALID, "stride must be >= 64 and a multiple of 4 (got %u)", stride);
    if (n == 0) return 0;
    tbvh_context* c = s->ctx;
    if (int r = setDevice(c)) return r;
    if (int r = ensureStage(c, n)) return r;
    if (n >= kPipeMinRays) {   // pinned, chunked, multi-threaded staging (see HostPipe)
        if (int r = ensurePipe(c, n)) return r;
        if (int r = pipeUpload(c, (const char*)rays, n, stride, c->stageRays)) return r;
        if (int r = launchQuery(s, c->stageRays, n, nullptr)) return r;
        if (int r = pipeDownloadHits(c, (char*)rays, n, stride, c->stageRays)) return r;
        return checkStatus(c);
    }
    HIP_TRY(hipMemcpy2DAsync(c->stageRays, 64, rays, stride, 64, n, hipMemcpyHostToDevice, c->stream));
    if (int r = launchQuery(s, c->stageRays, n, nullptr)) return r;
    // copy back bytes 44..63 of every record (hit.inst + hit)
    HIP_TRY(hipMemcpy2DAsync((char*)rays + 44, stride, (char*)c->stageRays + 44, 64, 20, n, hipMemcpyDeviceToHost, c->stream));
    return checkStatus(c);
}

int tbvh_occluded(tbvh_scene* s, const void* rays, uint64_t n, uint32_t stride, uint8_t* occ) {
    if (!s || ((!rays || !occ) && n)) return fail(TBVH_E_INVALID, "tbvh_occluded: null argument");
    if (stride < 64 || (stride & 3)) return fail(TBVH_E_INVALID, "stride must be >= 64 and a multiple of 4 (got %u)", stride);
    if (n == 0) return 0;
    tbvh_context* c = s->ctx;
    if (int r = setDevice(c)) return r;
    if (int r = ensureStage(c, n)) return r;
    if (int r = ensureStageOcc(c, n)) return r;
    if (n >= kPipeMinRays) {
        if (int r = ensurePipe(c, n)) return r;
        if (int r = pipeUpload(c, (const char*)rays, n, stride, c->stageRays)) return r;
    } else HIP_TRY(hipMemcpy2DAsync(c->stageRays, 64, rays, stride, 64, n, hipMemcpyHostToDevice, c->stream));
    if (int r = launchQuery(s, c->stageRays, n, c->stageOcc)) return r;
    HIP_TRY(hipMemcpyAsync(occ, c->stageOcc, n, hipMemcpyDeviceToHost, c->stream));
    return checkStatus(c);
}

// ---- one ray array over several devices -----------------------------------------------------------
// The BVH is replicated (scenes[i] = the same blobs uploaded through context i), the ray array is cut into contiguous,
// wave-aligned shards (the same arithmetic as tinybvh_amd/sharding.py: shard_range), one host thread per device drives
// that device's staging + kernel + read-back, results land in the caller's array in place.  No collective.
void tbvh_shard_range(uint64_t n_rays, uint32_t rank, uint32_t world, uint64_t* begin, uint64_t* end) {
    const uint64_t align = 64, units = (n_rays + align - 1) / align;
    const uint64_t base = world ? units / world : 0, extra = world ? units % world : 0;
    const uint64_t b = (uint64_t)rank * base + (rank < extra ? rank : extra), e = b + base + (rank < extra ? 1 : 0);
    if (begin) *begin = b * align < n_rays ? b * align : n_rays;
    if (end) *end = e * align < n_rays ? e * align : n_rays;
}

namespace {
int shardedQuery(tbvh_scene* const* scenes, uint32_t nDev, void* rays, uint64_t n, uint32_t stride, uint8_t* occ, const char* who) {
    if (!scenes || nDev == 0 || (!rays && n)) return fail(TBVH_E_INVALID, "%s: null/empty argument", who);
    if (stride < 64 || (stride & 3)) return fail(TBVH_E_INVALID, "stride must be >= 64 and a multiple of 4 (got %u)", stride);
    for (uint32_t i = 0; i < nDev; i++) {
        if (!scenes[i]) return fail(TBVH_E_INVALID, "%s: scene %u is null", who, i);
        if (scenes[i]->layout != scenes[0]->layout || scenes[i]->isTlas != scenes[0]->isTlas) return fail(TBVH_E_INVALID, "%s: scene %u is not a replica of scene 0 (layout differs)", who, i);
        for (uint32_t j = 0; j < i; j++) if (scenes[j]->ctx == scenes[i]->ctx) return fail(TBVH_E_INVALID, "%s: scenes %u and %u share a context (one context, i.e. one stream and staging area, per shard)", who, j, i);
    }
    if (n == 0) return 0;
    std::vector<int> rc(nDev, 0);
    std::vector<std::string> msg(nDev);
    auto work = [&](uint32_t i) {
        uint64_t b, e;
        tbvh_shard_range(n, i, nDev, &b, &e);
        if (e == b) return;
        char* base = (char*)rays + b * stride;
        rc[i] = occ ? tbvh_occluded(scenes[i], base, e - b, stride, occ + b) : tbvh_intersect(scenes[i], base, e - b, stride);
        if (rc[i]) msg[i] = tbvh_last_error();   // the worker's thread-local message
    };
    std::vector<std::thread> th;
    for (uint32_t i = 1; i < nDev; i++) th.emplace_back(work, i);
    work(0);
    for (auto& t : th) t.join();
    for (uint32_t i = 0; i < nDev; i++) if (rc[i]) return fail(rc[i], "%s: shard %u (device %d): %s", who, i, scenes[i]->ctx->device, msg[i].c_str());
    return 0;
}
}  // namespace

int tbvh_intersect_sharded(tbvh_scene* const* scenes, uint32_t nDev, void* rays, uint64_t n, uint32_t stride) {
    return shardedQuery(scenes, nDev, rays, n, stride, nullptr, "tbvh_intersect_sharded");
}
int tbvh_occluded_sharded(tbvh_scene* const* scenes, uint32_t nDev, const void* rays, uint64_t n, uint32_t stride, uint8_t* occ) {
    if (!occ && n) return fail(TBVH_E_INVALID, "tbvh_occluded_sharded: null output");
    return shardedQuery(scenes, nDev, (void*)rays, n, stride, occ, "tbvh_occluded_sharded");
}

// ---- device-resident rays over several devices: nothing crosses the host -------------------------------------------------------------
// Every device's launch is asynchronous on its context's stream, so ONE host thread enqueues them all (a few tens of microseconds each:
// dispatch_ms[i] = host time spent enqueueing device i's launch), then waits for all.  With the rays produced and consumed where they are
// traced — a wavefront path tracer per device (tbvh_wavefront_render_sharded), or rays a kernel of the caller's wrote — the devices run
// at their device-resident rate; tbvh_intersect_sharded above moves HOST rays and is bound by the host (DESIGN.md par. 7).
namespace {
int shardedDeviceQuery(tbvh_scene* const* scenes, uint32_t nDev, void* const* dRays, const uint64_t* nRays, uint8_t* const* dOcc, int fresh, float tmax,
                       float* kernelMs, float* dispatchMs, const char* who) {
    if (!scenes || !nDev || !dRays || !nRays) return fail(TBVH_E_INVALID, "%s: null argument", who);
    for (uint32_t i = 0; i < nDev; i++) {
        if (!scenes[i]) return fail(TBVH_E_INVALID, "%s: scenes[%u] is null", who, i);
        if (nRays[i] && (!dRays[i] || (dOcc && !dOcc[i]))) return fail(TBVH_E_INVALID, "%s: null ray / output pointer for device %u", who, i);
        for (uint32_t k = 0; k < i; k++) if (scenes[k]->ctx == scenes[i]->ctx) return fail(TBVH_E_INVALID, "%s: scenes %u and %u share a context (one scene per context)", who, k, i);
    }
    int rc = 0;
    uint32_t launched = 0;
    for (; launched < nDev && !rc; launched++) {
        const uint32_t i = launched;
        const auto t0 = std::chrono::steady_clock::now();
        rc = launchQuery(scenes[i], (RayRec*)dRays[i], nRays[i], dOcc ? dOcc[i] : nullptr, fresh != 0, tmax);
        if (dispatchMs) dispatchMs[i] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    // wait for everything that was enqueued, also after a failure: the caller gets its buffers back quiescent (the first error is reported)
    std::string firstErr = rc ? tbvh_last_error() : "";
    for (uint32_t i = 0; i < launched; i++) {
        if (!nRays[i]) { if (kernelMs) kernelMs[i] = 0.f; continue; }
        const int r = checkStatus(scenes[i]->ctx);   // synchronizes device i's stream
        if (r && !rc) { rc = r; firstErr = tbvh_last_error(); }
        if (kernelMs) kernelMs[i] = r ? -1.f : tbvh_time_last_ms(scenes[i]->ctx);
    }
    if (rc) return fail(rc, "%s: %s", who, firstErr.c_str());
    return 0;
}
}  // namespace

int tbvh_intersect_sharded_device(tbvh_scene* const* scenes, uint32_t nDev, void* const* dRays, const uint64_t* nRays, int fresh, float tmax,
                                  float* kernelMs, float* dispatchMs) {
    return shardedDeviceQuery(scenes, nDev, dRays, nRays, nullptr, fresh, tmax, kernelMs, dispatchMs, "tbvh_intersect_sharded_device");
}
int tbvh_occluded_sharded_device(tbvh_scene* const* scenes, uint32_t nDev, const void* const* dRays, const uint64_t* nRays, uint8_t* const* dOcc,
                                 float* kernelMs, float* dispatchMs) {
    if (!dOcc) return fail(TBVH_E_INVALID, "tbvh_occluded_sharded_device: null output");
    return shardedDeviceQuery(scenes, nDev, (void* const*)dRays, nRays, dOcc, 0, 1e30f, kernelMs, dispatchMs, "tbvh_occluded_sharded_device");
}

float tbvh_time_last_ms(tbvh_context* c) {
    if (!c || !c->timed) return -1.0f;
    hipSetDevice(c->device);
    if (hipEventSynchronize(c->ev1) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess) return -1.0f;
    return ms;
}

int tbvh_debug_stats(tbvh_context* c, uint64_t out[8], int reset) {
    if (!c || !out) return fail(TBVH_E_INVALID, "tbvh_debug_stats: null argument");
    if (int r = setDevice(c)) return r;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, c->counter + 8, 64, hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(hipMemset(c->counter + 8, 0, 64));
    return 0;
}

int tbvh_bin_rays_device(tbvh_context* c, const void* dIn, void* dOut, uint64_t n, const float bounds6[6], uint32_t cellBits, uint32_t flags, uint32_t* dPerm) {
    if (!c || !bounds6 || ((!dIn || !dOut) && n)) return fail(TBVH_E_INVALID, "tbvh_bin_rays_device: null argument");
    if (dIn == dOut) return fail(TBVH_E_INVALID, "tbvh_bin_rays_device: the batch cannot be binned in place");
    if (cellBits > 6 || (flags & ~3u) || (flags & 3u) == 3u) return fail(TBVH_E_INVALID, "tbvh_bin_rays_device: cell_bits 0..6, flags 0, 1 or 2");
    if (n > 0xffffffffull) return fail(TBVH_E_INVALID, "tbvh_bin_rays_device: at most 2^32 - 1 rays per call");
    if (int r = setDevice(c)) return r;
    if (!n) return 0;
    size_t scanTemp = 0;
    const size_t need = ray_bin_scratch_bytes(n, cellBits, flags, &scanTemp);
    if (need > c->binScratchBytes) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->binScratch) hipFree(c->binScratch);
        c->binScratch = nullptr; c->binScratchBytes = 0;
        HIP_TRY(hipMalloc(&c->binScratch, need));
        c->binScratchBytes = need;
    }
    RayBinArgs a;
    for (int k = 0; k < 3; k++) {
        a.lo[k] = bounds6[k];
        const float ext = bounds6[3 + k] - bounds6[k];
        a.scale[k] = ext > 0 ? (float)(1u << cellBits) / ext : 0.f;
    }
    a.cellBits = cellBits; a.flags = flags;
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    HIP_TRY(launch_ray_bin((const RayRec*)dIn, (RayRec*)dOut, dPerm, n, nullptr, a, c->binScratch, scanTemp, (uint32_t)c->numCUs * 16u, c->stream));
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    c->timed = true;
    return 0;
}

int tbvh_cwbvh_set_hybrid(tbvh_scene* s, int64_t packedNodes) {
    if (!s || s->isTlas || s->layout != TBVH_LAYOUT_CWBVH) return fail(TBVH_E_INVALID, "tbvh_cwbvh_set_hybrid: not a BVH8_CWBVH scene");
    tbvh_context* c = s->ctx;
    if (int r = setDevice(c)) return r;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (s->nodesHy) { s->bytes -= hybridBytes(s->nNodes, s->hybridK); hipFree(s->nodesHy); s->nodesHy = nullptr; }
    if (packedNodes < 0) return 0;
    const uint32_t K = (uint32_t)std::min<uint64_t>((uint64_t)packedNodes, s->nNodes) & ~7u;   // the padded part starts on a 128-byte line
    if (!s->hyPerm) {
        std::vector<Vec4> host((size_t)s->nNodes * 5);
        HIP_TRY(hipMemcpy(host.data(), s->nodes, host.size() * 16, hipMemcpyDeviceToHost));
        std::vector<uint32_t> perm;
        cwbvh_priority_order(host.data(), s->nNodes, perm);
        HIP_TRY(hipMalloc((void**)&s->hyPerm, (size_t)s->nNodes * 4));
        HIP_TRY(hipMemcpy(s->hyPerm, perm.data(), (size_t)s->nNodes * 4, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMalloc((void**)&s->nodesHy, hybridBytes(s->nNodes, K)));
    HIP_TRY(hipMemsetAsync(s->nodesHy, 0, hybridBytes(s->nNodes, K), c->stream));
    s->hybridK = K;
    launch_cwbvh_derive_hybrid(s->nodes, s->hyPerm, s->nodesHy, s->nNodes, K, c->stream);
    HIP_TRY(hipGetLastError());
    s->bytes += hybridBytes(s->nNodes, K);
    if (!s->tris64 && s->nTriBlocks) {
        const uint64_t nT = s->nTriBlocks / 3;
        HIP_TRY(hipMalloc((void**)&s->tris64, nT * 64));
        launch_cwbvh_pad_tris(s->tris, s->tris64, nT, c->stream);
        HIP_TRY(hipGetLastError());
        s->bytes += nT * 64;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int tbvh_debug_set_flags(tbvh_context* c, uint32_t flags) {
    if (!c) return fail(TBVH_E_INVALID, "tbvh_debug_set_flags: null context");
    c->expFlags = flags;
    return 0;
}

int tbvh_debug_last_probe(tbvh_context* c, uint32_t out[3]) {
    if (!c || !out) return fail(TBVH_E_INVALID, "tbvh_debug_last_probe: null argument");
    if (int r = setDevice(c)) return r;
    out[0] = out[1] = out[2] = 0;
    if (!c->lastProbed) return 0;
    HIP_TRY(hipStreamSynchronize(c->stream));
    // (the area the last launch drew from: the next launch's kernels will zero it)
    HIP_TRY(hipMemcpy(out, (uint32_t*)c->pool + (size_t)(c->poolCur ^ 1) * ((size_t)(kPoolParts + 1) * kPoolCounterStride) + (size_t)kPoolParts * kPoolCounterStride, 8, hipMemcpyDeviceToHost));
    out[2] = (out[1] != 0 && out[0] * 10u >= out[1] * 6u) ? 2u : 1u;   // the rule of k_cwbvh (kernels_cwbvh.hip)
    return 0;
}

// ---- ray generators ----------------------------------------------------------------------

int tbvh_generate_primary_device(tbvh_context* c, const tbvh_camera* cam, void* dRays, uint64_t first, uint64_t n) {
    if (!c || !cam || (!dRays && n)) return fail(TBVH_E_INVALID, "tbvh_generate_primary_device: null argument");
    if (cam->width % 4 || cam->height % 4 || !cam->spp_x || !cam->spp_y) return fail(TBVH_E_INVALID, "camera: width/height must be multiples of 4, spp > 0");
    if (int r = setDevice(c)) return r;
    if (!n) return 0;
    CameraArgs a;
    memcpy(a.eye, cam->eye, 12); memcpy(a.p1, cam->p1, 12); memcpy(a.p2, cam->p2, 12); memcpy(a.p3, cam->p3, 12);
    a.width = cam->width; a.height = cam->height; a.sppX = cam->spp_x; a.sppY = cam->spp_y;
    launch_gen_primary(a, (RayRec*)dRays, first, n, c->stream);
    HIP_TRY(hipGetLastError());
    return 0;
}

int tbvh_generate_bounce_device(tbvh_context* c, const void* dVerts, const void* dIn, void* dOut, uint64_t n, uint32_t seed) {
    if (!c || ((!dVerts || !dIn || !dOut) && n)) return fail(TBVH_E_INVALID, "tbvh_generate_bounce_device: null argument");
    if (int r = setDevice(c)) return r;
    if (!n) return 0;
    TriSource src; src.mode = 0; src.verts = (const float4*)dVerts;
    launch_gen_bounce(src, (const RayRec*)dIn, (RayRec*)dOut, n, seed, c->stream);
    HIP_TRY(hipGetLastError());
    return 0;
}

int tbvh_generate_shadow_device(tbvh_context* c, const void* dIn, void* dOut, uint64_t n, const float light[3], float eps) {
    if (!c || !light || ((!dIn || !dOut) && n)) return fail(TBVH_E_INVALID, "tbvh_generate_shadow_device: null argument");
    if (int r = setDevice(c)) return r;
    if (!n) return 0;
    launch_gen_shadow((const RayRec*)dIn, (RayRec*)dOut, n, light[0], light[1], light[2], eps, c->stream);
    HIP_TRY(hipGetLastError());
    return 0;
}

int tbvh_reset_hits_device(tbvh_context* c, void* dRays, uint64_t n, float tmax) {
    if (!c || (!dRays && n)) return fail(TBVH_E_INVALID, "tbvh_reset_hits_device: null argument");
    if (int r = setDevice(c)) return r;
    if (!n) return 0;
    launch_reset_hits((RayRec*)dRays, n, tmax, c->stream);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- wavefront path tracer (device-resident Generate / Extend / Shade / Connect) ----------------

constexpr uint32_t kWfCounterWords = 32u * 18u;   // 9 path-queue + 8 shadow-queue counters (max_depth <= 8), one 256-byte line each, 64-bit words

struct tbvh_wavefront {
    tbvh_context* ctx = nullptr;
    uint32_t width = 0, height = 0;   // of this object's accumulator: the image, or a band of it
    uint32_t firstRow = 0, fullHeight = 0;   // tbvh_wavefront_set_band: rows [firstRow, firstRow + height) of an image of fullHeight rows (0: the whole image)
    uint64_t n = 0;
    RayRec* rays[2] = {nullptr, nullptr};
    PathAux* aux[2] = {nullptr, nullptr};
    RayRec* shadow = nullptr;
    PathAux* shadowAux = nullptr;
    uint8_t* occ = nullptr;
    float* accum = nullptr;
    const float4** blasVerts = nullptr;        // device array: vertex array of every BLAS (TLAS scenes)
    uint32_t* blueNoise = nullptr;             // device copy of the 128 x 128 x 8 table (optional)
    uint64_t nBlasVerts = 0;
    unsigned long long* counters = nullptr;   // [0],[1] path queues, [2] shadow queue, [8..] per-depth history
    hipEvent_t e0 = nullptr, e1 = nullptr;
};

int tbvh_wavefront_create(tbvh_context* c, uint32_t width, uint32_t height, tbvh_wavefront** out) {
    if (!c || !out || !width || !height || (width & 3) || (height & 3)) return fail(TBVH_E_INVALID, "tbvh_wavefront_create: null argument or size not a multiple of 4");
    if (int r = setDevice(c)) return r;
    tbvh_wavefront* w = new (std::nothrow) tbvh_wavefront;
    if (!w) return fail(TBVH_E_NOMEM, "out of host memory");
    w->ctx = c; w->width = width; w->height = height; w->n = (uint64_t)width * height;
    hipError_t e = hipSuccess;
    for (int i = 0; i < 2 && e == hipSuccess; i++) {
        e = hipMalloc((void**)&w->rays[i], w->n * 64);
        if (e == hipSuccess) e = hipMalloc((void**)&w->aux[i], w->n * sizeof(PathAux));
    }
    if (e == hipSuccess) e = hipMalloc((void**)&w->shadow, w->n * 64);
    if (e == hipSuccess) e = hipMalloc((void**)&w->shadowAux, w->n * sizeof(PathAux));
    if (e == hipSuccess) e = hipMalloc((void**)&w->occ, w->n);
    if (e == hipSuccess) e = hipMalloc((void**)&w->accum, w->n * 16);
    if (e == hipSuccess) e = hipMalloc((void**)&w->counters, (size_t)kWfCounterWords * 8);
    if (e == hipSuccess) e = hipMemset(w->accum, 0, w->n * 16);
    if (e == hipSuccess) e = hipEventCreate(&w->e0);
    if (e == hipSuccess) e = hipEventCreate(&w->e1);
    if (e != hipSuccess) { tbvh_wavefront_destroy(w); return fail(TBVH_E_NOMEM, "wavefront allocation failed: %s", hipGetErrorString(e)); }
    *out = w;
    return 0;
}

void tbvh_wavefront_destroy(tbvh_wavefront* w) {
    if (!w) return;
    hipSetDevice(w->ctx->device);
    hipStreamSynchronize(w->ctx->stream);
    for (int i = 0; i < 2; i++) { if (w->rays[i]) hipFree(w->rays[i]); if (w->aux[i]) hipFree(w->aux[i]); }
    if (w->shadow) hipFree(w->shadow);
    if (w->shadowAux) hipFree(w->shadowAux);
    if (w->occ) hipFree(w->occ);
    if (w->accum) hipFree(w->accum);
    if (w->counters) hipFree(w->counters);
    if (w->blasVerts) hipFree((void*)w->blasVerts);
    if (w->blueNoise) hipFree(w->blueNoise);
    if (w->e0) hipEventDestroy(w->e0);
    if (w->e1) hipEventDestroy(w->e1);
    delete w;
}

int tbvh_wavefront_set_blas_vertices(tbvh_wavefront* w, const void* const* dVertsPerBlas, uint64_t nBlas) {
    if (!w || !dVertsPerBlas || !nBlas) return fail(TBVH_E_INVALID, "tbvh_wavefront_set_blas_vertices: null/empty argument");
    if (int r = setDevice(w->ctx)) return r;
    HIP_TRY(hipStreamSynchronize(w->ctx->stream));
    if (w->blasVerts) { hipFree((void*)w->blasVerts); w->blasVerts = nullptr; w->nBlasVerts = 0; }
    HIP_TRY(hipMalloc((void**)&w->blasVerts, nBlas * sizeof(void*)));
    HIP_TRY(hipMemcpy((void*)w->blasVerts, dVertsPerBlas, nBlas * sizeof(void*), hipMemcpyHostToDevice));
    w->nBlasVerts = nBlas;
    return 0;
}

int tbvh_wavefront_render(tbvh_wavefront* w, tbvh_scene* scene, const void* dVerts, const tbvh_camera* cam, const tbvh_wf_params* p,
                          tbvh_wf_stats* stats) {
    if (!w || !scene || !cam || !p) return fail(TBVH_E_INVALID, "tbvh_wavefront_render: null argument");
    if (scene->isTlas) {
        if (w->nBlasVerts < scene->nBlas) return fail(TBVH_E_INVALID, "tbvh_wavefront_render: a TLAS scene needs tbvh_wavefront_set_blas_vertices (%llu BLASes)", (unsigned long long)scene->nBlas);
    } else if (!dVerts) return fail(TBVH_E_INVALID, "tbvh_wavefront_render: null vertex array");
    if (scene->ctx != w->ctx) return fail(TBVH_E_INVALID, "scene and wavefront belong to different contexts");
    const uint32_t fullH = w->fullHeight ? w->fullHeight : w->height;
    if (cam->width != w->width || cam->height != fullH) return fail(TBVH_E_INVALID, "camera size differs from the wavefront's (a band takes the FULL image's camera)");
    const uint32_t maxDepth = p->max_depth ? (p->max_depth > 8 ? 8 : p->max_depth) : 3;
    tbvh_context* c = w->ctx;
    if (int r = setDevice(c)) return r;
    hipStream_t st = c->stream;
    HIP_TRY(hipEventRecord(w->e0, st));
    if (p->clear) HIP_TRY(hipMemsetAsync(w->accum, 0, w->n * 16, st));
    // Queue counters: one per queue AND depth — the path queue that depth d reads (word 32 d; depth 0: the n camera rays) and the shadow queue
    // depth d fills (word 32 (9 + d)) —, each on its own 256-byte line (appends to different queues hit different lines; same-line atomics are
    // serialised memory-side).  Nothing is reused within a frame, so nothing has to be cleared or copied between the stages: k_wf_generate sets
    // them all, and the per-depth history of the statistics IS the counters.  (Until round 3 two path counters and one shadow counter were
    // recycled: 13 memsets and 7 copies per 3-bounce frame, each a launch of its own — 0.18 of the 0.92 ms of a 1280 x 720 frame.)
    auto QP = [&](uint32_t d) { return &w->counters[32u * d]; };
    auto QS = [&](uint32_t d) { return &w->counters[32u * (9u + d)]; };
    CameraArgs ca;
    memcpy(ca.eye, cam->eye, 12); memcpy(ca.p1, cam->p1, 12); memcpy(ca.p2, cam->p2, 12); memcpy(ca.p3, cam->p3, 12);
    ca.width = cam->width; ca.height = cam->height; ca.sppX = ca.sppY = 1;
    launch_wf_generate(ca, w->rays[0], w->aux[0], w->n, p->seed, w->firstRow, w->height, w->counters, kWfCounterWords, st);
    int cur = 0;
    for (uint32_t d = 0; d < maxDepth; d++) {
        const int nxt = cur ^ 1;
        // Extend: nearest hit of every live path; the batch size lives on the device
        if (int r = launchQuery(scene, w->rays[cur], w->n, nullptr, false, 1e30f, QP(d))) return r;
        ShadeArgs a;
        a.in = w->rays[cur]; a.auxIn = w->aux[cur]; a.nIn = QP(d);
        a.out = w->rays[nxt]; a.auxOut = w->aux[nxt]; a.nOut = QP(d + 1);
        a.shadow = w->shadow; a.shadowAux = w->shadowAux; a.nShadow = QS(d);
        a.verts = (const float4*)dVerts; a.accum = w->accum;
        a.blasVerts = scene->isTlas ? w->blasVerts : nullptr; a.instances = scene->isTlas ? scene->instances : nullptr;
        a.blueNoise = w->blueNoise; a.sampleIdx = p->sample_index; a.width = w->width; a.height = fullH; a.pixelOffset = w->firstRow * w->width;
        memcpy(a.lightPos, p->light_pos, 12); memcpy(a.lightColor, p->light_color, 12); memcpy(a.skyLo, p->sky_lo, 12); memcpy(a.skyHi, p->sky_hi, 12);
        a.lightSize[0] = p->light_size[0]; a.lightSize[1] = p->light_size[1]; a.flags = p->flags;
        a.eps = p->eps; a.depth = d; a.maxDepth = maxDepth; a.seed = p->seed;
        launch_wf_shade(a, w->n, st);
        // Connect: any-hit over the shadow queue, then add what is unoccluded
        if (int r = launchQuery(scene, w->shadow, w->n, w->occ, false, 1e30f, QS(d))) return r;
        launch_wf_connect(w->occ, w->shadowAux, QS(d), w->accum, w->n, st);
        cur = nxt;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(w->e1, st));
    if (stats) {
        std::vector<unsigned long long> h(kWfCounterWords);
        HIP_TRY(hipMemcpyAsync(h.data(), w->counters, (size_t)kWfCounterWords * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        memset(stats, 0, sizeof *stats);
        for (uint32_t d = 0; d < maxDepth; d++) { stats->extend_rays[d] = h[32u * d]; stats->shadow_rays[d] = h[32u * (9u + d)]; }
        HIP_TRY(hipEventElapsedTime(&stats->frame_ms, w->e0, w->e1));
        if (int r = checkStatus(c)) return r;
    }
    return 0;
}

int tbvh_wavefront_set_band(tbvh_wavefront* w, uint32_t firstRow, uint32_t fullHeight) {
    if (!w) return fail(TBVH_E_INVALID, "tbvh_wavefront_set_band: null wavefront");
    if (fullHeight == 0) { w->firstRow = 0; w->fullHeight = 0; return 0; }
    if ((firstRow & 3u) || (fullHeight & 3u) || (uint64_t)firstRow + w->height > fullHeight) return fail(TBVH_E_INVALID, "tbvh_wavefront_set_band: rows %u + %u of %u (multiples of 4, inside the image)", firstRow, w->height, fullHeight);
    w->firstRow = firstRow; w->fullHeight = fullHeight;
    return 0;
}

// One frame over several devices: every wavefront object renders its band of the image on its own device with its own copy of the scene; the
// frames are enqueued by this thread one after the other (each enqueue is asynchronous) and run concurrently.  No exchange between devices:
// a band's accumulator stays on its device until tbvh_wavefront_read_sharded gathers the image.
int tbvh_wavefront_render_sharded(tbvh_wavefront* const* wfs, tbvh_scene* const* scenes, const void* const* dVerts, uint32_t nDev, const tbvh_camera* cam,
                                  const tbvh_wf_params* p, tbvh_wf_stats* stats, float* dispatchMs) {
    if (!wfs || !scenes || !nDev || !cam || !p) return fail(TBVH_E_INVALID, "tbvh_wavefront_render_sharded: null argument");
    uint32_t row = 0;
    for (uint32_t i = 0; i < nDev; i++) {
        if (!wfs[i] || !scenes[i]) return fail(TBVH_E_INVALID, "tbvh_wavefront_render_sharded: null wavefront / scene %u", i);
        const uint32_t fullH = wfs[i]->fullHeight ? wfs[i]->fullHeight : wfs[i]->height;
        if (fullH != cam->height || wfs[i]->firstRow != row) return fail(TBVH_E_INVALID, "tbvh_wavefront_render_sharded: band %u covers rows %u.. of %u, expected rows %u.. of %u (bands in order, tiling the image)", i, wfs[i]->firstRow, fullH, row, cam->height);
        row += wfs[i]->height;
        for (uint32_t k = 0; k < i; k++) if (wfs[k]->ctx == wfs[i]->ctx) return fail(TBVH_E_INVALID, "tbvh_wavefront_render_sharded: bands %u and %u share a context", k, i);
    }
    if (row != cam->height) return fail(TBVH_E_INVALID, "tbvh_wavefront_render_sharded: the bands cover %u of %u rows", row, cam->height);
    for (uint32_t i = 0; i < nDev; i++) {
        const auto t0 = std::chrono::steady_clock::now();
        if (int r = tbvh_wavefront_render(wfs[i], scenes[i], dVerts ? dVerts[i] : nullptr, cam, p, nullptr)) return r;
        if (dispatchMs) dispatchMs[i] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    for (uint32_t i = 0; i < nDev; i++) {
        tbvh_wavefront* w = wfs[i];
        tbvh_context* c = w->ctx;
        if (int r = setDevice(c)) return r;
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (int r = checkStatus(c)) return r;
        if (stats) {
            const uint32_t maxDepth = p->max_depth ? (p->max_depth > 8 ? 8 : p->max_depth) : 3;
            std::vector<unsigned long long> h(kWfCounterWords);
            HIP_TRY(hipMemcpy(h.data(), w->counters, (size_t)kWfCounterWords * 8, hipMemcpyDeviceToHost));
            memset(&stats[i], 0, sizeof stats[i]);
            for (uint32_t d = 0; d < maxDepth; d++) { stats[i].extend_rays[d] = h[32u * d]; stats[i].shadow_rays[d] = h[32u * (9u + d)]; }
            HIP_TRY(hipEventElapsedTime(&stats[i].frame_ms, w->e0, w->e1));
        }
    }
    return 0;
}

int tbvh_wavefront_read_sharded(tbvh_wavefront* const* wfs, uint32_t nDev, float* rgba) {
    if (!wfs || !nDev || !rgba) return fail(TBVH_E_INVALID, "tbvh_wavefront_read_sharded: null argument");
    for (uint32_t i = 0; i < nDev; i++) {
        if (!wfs[i]) return fail(TBVH_E_INVALID, "tbvh_wavefront_read_sharded: null wavefront %u", i);
        if (int r = tbvh_wavefront_read(wfs[i], rgba + (size_t)wfs[i]->firstRow * wfs[i]->width * 4)) return r;
    }
    return 0;
}

int tbvh_wavefront_set_blue_noise(tbvh_wavefront* w, const uint32_t* table, uint64_t nWords) {
    if (!w) return fail(TBVH_E_INVALID, "tbvh_wavefront_set_blue_noise: null wavefront");
    if (table && nWords != 128ull * 128 * 8) return fail(TBVH_E_INVALID, "tbvh_wavefront_set_blue_noise: the table is 128 x 128 x 8 = 131072 words (got %llu)", (unsigned long long)nWords);
    if (int r = setDevice(w->ctx)) return r;
    HIP_TRY(hipStreamSynchronize(w->ctx->stream));
    if (w->blueNoise) { hipFree(w->blueNoise); w->blueNoise = nullptr; }
    if (!table) return 0;
    HIP_TRY(hipMalloc((void**)&w->blueNoise, nWords * 4));
    HIP_TRY(hipMemcpy(w->blueNoise, table, nWords * 4, hipMemcpyHostToDevice));
    return 0;
}

int tbvh_wavefront_read(tbvh_wavefront* w, float* rgba) {
    if (!w || !rgba) return fail(TBVH_E_INVALID, "tbvh_wavefront_read: null argument");
    if (int r = setDevice(w->ctx)) return r;
    HIP_TRY(hipMemcpyAsync(rgba, w->accum, w->n * 16, hipMemcpyDeviceToHost, w->ctx->stream));
    HIP_TRY(hipStreamSynchronize(w->ctx->stream));
    return 0;
}

int tbvh_wavefront_finalize(tbvh_wavefront* w, float scale, uint32_t* pixels) {
    if (!w || !pixels) return fail(TBVH_E_INVALID, "tbvh_wavefront_finalize: null argument");
    if (int r = setDevice(w->ctx)) return r;
    uint32_t* d = (uint32_t*)w->shadow;   // 4 bytes per pixel in the shadow-ray buffer (64 bytes per pixel, idle between frames)
    launch_wf_finalize(w->accum, scale, d, w->n, w->ctx->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(pixels, d, w->n * 4, hipMemcpyDeviceToHost, w->ctx->stream));
    HIP_TRY(hipStreamSynchronize(w->ctx->stream));
    return 0;
}

// ---- device buffers ------------------------------------------------------------------------

int tbvh_device_malloc(tbvh_context* c, uint64_t bytes, void** out) {
    if (!c || !out) return fail(TBVH_E_INVALID, "tbvh_device_malloc: null argument");
    if (int r = setDevice(c)) return r;
    hipError_t e = hipMalloc(out, bytes ? bytes : 16);
    if (e != hipSuccess) return fail(TBVH_E_NOMEM, "hipMalloc(%llu): %s", (unsigned long long)bytes, hipGetErrorString(e));
    return 0;
}
int tbvh_device_free(tbvh_context* c, void* p) {
    if (!c) return fail(TBVH_E_INVALID, "null context");
    if (int r = setDevice(c)) return r;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipFree(p));
    return 0;
}
int tbvh_copy_to_device(tbvh_context* c, void* d, const void* src, uint64_t bytes) {
    if (!c || ((!d || !src) && bytes)) return fail(TBVH_E_INVALID, "tbvh_copy_to_device: null argument");
    if (int r = setDevice(c)) return r;
    HIP_TRY(hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}
int tbvh_copy_from_device(tbvh_context* c, void* dst, const void* d, uint64_t bytes) {
    if (!c || ((!d || !dst) && bytes)) return fail(TBVH_E_INVALID, "tbvh_copy_from_device: null argument");
    if (int r = setDevice(c)) return r;
    HIP_TRY(hipMemcpyAsync(dst, d, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

// Device memory bandwidth as this GPU delivers it today: a streaming copy (one float4 per thread, non-temporal; read + written bytes
// counted) and a read-only sweep over `bytes`, best of `reps` launches each.  The denominators of the roofline lines in bench.py.
static int timeBest(tbvh_context* c, uint32_t reps, const std::function<void()>& launch, double* bestMs) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { hipEventDestroy(e0); return fail(TBVH_E_HIP, "hipEventCreate failed"); }
    double best = 0;
    hipError_t err = hipSuccess;
    for (uint32_t i = 0; i <= reps && err == hipSuccess; i++) {   // the first launch warms up
        err = hipEventRecord(e0, c->stream);
        launch();
        if (err == hipSuccess) err = hipGetLastError();
        if (err == hipSuccess) err = hipEventRecord(e1, c->stream);
        if (err == hipSuccess) err = hipEventSynchronize(e1);
        float ms = 0;
        if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
        if (err == hipSuccess && i && ms > 0 && (best == 0 || ms < best)) best = ms;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (err != hipSuccess) return fail(TBVH_E_HIP, "measurement launch failed: %s", hipGetErrorString(err));
    if (best <= 0) return fail(TBVH_E_HIP, "measurement produced no timing");
    *bestMs = best;
    return 0;
}

int tbvh_measure_copy_bandwidth(tbvh_context* c, uint64_t bytes, uint32_t reps, double* gbps) {
    if (!c || !gbps || bytes < (1u << 20)) return fail(TBVH_E_INVALID, "tbvh_measure_copy_bandwidth: null argument or under 1 MB");
    if (int r = setDevice(c)) return r;
    void *a = nullptr, *b = nullptr;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) { if (a) hipFree(a); return fail(TBVH_E_NOMEM, "tbvh_measure_copy_bandwidth: cannot allocate 2 x %llu bytes", (unsigned long long)bytes); }
    int r = 0;
    if (hipMemsetAsync(a, 1, bytes, c->stream) != hipSuccess) r = fail(TBVH_E_HIP, "hipMemsetAsync failed");
    double ms = 0;
    if (!r) r = timeBest(c, reps ? reps : 3, [&] { launch_stream_copy((const float4*)a, (float4*)b, bytes / 16, c->stream); }, &ms);
    hipFree(a); hipFree(b);
    if (r) return r;
    *gbps = 2.0 * (double)bytes / (ms * 1e-3) / 1e9;
    return 0;
}

int tbvh_measure_read_bandwidth(tbvh_context* c, uint64_t bytes, uint32_t reps, double* gbps) {
    if (!c || !gbps || bytes < (1u << 20)) return fail(TBVH_E_INVALID, "tbvh_measure_read_bandwidth: null argument or under 1 MB");
    if (int r = setDevice(c)) return r;
    void *a = nullptr, *sink = nullptr;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&sink, 256) != hipSuccess) { if (a) hipFree(a); return fail(TBVH_E_NOMEM, "tbvh_measure_read_bandwidth: cannot allocate %llu bytes", (unsigned long long)bytes); }
    int r = 0;
    if (hipMemsetAsync(a, 1, bytes, c->stream) != hipSuccess) r = fail(TBVH_E_HIP, "hipMemsetAsync failed");
    double ms = 0;
    if (!r) r = timeBest(c, reps ? reps : 3, [&] { launch_stream_read((const float4*)a, (float*)sink, bytes / 16, (uint32_t)c->numCUs * 32u, c->stream); }, &ms);
    hipFree(a); hipFree(sink);
    if (r) return r;
    *gbps = (double)bytes / (ms * 1e-3) / 1e9;
    return 0;
}

// VALU issue ceiling of this GPU for the instruction mix of the CWBVH node test (kernels_raygen.hip: k_valu_mix), 8 waves per SIMD:
// wave64 VALU instructions per second over the whole chip, in units of 1e9.
int tbvh_measure_valu_issue(tbvh_context* c, uint32_t reps, double* ginstr_per_s) {
    if (!c || !ginstr_per_s) return fail(TBVH_E_INVALID, "tbvh_measure_valu_issue: null argument");
    if (int r = setDevice(c)) return r;
    const uint32_t blocks = (uint32_t)c->numCUs * 32u;
    const int iters = 20000;
    void* out = nullptr;
    if (hipMalloc(&out, (size_t)blocks * 64 * 4) != hipSuccess) return fail(TBVH_E_NOMEM, "tbvh_measure_valu_issue: out of device memory");
    double ms = 0;
    const int r = timeBest(c, reps ? reps : 3, [&] { launch_valu_mix((float*)out, iters, blocks, c->stream); }, &ms);
    hipFree(out);
    if (r) return r;
    *ginstr_per_s = (double)blocks * iters * 32.0 / (ms * 1e-3) / 1e9;
    return 0;
}

// ---- host builder ------------------------------------------------------------------------

int tbvh_host_build(const void* verts16, uint64_t nTris, int layout, const tbvh_build_params* p, tbvh_hostbvh** out) {
    if (!verts16 || !out || nTris == 0) return fail(TBVH_E_INVALID, "tbvh_host_build: null/empty argument");
    if (nTris > 0x3fffffffull) return fail(TBVH_E_INVALID, "too many triangles");
    if (layout != TBVH_LAYOUT_BVH2_WALD && layout != TBVH_LAYOUT_BVH_GPU && layout != TBVH_LAYOUT_BVH4_GPU && layout != TBVH_LAYOUT_CWBVH)
        return fail(TBVH_E_INVALID, "unknown layout %d", layout);
    tbvh_hostbvh* h = new (std::nothrow) tbvh_hostbvh;
    if (!h) return fail(TBVH_E_NOMEM, "out of host memory");
    h->layout = layout;
    BuildParams bp;
    // CWBVH default: SAH-optimal collapse with a triangle test priced like a node visit, one triangle per BVH2 leaf (the DP
    // forms the leaves).  On the MI355X kernel a triangle test costs about as much as a node visit (the triangle phase runs
    // at ~20 % lane utilisation); against the greedy collapse with 3-triangle leaves: Bistro stand-in camera rays equal,
    // bounce rays +3 % (depth 1) / +6 % (depth 2), 6 % less memory.
    if (layout == TBVH_LAYOUT_CWBVH) { bp.greedyCollapse = false; bp.cPrim = 1.0f; }
    // BVH4_GPU: the same collapse (leaves of <= 4 triangles as the BVH2 builder made them): 1-3 % fewer node visits + triangle
    // tests per ray on both stand-in scenes, measured +1-3 % on the GPU
    if (layout == TBVH_LAYOUT_BVH4_GPU) { bp.greedyCollapse = false; bp.cPrim = 1.0f; }
    if (p) {
        bp.bins = p->bins ? p->bins : 8; bp.threads = p->threads; bp.maxLeafTris = p->max_leaf_tris;
        if (p->flags & TBVH_BUILD_OPTIMAL_COLLAPSE) bp.greedyCollapse = false;
        if (p->flags & TBVH_BUILD_GREEDY_COLLAPSE) bp.greedyCollapse = true;
        if (p->flags >> 8) bp.cPrim = (float)((p->flags >> 8) & 0xffff) * 0.01f;
    }
    if (!bp.maxLeafTris) bp.maxLeafTris = layout == TBVH_LAYOUT_CWBVH ? (bp.greedyCollapse ? 3 : 1) : 4;
    if (layout == TBVH_LAYOUT_CWBVH && bp.maxLeafTris > 3) bp.maxLeafTris = 3;
    try {
        const Vec4* v = (const Vec4*)verts16;
        build_bvh2(v, (uint32_t)nTris, bp, h->bvh2);
        if (layout == TBVH_LAYOUT_BVH_GPU) encode_bvh_gpu(h->bvh2, h->al);
        else if (layout == TBVH_LAYOUT_BVH4_GPU) encode_bvh4_gpu(h->bvh2, v, bp, h->blocksA);
        else if (layout == TBVH_LAYOUT_CWBVH) encode_cwbvh(h->bvh2, v, bp, h->blocksA, h->blocksB);
    } catch (const std::bad_alloc&) {
        delete h;
        return fail(TBVH_E_NOMEM, "out of host memory while building");
    }
    *out = h;
    return 0;
}

int tbvh_host_build_tlas(void* instances192, uint64_t nInst, const float* blasBounds6, uint64_t nBlas, tbvh_hostbvh** out) {
    if (!instances192 || !blasBounds6 || !out || nInst == 0 || nBlas == 0) return fail(TBVH_E_INVALID, "tbvh_host_build_tlas: null/empty argument");
    Instance192* inst = (Instance192*)instances192;
    std::vector<float> boxes(nInst * 6);
    for (uint64_t i = 0; i < nInst; i++) {
        if (inst[i].blasIdx >= nBlas) return fail(TBVH_E_INVALID, "instance %llu: blasIdx %u out of range", (unsigned long long)i, inst[i].blasIdx);
        update_instance(inst[i], blasBounds6 + 6 * (size_t)inst[i].blasIdx);
        for (int a = 0; a < 3; a++) boxes[i * 6 + a] = inst[i].aabbMin[a], boxes[i * 6 + 3 + a] = inst[i].aabbMax[a];
    }
    tbvh_hostbvh* h = new (std::nothrow) tbvh_hostbvh;
    if (!h) return fail(TBVH_E_NOMEM, "out of host memory");
    h->layout = TBVH_LAYOUT_BVH_GPU;
    BuildParams bp; bp.maxLeafTris = 1; bp.threads = 1;
    build_bvh2_boxes(boxes.data(), (uint32_t)nInst, bp, h->bvh2);
    encode_bvh_gpu(h->bvh2, h->al);
    *out = h;
    return 0;
}

// ---- BVH8_CWBVH::Save / Load compatible files (tiny_bvh.h:5786-5820) -----------------------------------------------
// File: u32 header = sub | minor << 8 | major << 16 | layout << 24, u32 triCount, a raw dump of the C++ object
// (sizeof(BVH8_CWBVH) bytes), usedBlocks x 16 bytes of nodes, idxCount x 64 bytes of triangle space (48 used per entry).
// The object dump makes the format specific to one tinybvh version and C++ ABI: the constants below are tinybvh 1.6.7
// built for x86-64 / LP64 (g++ and clang lay the class out identically: Itanium ABI), checked against the real header by
// tests/test_cwbvh_file.py (oracle/ref_shim.cpp: ref_cwbvh_object_layout) and, on every read, against the file length.
namespace {
constexpr uint32_t kCwFileHeader = 7u | (6u << 8) | (1u << 16) | (10u << 24);   // 1.6.7, LAYOUT_CWBVH (tiny_bvh.h:92-94, 788)
constexpr uint32_t kCwObjBytes = 560, kCwOffRefittable = 1, kCwOffLayout = 32, kCwOffTriCount = 44, kCwOffIdxCount = 48,
                   kCwOffCTrav = 52, kCwOffCInt = 56, kCwOffBins = 64, kCwOffAabbMin = 72, kCwOffAabbMax = 84,
                   kCwOffAllocatedBlocks = 128, kCwOffUsedBlocks = 132, kCwOffBvh8IdxCount = 184, kCwOffOwnBvh8 = 552;
struct FileCloser { FILE* f; ~FileCloser() { if (f) fclose(f); } };
}  // namespace

int tbvh_cwbvh_file_write(const char* path, const void* nodes16, uint64_t nNodeBlocks, const void* tris16, uint64_t nTriBlocks,
                          uint64_t nTris, const float* bounds6) {
    if (!path || !nodes16 || !tris16) return fail(TBVH_E_INVALID, "tbvh_cwbvh_file_write: null argument");
    if (nNodeBlocks == 0 || nNodeBlocks % 5 || nTriBlocks % 3 || nNodeBlocks > 0xffffffffull || nTriBlocks / 3 > 0xffffffffull || nTris > 0xffffffffull)
        return fail(TBVH_E_INVALID, "tbvh_cwbvh_file_write: node blocks must be a multiple of 5, triangle blocks of 3, counts 32-bit");
    if (const char* e = validate_cwbvh((const Vec4*)nodes16, nNodeBlocks / 5, nTriBlocks)) return fail(TBVH_E_FORMAT, "tbvh_cwbvh_file_write: %s", e);
    const uint32_t idxCount = (uint32_t)(nTriBlocks / 3), usedBlocks = (uint32_t)nNodeBlocks, triCount = (uint32_t)nTris;
    unsigned char obj[kCwObjBytes];
    std::memset(obj, 0, sizeof obj);                 // pointers, context, the embedded MBVH<8>: all rebuilt by Load
    obj[kCwOffRefittable] = 1;
    auto put32 = [&](uint32_t off, uint32_t v) { std::memcpy(obj + off, &v, 4); };
    auto putf = [&](uint32_t off, float v) { std::memcpy(obj + off, &v, 4); };
    put32(kCwOffLayout, 10u); put32(kCwOffTriCount, triCount); put32(kCwOffIdxCount, idxCount);
    putf(kCwOffCTrav, 1.0f); putf(kCwOffCInt, 1.0f); put32(kCwOffBins, 8u);
    float b[6];
    if (bounds6) std::memcpy(b, bounds6, sizeof b);
    else {   // the root node's own box: origin + 255 quantisation steps of 2^e per axis (a superset of the true bounds)
        const float* n0 = (const float*)nodes16;
        uint32_t ew; std::memcpy(&ew, n0 + 3, 4);
        for (int a = 0; a < 3; a++) { b[a] = n0[a]; b[3 + a] = n0[a] + 255.0f * std::ldexp(1.0f, (int)(int8_t)(ew >> (8 * a))); }
    }
    for (int a = 0; a < 3; a++) { putf(kCwOffAabbMin + 4 * a, b[a]); putf(kCwOffAabbMax + 4 * a, b[3 + a]); }
    put32(kCwOffAllocatedBlocks, usedBlocks); put32(kCwOffUsedBlocks, usedBlocks); put32(kCwOffBvh8IdxCount, idxCount);
    obj[kCwOffOwnBvh8] = 1;
    FileCloser fc{fopen(path, "wb")};
    if (!fc.f) return fail(TBVH_E_INVALID, "tbvh_cwbvh_file_write: cannot open %s", path);
    const uint32_t head[2] = {kCwFileHeader, triCount};
    bool ok = fwrite(head, 4, 2, fc.f) == 2 && fwrite(obj, 1, sizeof obj, fc.f) == sizeof obj &&
              fwrite(nodes16, 16, usedBlocks, fc.f) == usedBlocks;
    // the reference writes idxCount x 4 blocks of triangle space, of which 3 per entry are used (uncompressed triangles)
    ok = ok && fwrite(tris16, 16, (size_t)idxCount * 3, fc.f) == (size_t)idxCount * 3;
    const std::vector<unsigned char> pad(1 << 16, 0);
    for (uint64_t left = (uint64_t)idxCount * 16; ok && left;) {
        const size_t k = (size_t)(left < pad.size() ? left : pad.size());
        ok = fwrite(pad.data(), 1, k, fc.f) == k; left -= k;
    }
    if (!ok) return fail(TBVH_E_INVALID, "tbvh_cwbvh_file_write: short write to %s", path);
    return 0;
}

int tbvh_cwbvh_file_read(const char* path, uint64_t expectedTris, tbvh_hostbvh** out, uint64_t* nTrisOut) {
    if (!path || !out) return fail(TBVH_E_INVALID, "tbvh_cwbvh_file_read: null argument");
    FileCloser fc{fopen(path, "rb")};
    if (!fc.f) return fail(TBVH_E_INVALID, "tbvh_cwbvh_file_read: cannot open %s", path);
    uint32_t head[2];
    unsigned char obj[kCwObjBytes];
    if (fread(head, 4, 2, fc.f) != 2 || fread(obj, 1, sizeof obj, fc.f) != sizeof obj) return fail(TBVH_E_FORMAT, "tbvh_cwbvh_file_read: %s is too short", path);
    // the checks of BVH8_CWBVH::Load (tiny_bvh.h:5806-5812): version, layout, triangle count
    if (head[0] != kCwFileHeader)
        return fail(TBVH_E_FORMAT, "tbvh_cwbvh_file_read: header %08x is not tinybvh 1.6.7 / LAYOUT_CWBVH (%08x)", head[0], kCwFileHeader);
    if (expectedTris && head[1] != expectedTris) return fail(TBVH_E_FORMAT, "tbvh_cwbvh_file_read: file holds %u triangles, expected %llu", head[1], (unsigned long long)expectedTris);
    uint32_t usedBlocks, idxCount;
    std::memcpy(&usedBlocks, obj + kCwOffUsedBlocks, 4); std::memcpy(&idxCount, obj + kCwOffBvh8IdxCount, 4);
    if (fseek(fc.f, 0, SEEK_END)) return fail(TBVH_E_FORMAT, "tbvh_cwbvh_file_read: cannot seek in %s", path);
    const long long len = ftell(fc.f);
    const long long want = 8ll + kCwObjBytes + (long long)usedBlocks * 16 + (long long)idxCount * 64;
    if (len != want || usedBlocks == 0 || usedBlocks % 5)
        return fail(TBVH_E_FORMAT, "tbvh_cwbvh_file_read: %s: length %lld does not match the object dump (usedBlocks %u, idxCount %u): "
                                   "written by a build with a different object layout?", path, len, usedBlocks, idxCount);
    if (fseek(fc.f, 8 + kCwObjBytes, SEEK_SET)) return fail(TBVH_E_FORMAT, "tbvh_cwbvh_file_read: cannot seek in %s", path);
    tbvh_hostbvh* h = new (std::nothrow) tbvh_hostbvh;
    if (!h) return fail(TBVH_E_NOMEM, "out of host memory");
    h->layout = TBVH_LAYOUT_CWBVH;
    try {
        h->blocksA.resize(usedBlocks); h->blocksB.resize((size_t)idxCount * 3);
    } catch (const std::bad_alloc&) { delete h; return fail(TBVH_E_NOMEM, "out of host memory"); }
    if (fread(h->blocksA.data(), 16, usedBlocks, fc.f) != usedBlocks || fread(h->blocksB.data(), 16, (size_t)idxCount * 3, fc.f) != (size_t)idxCount * 3) {
        delete h; return fail(TBVH_E_FORMAT, "tbvh_cwbvh_file_read: short read from %s", path);
    }
    if (const char* e = validate_cwbvh(h->blocksA.data(), usedBlocks / 5, (uint64_t)idxCount * 3)) { delete h; return fail(TBVH_E_FORMAT, "tbvh_cwbvh_file_read: %s", e); }
    if (nTrisOut) *nTrisOut = head[1];
    *out = h;
    return 0;
}

void tbvh_host_free(tbvh_hostbvh* h) { delete h; }
int tbvh_host_layout(const tbvh_hostbvh* h) { return h ? h->layout : TBVH_E_INVALID; }

const void* tbvh_host_blob(const tbvh_hostbvh* h, int which) {
    if (!h) return nullptr;
    switch (h->layout) {
    case TBVH_LAYOUT_BVH2_WALD: return which == 0 ? (const void*)h->bvh2.nodes.data() : which == 1 ? (const void*)h->bvh2.primIdx.data() : nullptr;
    case TBVH_LAYOUT_BVH_GPU: return which == 0 ? (const void*)h->al.data() : which == 1 ? (const void*)h->bvh2.primIdx.data() : which == 2 ? (const void*)h->bvh2.nodes.data() : nullptr;
    case TBVH_LAYOUT_BVH4_GPU: return which == 0 ? (const void*)h->blocksA.data() : which == 2 ? (const void*)h->bvh2.nodes.data() : which == 3 ? (const void*)h->bvh2.primIdx.data() : nullptr;
    case TBVH_LAYOUT_CWBVH: return which == 0 ? (const void*)h->blocksA.data() : which == 1 ? (const void*)h->blocksB.data() : which == 2 ? (const void*)h->bvh2.nodes.data() : which == 3 ? (const void*)h->bvh2.primIdx.data() : nullptr;
    }
    return nullptr;
}
uint64_t tbvh_host_blob_count(const tbvh_hostbvh* h, int which) {
    if (!h) return 0;
    switch (h->layout) {
    case TBVH_LAYOUT_BVH2_WALD: return which == 0 ? h->bvh2.nodes.size() : which == 1 ? h->bvh2.primIdx.size() : 0;
    case TBVH_LAYOUT_BVH_GPU: return which == 0 ? h->al.size() : which == 1 ? h->bvh2.primIdx.size() : which == 2 ? h->bvh2.nodes.size() : 0;
    case TBVH_LAYOUT_BVH4_GPU: return which == 0 ? h->blocksA.size() : which == 2 ? h->bvh2.nodes.size() : which == 3 ? h->bvh2.primIdx.size() : 0;
    case TBVH_LAYOUT_CWBVH: return which == 0 ? h->blocksA.size() : which == 1 ? h->blocksB.size() : which == 2 ? h->bvh2.nodes.size() : which == 3 ? h->bvh2.primIdx.size() : 0;
    }
    return 0;
}

int tbvh_upload_host(tbvh_context* c, const tbvh_hostbvh* h, const void* verts16, uint64_t nTris, tbvh_scene** out) {
    if (!c || !h || !out) return fail(TBVH_E_INVALID, "tbvh_upload_host: null argument");
    switch (h->layout) {
    case TBVH_LAYOUT_BVH_GPU:
        return tbvh_upload_bvh_gpu(c, h->al.data(), h->al.size(), h->bvh2.primIdx.data(), h->bvh2.primIdx.size(), verts16, nTris, out);
    case TBVH_LAYOUT_BVH4_GPU:
        return tbvh_upload_bvh4_gpu(c, h->blocksA.data(), h->blocksA.size(), out);
    case TBVH_LAYOUT_CWBVH:
        return tbvh_upload_cwbvh(c, h->blocksA.data(), h->blocksA.size(), h->blocksB.data(), h->blocksB.size(), out);
    }
    return fail(TBVH_E_INVALID, "layout %d cannot be uploaded", h->layout);
}

}  // extern "C"
