// capi_internal.h — what the translation units behind include/tinybvh_amd.h share: the context and scene objects, the error
// helper, and the few internal entry points that cross files (capi_context / capi_scene / capi_query / capi_wavefront / capi_host).
// Not installed; nothing outside tinybvh_amd/csrc includes it.
#pragma once
#include "../../include/tinybvh_amd.h"
#include "../../include/tinybvh_amd_debug.h"

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
#include <memory>
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

namespace tbvh_capi {
int fail(int code, const char* fmt, ...);   // sets tbvh_last_error() of the calling thread, returns code
}  // namespace tbvh_capi

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) return fail(TBVH_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct tbvh_context {
    // One lock per context: every entry point that touches the context (its stream position, staging buffers, ray-pool counter areas, event
    // ring, scenes) holds it, so host threads may share a context and a scene the way the reference's callers share a BVH
    // (tiny_bvh_speedtest.cpp:1077-1083 runs the const Intersect from 8 threads): calls on ONE context serialise — correct, not concurrent;
    // threads that want their queries to overlap use one context each (include/tiny_hip.h: tinyhip::Scene::ForThread).  Recursive: entry
    // points call each other (tbvh_upload_host -> tbvh_upload_cwbvh, the sharded calls -> the per-device ones).
    std::recursive_mutex mu;
    int device = 0;
    hipStream_t ownStream = nullptr;
    hipStream_t stream = nullptr;
    // HIP-event timing: every timed operation (query, refit, build, ...) takes the next of kTimeRing event pairs, so a caller can enqueue many
    // operations back to back and read all their durations afterwards (tbvh_time_history) instead of synchronizing after each one
    // (tbvh_time_last_ms); ev0 / ev1 = the pair of the most recent operation.
    static constexpr uint32_t kTimeRing = 256;
    hipEvent_t evRing[kTimeRing][2] = {};
    bool evDone[kTimeRing] = {};
    int cohTunerMode = 0;         // TBVH_COHERENT_TUNER: 0 = measure per scene (default), 1 = always the deferred + gated schedule, 2 = always the strict one, 3 = always one traversal per wave
    uint64_t evSeq = 0;           // timed operations begun on this context
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    int numCUs = 0;
    uint32_t blocks = 0;          // persistent grid size (64-thread workgroups)
    uint32_t* spill = nullptr;    // stack spill area
    uint32_t spillEntries = 0;    // 32-bit entries per lane
    unsigned long long* counter = nullptr;  // status word, instrumentation counters
    uint32_t poolParts = 5;   // log2: 32 partitions
    bool embedTris = true;          // TBVH_EMBED_TRIS=0: the hybrid node copy without a triangle in each node's line (A/B: tools/ab_configs.py)
    bool incoherentCopies = true;   // TBVH_INCOHERENT_COPIES=0: no hybrid node copy / 64-byte triangle records (prepareIncoherentCopies)
    uint32_t expFlags = 0;     // tbvh_debug_set_flags: QueryArgs::flags of the next launches (experiments)
    bool skipTiming = false;   // launches enqueued by a stage loop of the library itself (the wavefront frame): no event pair per query
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
    // page-locked host memory handed out by tbvh_pinned_malloc: a packed (64-byte) ray array inside such a range goes up by DMA straight from there,
    // without the packing pass through the library's pinned ring (capi_query.hip: hostQuery)
    struct PinnedRange { char* host; uint64_t bytes; };
    float hostQueryMs = -1.f;     // device time of the most recent host-array query (the sum over its groups' launches) ...
    uint64_t hostQuerySeq = ~0ull; // ... valid while no later operation was timed (evSeq still equals this)
    std::vector<PinnedRange> pinned;
    void* binScratch = nullptr;   // tbvh_bin_rays_device
    size_t binScratchBytes = 0;
    std::vector<tbvh_scene*> scenes;
};

// Which schedule runs COHERENT batches of a two-flavor launch on this scene: the deferred-triangles + gated schedule on a third more waves
// (kernels_cwbvh.hip: PROBED == 3; +1 ... +8 % on camera and shadow rays of the street, foliage, soup stand-ins) or the strict one (PROBED == 4; +10 %
// on camera rays of the atrium generator at 1 M triangles, whose big occluders make every node visited ahead of a pending triangle test wasted work:
// profiles/r04_sensitivity.txt).  No static property of a blob told the two apart, so the library measures: while undecided the launches alternate
// between the two, an event between the two kernels times the first one, and after two coherent samples of each the faster (by 3 %) stays.
struct CohTuner {
    int decided = 0;                    // 0 measuring, 1 deferred + gated, 2 strict, 3 one traversal per wave (kernels_cwbvh_packet.hip)
    bool pinned = false;                // `decided` came from the caller (tbvh_scene_set_schedule_hint): not measured, not reset by a topology change
    uint32_t launches = 0;
    uint64_t refRays = 0;               // size of the first coherent batch sampled: only batches within 3/4 ... 4/3 of it are compared
    static constexpr int kModes = 3;
    uint32_t n[kModes] = {0, 0, 0};             // coherent-verdict samples per schedule
    float best[kModes] = {1e30f, 1e30f, 1e30f};     // ns per ray of the first kernel: best sample per schedule (other work on the GPU only ever ADDS time: the minimum is the robust statistic)
    // a launch being measured: the tuner's OWN event pair around the first kernel (so it measures with tbvh_set_timing(0) as well)
    struct Pending { hipEvent_t e0, e1; int mode; uint64_t rays; };
    std::vector<Pending> pending;
    static constexpr uint32_t kSamples = 3;   // per schedule, before the faster (by 3 %) stays
    void drop_pending() { for (Pending& p : pending) { hipEventDestroy(p.e0); hipEventDestroy(p.e1); } pending.clear(); }
    // (capi_query.hip) the measured launches that have finished since: their times go into n / best; first kernels that left within `minMs` (the probe found
    // the batch incoherent) and batches of another size than the first one sampled are not samples
    void harvest(float minMs);
    // once every schedule has its samples: `decided` is set.  packetOrStrict: the scene's per-lane kernel runs strict whatever (scenes under 48 MB and beyond
    // 384 MB) — the packet kernel must win by `margin` AND by 15 us per launch, which is what the second kernel of a probed launch costs
    void settle(bool packetOrStrict, float margin);
    int least_sampled() const;   // the schedule with the fewest samples taken or in flight
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
    CohTuner cohTuner[2][4];    // [any-hit][batch-size class: < 6 M, < 12 M, more rays; 3 = 768 k .. 1.5 M rays on a scene under 48 MB]: which schedule wins can depend on the batch size (the tail of a launch weighs differently)
    uint8_t cohLastClass[2] = {2, 2};   // the class of the most recent two-flavor launch (tbvh_debug_coherent_schedule reports that one)
    bool hyTried = false;       // the incoherent-batch copies were built, or found impossible / unwanted: launchQuery does not try again
    bool hyLevelOrder = false;  // the node array is in level order (made on the device): the hybrid copy needs no renumbering
    uint32_t nNodes = 0;
    uint64_t nNodeBlocks = 0, nTriBlocks = 0;
    uint64_t capNodeBlocks = 0, capTriBlocks = 0;   // what the allocations hold (tbvh_update_*: a re-converted blob of at most this size goes in place)
    uint64_t topoHash = 0;       // CWBVH: hash of every node's (imask, child base): an update with the same topology keeps the hybrid copy's numbering
    uint64_t bytes = 0;
    // TLAS (layout = BVH_GPU nodes in `nodes`)
    bool isTlas = false;
    uint32_t* tlasIdx = nullptr;
    float4* instances = nullptr;
    BlasDesc* blasDesc = nullptr;
    int blasLayout = 0;
    bool blasMixCw2 = false;          // blasLayout == 0 and every BLAS is BVH8_CWBVH or BVH_GPU: the reference's two BLAS types (traverse_tlas.cl:50-72)
    // any-hit queries may enter the BLASes through other arrays than closest-hit ones (BVH4_GPU BLASes: their own stream for closest hits — k_tlas4 —, their
    // 8-wide copies for IsOccluded — k_tlas8, + 28 % on 1000 instances —: capi_scene.hip: reclassifyTlas); blasDescAny == nullptr: the same as above
    BlasDesc* blasDescAny = nullptr;
    bool anyHitSeen = false;          // the TLAS has had an any-hit query: only then do its BVH4_GPU BLASes get their copies and the second wide tree is kept (a frame loop that only
                                      // ever calls Intersect pays for neither: per frame the second tree's rebuild and the copies' refit cost 0.27 ms at 1000 instances)
    int blasLayoutAny = -1;
    bool blasMixCw2Any = false;
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
    // BVH_GPU / BVH4_GPU: the same tree collapsed 8-wide into the BVH8_CWBVH format (capi_scene.hip: makeWideCopy; made by the first query), kept current by update / refit / micromap
    // calls and traced INSTEAD of `nodes` by the queries on this scene: hit records do not depend on the layout (device_common.h: hit_wins), and the
    // compressed wide kernels trace the same rays 1.6-2.9 x faster than the 2-wide one (profiles/r06_bvh2.txt).  Owned by this scene, not listed in
    // the context's scene table; TLASes over this BLAS enter it through the copies as well (capi_scene.hip: blasView).
    tbvh_scene* wide = nullptr;
    bool wideTried = false;      // the copy was made, or found unwanted / impossible: launchQuery does not try again
    bool wideTlasOnly = false;   // the copy is a small one made for the TLASes over this scene: the scene's own queries keep the uploaded nodes
    // tbvh_update_* (the reference's animation flow: BVH::Refit + ConvertFrom on the host, the blob re-uploaded) DROPS the copies — making them again costs
    // milliseconds, more than a frame's queries gain — and they come back once the scene has answered `recopyAfter` queries without another update; an
    // update that arrives soon after they came back quadruples that number (a blob that keeps changing ends up without copies, a blob updated once has
    // them again after four queries).  tbvh_refit keeps the copies: it refits them in place.
    uint8_t pendingCopies = 0;           // bit 0: the 8-wide copy, bit 1: the 4-wide one — dropped by an update, to be made again
    uint32_t recopyAfter = 4, queriesSinceUpdate = 0;
    bool remadeSinceUpdate = false;
    bool blasRecopyPending = false;      // TLAS: some BLAS has pendingCopies
    // tbvh_refit refits the copies in place (0.3-0.5 ms each for 100 k triangles) — unless fewer than kRefitKeepRays rays were traced through the scene (or the
    // TLASes over it) since the previous refit: then the copies cost a frame more than they save and are dropped like after an update
    uint64_t raysTraced = 0;             // rays of every query launched on this scene (a TLAS counts its own)
    uint64_t raysAtRefit = 0;            // raysTraced of this scene + of the TLASes over it, at the previous tbvh_refit
    bool refitSeen = false;
    // ... and a 4-wide one (BVH4_GPU format) of a BVH_GPU / BVH8_CWBVH BLAS, made when a TLAS is uploaded over it: under a TLAS k_tlas4 is the fastest kernel for
    // closest hits (1000 instances, camera rays: 4650 MRays/s against 4190 through BVH8_CWBVH BLASes and 3840 through BVH_GPU ones), k_tlas8 for any-hit queries
    tbvh_scene* wide4 = nullptr;
    bool wide4Tried = false;
};

struct BLASInstanceCheck { float m[32]; float mn[3]; uint32_t blasIdx; float mx[3]; uint32_t mask; uint32_t pad[8]; };
static_assert(sizeof(BLASInstanceCheck) == 192, "BLASInstance is 192 bytes");

struct tbvh_hostbvh {
    int layout = 0;
    BVH2 bvh2;
    std::vector<NodeAL> al;
    std::vector<Vec4> blocksA, blocksB;
};

struct HostPipe {
    static constexpr uint64_t kChunk = 1ull << 18;   // rays per chunk: 16 MB up, 5 MB down
    void* pinUp[2] = {nullptr, nullptr};
    hipEvent_t evUp[2] = {nullptr, nullptr};
    hipStream_t down = nullptr;   // the results' way back: pack kernel + device-to-host copies, beside the uploads and kernels on the context's stream
    hipEvent_t evKernel = nullptr;
    std::vector<hipEvent_t> evGroup;   // group g's results have landed in pinDown
    uint32_t* packed = nullptr;   // device: 5 dwords per ray (bytes 44..63 of the record)
    void* pinDown = nullptr;      // pinned host: the same, for the whole batch
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
    void parallel_for(uint64_t items, const std::function<void(uint32_t, uint32_t)>& f) {
        if (items < 16384 || workers.empty()) { f(0, 1); return; }   // (a few thousand records: waking the workers costs more than the copy)
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
            if (evUp[i]) hipEventDestroy(evUp[i]);
        }
        for (hipEvent_t e : evGroup) hipEventDestroy(e);
        if (evKernel) hipEventDestroy(evKernel);
        if (down) hipStreamDestroy(down);
        if (pinDown) hipHostFree(pinDown);
        if (packed) hipFree(packed);
    }
};

namespace tbvh_capi {
int setDevice(tbvh_context* c);
// first statement of an entry point once its arguments are known to be non-null: take the context's lock for the rest of the call, make its device current
#define TBVH_LOCK(ctxptr) std::lock_guard<std::recursive_mutex> ctx_lock_((ctxptr)->mu)
#define TBVH_ENTER(ctxptr) std::lock_guard<std::recursive_mutex> ctx_lock_((ctxptr)->mu); if (int r_ = tbvh_capi::setDevice(ctxptr)) return r_
hipError_t timedBegin(tbvh_context* c);   // next event pair of the ring, start event recorded on the context's stream
hipError_t timedEnd(tbvh_context* c);     // end event recorded; the operation counts as timed
// one query launch (probe-in-kernel + traversal kernel(s)) on the context's stream; asynchronous.  nDev: batch size in device memory (wavefront queues)
int launchQuery(tbvh_scene* s, tbvh::RayRec* d_rays, uint64_t n, uint8_t* d_occ, bool fresh = false, float freshTmax = 1e30f,
                const unsigned long long* nDev = nullptr);
int checkStatus(tbvh_context* c);   // synchronizes the stream, turns the device status word into an error code
tbvh_scene* newScene(tbvh_context* c, int layout);
uint64_t cwbvhTopologyHash(const tbvh::Vec4* nodes, uint32_t nNodes);
int padCwbvhIfLarge(tbvh_scene* s);
size_t hybridBytes(uint32_t nNodes, uint32_t K);
bool wantsIncoherentCopies(const tbvh_scene* s);
int prepareIncoherentCopies(tbvh_scene* s);
void freeWideCopy(tbvh_scene* s);
void freeWide4Copy(tbvh_scene* s);
int makeWide4Copy(tbvh_scene* s);   // the 4-wide copy of a BVH_GPU / BVH8_CWBVH BLAS (closest-hit queries of the TLASes over it)
int reclassifyTlas(tbvh_scene* t);
void dropCopiesAfterUpdate(tbvh_scene* s);   // (capi_scene.hip) tbvh_update_*: see tbvh_scene::pendingCopies
void countQueryForRecopy(tbvh_scene* s);     // ... and the query side of it (launchQuery)   // (capi_scene.hip) descriptors, kernel class and wide trees of a TLAS from its BLASes as they are now
int makeWideCopy(tbvh_scene* s);   // (lazily, from launchQuery) the 8-wide copy of a BVH_GPU / BVH4_GPU scene   // (lazily, from launchQuery) hybrid node copy + 64-byte triangle records for incoherent batches
}  // namespace tbvh_capi
