// capi_query.hip — the query path: launchQuery (grid shape, schedule choice, kernel flavors), host-array staging, multi-device sharding, ray generators.
#include "capi_internal.h"

using namespace tbvh;
using namespace tbvh_capi;

namespace tbvh_capi {
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
}  // namespace tbvh_capi

void CohTuner::harvest(float minMs) {
    for (size_t k = 0; k < pending.size();) {
        const Pending pe = pending[k];
        const hipError_t qe = hipEventQuery(pe.e1);
        if (qe == hipErrorNotReady) { (void)hipGetLastError(); k++; continue; }
        float t1 = 0.f;
        if (qe == hipSuccess && hipEventElapsedTime(&t1, pe.e0, pe.e1) == hipSuccess && t1 > minMs) {
            // time per ray depends on the batch size (the tail of a launch): only batches of about one size are compared
            if (!refRays) refRays = pe.rays;
            if (pe.rays * 4 >= refRays * 3 && pe.rays * 3 <= refRays * 4) {
                const float perRay = t1 * 1e6f / (float)pe.rays;
                n[pe.mode - 1]++;
                if (perRay < best[pe.mode - 1]) best[pe.mode - 1] = perRay;
            }
        } else (void)hipGetLastError();
        hipEventDestroy(pe.e0); hipEventDestroy(pe.e1);
        pending.erase(pending.begin() + k);
    }
}

void CohTuner::settle(bool packetOrStrict, float margin) {
    if (n[0] >= kSamples && n[1] >= kSamples && n[2] >= kSamples) {
        int win = 0;   // the deferred + gated schedule unless another one beats it by 3 %
        for (int m = 1; m < kModes; m++) if (best[m] < 0.97f * best[0] && best[m] < best[win]) win = m;
        if (packetOrStrict) {   // strict (= from now on the unprobed single kernel) unless the packet kernel wins by the margin AND by more than the second launch costs
            const float gainMs = (best[1] - best[2]) * (float)refRays * 1e-6f;
            win = (best[2] < margin * best[1] && gainMs >= 0.015f) ? 2 : 1;
        }
        decided = win + 1; drop_pending();
    } else if (launches >= 96) { decided = 1; drop_pending(); }   // batches too varied to compare: the schedule that wins on most scenes
}

int CohTuner::least_sampled() const {
    uint32_t cnt[kModes];
    for (int m = 0; m < kModes; m++) cnt[m] = n[m];
    for (const Pending& pe : pending) cnt[pe.mode - 1]++;
    int least = 0;
    for (int m = 1; m < kModes; m++) if (cnt[m] < cnt[least]) least = m;
    return least;
}

namespace tbvh_capi {

int ensurePipe(tbvh_context* c, uint64_t nHits) {
    if (!c->pipe) {
        // built aside and published only when complete: a failure half way (pinned memory is a scarce resource) must leave the context without a
        // pipe, so that the next host query tries again instead of running on null buffers (HIP_TRY returns; ~HostPipe releases what was made)
        std::unique_ptr<HostPipe> p(new (std::nothrow) HostPipe);
        if (!p) return fail(TBVH_E_NOMEM, "out of host memory");
        for (int i = 0; i < 2; i++) {
            HIP_TRY(hipHostMalloc(&p->pinUp[i], HostPipe::kChunk * 64, hipHostMallocDefault));
            HIP_TRY(hipEventCreateWithFlags(&p->evUp[i], hipEventDisableTiming));
        }
        HIP_TRY(hipStreamCreateWithFlags(&p->down, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&p->evKernel, hipEventDisableTiming));
        // The packing and scattering of the caller's records is host memcpy work: 16.7 M records of a tinybvh::Ray[] pack in 78 / 22 / 17 / 13 ms on
        // 1 / 4 / 8 / 16 threads of the round-5 box (tools/ubench/link_rate.hip) against 19 ms the link needs for them — every core the process
        // may use takes part (up to 16), the calling thread being one of them
        const uint32_t hw = usable_host_threads();
        uint32_t t = hw > 16 ? 15 : hw > 1 ? hw - 1 : 0;
        if (const char* e = getenv("TBVH_HOST_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 64) t = (uint32_t)v - 1; }
        try { p->start(t); }
        catch (const std::exception&) { return fail(TBVH_E_NOMEM, "cannot start the host staging threads"); }   // (~HostPipe joins the ones that did start)
        c->pipe = p.release();
    }
    HostPipe* p = c->pipe;
    if (p->packedCap < nHits) {   // nHits: 20-byte result records the two buffers must hold (two groups of a closest-hit batch, or an any-hit batch's flags)
        HIP_TRY(hipStreamSynchronize(p->down));
        if (p->packed) hipFree(p->packed);
        if (p->pinDown) hipHostFree(p->pinDown);
        p->packed = nullptr; p->pinDown = nullptr; p->packedCap = 0;
        HIP_TRY(hipMalloc((void**)&p->packed, nHits * 20));
        HIP_TRY(hipHostMalloc(&p->pinDown, nHits * 20, hipHostMallocDefault));
        p->packedCap = nHits;
    }
    return 0;
}

static bool isPinned(tbvh_context* c, const void* p, uint64_t bytes) {
    for (const tbvh_context::PinnedRange& r : c->pinned)
        if ((const char*)p >= r.host && (const char*)p + bytes <= r.host + r.bytes) return true;
    return false;
}

// ---- a host ray array through the device, pipelined -----------------------------------------------------------------------------------------
// tiny_bvh_speedtest.cpp:1110-1137 does, one after the other: a memcpy loop (64 of every 128 bytes into a tinyocl::Buffer), CopyToDevice, Kernel::Run,
// CopyFromDevice, and the caller's loop over the results.  Here the batch is cut into groups of ~4 M rays and the stages of consecutive groups
// overlap: while the host threads pack group g + 1 into the pinned upload ring (chunks of 256 k rays, two buffers) and the link carries them up,
// the device traces group g and a second stream packs its 20 result bytes per ray and carries them DOWN (the link is full duplex:
// 48 GB/s each way at once against 57 one way, tools/ubench/link_rate.hip); the host scatters group g - 1's results into the caller's records.
// What was measured and NOT taken (profiles/r05_link_rate.txt): hipMemcpy2D for the strided side (19 GB/s up, 3-4 GB/s down), and the device
// reading / writing the caller's pinned array itself (a 128-byte record costs a 128-byte read for its 64 useful bytes: 27 GB/s; 20-byte writes
// cost a 64-byte line each: 20 GB/s of results).  A packed (64-byte) array in page-locked memory of the library's (tbvh_pinned_malloc) needs no packing: it goes up
// by DMA straight from there.
// Groups of ~1 M rays (round 6; 4 M before): the first group's packing and the last group's scatter overlap with nothing, so the smaller the group
// the shorter the pipeline's fill and drain (~3 ms each of a 34 ms call at 4 M); a 1 M-ray launch still runs at > 3 GRays/s, far above the link.
// The result buffers (device: packed hits, host: pinned) hold TWO groups and are used alternately — group g's results land in one while the host
// scatters group g - 1 out of the other (16.7 M rays: 2 x 21 MB page-locked where the whole batch took 335 MB).
constexpr uint64_t kGroupRays = 1ull << 20;
constexpr uint64_t kDirectRays = 16384;   // batches up to this size skip the pipeline (no worker threads, no pinned ring): two strided copies around the launch

// a small host batch: strided copy up, launch, strided copy of bytes 44..63 back; synchronous
static int hostQuerySmall(tbvh_scene* s, const char* raysIn, char* raysOut, uint64_t n, uint32_t stride, uint8_t* occ) {
    tbvh_context* c = s->ctx;
    if (int r = ensureStage(c, n)) return r;
    if (occ) if (int r = ensureStageOcc(c, n)) return r;
    HIP_TRY(hipMemcpy2DAsync(c->stageRays, 64, raysIn, stride, 64, n, hipMemcpyHostToDevice, c->stream));
    if (int r = launchQuery(s, c->stageRays, n, occ ? c->stageOcc : nullptr)) return r;
    if (occ) HIP_TRY(hipMemcpyAsync(occ, c->stageOcc, n, hipMemcpyDeviceToHost, c->stream));
    else HIP_TRY(hipMemcpy2DAsync(raysOut + 44, stride, (const char*)c->stageRays + 44, 64, 20, n, hipMemcpyDeviceToHost, c->stream));
    return checkStatus(c);   // (synchronizes the stream)
}

int hostQuery(tbvh_scene* s, const char* raysIn, char* raysOut, uint64_t n, uint32_t stride, uint8_t* occ) {
    tbvh_context* c = s->ctx;
    if (n <= kDirectRays) return hostQuerySmall(s, raysIn, raysOut, n, stride, occ);
    if (int r = ensureStage(c, n)) return r;
    if (occ) if (int r = ensureStageOcc(c, n)) return r;
    const uint64_t G = n <= kGroupRays + kGroupRays / 2 ? 1 : (n + kGroupRays - 1) / kGroupRays;
    const uint64_t per = (((n + G - 1) / G) + HostPipe::kChunk - 1) / HostPipe::kChunk * HostPipe::kChunk;
    if (int r = ensurePipe(c, occ ? (n + 19) / 20 : 2 * per)) return r;   // (two groups of 20-byte closest-hit records, or one byte per any-hit flag)
    HostPipe* p = c->pipe;
    const bool direct = stride == 64 && isPinned(c, raysIn, n * 64);
    while (p->evGroup.size() < G) { hipEvent_t e = nullptr; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); p->evGroup.push_back(e); }
    const uint64_t seq0 = c->evSeq;
    uint64_t chunkNo = 0;
    auto drain = [&](uint64_t g) -> int {   // group g's results into the caller's records
        const uint64_t b = g * per, e = b + per < n ? b + per : n;
        HIP_TRY(hipEventSynchronize(p->evGroup[g]));
        const char* pin = (const char*)p->pinDown + (g & 1) * per * 20;
        p->parallel_for(e - b, [=](uint32_t part, uint32_t parts) {
            const uint64_t lo = (e - b) * part / parts, hi = (e - b) * (part + 1) / parts;
            for (uint64_t i = lo; i < hi; i++) std::memcpy(raysOut + (b + i) * stride + 44, pin + i * 20, 20);
        });
        return 0;
    };
    for (uint64_t g = 0; g < G; g++) {
        const uint64_t b = g * per, e = b + per < n ? b + per : n;
        for (uint64_t first = b; first < e; first += HostPipe::kChunk, chunkNo++) {
            const uint64_t cnt = e - first < HostPipe::kChunk ? e - first : HostPipe::kChunk;
            if (direct) { HIP_TRY(hipMemcpyAsync(c->stageRays + first, raysIn + first * 64, cnt * 64, hipMemcpyHostToDevice, c->stream)); continue; }
            const int k = (int)(chunkNo & 1);
            if (chunkNo >= 2) HIP_TRY(hipEventSynchronize(p->evUp[k]));   // the DMA that last read this buffer is done
            char* pin = (char*)p->pinUp[k];
            const char* src = raysIn + first * stride;
            p->parallel_for(cnt, [=](uint32_t part, uint32_t parts) {
                const uint64_t lo = cnt * part / parts, hi = cnt * (part + 1) / parts;
                if (stride == 64) std::memcpy(pin + lo * 64, src + lo * 64, (hi - lo) * 64);
                else for (uint64_t i = lo; i < hi; i++) std::memcpy(pin + i * 64, src + i * stride, 64);
            });
            HIP_TRY(hipMemcpyAsync(c->stageRays + first, pin, cnt * 64, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipEventRecord(p->evUp[k], c->stream));
        }
        if (int r = launchQuery(s, c->stageRays + b, e - b, occ ? c->stageOcc + b : nullptr)) return r;
        if (!occ) {
            // (the half of the result buffers this group uses was last read by drain(g - 2), which returned before this point was reached)
            uint32_t* packed = p->packed + (g & 1) * per * 5;
            HIP_TRY(hipEventRecord(p->evKernel, c->stream));
            HIP_TRY(hipStreamWaitEvent(p->down, p->evKernel, 0));
            launch_pack_hits(c->stageRays + b, packed, e - b, p->down);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync((char*)p->pinDown + (g & 1) * per * 20, packed, (e - b) * 20, hipMemcpyDeviceToHost, p->down));
            HIP_TRY(hipEventRecord(p->evGroup[g], p->down));
            if (g) if (int r = drain(g - 1)) return r;
        }
    }
    if (occ) HIP_TRY(hipMemcpyAsync(p->pinDown, c->stageOcc, n, hipMemcpyDeviceToHost, c->stream));
    else if (int r = drain(G - 1)) return r;
    if (int r = checkStatus(c)) return r;
    if (occ) std::memcpy(occ, p->pinDown, n);
    // the query's device time = the sum over its groups' launches (what tbvh_time_last_ms reports for a host-array query)
    if (!c->skipTiming && c->evSeq - seq0 == G && G <= tbvh_context::kTimeRing) {
        float sum = 0.f; bool ok = true;
        for (uint64_t q = seq0; q < c->evSeq && ok; q++) {
            const uint32_t slot = (uint32_t)(q % tbvh_context::kTimeRing);
            float t = 0.f;
            ok = c->evDone[slot] && hipEventElapsedTime(&t, c->evRing[slot][0], c->evRing[slot][1]) == hipSuccess;
            sum += t;
        }
        if (ok) { c->hostQueryMs = sum; c->hostQuerySeq = c->evSeq; } else (void)hipGetLastError();
    }
    return 0;
}


// The two-level kernels: one per BLAS layout (and one for BVH8_CWBVH and BVH_GPU BLASes mixed), each walking the TLAS form made for it at upload.
static void launchTlasKernels(tbvh_scene* s, QueryArgs& q, bool any, uint32_t blocks) {
    tbvh_context* c = s->ctx;
    const uint32_t blocks7 = (!c->gridOverride && blocks == c->blocks) ? (uint32_t)c->numCUs * 28u : blocks;   // the full grid of the kernels built for 7 waves per SIMD
    // any-hit queries may have a class of their own (capi_scene.hip: reclassifyTlas)
    const bool own = any && s->blasDescAny != nullptr;
    const int layout = own ? s->blasLayoutAny : s->blasLayout;
    const bool mix = own ? s->blasMixCw2Any : s->blasMixCw2;
    const BlasDesc* desc = own ? s->blasDescAny : s->blasDesc;
    if (s->tlas4 && layout == TBVH_LAYOUT_BVH4_GPU) {   // BVH4_GPU BLASes: the unified 4-wide kernel
        q.spillStride = c->spillEntries;   // 32-bit stack entries
        launch_tlas4(any, 0, s->tlas4, s->instances, desc, q, c->status, blocks, c->stream, blocks7);
    } else if (s->tlas8 && (layout == TBVH_LAYOUT_CWBVH || mix)) {   // BVH8_CWBVH BLASes (or those and BVH_GPU ones): the unified 8-wide kernel
        q.spillStride = c->spillEntries / 2;   // 8-byte stack entries
        launch_tlas8(any, 0, s->tlas8, s->tlas8Refs, s->instances, desc, q, c->status, blocks, c->stream, blocks7, mix);
    } else if (layout == TBVH_LAYOUT_BVH_GPU) {   // BVH_GPU BLASes: the TLAS already has their node format (kernels_tlas2.hip)
        q.spillStride = c->spillEntries;
        launch_tlas2(any, 0, s->nodes, s->tlasIdx, s->instances, desc, q, c->status, blocks, c->stream, blocks7);
    } else {
        q.spillStride = c->spillEntries / 2;
        launch_tlas(any, layout, s->nodes, s->tlasIdx, s->instances, desc, q, c->status, blocks, c->stream);
    }
}

// What launchQuery has worked out about a launch on a BVH8_CWBVH scene before the kernels are chosen.
struct CwbvhLaunch {
    bool any;              // IsOccluded
    bool small;            // the scene lives in the L2s (< 48 MB)
    bool probedSmall;      // a scene under 48 MB or beyond 384 MB whose batch is probed for the packet kernel
    int sizeClass;         // which of the scene's tuners the batch belongs to
    uint32_t blocks;       // grid of a coherent batch (a third more waves when probed)
    uint32_t blocksBase;   // grid of an incoherent one
    uint32_t* probeWords;  // the coherence probe's counters in this launch's pool area
};

// One coherent-flavor kernel of a two-flavor launch, in the schedule the scene's tuner asks for; timed by the tuner's own events while it still measures.
static int launchCoherentFlavor(tbvh_scene* s, const QueryArgs& q, const CwbvhLaunch& L, const float4* tris, uint32_t blocks7) {
    tbvh_context* c = s->ctx;
    QueryArgs qa = q;
    qa.baseBlocks = 0;   // every wave of the coherent flavor leaves unless the batch is coherent
    if (c->expFlags & 16u) qa.flags |= 16u;   // (debug flag 16: the coherent flavor takes the batch whatever the probe finds: tests put incoherent rays through it)
    // which schedule serves a coherent batch on this scene is measured, not assumed (CohTuner, capi_internal.h)
    s->cohLastClass[L.any ? 1 : 0] = (uint8_t)L.sizeClass;
    CohTuner& tu = s->cohTuner[L.any ? 1 : 0][L.sizeClass];
    const bool sizeKnown = q.nRaysDev == nullptr;   // a batch whose size only the device knows (the wavefront stages) cannot be priced per ray: the default schedule, no sample
    if (L.probedSmall && !tu.decided && tu.n[0] == 0) { tu.n[0] = CohTuner::kSamples; tu.best[0] = 1e30f; }   // (no deferred schedule on such a scene: strict or packet)
    if (!tu.decided && !c->cohTunerMode) {
        tu.harvest(L.probedSmall ? 0.015f : 0.05f);
        tu.settle(L.probedSmall, L.small ? 0.97f : 0.92f);   // (beyond 384 MB the per-lane flavor measured here walks the packed nodes; a scene with the padded node copy runs a few % faster unprobed)
    }
    const bool measuring = !tu.decided && !c->cohTunerMode && sizeKnown;
    // while undecided: the schedule with the fewest samples taken or in flight (round-robin by launch count aliased with callers whose coherent
    // launches come every third time: one schedule got every sample, the others none, and the tuner idled into its fallback)
    int mode = c->cohTunerMode ? c->cohTunerMode : tu.decided ? tu.decided : measuring ? 1 + tu.least_sampled() : 1;
    if (L.probedSmall && mode == 1) mode = 2;
    if (sizeKnown) tu.launches++;
    if (mode == 2) qa.flags |= 32u;
    CohTuner::Pending pe{nullptr, nullptr, mode, q.nRays};
    if (measuring && tu.pending.size() < 16) {
        if (hipEventCreate(&pe.e0) != hipSuccess || hipEventCreate(&pe.e1) != hipSuccess || hipEventRecord(pe.e0, c->stream) != hipSuccess) {   // no sample then
            if (pe.e0) hipEventDestroy(pe.e0);
            if (pe.e1) hipEventDestroy(pe.e1);
            pe.e0 = pe.e1 = nullptr; (void)hipGetLastError();
        }
    }
    if (mode == 3) launch_cwbvh_packet(L.any, s->nodes, s->tris, qa, c->status, L.blocks, c->stream);   // one traversal per wave of 64 consecutive rays
    else launch_cwbvh(L.any, 0, s->nodes, tris, qa, c->status, mode == 2 ? L.blocksBase : L.blocks, c->stream, 5, L.small, blocks7);
    const hipError_t le = hipGetLastError();
    if (pe.e0) {
        if (le == hipSuccess && hipEventRecord(pe.e1, c->stream) == hipSuccess) tu.pending.push_back(pe);
        else { hipEventDestroy(pe.e0); hipEventDestroy(pe.e1); }
    }
    HIP_TRY(le);
    return 0;
}

// The kernels of one launch on a BVH8_CWBVH scene.  A probed launch on a scene with the incoherent-batch copies (prepareIncoherentCopies) is TWO kernels
// back to back, each for one verdict of the probe; the one the verdict is not for leaves at once (~10 us).  The coherent flavor keeps the packed arrays as
// uploaded (its working set lives in the L2s); the incoherent one walks the hybrid node copy and the 64-byte triangle records.
// Bistro stand-in, 16.7 M rays, interleaved medians (profiles/r03_ab_16m.txt): bounce rays +10 %, camera and shadow rays unchanged.
static int launchCwbvhKernels(tbvh_scene* s, QueryArgs& q, const CwbvhLaunch& L) {
    tbvh_context* c = s->ctx;
    const bool autoPad = s->variant == 0 && s->nodes128 != nullptr;   // nodes beyond the Infinity Cache: the padded copy (padCwbvhIfLarge)
    const uint32_t blocks7 = c->gridOverride ? 0xFFFFFFFFu : (uint32_t)c->numCUs * 28u;
    const float4* tris = s->tris;
#ifdef TBVH_EXPERIMENTS
    if ((c->expFlags & 2u) && s->tris64) { tris = s->tris64; q.flags |= 2u; }   // experiment: 64-byte triangle records in the ordinary kernels too
#endif
    const bool twoFlavors = q.probe && s->variant == 0 && ((!autoPad && s->nodesHy && s->tris64) || L.probedSmall) && !(c->expFlags & 4u);
    if (s->variant == 90 && s->nodesHy && s->tris64) {   // diagnostic: the incoherent flavor whatever the batch (tests, tools/ab_configs.py)
        launch_cwbvh(L.any, 0, s->nodesHy, s->tris64, q, c->status, L.blocksBase, c->stream, 13, L.small, blocks7);
    } else if (s->variant == 91) {   // diagnostic: the coherent flavor (deferred triangles, gated triangle phase) whatever the batch and whatever its probe says
        QueryArgs qa = q;
        qa.probe = L.probeWords; qa.baseBlocks = 0; qa.flags |= 16u;
        launch_cwbvh(L.any, 0, s->nodes, tris, qa, c->status, L.blocks, c->stream, 5, L.small, blocks7);
    } else if (s->variant == 92) {   // diagnostic: one traversal per wave of 64 consecutive rays (kernels_cwbvh_packet.hip) whatever the batch, the scene's size and the tuner
        launch_cwbvh_packet(L.any, s->nodes, s->tris, q, c->status, L.blocks, c->stream);
    } else if (twoFlavors) {
        if (int r = launchCoherentFlavor(s, q, L, tris, blocks7)) return r;
        if (L.probedSmall) {   // behind it: the scene's unprobed kernel, as launched without a probe
            QueryArgs qp = q;
            qp.probe = nullptr; qp.baseBlocks = 0;
            launch_cwbvh(L.any, s->variant, autoPad ? s->nodes128 : s->nodes, tris, qp, c->status, L.blocks, c->stream, autoPad ? 8 : 5, L.small, blocks7);
        } else {
            // the incoherent flavor on 28 one-wave workgroups per CU when the batch fills the grid (24 is the persistent grid's size: 20 / 26 / 28 / 30 /
            // 32 per CU trace bounce rays at -4.4 / +0.6 / +0.9 / +0.5 / +0.5 %, interleaved medians of 13 rounds)
            uint32_t wX = (c->expFlags >> 8) & 0xffu;   // experiment: another number of waves per CU
            if (wX > 32u) wX = 32u;                     // (the spill area holds blocks + blocks / 3 = 32 workgroups per CU: LaneStack strides by gridDim)
            const uint32_t wB = wX ? wX : (c->gridOverride ? 0u : 28u);
            launch_cwbvh(L.any, 0, s->nodesHy, s->tris64, q, c->status, (wB && L.blocksBase == c->blocks) ? (uint32_t)c->numCUs * wB : L.blocksBase, c->stream, 13, L.small, blocks7);
        }
    } else {
        launch_cwbvh(L.any, s->variant, autoPad ? s->nodes128 : s->nodes, tris, q, c->status, L.blocks, c->stream, autoPad ? 8 : 5, L.small, blocks7);
    }
    return 0;
}

int launchQuery(tbvh_scene* s, RayRec* d_rays, uint64_t n, uint8_t* d_occ, bool fresh, float freshTmax, const unsigned long long* nDev) {
    tbvh_context* c = s->ctx;
    TBVH_ENTER(c);
    if (n == 0) return 0;
    s->raysTraced += n;   // (what tbvh_refit weighs the refit of a scene's copies against)
    if (s->pendingCopies || s->blasRecopyPending) countQueryForRecopy(s);   // copies dropped by a tbvh_update_* come back once the blob has settled
    if (!s->wideTried && !s->isTlas && s->variant == 0 && (nDev || n >= 1024u) && (s->layout == TBVH_LAYOUT_BVH_GPU || s->layout == TBVH_LAYOUT_BVH4_GPU)) makeWideCopy(s);   // first query of this scene (not part of its time)
    if (s->wide && s->variant == 0 && !s->wideTlasOnly) return launchQuery(s->wide, d_rays, n, d_occ, fresh, freshTmax, nDev);   // BVH_GPU with an 8-wide copy (capi_scene.hip: makeWideCopy)
    const bool any = d_occ != nullptr;
    if (any && s->isTlas && !s->anyHitSeen) {   // the first IsOccluded through this TLAS (not part of its time): its BVH4_GPU and BVH_GPU BLASes are entered through 8-wide copies by any-hit queries
        s->anyHitSeen = true;
        for (tbvh_scene* b : s->blasList)
            if ((b->layout == TBVH_LAYOUT_BVH4_GPU || b->layout == TBVH_LAYOUT_BVH_GPU) && !b->wideTried && b->variant == 0) makeWideCopy(b);   // (re-classifies the TLASes over b, this one included)
        if (int r = reclassifyTlas(s)) return r;
    }
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
    q.probe = nullptr; q.baseBlocks = 0; q.hybridK = s->hybridK; q.flags = 0u;
#ifdef TBVH_EXPERIMENTS
    q.flags = c->expFlags & 0x3FF0001u;
#endif
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
    const bool probed = !s->isTlas && !small && blobBytes <= (384ull << 20) && s->layout == TBVH_LAYOUT_CWBVH && n >= (1ull << 21) && (s->variant == 0 || s->variant == 88) && !(c->expFlags & 64u);
    if (probed && !s->hyTried) { if (int r = prepareIncoherentCopies(s)) return r; }   // first launch of this class on the scene: the derived copies for incoherent batches (not part of the query's time)
    q.hybridK = s->hybridK;
    HIP_TRY(timedBegin(c));
    // BVH8_CWBVH scenes beyond the L2s (the `small` class runs dense triangle phases, where the gated schedule loses 15 %) but within reach
    // of the Infinity Cache (beyond it camera rays are bound by memory too: 30 M / 60 M triangles lose 11 / 19 % under the gate), batches of 2 M
    // rays and more: a probe of the batch's coherence (256 neighbouring ray pairs, sampled by every wave of the traversal kernel itself: kernels_cwbvh.hip)
    // lets the traversal kernel pick its schedule for the launch; a coherent batch also gets a third more waves (the surplus leaves at once
    // otherwise).  Bistro stand-in, 16.7 M rays: camera rays +4.5 %, shadow rays +6 %, bounce rays unchanged.
    uint32_t blocksBase = blocks;
    // Scenes under 48 MB (round 6; `probedSmall` also covers the scenes beyond 384 MB, below): batches of 768 k rays and more whose size the host knows are probed too — not for the deferred schedule (it loses
    // there) but for the PACKET kernel: one traversal per wave traces the Sponza stand-in's camera rays 1.23 / 1.64 / 1.56 x as fast at 1 / 4.2 / 16.7 M
    // rays, its shadow rays 0.97 / 1.51 / 1.74 x (profiles/r06_small_scene_packet.txt).  Two kernels back to back as on the larger scenes: the coherent
    // flavor the scene's tuner has settled on (strict per-lane, or packet) leaves at once unless the batch is coherent, the unprobed kernel behind it
    // takes what the pool still holds.
    // The two-kernel launch costs a small launch 10-20 us (the probe, the second kernel's start and exit): where the scene's tuner has measured the per-lane
    // kernel faster — the Dragon stand-in: a 16 x 4-pixel chunk of camera rays covers hundreds of its triangles, the packet walk is 5-8 x slower — or only
    // a few us slower, the batch runs as before round 6: ONE unprobed kernel.  Only the measuring launches (6-9 per class of batch size) pay.
    // ... and scenes beyond 384 MB (never probed either: the deferred schedule loses there), from 6 M rays on: the street at 12 M triangles traces 16.7 M camera
    // rays at 7094 instead of 4718 MRays/s with the packet kernel — and 4.2 M at 3383 instead of 3733, shadow rays slower throughout: measured per class likewise.
    const bool huge = !small && blobBytes > (384ull << 20);
    const int sizeClass = (small && n < (3ull << 19)) ? 3 : n < (6ull << 20) ? 0 : n < (12ull << 20) ? 1 : 2;
    bool probedSmall = !s->isTlas && (small || huge) && s->layout == TBVH_LAYOUT_CWBVH && s->variant == 0 && !nDev && n >= (small ? (3ull << 18) : (6ull << 20)) && !(c->expFlags & 64u) && !c->gridOverride;
    if (probedSmall) {
        const int m = c->cohTunerMode ? c->cohTunerMode : s->cohTuner[any ? 1 : 0][sizeClass].decided;
        if (m == 1 || m == 2) { probedSmall = false; s->cohLastClass[any ? 1 : 0] = (uint8_t)sizeClass; }
    }
    if (probedSmall) {
        q.probe = poolArea + (size_t)kPoolParts * kPoolCounterStride; q.baseBlocks = blocks;
        c->lastProbed = true;
    }
    if (probed) {
        uint32_t* probe = poolArea + (size_t)kPoolParts * kPoolCounterStride;
        q.probe = probe; q.baseBlocks = blocks;
        c->lastProbed = true;
        if (!c->gridOverride && blocks == c->blocks) blocks = c->blocks + c->blocks / 3u;   // 24 -> 32 one-wave workgroups per CU
    }
    if (s->isTlas) {
        launchTlasKernels(s, q, any, blocks);
        HIP_TRY(hipGetLastError());
        HIP_TRY(timedEnd(c));
        c->poolCur ^= 1; c->poolClean = true;   // (the other counter area has been zeroed by what was just enqueued)
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
        if (int r = launchCwbvhKernels(s, q, CwbvhLaunch{any, small, probedSmall, sizeClass, blocks, blocksBase, poolArea + (size_t)kPoolParts * kPoolCounterStride})) return r;
        break;
    default:
        return fail(TBVH_E_INVALID, "scene layout %d has no query kernel", s->layout);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(timedEnd(c));
    c->poolCur ^= 1; c->poolClean = true;   // (the other counter area has been zeroed by what was just enqueued)
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
}  // namespace tbvh_capi

extern "C" {

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
    TBVH_ENTER(c);
    return hostQuery(s, (const char*)rays, (char*)rays, n, stride, nullptr);   // pinned, chunked, multi-threaded, pipelined staging (capi_query.hip: hostQuery)
}

int tbvh_occluded(tbvh_scene* s, const void* rays, uint64_t n, uint32_t stride, uint8_t* occ) {
    if (!s || ((!rays || !occ) && n)) return fail(TBVH_E_INVALID, "tbvh_occluded: null argument");
    if (stride < 64 || (stride & 3)) return fail(TBVH_E_INVALID, "stride must be >= 64 and a multiple of 4 (got %u)", stride);
    if (n == 0) return 0;
    tbvh_context* c = s->ctx;
    TBVH_ENTER(c);
    return hostQuery(s, (const char*)rays, nullptr, n, stride, occ);
}

// ---- page-locked host memory FROM the library: the tinyocl::Buffer( bytes ) of this boundary (tiny_ocl.h: a Buffer made without a host pointer owns
// ---- its host side; tiny_bvh_speedtest.cpp:1101-1108 wraps its ray array in one before every GPU block) -----------------------------------------
// Round 5 first page-locked the CALLER's memory (hipHostRegister): with that in the process, later plain hipMemcpy calls from pageable memory that
// had been registered, unregistered, freed and handed out again by the allocator faulted the GPU ("Memory access fault ... Reason: Unknown", 3 runs of the
// GPU suite in 9; 0 in 6 without the registering tests).  The library therefore never registers foreign memory: it hands out memory of its own.
int tbvh_pinned_malloc(tbvh_context* c, uint64_t bytes, void** out) {
    if (!c || !out || !bytes) return fail(TBVH_E_INVALID, "tbvh_pinned_malloc: null/empty argument");
    TBVH_ENTER(c);
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess || !p) { (void)hipGetLastError(); return fail(TBVH_E_NOMEM, "tbvh_pinned_malloc: cannot page-lock %llu bytes", (unsigned long long)bytes); }
    try { c->pinned.push_back(tbvh_context::PinnedRange{(char*)p, bytes}); }
    catch (const std::bad_alloc&) { hipHostFree(p); return fail(TBVH_E_NOMEM, "out of host memory"); }
    *out = p;
    return 0;
}

int tbvh_pinned_free(tbvh_context* c, void* ptr) {
    if (!c || !ptr) return fail(TBVH_E_INVALID, "tbvh_pinned_free: null argument");
    TBVH_ENTER(c);
    for (size_t i = 0; i < c->pinned.size(); i++)
        if (c->pinned[i].host == (char*)ptr) {
            HIP_TRY(hipStreamSynchronize(c->stream));   // nothing of this context may still be reading the range
            c->pinned.erase(c->pinned.begin() + i);
            HIP_TRY(hipHostFree(ptr));
            return 0;
        }
    return fail(TBVH_E_INVALID, "tbvh_pinned_free: %p did not come from tbvh_pinned_malloc of this context", ptr);
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

int tbvh_bin_rays_device(tbvh_context* c, const void* dIn, void* dOut, uint64_t n, const float bounds6[6], uint32_t cellBits, uint32_t flags, uint32_t* dPerm) {
    if (!c || !bounds6 || ((!dIn || !dOut) && n)) return fail(TBVH_E_INVALID, "tbvh_bin_rays_device: null argument");
    if (dIn == dOut) return fail(TBVH_E_INVALID, "tbvh_bin_rays_device: the batch cannot be binned in place");
    if (cellBits > 6 || (flags & ~3u) || (flags & 3u) == 3u) return fail(TBVH_E_INVALID, "tbvh_bin_rays_device: cell_bits 0..6, flags 0, 1 or 2");
    if (n > 0xffffffffull) return fail(TBVH_E_INVALID, "tbvh_bin_rays_device: at most 2^32 - 1 rays per call");
    TBVH_ENTER(c);
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
    HIP_TRY(timedBegin(c));
    HIP_TRY(launch_ray_bin((const RayRec*)dIn, (RayRec*)dOut, dPerm, n, nullptr, a, c->binScratch, scanTemp, (uint32_t)c->numCUs * 16u, c->stream));
    HIP_TRY(timedEnd(c));
    return 0;
}

// ---- ray generators ----------------------------------------------------------------------

int tbvh_generate_primary_device(tbvh_context* c, const tbvh_camera* cam, void* dRays, uint64_t first, uint64_t n) {
    if (!c || !cam || (!dRays && n)) return fail(TBVH_E_INVALID, "tbvh_generate_primary_device: null argument");
    if (cam->width % 4 || cam->height % 4 || !cam->spp_x || !cam->spp_y) return fail(TBVH_E_INVALID, "camera: width/height must be multiples of 4, spp > 0");
    TBVH_ENTER(c);
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
    TBVH_ENTER(c);
    if (!n) return 0;
    TriSource src; src.mode = 0; src.verts = (const float4*)dVerts;
    launch_gen_bounce(src, (const RayRec*)dIn, (RayRec*)dOut, n, seed, c->stream);
    HIP_TRY(hipGetLastError());
    return 0;
}

int tbvh_generate_shadow_device(tbvh_context* c, const void* dIn, void* dOut, uint64_t n, const float light[3], float eps) {
    if (!c || !light || ((!dIn || !dOut) && n)) return fail(TBVH_E_INVALID, "tbvh_generate_shadow_device: null argument");
    TBVH_ENTER(c);
    if (!n) return 0;
    launch_gen_shadow((const RayRec*)dIn, (RayRec*)dOut, n, light[0], light[1], light[2], eps, c->stream);
    HIP_TRY(hipGetLastError());
    return 0;
}

int tbvh_reset_hits_device(tbvh_context* c, void* dRays, uint64_t n, float tmax) {
    if (!c || (!dRays && n)) return fail(TBVH_E_INVALID, "tbvh_reset_hits_device: null argument");
    TBVH_ENTER(c);
    if (!n) return 0;
    launch_reset_hits((RayRec*)dRays, n, tmax, c->stream);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
