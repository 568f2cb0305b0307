// capi_wavefront.hip — the device-resident wavefront path tracer (Generate, {Extend, Shade} x depth, Connect) and its multi-device bands.
#include "capi_internal.h"

using namespace tbvh;
using namespace tbvh_capi;

extern "C" {

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
    TBVH_ENTER(c);
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
    TBVH_LOCK(w->ctx);
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
    if (w->ctx->ev0 == w->e0 || w->ctx->ev1 == w->e1) { w->ctx->timed = false; w->ctx->ev0 = w->ctx->ev1 = nullptr; }   // (tbvh_time_last_ms pointed at this frame)
    if (w->e0) hipEventDestroy(w->e0);
    if (w->e1) hipEventDestroy(w->e1);
    delete w;
}

int tbvh_wavefront_set_blas_vertices(tbvh_wavefront* w, const void* const* dVertsPerBlas, uint64_t nBlas) {
    if (!w || !dVertsPerBlas || !nBlas) return fail(TBVH_E_INVALID, "tbvh_wavefront_set_blas_vertices: null/empty argument");
    TBVH_ENTER(w->ctx);
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
    TBVH_ENTER(c);
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
    // the stage loop brackets the FRAME with one event pair (e0 / e1: tbvh_wf_stats::frame_ms, tbvh_time_last_ms), not every query of it: twelve
    // event records fewer per 3-bounce frame, 0.871 -> 0.837 ms at 1280 x 720 (tools/wavefront_small_frame.py)
    struct Untimed { tbvh_context* c; bool was; ~Untimed() { c->skipTiming = was; } } untimed{c, c->skipTiming};
    c->skipTiming = true;
    for (uint32_t d = 0; d < maxDepth; d++) {
        const int nxt = cur ^ 1;
        // Extend: nearest hit of every live path; the batch size lives on the device — except at depth 0, where it is the n camera rays k_wf_generate
        // made: that launch goes out with a size the host knows, so it is probed, sampled by the scene's schedule tuner and traced by the kernel the
        // tuner settles on (the packet kernel on most scenes: capi_query.hip) like any camera batch
        if (int r = launchQuery(scene, w->rays[cur], w->n, nullptr, false, 1e30f, d == 0 ? nullptr : QP(d))) return r;
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
    c->ev0 = w->e0; c->ev1 = w->e1; c->timed = true;   // tbvh_time_last_ms: the frame (tbvh_wavefront_destroy withdraws the pair)
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
    TBVH_LOCK(w->ctx);
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
    // enqueue every band; after a failure stop enqueueing, but still wait for everything that WAS enqueued (the caller — tinyhip::PathTracer,
    // Python — may free or reuse accumulators and vertex buffers as soon as this returns) and report the first error
    int rc = 0;
    std::string firstErr;
    uint32_t launched = 0;
    for (; launched < nDev && !rc; launched++) {
        const uint32_t i = launched;
        const auto t0 = std::chrono::steady_clock::now();
        rc = tbvh_wavefront_render(wfs[i], scenes[i], dVerts ? dVerts[i] : nullptr, cam, p, nullptr);
        if (rc) firstErr = tbvh_last_error();
        if (dispatchMs) dispatchMs[i] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    auto finish = [&](uint32_t i) -> int {
        tbvh_wavefront* w = wfs[i];
        tbvh_context* c = w->ctx;
        TBVH_ENTER(c);
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (int r = checkStatus(c)) return r;
        if (stats && !rc) {
            const uint32_t maxDepth = p->max_depth ? (p->max_depth > 8 ? 8 : p->max_depth) : 3;
            std::vector<unsigned long long> h(kWfCounterWords);
            HIP_TRY(hipMemcpy(h.data(), w->counters, (size_t)kWfCounterWords * 8, hipMemcpyDeviceToHost));
            memset(&stats[i], 0, sizeof stats[i]);
            for (uint32_t d = 0; d < maxDepth; d++) { stats[i].extend_rays[d] = h[32u * d]; stats[i].shadow_rays[d] = h[32u * (9u + d)]; }
            HIP_TRY(hipEventElapsedTime(&stats[i].frame_ms, w->e0, w->e1));
        }
        return 0;
    };
    for (uint32_t i = 0; i < launched; i++) {
        const int r = finish(i);
        if (r && !rc) { rc = r; firstErr = tbvh_last_error(); }
    }
    if (rc) return fail(rc, "tbvh_wavefront_render_sharded: %s", firstErr.c_str());
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
    TBVH_ENTER(w->ctx);
    HIP_TRY(hipStreamSynchronize(w->ctx->stream));
    if (w->blueNoise) { hipFree(w->blueNoise); w->blueNoise = nullptr; }
    if (!table) return 0;
    HIP_TRY(hipMalloc((void**)&w->blueNoise, nWords * 4));
    HIP_TRY(hipMemcpy(w->blueNoise, table, nWords * 4, hipMemcpyHostToDevice));
    return 0;
}

int tbvh_wavefront_read(tbvh_wavefront* w, float* rgba) {
    if (!w || !rgba) return fail(TBVH_E_INVALID, "tbvh_wavefront_read: null argument");
    TBVH_ENTER(w->ctx);
    HIP_TRY(hipMemcpyAsync(rgba, w->accum, w->n * 16, hipMemcpyDeviceToHost, w->ctx->stream));
    HIP_TRY(hipStreamSynchronize(w->ctx->stream));
    return 0;
}

int tbvh_wavefront_finalize(tbvh_wavefront* w, float scale, uint32_t* pixels) {
    if (!w || !pixels) return fail(TBVH_E_INVALID, "tbvh_wavefront_finalize: null argument");
    TBVH_ENTER(w->ctx);
    uint32_t* d = (uint32_t*)w->shadow;   // 4 bytes per pixel in the shadow-ray buffer (64 bytes per pixel, idle between frames)
    launch_wf_finalize(w->accum, scale, d, w->n, w->ctx->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(pixels, d, w->n * 4, hipMemcpyDeviceToHost, w->ctx->stream));
    HIP_TRY(hipStreamSynchronize(w->ctx->stream));
    return 0;
}

}  // extern "C"
