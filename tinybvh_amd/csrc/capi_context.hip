// capi_context.hip — contexts, streams, timing, debug knobs, device buffers and the ceilings bench.py measures (include/tinybvh_amd.h).
#include "capi_internal.h"

using namespace tbvh;
using namespace tbvh_capi;

namespace {
thread_local char g_err[512] = "";
}  // namespace

namespace tbvh_capi {
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

int setDevice(tbvh_context* c) {
    HIP_TRY(hipSetDevice(c->device));
    return 0;
}
hipError_t timedBegin(tbvh_context* c) {
    if (c->skipTiming) return hipSuccess;
    const uint32_t slot = (uint32_t)(c->evSeq % tbvh_context::kTimeRing);
    for (int k = 0; k < 2; k++)
        if (!c->evRing[slot][k]) { const hipError_t e = hipEventCreate(&c->evRing[slot][k]); if (e != hipSuccess) return e; }
    c->evDone[slot] = false;
    c->ev0 = c->evRing[slot][0]; c->ev1 = c->evRing[slot][1];
    c->evSeq++;
    return hipEventRecord(c->ev0, c->stream);
}
hipError_t timedEnd(tbvh_context* c) {
    if (c->skipTiming) return hipSuccess;
    const hipError_t e = hipEventRecord(c->ev1, c->stream);
    if (e == hipSuccess && c->evSeq) { c->evDone[(c->evSeq - 1) % tbvh_context::kTimeRing] = true; c->timed = true; }
    return e;
}
}  // namespace tbvh_capi

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
    if (const char* e = getenv("TBVH_EMBED_TRIS")) { if (atoi(e) == 0) c->embedTris = false; }
    if (const char* e = getenv("TBVH_DEBUG_FLAGS")) c->expFlags = (uint32_t)strtoul(e, nullptr, 0);   // tbvh_debug_set_flags from the environment (counter runs of tools/)
    if (const char* e = getenv("TBVH_COHERENT_TUNER")) { const int v = atoi(e); c->cohTunerMode = v == 0 ? 1 : (v == 2 ? 2 : (v == 3 ? 3 : 0)); }   // 0: off (always deferred + gated), 2: always strict, 3: always one traversal per wave
    if (const char* e = getenv("TBVH_POOL_PARTS")) {  // experiment knob
        const int b = atoi(e);
        if (b >= 0 && (1 << b) <= kPoolParts) c->poolParts = (uint32_t)b;   // log2 of the partition count
    }
    c->spillEntries = 232;  // 32-bit entries per lane beyond the LDS part of the stack
    if (const char* e2 = getenv("TBVH_SPILL_ENTRIES")) {   // a SMALLER spill area: how the tests reach the "traversal stack overflow" path with a tree of a few hundred levels
        const int v = atoi(e2);
        if (v >= 2 && v <= 232) c->spillEntries = (uint32_t)v & ~1u;
    }
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
    for (const tbvh_context::PinnedRange& r : c->pinned) hipHostFree(r.host);   // (memory of tbvh_pinned_malloc the caller never gave back goes with the context)
    c->pinned.clear();
    if (c->stageRays) hipFree(c->stageRays);
    if (c->stageOcc) hipFree(c->stageOcc);
    if (c->binScratch) hipFree(c->binScratch);
    delete c->pipe;
    for (auto& pair : c->evRing) for (hipEvent_t ev : pair) if (ev) hipEventDestroy(ev);
    if (c->ownStream) hipStreamDestroy(c->ownStream);
    delete c;
}

int tbvh_synchronize(tbvh_context* c) {
    if (!c) return fail(TBVH_E_INVALID, "null context");
    TBVH_ENTER(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int tbvh_set_stream(tbvh_context* c, void* s) {
    if (!c) return fail(TBVH_E_INVALID, "null context");
    TBVH_LOCK(c);
    c->stream = s ? (hipStream_t)s : c->ownStream;
    return 0;
}

float tbvh_time_last_ms(tbvh_context* c) {
    if (!c) return -1.0f;
    TBVH_LOCK(c);
    if (!c->timed) return -1.0f;
    if (c->hostQuerySeq == c->evSeq && c->hostQueryMs >= 0.f) return c->hostQueryMs;   // a host-array query runs as several launches: their sum
    hipSetDevice(c->device);
    if (hipEventSynchronize(c->ev1) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess) return -1.0f;
    return ms;
}

int tbvh_time_history(tbvh_context* c, float* ms, uint32_t cap, uint32_t* count) {
    if (!c || !count || (!ms && cap)) return fail(TBVH_E_INVALID, "tbvh_time_history: null argument");
    *count = 0;
    TBVH_ENTER(c);
    uint64_t n = c->evSeq < tbvh_context::kTimeRing ? c->evSeq : tbvh_context::kTimeRing;
    if (n > cap) n = cap;
    for (uint64_t i = 0; i < n; i++) {   // oldest first
        const uint32_t slot = (uint32_t)((c->evSeq - n + i) % tbvh_context::kTimeRing);
        float t = -1.0f;
        if (c->evDone[slot]) {
            HIP_TRY(hipEventSynchronize(c->evRing[slot][1]));
            HIP_TRY(hipEventElapsedTime(&t, c->evRing[slot][0], c->evRing[slot][1]));
        }
        ms[i] = t;
    }
    *count = (uint32_t)n;
    return 0;
}

int tbvh_debug_stats(tbvh_context* c, uint64_t out[8], int reset) {
    if (!c || !out) return fail(TBVH_E_INVALID, "tbvh_debug_stats: null argument");
    TBVH_ENTER(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, c->counter + 8, 64, hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(hipMemset(c->counter + 8, 0, 64));
    return 0;
}

int tbvh_set_timing(tbvh_context* c, int enabled) {
    if (!c) return fail(TBVH_E_INVALID, "tbvh_set_timing: null context");
    TBVH_LOCK(c);
    c->skipTiming = enabled == 0;
    return 0;
}

int tbvh_debug_set_flags(tbvh_context* c, uint32_t flags) {
    if (!c) return fail(TBVH_E_INVALID, "tbvh_debug_set_flags: null context");
    TBVH_LOCK(c);
    c->expFlags = flags;
    return 0;
}

int tbvh_debug_last_probe(tbvh_context* c, uint32_t out[3]) {
    if (!c || !out) return fail(TBVH_E_INVALID, "tbvh_debug_last_probe: null argument");
    TBVH_ENTER(c);
    out[0] = out[1] = out[2] = 0;
    if (!c->lastProbed) return 0;
    HIP_TRY(hipStreamSynchronize(c->stream));
    // (the area the last launch drew from: the next launch's kernels will zero it)
    HIP_TRY(hipMemcpy(out, (uint32_t*)c->pool + (size_t)(c->poolCur ^ 1) * ((size_t)(kPoolParts + 1) * kPoolCounterStride) + (size_t)kPoolParts * kPoolCounterStride, 8, hipMemcpyDeviceToHost));
    out[2] = (out[1] != 0 && out[0] * 10u >= out[1] * 6u) ? 2u : 1u;   // the rule of k_cwbvh (kernels_cwbvh.hip)
    return 0;
}

// ---- device buffers ------------------------------------------------------------------------

int tbvh_device_malloc(tbvh_context* c, uint64_t bytes, void** out) {
    if (!c || !out) return fail(TBVH_E_INVALID, "tbvh_device_malloc: null argument");
    TBVH_ENTER(c);
    hipError_t e = hipMalloc(out, bytes ? bytes : 16);
    if (e != hipSuccess) return fail(TBVH_E_NOMEM, "hipMalloc(%llu): %s", (unsigned long long)bytes, hipGetErrorString(e));
    return 0;
}
int tbvh_device_free(tbvh_context* c, void* p) {
    if (!c) return fail(TBVH_E_INVALID, "null context");
    TBVH_ENTER(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipFree(p));
    return 0;
}
int tbvh_copy_to_device(tbvh_context* c, void* d, const void* src, uint64_t bytes) {
    if (!c || ((!d || !src) && bytes)) return fail(TBVH_E_INVALID, "tbvh_copy_to_device: null argument");
    TBVH_ENTER(c);
    HIP_TRY(hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}
int tbvh_copy_from_device(tbvh_context* c, void* dst, const void* d, uint64_t bytes) {
    if (!c || ((!d || !dst) && bytes)) return fail(TBVH_E_INVALID, "tbvh_copy_from_device: null argument");
    TBVH_ENTER(c);
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
    TBVH_ENTER(c);
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
    TBVH_ENTER(c);
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

// Host link as this box delivers it: one pinned buffer, hipMemcpyAsync up and down, best of `reps` each.  The denominator of bench.py's
// detail.host_rays (a host tinybvh::Ray[] costs 64 bytes up and 20 bytes down per ray).
int tbvh_measure_link_bandwidth(tbvh_context* c, uint64_t bytes, uint32_t reps, double* h2d_gbps, double* d2h_gbps) {
    if (!c || !h2d_gbps || !d2h_gbps || bytes < (1u << 20)) return fail(TBVH_E_INVALID, "tbvh_measure_link_bandwidth: null argument or under 1 MB");
    TBVH_ENTER(c);
    void *h = nullptr, *d = nullptr;
    if (hipHostMalloc(&h, bytes, hipHostMallocDefault) != hipSuccess) return fail(TBVH_E_NOMEM, "tbvh_measure_link_bandwidth: cannot pin %llu bytes", (unsigned long long)bytes);
    if (hipMalloc(&d, bytes) != hipSuccess) { hipHostFree(h); return fail(TBVH_E_NOMEM, "tbvh_measure_link_bandwidth: cannot allocate %llu device bytes", (unsigned long long)bytes); }
    memset(h, 1, bytes);
    double up = 0, down = 0;
    int r = timeBest(c, reps ? reps : 3, [&] { (void)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream); }, &up);
    if (!r) r = timeBest(c, reps ? reps : 3, [&] { (void)hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream); }, &down);
    hipFree(d); hipHostFree(h);
    if (r) return r;
    *h2d_gbps = (double)bytes / (up * 1e-3) / 1e9; *d2h_gbps = (double)bytes / (down * 1e-3) / 1e9;
    return 0;
}

// VALU issue ceiling of this GPU for the instruction mix of the CWBVH node test (kernels_raygen.hip: k_valu_mix), 8 waves per SIMD:
// wave64 VALU instructions per second over the whole chip, in units of 1e9.
int tbvh_measure_valu_issue(tbvh_context* c, uint32_t reps, double* ginstr_per_s) {
    if (!c || !ginstr_per_s) return fail(TBVH_E_INVALID, "tbvh_measure_valu_issue: null argument");
    TBVH_ENTER(c);
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

}  // extern "C"
