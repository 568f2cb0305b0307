// capi_scene.hip — scenes: uploads and their validation, derived copies, device conversion / build / refit, TLAS upload and rebuild, opacity maps.
#include "capi_internal.h"

using namespace tbvh;
using namespace tbvh_capi;

namespace tbvh_capi {
// A BVH8_CWBVH scene whose node array is larger than twice the 256 MB Infinity Cache is traversed through a copy with one node per
// 128-byte line: an 80-byte node straddles 1.6 lines on average, and once the lines come from HBM that is 17 % more traffic than the
// 60 % larger array costs (tools/size_sweep.py, 60 M triangles: bounce rays +6 %; below that size the smaller footprint wins).
int padCwbvhIfLarge(tbvh_scene* s) {
    if (s->layout != TBVH_LAYOUT_CWBVH || s->isTlas || s->nodes128 || (uint64_t)s->nNodes * 80 < (512ull << 20)) return 0;
    if ((uint64_t)s->nNodes * 8 >> 32) return 0;   // (cw_load_node addresses float4s with 32 bits: beyond 2^29 nodes — 64 GB padded — the packed array serves)
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
// first kHybridPacked packed and the others one per 128-byte line (each with one of its triangles in the line's spare 48 bytes), and the
// triangle records padded to 64 bytes.  Built LAZILY by the first launch that would use them (launchQuery: a batch of 2 M rays or more) — a
// scene that is only ever a BLAS under a TLAS, or only traced with small batches, never pays the 2.3 x memory and the host pass; that first
// launch waits for the build (~0.1 s for 600 k nodes: the node array is read back, ordered on the host, scattered on the device).  Trees made
// on the device (tbvh_convert_bvh2_device, tbvh_build_device) are in level order, which already is close to priority order: no renumbering.
// A blob that is not a strict tree (cwbvh_priority_order), one with 2^27 triangle records or more, or a failed allocation is not an error:
// the scene then runs the one-kernel path.  TBVH_INCOHERENT_COPIES=0 turns the copies off.
constexpr uint32_t kHybridPacked = 8192;
bool wantsIncoherentCopies(const tbvh_scene* s) {
    const uint64_t blobBytes = (s->nNodeBlocks + s->nTriBlocks) * 16;
    return s->layout == TBVH_LAYOUT_CWBVH && !s->isTlas && s->ctx->incoherentCopies && blobBytes >= (48ull << 20) && blobBytes <= (384ull << 20) && s->nNodes > kHybridPacked &&
           s->nTriBlocks != 0 && s->nTriBlocks / 3 < (1ull << 27);
}
int prepareIncoherentCopies(tbvh_scene* s) {
    tbvh_context* c = s->ctx;
    s->hyTried = true;
    if (!wantsIncoherentCopies(s)) return 0;
    const uint32_t K = kHybridPacked;
    const uint64_t nT = s->nTriBlocks / 3;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!s->hyLevelOrder && !s->hyPerm) {
        std::vector<Vec4> host((size_t)s->nNodes * 5);
        HIP_TRY(hipMemcpy(host.data(), s->nodes, host.size() * 16, hipMemcpyDeviceToHost));
        std::vector<uint32_t> perm;
        if (!cwbvh_priority_order(host.data(), s->nNodes, perm)) return 0;   // not a strict tree: traversed as uploaded
        if (hipMalloc((void**)&s->hyPerm, (size_t)s->nNodes * 4) != hipSuccess) { s->hyPerm = nullptr; (void)hipGetLastError(); return 0; }
        HIP_TRY(hipMemcpy(s->hyPerm, perm.data(), (size_t)s->nNodes * 4, hipMemcpyHostToDevice));
    }
    if (!s->nodesHy && hipMalloc((void**)&s->nodesHy, hybridBytes(s->nNodes, K)) != hipSuccess) { s->nodesHy = nullptr; (void)hipGetLastError(); return 0; }
    if (!s->tris64 && hipMalloc((void**)&s->tris64, nT * 64) != hipSuccess) { s->tris64 = nullptr; (void)hipGetLastError(); hipFree(s->nodesHy); s->nodesHy = nullptr; return 0; }
    s->hybridK = K;
    HIP_TRY(hipMemsetAsync(s->nodesHy, 0, hybridBytes(s->nNodes, K), c->stream));
    launch_cwbvh_derive_hybrid(s->nodes, s->hyPerm, s->nodesHy, s->nNodes, K, (c->embedTris && !(c->expFlags & 8u)) ? s->tris : nullptr, c->stream);
    launch_cwbvh_pad_tris(s->tris, s->tris64, nT, c->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    s->bytes += hybridBytes(s->nNodes, K) + nT * 64;
    return 0;
}

// order-dependent hash of what the hybrid copy's numbering depends on: which slots of every node are interior children and where they start
uint64_t cwbvhTopologyHash(const Vec4* nodes, uint32_t nNodes) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ nNodes;
    for (uint32_t i = 0; i < nNodes; i++) {
        uint32_t w[2];
        std::memcpy(&w[0], &nodes[(size_t)i * 5].w, 4); std::memcpy(&w[1], &nodes[(size_t)i * 5 + 1].x, 4);
        const uint64_t k = ((uint64_t)(w[0] >> 24) << 32) | ((w[0] >> 24) ? w[1] : 0u);
        h = (h ^ k) * 0x100000001B3ull; h ^= h >> 29;
    }
    return h;
}

tbvh_scene* newScene(tbvh_context* c, int layout) {
    tbvh_scene* s = new (std::nothrow) tbvh_scene;
    if (!s) return nullptr;
    s->ctx = c; s->layout = layout;
    c->scenes.push_back(s);
    return s;
}

constexpr uint64_t kWideCopyMin = 32768;   // blob entries from which a scene's own queries go through its 8-wide copy (TBVH_WIDE_COPY_MIN)

void freeWideCopy(tbvh_scene* s) {
    if (!s || !s->wide) return;
    tbvh_scene* w = s->wide;
    s->wide = nullptr;
    s->bytes -= w->bytes < s->bytes ? w->bytes : 0;
    w->opmap = nullptr; w->opmapBytes = 0;   // (shared with the owner, never owned)
    tbvh_free_scene(w);
}

void freeWide4Copy(tbvh_scene* s) {
    if (!s || !s->wide4) return;
    tbvh_scene* w = s->wide4;
    s->wide4 = nullptr;
    s->bytes -= w->bytes < s->bytes ? w->bytes : 0;
    w->opmap = nullptr; w->opmapBytes = 0;   // (shared with the owner, never owned)
    tbvh_free_scene(w);
}

// The 8-wide copy of a BVH_GPU / BVH4_GPU scene (tbvh_scene::wide), made LAZILY by the scene's first query of 1024 rays or more (launchQuery) — a BLAS
// that is only ever traced through a TLAS never pays for it — from what the scene keeps on the device: the blob is read back, the host turns it into a
// Wald-layout BVH2 with leaves of at most 3 entries (host_builder.cpp: bvh_gpu_to_bvh2 in record mode / bvh4_gpu_to_bvh2), the device converter every
// BVH8_CWBVH conversion uses collapses and encodes it in ITS record mode (kernels_convert.hip; the greedy collapse of MBVH<8>::ConvertFrom,
// tiny_bvh.h:4975-5048): triangle records are carried over bit for bit.  Blobs below TBVH_WIDE_COPY_MIN entries / triangles (default 32768; 0 = never)
// keep their own kernel.  A failure here is never an error of the query: the scene then simply traces its own nodes.
static int convertDeviceImpl(tbvh_context* c, int layout, const float4* dN2, uint64_t nNodes2, const uint32_t* dIdx, uint64_t nIdx, const float4* dV, uint64_t nTris, tbvh_scene** out);
// One copy of scene s in the `target` layout (BVH8_CWBVH from a BVH_GPU / BVH4_GPU scene, BVH4_GPU from a BVH_GPU / BVH8_CWBVH one), or nullptr (too small,
// too large, out of memory: never an error of the caller's operation).  Not listed in the context's scene table; shares the owner's opacity maps.
// forTlas: the copy is wanted by a TLAS over s — there ONE kernel class for all BLASes is worth more than any single BLAS's speed (a BLAS without the copy
// puts the whole TLAS on the flat loop), so small blobs get one too (from 64 entries; TBVH_WIDE_COPY_MIN still rules when set).
static tbvh_scene* buildCopy(tbvh_scene* s, int target, bool forTlas) {
    tbvh_context* c = s->ctx;
    uint64_t minIdx = forTlas ? 64 : kWideCopyMin;
    if (const char* e = getenv("TBVH_WIDE_COPY_MIN")) { const long long v = atoll(e); minIdx = v <= 0 ? ~0ull : (uint64_t)v; }
    std::vector<Node2> n2;
    std::vector<Vec4> blob, recs;
    const float4* dRecs = nullptr;
    uint64_t nRecs = 0;
    struct Tmp { void *n2 = nullptr, *r = nullptr; ~Tmp() { if (n2) hipFree(n2); if (r) hipFree(r); } } t;
    try {
        if (s->layout == TBVH_LAYOUT_BVH_GPU) {
            const uint64_t nNodes = s->nNodeBlocks / 4, nIdx = s->nTriBlocks / 3;
            if (nIdx < minIdx || nIdx > 0x7fffffffull || nNodes > 0x3fffffffull) return nullptr;
            blob.resize(s->nNodeBlocks); recs.resize(s->nTriBlocks);
            if (hipMemcpyAsync(blob.data(), s->nodes, s->nNodeBlocks * 16, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                hipMemcpyAsync(recs.data(), s->tris, s->nTriBlocks * 16, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                hipStreamSynchronize(c->stream) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            if (!bvh_gpu_to_bvh2((const NodeAL*)blob.data(), nNodes, nullptr, nIdx, nullptr, 0, 3u, n2, recs.data())) return nullptr;
            dRecs = s->tris; nRecs = nIdx;       // (the gathered records are on the device already, in leaf order)
        } else {
            if (s->layout == TBVH_LAYOUT_BVH4_GPU) {
                if (s->nNodeBlocks / 4 < minIdx || s->nNodeBlocks > 0x7fffffffull) return nullptr;   // (a stream of n triangles has at least 3 n blocks: a cheap first cut)
                blob.resize(s->nNodeBlocks);
                if (hipMemcpyAsync(blob.data(), s->nodes, s->nNodeBlocks * 16, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                    hipStreamSynchronize(c->stream) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
                if (!bvh4_gpu_to_bvh2(blob.data(), s->nNodeBlocks, 3u, n2, recs)) return nullptr;
            } else {   // BVH8_CWBVH
                if (s->nTriBlocks / 3 < minIdx || s->nTriBlocks > 0x7fffffffull) return nullptr;
                std::vector<Vec4> tris(s->nTriBlocks);
                blob.resize(s->nNodeBlocks);
                if (hipMemcpyAsync(blob.data(), s->nodes, s->nNodeBlocks * 16, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                    hipMemcpyAsync(tris.data(), s->tris, s->nTriBlocks * 16, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                    hipStreamSynchronize(c->stream) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
                if (!cwbvh_to_bvh2(blob.data(), s->nNodeBlocks / 5, tris.data(), s->nTriBlocks, n2, recs)) return nullptr;
            }
            nRecs = recs.size() / 3;
            if (nRecs < minIdx || nRecs > 0x7fffffffull) return nullptr;
            if (hipMalloc(&t.r, recs.size() * 16) != hipSuccess ||
                hipMemcpyAsync(t.r, recs.data(), recs.size() * 16, hipMemcpyHostToDevice, c->stream) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            dRecs = (const float4*)t.r;
        }
    } catch (const std::bad_alloc&) { return nullptr; }
    if (n2.size() > 0x7fffffffull) return nullptr;
    if (hipMalloc(&t.n2, n2.size() * 32) != hipSuccess ||
        hipMemcpyAsync(t.n2, n2.data(), n2.size() * 32, hipMemcpyHostToDevice, c->stream) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    tbvh_scene* w = nullptr;
    if (convertDeviceImpl(c, target, (const float4*)t.n2, n2.size(), nullptr, nRecs, dRecs, nRecs, &w) != 0 || !w) { (void)hipGetLastError(); return nullptr; }
    for (size_t i = 0; i < c->scenes.size(); i++)
        if (c->scenes[i] == w) { c->scenes.erase(c->scenes.begin() + i); break; }   // owned by `s`, freed with it
    w->opmap = s->opmap; w->opmapN = s->opmapN;
    return w;
}

static int makeWideCopyImpl(tbvh_scene* s) {
    freeWideCopy(s);
    s->wideTried = true;
    if (s->isTlas || (s->layout != TBVH_LAYOUT_BVH_GPU && s->layout != TBVH_LAYOUT_BVH4_GPU)) return 0;
    if (tbvh_scene* w = buildCopy(s, TBVH_LAYOUT_CWBVH, !s->usedBy.empty())) {
        s->wide = w; s->bytes += w->bytes;
        // a copy below the size at which the scene's OWN queries gain from it (made for the TLASes over the scene): those queries keep the uploaded nodes
        const uint64_t entries = s->layout == TBVH_LAYOUT_BVH_GPU ? s->nTriBlocks / 3 : w->nTriBlocks / 3;
        s->wideTlasOnly = entries < kWideCopyMin && !getenv("TBVH_WIDE_COPY_MIN");
    }
    return 0;
}

static int makeWide4CopyImpl(tbvh_scene* s) {
    freeWide4Copy(s);
    s->wide4Tried = true;
    if (s->isTlas || (s->layout != TBVH_LAYOUT_BVH_GPU && s->layout != TBVH_LAYOUT_CWBVH)) return 0;
    if (tbvh_scene* w = buildCopy(s, TBVH_LAYOUT_BVH4_GPU, true)) { s->wide4 = w; s->bytes += w->bytes; }
    return 0;
}
}  // namespace tbvh_capi

extern "C" {

// ---- uploads ---------------------------------------------------------------------------

int tbvh_upload_bvh_gpu(tbvh_context* c, const void* nodes64, uint64_t nNodes, const uint32_t* primIdx, uint64_t nIdx,
                        const void* verts16, uint64_t nTris, tbvh_scene** out) {
    if (!c || !nodes64 || !primIdx || !verts16 || !out || nNodes == 0) return fail(TBVH_E_INVALID, "tbvh_upload_bvh_gpu: null/empty argument");
    if (const char* why = validate_bvh_gpu((const NodeAL*)nodes64, nNodes, nIdx)) return fail(why == kValidateNoMemory ? TBVH_E_NOMEM : TBVH_E_FORMAT, "%s", why);
    TBVH_ENTER(c);
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
    s->capNodeBlocks = s->nNodeBlocks; s->capTriBlocks = s->nTriBlocks;
    s->bytes = nNodes * 64 + nIdx * 48;
    *out = s;
    return 0;
}

int tbvh_upload_bvh4_gpu(tbvh_context* c, const void* blocks16, uint64_t nBlocks, tbvh_scene** out) {
    if (!c || !blocks16 || !out || nBlocks < 4) return fail(TBVH_E_INVALID, "tbvh_upload_bvh4_gpu: null/empty argument");
    if (const char* why = validate_bvh4_gpu((const Vec4*)blocks16, nBlocks)) return fail(why == kValidateNoMemory ? TBVH_E_NOMEM : TBVH_E_FORMAT, "%s", why);
    TBVH_ENTER(c);
    tbvh_scene* s = newScene(c, TBVH_LAYOUT_BVH4_GPU);
    if (!s) return fail(TBVH_E_NOMEM, "out of host memory");
    hipError_t e = hipMalloc((void**)&s->nodes, nBlocks * 16);
    if (e == hipSuccess) e = hipMemcpyAsync(s->nodes, blocks16, nBlocks * 16, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { tbvh_free_scene(s); return fail(TBVH_E_HIP, "BVH4_GPU upload failed: %s", hipGetErrorString(e)); }
    s->nNodeBlocks = nBlocks; s->capNodeBlocks = nBlocks; s->bytes = nBlocks * 16;
    *out = s;
    return 0;
}

int tbvh_upload_cwbvh(tbvh_context* c, const void* nodes16, uint64_t nNodeBlocks, const void* tris16, uint64_t nTriBlocks,
                      tbvh_scene** out) {
    if (!c || !nodes16 || !out || nNodeBlocks < 5 || (nTriBlocks && !tris16)) return fail(TBVH_E_INVALID, "tbvh_upload_cwbvh: null/empty argument");
    if (nNodeBlocks % 5) return fail(TBVH_E_FORMAT, "CWBVH node blocks (%llu) not a multiple of 5", (unsigned long long)nNodeBlocks);
    if (nNodeBlocks >> 32) return fail(TBVH_E_FORMAT, "CWBVH node blocks (%llu) beyond the layout's 32-bit block index", (unsigned long long)nNodeBlocks);   // (cw_load_node: 32-bit float4 offsets)
    if (const char* why = validate_cwbvh((const Vec4*)nodes16, nNodeBlocks / 5, nTriBlocks)) return fail(why == kValidateNoMemory ? TBVH_E_NOMEM : TBVH_E_FORMAT, "%s", why);
    TBVH_ENTER(c);
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
    s->capNodeBlocks = nNodeBlocks; s->capTriBlocks = nTriBlocks ? nTriBlocks : 1;
    s->topoHash = cwbvhTopologyHash((const Vec4*)nodes16, s->nNodes);
    s->bytes = (nNodeBlocks + nTriBlocks) * 16;
    if (int r = padCwbvhIfLarge(s)) { tbvh_free_scene(s); return r; }
    *out = s;
    return 0;
}

namespace {
// (re)build the wide TLAS(es) from the BVH_GPU nodes on the device — 8-wide in the BVH8_CWBVH node format, 4-wide in the BVH4_GPU one, whichever the
// two-level kernels of this TLAS's closest-hit and any-hit queries walk; asynchronous on the context's stream
int buildTlasWide8(tbvh_scene* s) {
    tbvh_context* c = s->ctx;
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

int buildTlasWide4(tbvh_scene* s) {
    tbvh_context* c = s->ctx;
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

int buildTlas4(tbvh_scene* s) {
    const bool any2 = s->blasDescAny != nullptr;
    const bool want8 = s->blasLayout == TBVH_LAYOUT_CWBVH || s->blasMixCw2 || (any2 && (s->blasLayoutAny == TBVH_LAYOUT_CWBVH || s->blasMixCw2Any));
    const bool want4 = s->blasLayout == TBVH_LAYOUT_BVH4_GPU || (any2 && s->blasLayoutAny == TBVH_LAYOUT_BVH4_GPU);
    if (want8) if (int r = buildTlasWide8(s)) return r;   // (the two builds share the scratch area: in order on one stream)
    if (want4) if (int r = buildTlasWide4(s)) return r;
    return 0;
}

int tlasCopy(tbvh_scene* s, const void* nodes64, uint64_t nNodes, const uint32_t* idx, uint64_t nIdx, const void* inst, uint64_t nInst) {
    tbvh_context* c = s->ctx;
    // same hardening as the BLAS uploads: the TLAS kernels index instances[idx[]] and blas[blasIdx] unguarded
    if (const char* why = validate_bvh_gpu((const NodeAL*)nodes64, nNodes, nIdx)) return fail(why == kValidateNoMemory ? TBVH_E_NOMEM : TBVH_E_FORMAT, "TLAS: %s", why);
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

extern "C++" {
namespace tbvh_capi {
// What a TLAS traverses for BLAS b, by the kind of query (round 6; 1000 instances of a 100 k-triangle BLAS, camera / shadow / random MRays/s in DESIGN.md par. 3.5):
// closest hits through a 4-wide stream — a BVH4_GPU BLAS's own, the 4-wide copy of a BVH_GPU / BVH8_CWBVH one (k_tlas4 is the fastest two-level kernel for
// closest hits) —, any-hit queries through 8-wide nodes — a BVH8_CWBVH BLAS's own, the 8-wide copy of the others (k_tlas8 is the fastest there).  A forced
// variant on the BLAS (tbvh_set_variant) pins the uploaded nodes; allow4 = false: the closest-hit view without the 4-wide copies (reclassifyTlas's fallback).
static const tbvh_scene* blasView(const tbvh_scene* b, bool any, bool allow4 = true) {
    if (b->variant != 0) return b;
    if (!any && allow4 && b->wide4 && (b->layout == TBVH_LAYOUT_BVH_GPU || b->layout == TBVH_LAYOUT_CWBVH)) return b->wide4;   // closest hits: the 4-wide kernel
    const bool viaCopy = b->wide && (b->layout == TBVH_LAYOUT_BVH_GPU || (any && b->layout == TBVH_LAYOUT_BVH4_GPU));
    return viaCopy ? b->wide : b;
}

// The BLAS descriptors of TLAS t and the class of two-level kernel that serves each kind of query, from its BLASes as they are NOW (their copies come and
// go: a tbvh_update_* drops them, queries bring them back, tbvh_set_variant switches between copy and nodes); builds the wide TLAS(es) those kernels walk.
// With every BLAS copied, BLASes of different layouts under one TLAS share one kernel class per kind of query instead of the flat three-state loop.
int reclassifyTlas(tbvh_scene* t) {
    const size_t nBlas = t->blasList.size();
    std::vector<BlasDesc> desc[2] = {std::vector<BlasDesc>(nBlas), std::vector<BlasDesc>(nBlas)};
    int layout[2] = {0, 0};
    bool mix[2] = {false, false}, same = true;
    for (int pass = 0; pass < 3; pass++) {
        // pass 0: closest hits, BVH_GPU / BVH8_CWBVH BLASes through their 4-wide copies; pass 1 (only if pass 0 ended in the flat loop: a BLAS too small
        // for a copy next to copied ones): closest hits without the 4-wide copies; pass 2: any-hit queries
        const int any = pass == 2;
        if (pass == 1 && (layout[0] != 0 || mix[0])) continue;
        bool anyBvh4 = false;
        for (size_t i = 0; i < nBlas; i++) {
            const tbvh_scene* b = t->blasList[i];
            const tbvh_scene* v = blasView(b, any != 0, pass == 0);
            layout[any] = i == 0 ? v->layout : (layout[any] == v->layout ? layout[any] : 0);   // 0: the BLASes mix layouts (traverse_tlas.cl:50-72)
            anyBvh4 |= v->layout == TBVH_LAYOUT_BVH4_GPU;
            desc[any][i].nodes = v->nodes; desc[any][i].tris = v->tris; desc[any][i].opmap = b->opmap; desc[any][i].opmapN = b->opmapN; desc[any][i].layout = (uint32_t)v->layout;
            if (any) same &= desc[1][i].nodes == desc[0][i].nodes;
        }
        mix[any] = layout[any] == 0 && !anyBvh4;
    }
    // a class of its own for any-hit queries only where it is served by the 8-wide kernel (some BVH4_GPU BLASes with a copy and some without would
    // take the flat loop: then IsOccluded stays with Intersect's arrays)
    const bool any2 = t->anyHitSeen && !same && (layout[1] == TBVH_LAYOUT_CWBVH || mix[1]);
    t->blasLayout = layout[0]; t->blasMixCw2 = mix[0];
    t->blasLayoutAny = any2 ? layout[1] : -1; t->blasMixCw2Any = any2 && mix[1];
    if (!t->blasDesc) HIP_TRY(hipMalloc((void**)&t->blasDesc, nBlas * sizeof(BlasDesc)));
    HIP_TRY(hipStreamSynchronize(t->ctx->stream));   // (launches in flight read the old descriptors)
    HIP_TRY(hipMemcpy(t->blasDesc, desc[0].data(), nBlas * sizeof(BlasDesc), hipMemcpyHostToDevice));
    if (any2) {
        if (!t->blasDescAny) HIP_TRY(hipMalloc((void**)&t->blasDescAny, nBlas * sizeof(BlasDesc)));
        HIP_TRY(hipMemcpy(t->blasDescAny, desc[1].data(), nBlas * sizeof(BlasDesc), hipMemcpyHostToDevice));
    } else if (t->blasDescAny) { hipFree(t->blasDescAny); t->blasDescAny = nullptr; }
    if (t->nodes) return buildTlas4(t);
    return 0;
}

int makeWideCopy(tbvh_scene* s) {
    const int r = makeWideCopyImpl(s);
    for (size_t i = 0; i < s->usedBy.size(); i++) {
        bool seen = false;
        for (size_t k = 0; k < i; k++) seen |= s->usedBy[k] == s->usedBy[i];
        if (!seen) (void)reclassifyTlas(s->usedBy[i]);   // (the copy's arrays are new ones — or gone)
    }
    return r;
}

int makeWide4Copy(tbvh_scene* s) {
    const int r = makeWide4CopyImpl(s);
    for (size_t i = 0; i < s->usedBy.size(); i++) {
        bool seen = false;
        for (size_t k = 0; k < i; k++) seen |= s->usedBy[k] == s->usedBy[i];
        if (!seen) (void)reclassifyTlas(s->usedBy[i]);   // (the copy's arrays are new ones — or gone)
    }
    return r;
}

void dropCopiesAfterUpdate(tbvh_scene* s) {
    const uint8_t had = (uint8_t)((s->wide ? 1 : 0) | (s->wide4 ? 2 : 0) | s->pendingCopies);
    if (!had) return;
    if (s->remadeSinceUpdate && s->recopyAfter < (1u << 20)) s->recopyAfter *= 4u;   // updated again soon after the copies came back: a blob that keeps changing
    s->remadeSinceUpdate = false;
    hipStreamSynchronize(s->ctx->stream);
    freeWideCopy(s); freeWide4Copy(s);
    s->pendingCopies = had; s->queriesSinceUpdate = 0;
    for (size_t i = 0; i < s->usedBy.size(); i++) {
        bool seen = false;
        for (size_t k = 0; k < i; k++) seen |= s->usedBy[k] == s->usedBy[i];
        if (!seen) { (void)reclassifyTlas(s->usedBy[i]); s->usedBy[i]->blasRecopyPending = true; }   // (the TLASes enter this BLAS through its own nodes meanwhile)
    }
}

static void remakePendingCopies(tbvh_scene* b) {
    const uint8_t kinds = b->pendingCopies;
    b->pendingCopies = 0; b->remadeSinceUpdate = true;
    if (kinds & 1u) makeWideCopy(b);
    if (kinds & 2u) makeWide4Copy(b);
}

void countQueryForRecopy(tbvh_scene* s) {
    if (!s->isTlas) {
        if (s->pendingCopies && ++s->queriesSinceUpdate >= s->recopyAfter) remakePendingCopies(s);
        return;
    }
    if (!s->blasRecopyPending) return;
    bool still = false;
    for (tbvh_scene* b : s->blasList)
        if (b->pendingCopies) { if (++b->queriesSinceUpdate >= b->recopyAfter) remakePendingCopies(b); else still = true; }
    s->blasRecopyPending = still;
}
}  // namespace tbvh_capi
}  // extern "C++"

int tbvh_upload_tlas(tbvh_context* c, const void* nodes64, uint64_t nNodes, const uint32_t* idx, uint64_t nIdx, const void* inst,
                     uint64_t nInst, tbvh_scene* const* blas, uint64_t nBlas, tbvh_scene** out) {
    if (!c || !nodes64 || !idx || !inst || !blas || !out || !nNodes || !nIdx || !nInst || !nBlas) return fail(TBVH_E_INVALID, "tbvh_upload_tlas: null/empty argument");
    for (uint64_t i = 0; i < nBlas; i++) {
        const tbvh_scene* b = blas[i];
        if (!b || b->ctx != c || b->isTlas || b->zombie) return fail(TBVH_E_INVALID, "BLAS %llu is null, freed, a TLAS, or from another context", (unsigned long long)i);
        if (b->layout != TBVH_LAYOUT_CWBVH && b->layout != TBVH_LAYOUT_BVH4_GPU && b->layout != TBVH_LAYOUT_BVH_GPU)
            return fail(TBVH_E_INVALID, "BLAS %llu: layout %d cannot be a BLAS", (unsigned long long)i, b->layout);
    }
    TBVH_ENTER(c);
    for (uint64_t i = 0; i < nBlas; i++)   // closest-hit queries enter BVH_GPU and BVH8_CWBVH BLASes through 4-wide copies (blasView), made now; the 8-wide copies any-hit queries
                                           // enter BVH_GPU and BVH4_GPU BLASes through are made by the TLAS's first any-hit query (launchQuery)
        if ((blas[i]->layout == TBVH_LAYOUT_BVH_GPU || blas[i]->layout == TBVH_LAYOUT_CWBVH) && !blas[i]->wide4Tried && blas[i]->variant == 0) makeWide4Copy(blas[i]);
    tbvh_scene* s = newScene(c, TBVH_LAYOUT_BVH_GPU);
    if (!s) return fail(TBVH_E_NOMEM, "out of host memory");
    s->isTlas = true; s->nBlas = nBlas;
    s->blasLayout = -1;   // (not classified yet)
    for (uint64_t i = 0; i < nBlas; i++) { s->blasList.push_back(blas[i]); blas[i]->usedBy.push_back(s); }
    if (int r = reclassifyTlas(s)) { tbvh_free_scene(s); return r; }
    if (int r = tlasCopy(s, nodes64, nNodes, idx, nIdx, inst, nInst)) { tbvh_free_scene(s); return r; }
    *out = s;
    return 0;
}

int tbvh_update_tlas(tbvh_scene* s, const void* nodes64, uint64_t nNodes, const uint32_t* idx, uint64_t nIdx, const void* inst, uint64_t nInst) {
    if (!s || !s->isTlas || !nodes64 || !idx || !inst || !nNodes || !nIdx || !nInst) return fail(TBVH_E_INVALID, "tbvh_update_tlas: not a TLAS or null/empty argument");
    TBVH_ENTER(s->ctx);
    return tlasCopy(s, nodes64, nNodes, idx, nIdx, inst, nInst);
}

// ---- in-place re-upload of a BLAS whose blob the caller refitted / re-converted on the host ---------------------------------------------
// (BVH::Refit tiny_bvh.h:3055-3093 + X::ConvertFrom again: the reference's flow for animated geometry.)  The device allocations, the scene
// handle and the pointers the TLASes over this BLAS hold stay as they are; the library's derived copies follow.
int tbvh_update_bvh_gpu(tbvh_scene* s, const void* nodes64, uint64_t nNodes, const uint32_t* primIdx, uint64_t nIdx, const void* verts16, uint64_t nTris) {
    if (!s || s->isTlas || s->layout != TBVH_LAYOUT_BVH_GPU || !nodes64 || !primIdx || !verts16 || !nNodes) return fail(TBVH_E_INVALID, "tbvh_update_bvh_gpu: not a BVH_GPU scene or null/empty argument");
    if (nNodes * 4 > s->capNodeBlocks || nIdx * 3 > s->capTriBlocks) return fail(TBVH_E_INVALID, "tbvh_update_bvh_gpu: the blob (%llu nodes, %llu indices) is larger than the one uploaded: free the scene and upload", (unsigned long long)nNodes, (unsigned long long)nIdx);
    if (const char* why = validate_bvh_gpu((const NodeAL*)nodes64, nNodes, nIdx)) return fail(why == kValidateNoMemory ? TBVH_E_NOMEM : TBVH_E_FORMAT, "%s", why);
    tbvh_context* c = s->ctx;
    TBVH_ENTER(c);
    uint32_t* dIdx = nullptr; float4* dVerts = nullptr;
    hipError_t e = hipMalloc((void**)&dIdx, (nIdx ? nIdx : 1) * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&dVerts, (nTris ? nTris : 1) * 48);
    if (e == hipSuccess) e = hipMemcpyAsync(s->nodes, nodes64, nNodes * 64, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dIdx, primIdx, nIdx * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dVerts, verts16, nTris * 48, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && nIdx) { launch_gather_tris(dIdx, dVerts, s->tris, nIdx, nTris, c->stream); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (dIdx) hipFree(dIdx);
    if (dVerts) hipFree(dVerts);
    if (e != hipSuccess) return fail(TBVH_E_HIP, "tbvh_update_bvh_gpu: %s", hipGetErrorString(e));
    s->nNodeBlocks = nNodes * 4; s->nTriBlocks = nIdx * 3;
    dropCopiesAfterUpdate(s);   // (the copies are of the old tree: they come back once the blob has settled — tbvh_scene::pendingCopies)
    return 0;
}

int tbvh_update_bvh4_gpu(tbvh_scene* s, const void* blocks16, uint64_t nBlocks) {
    if (!s || s->isTlas || s->layout != TBVH_LAYOUT_BVH4_GPU || !blocks16 || nBlocks < 4) return fail(TBVH_E_INVALID, "tbvh_update_bvh4_gpu: not a BVH4_GPU scene or null/empty argument");
    if (nBlocks > s->capNodeBlocks) return fail(TBVH_E_INVALID, "tbvh_update_bvh4_gpu: the blob (%llu blocks) is larger than the one uploaded (%llu): free the scene and upload", (unsigned long long)nBlocks, (unsigned long long)s->capNodeBlocks);
    if (const char* why = validate_bvh4_gpu((const Vec4*)blocks16, nBlocks)) return fail(why == kValidateNoMemory ? TBVH_E_NOMEM : TBVH_E_FORMAT, "%s", why);
    tbvh_context* c = s->ctx;
    TBVH_ENTER(c);
    HIP_TRY(hipMemcpyAsync(s->nodes, blocks16, nBlocks * 16, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    s->nNodeBlocks = nBlocks;
    s->b4Levels.clear();   // (the node list of a device refit is rebuilt by the next tbvh_refit)
    if (s->refitScratch) { hipFree(s->refitScratch); s->refitScratch = nullptr; }
    dropCopiesAfterUpdate(s);   // (the copy is of the old tree: it comes back once the blob has settled — tbvh_scene::pendingCopies)
    return 0;
}

static int updateCwbvhImpl(tbvh_scene* s, const void* nodes16, uint64_t nNodeBlocks, const void* tris16, uint64_t nTriBlocks) {
    if (!s || s->isTlas || s->layout != TBVH_LAYOUT_CWBVH || !nodes16 || nNodeBlocks < 5 || (nTriBlocks && !tris16)) return fail(TBVH_E_INVALID, "tbvh_update_cwbvh: not a BVH8_CWBVH scene or null/empty argument");
    if (nNodeBlocks % 5) return fail(TBVH_E_FORMAT, "CWBVH node blocks (%llu) not a multiple of 5", (unsigned long long)nNodeBlocks);
    if (nNodeBlocks > s->capNodeBlocks || nTriBlocks > s->capTriBlocks) return fail(TBVH_E_INVALID, "tbvh_update_cwbvh: the blob (%llu + %llu blocks) is larger than the one uploaded (%llu + %llu): free the scene and upload",
                                                                                    (unsigned long long)nNodeBlocks, (unsigned long long)nTriBlocks, (unsigned long long)s->capNodeBlocks, (unsigned long long)s->capTriBlocks);
    if (const char* why = validate_cwbvh((const Vec4*)nodes16, nNodeBlocks / 5, nTriBlocks)) return fail(why == kValidateNoMemory ? TBVH_E_NOMEM : TBVH_E_FORMAT, "%s", why);
    tbvh_context* c = s->ctx;
    TBVH_ENTER(c);
    HIP_TRY(hipMemcpyAsync(s->nodes, nodes16, nNodeBlocks * 16, hipMemcpyHostToDevice, c->stream));
    if (nTriBlocks) HIP_TRY(hipMemcpyAsync(s->tris, tris16, nTriBlocks * 16, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));   // the caller may reuse its arrays
    const uint32_t nNodes = (uint32_t)(nNodeBlocks / 5);
    const uint64_t hash = cwbvhTopologyHash((const Vec4*)nodes16, nNodes);
    const bool sameShape = nNodes == s->nNodes && nTriBlocks == s->nTriBlocks && hash == s->topoHash;
    s->bytes -= (s->nNodeBlocks + s->nTriBlocks) * 16; s->bytes += (nNodeBlocks + nTriBlocks) * 16;
    s->nNodes = nNodes; s->nNodeBlocks = nNodeBlocks; s->nTriBlocks = nTriBlocks; s->topoHash = hash;
    if (s->refitScratch) { hipFree(s->refitScratch); s->refitScratch = nullptr; }   // (sized and filled for the old tree)
    if (!sameShape) for (auto& kind : s->cohTuner) for (CohTuner& tu : kind) if (!tu.pinned) { tu.drop_pending(); tu = CohTuner(); }   // (its timings were taken on the old tree)
    if (sameShape) {   // boxes and vertices moved, the tree did not: the derived copies keep their numbering and are re-derived on the device
        if (s->nodes128) launch_cwbvh_pad(s->nodes, s->nodes128, nNodes, c->stream);
        if (s->nodesHy) launch_cwbvh_derive_hybrid(s->nodes, s->hyPerm, s->nodesHy, nNodes, s->hybridK, (c->embedTris && !(c->expFlags & 8u)) ? s->tris : nullptr, c->stream);
        if (s->tris64) launch_cwbvh_pad_tris(s->tris, s->tris64, nTriBlocks / 3, c->stream);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    // another tree in the same allocation: the derived copies go; they come back as at upload (padded nodes now, the incoherent-batch copies lazily)
    if (s->nodes128) { hipFree(s->nodes128); s->nodes128 = nullptr; }
    if (s->nodesHy) { hipFree(s->nodesHy); s->nodesHy = nullptr; }
    if (s->tris64) { hipFree(s->tris64); s->tris64 = nullptr; }
    if (s->hyPerm) { hipFree(s->hyPerm); s->hyPerm = nullptr; }
    s->hybridK = 0; s->hyTried = false; s->hyLevelOrder = false;
    s->bytes = (nNodeBlocks + nTriBlocks) * 16 + s->opmapBytes;
    return padCwbvhIfLarge(s);
}

int tbvh_update_cwbvh(tbvh_scene* s, const void* nodes16, uint64_t nNodeBlocks, const void* tris16, uint64_t nTriBlocks) {
    if (s && !s->isTlas && (s->wide4 || s->pendingCopies)) {   // the 4-wide copy TLASes enter this BLAS through is of the old tree (also if the update is refused: harmless)
        TBVH_ENTER(s->ctx);
        dropCopiesAfterUpdate(s);
    }
    return updateCwbvhImpl(s, nodes16, nNodeBlocks, tris16, nTriBlocks);
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
    s->nNodeBlocks = nBlocks; s->capNodeBlocks = nBlocks; s->bytes = nBlocks * 16;
    *out = s;
    return 0;
}

}  // namespace
namespace tbvh_capi {
static int convertDeviceImpl(tbvh_context* c, int layout, const float4* dN2, uint64_t nNodes2, const uint32_t* dIdx, uint64_t nIdx, const float4* dV, uint64_t nTris,
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
    if (((uint64_t)nWide * 5) >> 32) return fail(TBVH_E_FORMAT, "BVH2 -> CWBVH: %u nodes are beyond the layout's 32-bit block index", nWide);   // (as tbvh_upload_cwbvh refuses them)
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
    s->capNodeBlocks = s->nNodeBlocks; s->capTriBlocks = nWideTris ? s->nTriBlocks : 3;
    s->bytes = (s->nNodeBlocks + s->nTriBlocks) * 16;
    if (int r = padCwbvhIfLarge(s)) { tbvh_free_scene(s); return r; }
    s->hyLevelOrder = true;   // (level order is close to priority order: the incoherent-batch copies need no renumbering)
    *out = s;
    return 0;
}
}  // namespace

int tbvh_convert_bvh2_device(tbvh_context* c, const void* nodes32, uint64_t nNodes2, const uint32_t* primIdx, uint64_t nIdx, const void* verts16,
                             uint64_t nTris, int onDevice, int layout, tbvh_scene** out) {
    if (!c || !nodes32 || !primIdx || !verts16 || !out || nNodes2 == 0 || nIdx == 0 || nTris == 0) return fail(TBVH_E_INVALID, "tbvh_convert_bvh2_device: null/empty argument");
    if (layout != TBVH_LAYOUT_CWBVH && layout != TBVH_LAYOUT_BVH4_GPU) return fail(TBVH_E_INVALID, "tbvh_convert_bvh2_device: target layout %d not supported (BVH8_CWBVH and BVH4_GPU are)", layout);
    if (nNodes2 > 0x7fffffffull || nIdx > 0x7fffffffull) return fail(TBVH_E_INVALID, "tbvh_convert_bvh2_device: BVH2 too large for 32-bit node / triangle indices");
    TBVH_ENTER(c);
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
    HIP_TRY(timedBegin(c));
    const int r = convertDeviceImpl(c, layout, dN2, nNodes2, dIdx, nIdx, dV, nTris, out);
    HIP_TRY(timedEnd(c));
    return r;
}

namespace {
// builder: 0 = LBVH (maxLeafTris applies), 1 = PLOC (one triangle per leaf; radius = search window to each side)
int buildDeviceImpl(const char* who, tbvh_context* c, const void* verts16, uint64_t nTris, int onDevice, int layout, uint32_t maxLeafTris, int builder, uint32_t radius,
                    tbvh_scene** out) {
    if (!c || !verts16 || !out || nTris == 0) return fail(TBVH_E_INVALID, "%s: null/empty argument", who);
    if (layout != TBVH_LAYOUT_CWBVH && layout != TBVH_LAYOUT_BVH4_GPU) return fail(TBVH_E_INVALID, "%s: target layout %d not supported (BVH8_CWBVH and BVH4_GPU are)", who, layout);
    if (nTris > 0x3fffffffull) return fail(TBVH_E_INVALID, "%s: too many triangles for 32-bit node indices", who);
    TBVH_ENTER(c);
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
    HIP_TRY(timedBegin(c));
    if (builder == 1) HIP_TRY(launch_ploc_build(dV, (uint32_t)nTris, radius, (float4*)t.n2, (uint32_t*)t.idx, t.scratch, sortTemp, scanTemp, c->stream, nullptr));
    else HIP_TRY(launch_lbvh_build(dV, (uint32_t)nTris, maxLeafTris, (float4*)t.n2, (uint32_t*)t.idx, t.scratch, sortTemp, c->stream));
    const int r = convertDeviceImpl(c, layout, (const float4*)t.n2, nTris * 2, (const uint32_t*)t.idx, nTris, dV, nTris, out);
    HIP_TRY(timedEnd(c));
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
    for (size_t i = 0; i < b->usedBy.size(); i++) {
        bool seen = false;
        for (size_t k = 0; k < i; k++) seen |= b->usedBy[k] == b->usedBy[i];
        if (!seen) if (int r = reclassifyTlas(b->usedBy[i])) return r;
    }
    return 0;
}
}  // namespace

int tbvh_set_opacity_micromaps(tbvh_scene* s, const uint32_t* mapData, uint32_t N, uint64_t nTris, int onDevice) {
    if (!s || s->isTlas) return fail(TBVH_E_INVALID, "tbvh_set_opacity_micromaps: not a BLAS scene (set the maps on the BLASes before uploading their TLAS)");
    tbvh_context* c = s->ctx;
    TBVH_ENTER(c);
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
    if (s->wide) { s->wide->opmap = s->opmap; s->wide->opmapN = s->opmapN; }   // (shared, owned here)
    if (s->wide4) { s->wide4->opmap = s->opmap; s->wide4->opmapN = s->opmapN; }
    const int r = refreshBlasDescs(s);   // the descriptors are rewritten before the old maps go
    if (old && r == 0) hipFree(old);   // (a failed refresh may have left a descriptor on the old maps: leak them rather than dangle)
    return r;
}

int tbvh_scene_download(tbvh_scene* s, int which, void* dst, uint64_t capBytes, uint64_t* bytesOut) {
    if (!s || s->isTlas || (which != 0 && which != 1)) return fail(TBVH_E_INVALID, "tbvh_scene_download: not a BLAS scene or bad blob selector");
    tbvh_context* c = s->ctx;
    TBVH_ENTER(c);
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

namespace {
constexpr uint64_t kRefitKeepRays = 8ull << 20;   // a copy's refit (0.3-0.5 ms per 100 k triangles) pays from about this many rays per refit on (0.04-0.08 ns gained per ray)
// a mesh refitted every frame with few rays traced in between: the copies are dropped (they come back like after an update: tbvh_scene::pendingCopies)
bool refitDropsCopies(tbvh_scene* s) {
    uint64_t total = s->raysTraced;
    for (size_t i = 0; i < s->usedBy.size(); i++) {
        bool seen = false;
        for (size_t k = 0; k < i; k++) seen |= s->usedBy[k] == s->usedBy[i];
        if (!seen) total += s->usedBy[i]->raysTraced;
    }
    const bool drop = (s->wide || s->wide4) && s->refitSeen && total - s->raysAtRefit < kRefitKeepRays;
    s->refitSeen = true; s->raysAtRefit = total;
    if (drop) dropCopiesAfterUpdate(s);
    return drop;
}
}  // namespace

int tbvh_refit(tbvh_scene* s, const void* verts16, uint64_t nTris, int onDevice) {
    if (!s || !verts16 || !nTris) return fail(TBVH_E_INVALID, "tbvh_refit: null/empty argument");
    if (s->isTlas) return fail(TBVH_E_INVALID, "tbvh_refit: a TLAS is rebuilt with tbvh_rebuild_tlas_device / tbvh_update_tlas");
    tbvh_context* c = s->ctx;
    TBVH_ENTER(c);
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
        HIP_TRY(timedBegin(c));
        HIP_TRY(run_refit_bvh4(s->nodes, s->nNodeBlocks, dv4, nTris, items, capNodes, counter, childBox, s->b4Levels, c->status, c->stream));
        HIP_TRY(timedEnd(c));
        if (refitDropsCopies(s)) return 0;
        if (s->wide) return tbvh_refit(s->wide, dv4, nTris, 1);   // the 8-wide copy follows
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
    HIP_TRY(timedBegin(c));
    HIP_TRY(launch_refit(s->layout, s->nodes, nNodes, s->tris, nRecords, dv, nTris, s->refitScratch, c->status, c->stream));
    HIP_TRY(timedEnd(c));
    // derived node layouts of the experiment kernels would be stale now
    if (s->nodes128) launch_cwbvh_pad(s->nodes, s->nodes128, nNodes, c->stream);   // keep the padded copy current
    if (s->nodesHy) launch_cwbvh_derive_hybrid(s->nodes, s->hyPerm, s->nodesHy, nNodes, s->hybridK, (c->embedTris && !(c->expFlags & 8u)) ? s->tris : nullptr, c->stream);
    if (s->tris64) launch_cwbvh_pad_tris(s->tris, s->tris64, s->nTriBlocks / 3, c->stream);
    if (refitDropsCopies(s)) return 0;
    if (s->wide) if (int r = tbvh_refit(s->wide, dv, nTris, 1)) return r;     // the 8-wide copy follows (same vertices, already on the device)
    if (s->wide4) return tbvh_refit(s->wide4, dv, nTris, 1);                 // ... and the 4-wide one
    return 0;
}

int tbvh_rebuild_tlas_device(tbvh_scene* s, const void* transforms, int onDevice, const float* blasBounds6, uint64_t nBlas) {
    if (!s || !s->isTlas) return fail(TBVH_E_INVALID, "tbvh_rebuild_tlas_device: not a TLAS");
    tbvh_context* c = s->ctx;
    TBVH_ENTER(c);
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
    HIP_TRY(timedBegin(c));
    HIP_TRY(launch_tlas_rebuild(s->nodes, s->tlasIdx, s->instances, xf, s->blasBounds, (uint32_t)n, (uint32_t)s->nBlas, s->buildScratch, s->sortTempBytes, c->stream));
    s->bytes = nNodes * 64 + n * 4 + n * 192;
    s->nTlasNodes = nNodes; s->nTlasIdx = n;
    if (int r = buildTlas4(s)) return r;
    HIP_TRY(timedEnd(c));
    return 0;
}

int tbvh_tlas_download(tbvh_scene* s, void* nodes64, uint64_t capNodes, uint32_t* idx, uint64_t capIdx, void* instances192, uint64_t capInst,
                       uint64_t* nNodesOut) {
    if (!s || !s->isTlas) return fail(TBVH_E_INVALID, "tbvh_tlas_download: not a TLAS");
    tbvh_context* c = s->ctx;
    TBVH_ENTER(c);
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
    TBVH_LOCK(c);
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    if (!s->isTlas && !s->usedBy.empty()) { s->zombie = true; return; }   // a TLAS still points at this BLAS's memory: freed with the last such TLAS
    freeWideCopy(s);
    freeWide4Copy(s);
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
    if (s->blasDescAny) hipFree(s->blasDescAny);
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
    for (auto& kind : s->cohTuner) for (CohTuner& tu : kind) tu.drop_pending();
    for (size_t i = 0; i < c->scenes.size(); i++)
        if (c->scenes[i] == s) { c->scenes.erase(c->scenes.begin() + i); break; }
    delete s;
}
int tbvh_scene_layout(const tbvh_scene* s) { return s ? s->layout : TBVH_E_INVALID; }
uint64_t tbvh_scene_device_bytes(const tbvh_scene* s) { return s ? s->bytes : 0; }

int tbvh_debug_coherent_schedule(tbvh_scene* s, int anyhit, uint32_t out[4]) {
    if (!s || !out) return fail(TBVH_E_INVALID, "tbvh_debug_coherent_schedule: null argument");
    TBVH_LOCK(s->ctx);
    if (s->wide) s = s->wide;
    const CohTuner& t = s->cohTuner[anyhit ? 1 : 0][s->cohLastClass[anyhit ? 1 : 0]];   // (kept per batch-size class; this is the class of the most recent such launch)
    out[0] = s->ctx->cohTunerMode ? (uint32_t)s->ctx->cohTunerMode : (uint32_t)t.decided;
    out[1] = t.n[0]; out[2] = t.n[1];
    out[3] = (t.n[0] && t.n[1]) ? (uint32_t)(1000.f * t.best[1] / t.best[0]) : 0u;
    return 0;
}

// ---- the coherent-batch schedule as something a caller can read, keep and give back -------------------------------------------------------
int tbvh_scene_get_schedule_hint(tbvh_scene* s, tbvh_schedule_hint* out) {
    if (!s || !out) return fail(TBVH_E_INVALID, "tbvh_scene_get_schedule_hint: null argument");
    TBVH_LOCK(s->ctx);
    std::memset(out, 0, sizeof *out);
    if (s->wide) s = s->wide;   // (a BVH_GPU / BVH4_GPU scene: its queries run on the 8-wide copy, whose tuner decides)
    for (int k = 0; k < 3; k++) {
        out->closest_hit[k] = (uint8_t)(s->ctx->cohTunerMode ? s->ctx->cohTunerMode : s->cohTuner[0][k].decided);
        out->any_hit[k] = (uint8_t)(s->ctx->cohTunerMode ? s->ctx->cohTunerMode : s->cohTuner[1][k].decided);
    }
    // reserved[0 / 1]: the class of 768 k .. 1.5 M-ray batches on a scene under 48 MB (closest-hit / any-hit)
    out->reserved[0] = (uint8_t)(s->ctx->cohTunerMode ? s->ctx->cohTunerMode : s->cohTuner[0][3].decided);
    out->reserved[1] = (uint8_t)(s->ctx->cohTunerMode ? s->ctx->cohTunerMode : s->cohTuner[1][3].decided);
    return 0;
}

int tbvh_scene_set_schedule_hint(tbvh_scene* s, const tbvh_schedule_hint* hint) {
    if (!s || !hint) return fail(TBVH_E_INVALID, "tbvh_scene_set_schedule_hint: null argument");
    for (int k = 0; k < 3; k++) if (hint->closest_hit[k] > 3 || hint->any_hit[k] > 3) return fail(TBVH_E_INVALID, "tbvh_scene_set_schedule_hint: entries are 0 (measure), 1 (deferred + gated), 2 (strict) or 3 (one traversal per wave)");
    if (hint->reserved[0] > 3 || hint->reserved[1] > 3) return fail(TBVH_E_INVALID, "tbvh_scene_set_schedule_hint: entries are 0 (measure), 1 (deferred + gated), 2 (strict) or 3 (one traversal per wave)");
    TBVH_LOCK(s->ctx);
    if (s->wide) s = s->wide;
    for (int a = 0; a < 2; a++) for (int k = 0; k < 4; k++) {
        const uint8_t v = k == 3 ? hint->reserved[a] : a ? hint->any_hit[k] : hint->closest_hit[k];
        CohTuner& tu = s->cohTuner[a][k];
        tu.drop_pending();
        tu = CohTuner();          // (0: back to measuring, from scratch)
        if (v) { tu.decided = v; tu.pinned = true; }
    }
    return 0;
}

int tbvh_set_variant(tbvh_scene* s, int v) {
    if (!s) return fail(TBVH_E_INVALID, "null scene");
    TBVH_LOCK(s->ctx);
    // only the BVH8_CWBVH kernel keeps diagnostic variants (kernels_cwbvh.hip: forced schedules, instrumented kernels)
    // ... and BVH_GPU / BVH4_GPU scenes one: 1 = trace the nodes as uploaded (k_bvh2 / k_bvh4) even when the scene has an 8-wide copy (tests, A/B)
    const bool ok = v == 0 || (!s->isTlas && s->layout == TBVH_LAYOUT_CWBVH && cwbvh_variant_valid(v)) || (!s->isTlas && (s->layout == TBVH_LAYOUT_BVH_GPU || s->layout == TBVH_LAYOUT_BVH4_GPU) && v == 1);
    if (!ok) return fail(TBVH_E_INVALID, "unknown variant %d for layout %d", v, s->layout);
    const bool viewChanges = !s->isTlas && (s->wide || s->wide4) && (s->variant == 0) != (v == 0);
    s->variant = v;
    if (viewChanges) return refreshBlasDescs(s);   // (the TLASes over this BLAS enter it through the copy, or through its own nodes)
    return 0;
}

int tbvh_cwbvh_set_hybrid(tbvh_scene* s, int64_t packedNodes) {
    if (!s || s->isTlas || s->layout != TBVH_LAYOUT_CWBVH) return fail(TBVH_E_INVALID, "tbvh_cwbvh_set_hybrid: not a BVH8_CWBVH scene");
    tbvh_context* c = s->ctx;
    TBVH_ENTER(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (s->nodesHy) { s->bytes -= hybridBytes(s->nNodes, s->hybridK); hipFree(s->nodesHy); s->nodesHy = nullptr; }
    s->hyTried = true;   // the caller decides now: no lazy build behind its back
    if (packedNodes < 0) return 0;
    if (s->nTriBlocks / 3 >= (1ull << 27)) return fail(TBVH_E_INVALID, "tbvh_cwbvh_set_hybrid: 2^27 triangle records or more");
    if ((uint64_t)s->nNodes * 8 >> 32) return fail(TBVH_E_INVALID, "tbvh_cwbvh_set_hybrid: 2^29 nodes or more (the copy is addressed in 32-bit float4 offsets: cwbvh_node.h)");
    const uint32_t K = (uint32_t)std::min<uint64_t>((uint64_t)packedNodes, s->nNodes) & ~7u;   // the padded part starts on a 128-byte line
    if (!s->hyPerm && !s->hyLevelOrder) {
        std::vector<Vec4> host((size_t)s->nNodes * 5);
        HIP_TRY(hipMemcpy(host.data(), s->nodes, host.size() * 16, hipMemcpyDeviceToHost));
        std::vector<uint32_t> perm;
        if (!cwbvh_priority_order(host.data(), s->nNodes, perm)) return fail(TBVH_E_FORMAT, "tbvh_cwbvh_set_hybrid: the node array is not a strict tree (a child range shared by two parents or out of range)");
        HIP_TRY(hipMalloc((void**)&s->hyPerm, (size_t)s->nNodes * 4));
        HIP_TRY(hipMemcpy(s->hyPerm, perm.data(), (size_t)s->nNodes * 4, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMalloc((void**)&s->nodesHy, hybridBytes(s->nNodes, K)));
    HIP_TRY(hipMemsetAsync(s->nodesHy, 0, hybridBytes(s->nNodes, K), c->stream));
    s->hybridK = K;
    launch_cwbvh_derive_hybrid(s->nodes, s->hyPerm, s->nodesHy, s->nNodes, K, (c->embedTris && !(c->expFlags & 8u)) ? s->tris : nullptr, c->stream);
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

}  // extern "C"
