// capi_host.hip — host side: the library's own builder, BVH8_CWBVH::Save / Load compatible files, host blob accessors.
#include "capi_internal.h"

using namespace tbvh;
using namespace tbvh_capi;

extern "C" {


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
    // ... and triangles split ahead of the build, 30 % extra references (host_builder.cpp: presplit): the 2.83 M-triangle street, 16.7 M rays:
    // camera / bounce / shadow rays +2.6 / +1.7 / +1.2 %; the same street off the axes +8.2 / +1.5 / +5.6 % (profiles/r05_rotated.txt)
    if (layout == TBVH_LAYOUT_CWBVH) { bp.greedyCollapse = false; bp.cPrim = 1.0f; bp.splitBudget = 0.3f; }
    // BVH4_GPU: the same collapse (leaves of <= 4 triangles as the BVH2 builder made them): 1-3 % fewer node visits + triangle
    // tests per ray on both stand-in scenes, measured +1-3 % on the GPU
    if (layout == TBVH_LAYOUT_BVH4_GPU) { bp.greedyCollapse = false; bp.cPrim = 1.0f; }
    if (p) {
        bp.bins = p->bins ? p->bins : 8; bp.threads = p->threads; bp.maxLeafTris = p->max_leaf_tris;
        if (p->flags & TBVH_BUILD_OPTIMAL_COLLAPSE) bp.greedyCollapse = false;
        if (p->flags & TBVH_BUILD_GREEDY_COLLAPSE) bp.greedyCollapse = true;
        if ((p->flags >> 8) & 0xffff) bp.cPrim = (float)((p->flags >> 8) & 0xffff) * 0.01f;
        if (p->flags & TBVH_BUILD_SPLIT_TRIANGLES) bp.splitBudget = (p->flags >> 24) ? (float)(p->flags >> 24) * 0.01f : 0.3f;
        if (p->flags & TBVH_BUILD_WHOLE_TRIANGLES) bp.splitBudget = 0.f;
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
    if (const char* e = validate_cwbvh((const Vec4*)nodes16, nNodeBlocks / 5, nTriBlocks)) return fail(e == kValidateNoMemory ? TBVH_E_NOMEM : TBVH_E_FORMAT, "tbvh_cwbvh_file_write: %s", e);
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
    if (const char* e = validate_cwbvh(h->blocksA.data(), usedBlocks / 5, (uint64_t)idxCount * 3)) { delete h; return fail(e == kValidateNoMemory ? TBVH_E_NOMEM : TBVH_E_FORMAT, "tbvh_cwbvh_file_read: %s", e); }
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

int tbvh_debug_wide_copy_bvh2(int layout, const void* blob, uint64_t nBlob, const uint32_t* primIdx, uint64_t nIdx, const void* verts16, uint64_t nTris,
                              uint32_t maxLeaf, void* nodesOut, uint64_t capNodes, uint64_t* nNodesOut, void* recsOut, uint64_t capRecs, uint64_t* nRecsOut) {
    if (!blob || !nBlob || !maxLeaf) return fail(TBVH_E_INVALID, "tbvh_debug_wide_copy_bvh2: null/empty argument");
    std::vector<Node2> n2;
    std::vector<Vec4> recs;
    try {
        if (layout == TBVH_LAYOUT_BVH_GPU) {
            if (!verts16) return fail(TBVH_E_INVALID, "tbvh_debug_wide_copy_bvh2: BVH_GPU needs prim_idx and verts16, or (prim_idx NULL) the gathered records in verts16");
            if (const char* why = validate_bvh_gpu((const NodeAL*)blob, nBlob, nIdx)) return fail(TBVH_E_FORMAT, "%s", why);
            // prim_idx == NULL: RECORD MODE, what the library itself runs (capi_scene.hip: makeWideCopy) — verts16 = n_idx records {v0|prim, e1, e2}
            if (!bvh_gpu_to_bvh2((const NodeAL*)blob, nBlob, primIdx, nIdx, primIdx ? (const Vec4*)verts16 : nullptr, nTris, maxLeaf, n2, primIdx ? nullptr : (const Vec4*)verts16))
                return fail(TBVH_E_FORMAT, "the root is a leaf");
        } else if (layout == TBVH_LAYOUT_BVH4_GPU) {
            if (const char* why = validate_bvh4_gpu((const Vec4*)blob, nBlob)) return fail(TBVH_E_FORMAT, "%s", why);
            if (!bvh4_gpu_to_bvh2((const Vec4*)blob, nBlob, maxLeaf, n2, recs)) return fail(TBVH_E_FORMAT, "the root is a leaf or the stream is malformed");
        } else if (layout == TBVH_LAYOUT_CWBVH) {   // (the 4-wide copy a TLAS enters a BVH8_CWBVH BLAS through: blob = node blocks, verts16 = the n_tris * 3 triangle blocks)
            if (!verts16 || nBlob % 5) return fail(TBVH_E_INVALID, "tbvh_debug_wide_copy_bvh2: BVH8_CWBVH needs the node blocks (a multiple of 5) and the triangle blocks in verts16");
            if (const char* why = validate_cwbvh((const Vec4*)blob, nBlob / 5, nTris * 3)) return fail(TBVH_E_FORMAT, "%s", why);
            if (!cwbvh_to_bvh2((const Vec4*)blob, nBlob / 5, (const Vec4*)verts16, nTris * 3, n2, recs)) return fail(TBVH_E_FORMAT, "the root is a leaf or the blob is malformed");
        } else return fail(TBVH_E_INVALID, "tbvh_debug_wide_copy_bvh2: layout %d (BVH_GPU 5, BVH4_GPU 8 and BVH8_CWBVH 10 have copies)", layout);
    } catch (const std::bad_alloc&) { return fail(TBVH_E_NOMEM, "out of host memory"); }
    if (nNodesOut) *nNodesOut = n2.size();
    if (nRecsOut) *nRecsOut = recs.size() / 3;
    if ((nodesOut && capNodes < n2.size()) || (recsOut && capRecs < recs.size() / 3)) return fail(TBVH_E_INVALID, "tbvh_debug_wide_copy_bvh2: output capacity too small");
    if (nodesOut) std::memcpy(nodesOut, n2.data(), n2.size() * sizeof(Node2));
    if (recsOut && !recs.empty()) std::memcpy(recsOut, recs.data(), recs.size() * sizeof(Vec4));
    return 0;
}

}  // extern "C"
