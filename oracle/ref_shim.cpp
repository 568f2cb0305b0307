// ref_shim.cpp — TEST INFRASTRUCTURE ONLY.
//
// A thin extern "C" wrapper around the REAL reference (jbikker/tinybvh, tiny_bvh.h),
// compiled from the sources where they lie (-I$(REFERENCE), default /root/reference) into
// oracle/_ref/libtinybvh_ref.so by oracle/Makefile.  No reference source is copied into
// this repository; this file only calls the reference's public API.  It is used to
//   (1) pin the C restatement in tbvh_oracle.c against BVH::Intersect / IsOccluded,
//   (2) hand reference-BUILT blobs (BVH_GPU / BVH4_GPU / BVH8_CWBVH, Build and BuildHQ) to
//       the HIP kernels, which is the real drop-in situation,
//   (3) time the reference's BVH8_CPU (AVX2) path as bench.py's cpu_baseline ("reference").
// The .so travels to the GPU box inside the repo snapshot; /root/reference does not.
#define TINYBVH_IMPLEMENTATION
#include "tiny_bvh.h"

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

using namespace tinybvh;

static_assert(sizeof(Ray) == 128, "host Ray is 128 bytes (SURVEY.md §2a)");
static_assert(sizeof(BLASInstance) == 192, "BLASInstance is 192 bytes");
static_assert(INST_IDX_BITS == 32, "shim assumes the default INST_IDX_BITS");

namespace {

struct RefScene {
    bvhvec4* verts = nullptr;  // owned copy, 64-byte aligned
    uint32_t triCount = 0;
    bool hq = false;
    int optimize = 0;   // > 0: BVH8_CWBVH::Build followed by ::Optimize(optimize) (the reinsertion optimiser, tiny_bvh.h:4338-4449) — tools/ref_blobs_probe.py only
    BVH bvh;                   // the oracle tree (Build or BuildHQ)
    BVH_GPU* gpu2 = nullptr;
    BVH4_GPU* gpu4 = nullptr;
    BVH8_CWBVH* cw = nullptr;
    BVH8_CPU* cpu8 = nullptr;
};

struct RefTlas {
    std::vector<BLASInstance> inst;
    std::vector<BVHBase*> blas;
    BVH tlas;
    BVH_GPU* tlasGpu = nullptr;
};

// device ray (64 B) -> host Ray: the first 64 bytes of tinybvh::Ray are the device record
// (offsets checked in SURVEY.md §2a [probe] and by the static_asserts in ref_selfcheck()).
inline void load(Ray& r, const char* src) { std::memset((void*)&r, 0, sizeof(Ray)); std::memcpy((void*)&r, src, 64); }
inline void store(char* dst, const Ray& r) { std::memcpy(dst + 44, (const char*)&r + 44, 20); }

template <class F> void forRays(void* rays, uint64_t n, uint32_t stride, F f) {
    char* p = (char*)rays;
    for (uint64_t i = 0; i < n; i++, p += stride) { Ray r; load(r, p); f(r); store(p, r); }
}

}  // namespace

extern "C" {

int ref_selfcheck() {
    Ray r;
    const char* b = (const char*)&r;
    if ((const char*)&r.O != b || (const char*)&r.mask != b + 12) return 1;
    if ((const char*)&r.D != b + 16 || (const char*)&r.instIdx != b + 28) return 2;
    if ((const char*)&r.rD != b + 32 || (const char*)&r.hit.inst != b + 44) return 3;
    if ((const char*)&r.hit.t != b + 48 || (const char*)&r.hit.prim != b + 60) return 4;
    return 0;
}

const char* ref_version() {
    static char v[64];
    snprintf(v, sizeof v, "tinybvh %d.%d.%d", TINY_BVH_VERSION_MAJOR, TINY_BVH_VERSION_MINOR, TINY_BVH_VERSION_SUB);
    return v;
}

// Build the oracle BVH (BVH::Build, tiny_bvh.h:2124; or BVH::BuildHQ, 2623).
// threaded = 0 forces single-threaded builds for byte-stable output.
void* ref_build(const void* verts16, uint32_t triCount, int hq, int threaded) {
    RefScene* s = new RefScene;
    s->triCount = triCount; s->hq = hq == 1;
    if (hq >= 100) s->optimize = hq - 100;   // (hq = 100 + iterations: Build + Optimize, CWBVH layout only)
    s->verts = (bvhvec4*)malloc64((size_t)triCount * 48);
    std::memcpy((void*)s->verts, verts16, (size_t)triCount * 48);
    s->bvh.threadedBuild = threaded != 0;
    if (hq == 1) s->bvh.BuildHQ(s->verts, triCount); else s->bvh.Build(s->verts, triCount);
    return s;
}
void ref_free(void* h) {
    RefScene* s = (RefScene*)h;
    delete s->gpu2; delete s->gpu4; delete s->cw; delete s->cpu8;
    free64(s->verts);
    delete s;
}

// Convert to a GPU layout with the reference's own converter.  Each layout object builds
// its own tree with the same builder (Build/BuildHQ) the way tiny_bvh_speedtest.cpp does
// (:1098-1099, 1149-1150, 1196-1197).
// Layout codes are BVHBase::BVHType (tiny_bvh.h:773-791), the same values include/tinybvh_amd.h uses: 1 LAYOUT_BVH, 5 LAYOUT_BVH_GPU,
// 8 LAYOUT_BVH4_GPU, 10 LAYOUT_CWBVH, 11 LAYOUT_BVH8_AVX2 (BVH8_CPU); 110 = the BVH2 behind the CWBVH (ref_blob only).  Anything else is an error.
enum { L_BVH = 1, L_BVH_GPU = 5, L_BVH4_GPU = 8, L_CWBVH = 10, L_BVH8_CPU = 11, L_CWBVH_BVH2 = 110 };
static bool ensureLayout(RefScene* s, int layout) {
    if (layout != L_BVH && layout != L_BVH_GPU && layout != L_BVH4_GPU && layout != L_CWBVH && layout != L_BVH8_CPU) return false;
    if (layout == L_BVH_GPU && !s->gpu2) { s->gpu2 = new BVH_GPU(); if (s->hq) s->gpu2->BuildHQ(s->verts, s->triCount); else s->gpu2->Build(s->verts, s->triCount); }
    if (layout == L_BVH4_GPU && !s->gpu4) { s->gpu4 = new BVH4_GPU(); if (s->hq) s->gpu4->BuildHQ(s->verts, s->triCount); else s->gpu4->Build(s->verts, s->triCount); }
    if (layout == L_CWBVH && !s->cw) {
        s->cw = new BVH8_CWBVH();
        if (s->hq) s->cw->BuildHQ(s->verts, s->triCount); else s->cw->Build(s->verts, s->triCount);
        if (s->optimize > 0) s->cw->Optimize((uint32_t)s->optimize, false);
    }
    if (layout == L_BVH8_CPU && !s->cpu8) { s->cpu8 = new BVH8_CPU(); if (s->hq) s->cpu8->BuildHQ(s->verts, s->triCount); else s->cpu8->Build(s->verts, s->triCount); }
    return true;
}

// Blob access.  layout: 1 = BVH (Wald), 5 = BVH_GPU, 8 = BVH4_GPU, 10 = CWBVH, 110 = the BVH2 the CWBVH was
// converted from (bvh8.bvh after Compact + SplitLeafs(3), tiny_bvh.h:5829-5835): the input a device-side
// ConvertFrom replacement gets from a tinybvh user.
// which: 0 = nodes / blocks, 1 = primIdx (layouts 1, 5) or triangle blocks (layout 10).  Unknown layout: 0 elements, null pointer.
// Returns element count; *out receives the pointer (owned by the scene).
uint64_t ref_blob(void* h, int layout, int which, const void** out) {
    RefScene* s = (RefScene*)h;
    *out = nullptr;
    if (!ensureLayout(s, layout == L_CWBVH_BVH2 ? (int)L_CWBVH : layout)) return 0;
    switch (layout) {
    case L_BVH:
        if (which == 0) { *out = s->bvh.bvhNode; return s->bvh.usedNodes; }
        *out = s->bvh.primIdx; return s->bvh.idxCount;
    case L_BVH_GPU:
        if (which == 0) { *out = s->gpu2->bvhNode; return s->gpu2->usedNodes; }
        *out = s->gpu2->bvh.primIdx; return s->gpu2->bvh.idxCount;
    case L_BVH4_GPU:
        *out = s->gpu4->bvh4Data; return s->gpu4->usedBlocks;
    case L_CWBVH:
        if (which == 0) { *out = s->cw->bvh8Data; return s->cw->usedBlocks; }
        *out = s->cw->bvh8Tris; return (uint64_t)s->cw->bvh8.idxCount * 3;
    case L_CWBVH_BVH2:
        if (which == 0) { *out = s->cw->bvh8.bvh.bvhNode; return s->cw->bvh8.bvh.usedNodes; }
        *out = s->cw->bvh8.bvh.primIdx; return s->cw->bvh8.bvh.idxCount;
    }
    *out = nullptr; return 0;
}
const void* ref_verts(void* h) { return ((RefScene*)h)->verts; }
// BVHBase::SetOpacityMicroMaps on the scene's BVH (the oracle layout); mapData must outlive the queries.
void ref_set_opmap(void* h, uint32_t* mapData, uint32_t N) { ((RefScene*)h)->bvh.SetOpacityMicroMaps(mapData, N); }

// Per-ray queries through the reference's own traversal code.
// layout 1: BVH::Intersect (THE oracle, tiny_bvh.h:3222); 5 / 8 / 10: the CPU mirrors of the GPU
// layouts (4657 / 5252 / 7046); 11: BVH8_CPU::Intersect (7188).  -1: unknown layout.
int ref_intersect(void* h, int layout, void* rays, uint64_t n, uint32_t stride) {
    RefScene* s = (RefScene*)h;
    if (!ensureLayout(s, layout)) return -1;
    switch (layout) {
    case L_BVH: forRays(rays, n, stride, [&](Ray& r) { s->bvh.Intersect(r); }); return 0;
    case L_BVH_GPU: forRays(rays, n, stride, [&](Ray& r) { s->gpu2->Intersect(r); }); return 0;
    case L_BVH4_GPU: forRays(rays, n, stride, [&](Ray& r) { s->gpu4->Intersect(r); }); return 0;
    case L_CWBVH: forRays(rays, n, stride, [&](Ray& r) { s->cw->Intersect(r); }); return 0;
    case L_BVH8_CPU: forRays(rays, n, stride, [&](Ray& r) { s->cpu8->Intersect(r); }); return 0;
    }
    return -1;
}
int ref_occluded(void* h, int layout, const void* rays, uint64_t n, uint32_t stride, uint8_t* out) {
    RefScene* s = (RefScene*)h;
    if ((layout != L_BVH && layout != L_BVH8_CPU) || !ensureLayout(s, layout)) return -1;
    const char* p = (const char*)rays;
    for (uint64_t i = 0; i < n; i++, p += stride) {
        Ray r; load(r, p);
        bool o;
        if (layout == L_BVH) o = s->bvh.IsOccluded(r);
        else o = s->cpu8->IsOccluded(r);
        out[i] = o ? 1 : 0;
    }
    return 0;
}

// Node / triangle visit counts from the reference CPU mirrors: c_trav = 1024, c_int = 1
// (SURVEY.md §5 trick).  layouts 1, 5, 8 only (the CWBVH mirror returns 0); -1 otherwise.
int ref_counts(void* h, int layout, const void* rays, uint64_t n, uint32_t stride, uint64_t* steps, uint64_t* tris) {
    RefScene* s = (RefScene*)h;
    if ((layout != L_BVH && layout != L_BVH_GPU && layout != L_BVH4_GPU) || !ensureLayout(s, layout)) return -1;
    BVHBase* b = layout == L_BVH ? (BVHBase*)&s->bvh : layout == L_BVH_GPU ? (BVHBase*)s->gpu2 : (BVHBase*)s->gpu4;
    const float ct = b->c_trav, ci = b->c_int;
    b->c_trav = 65536.0f; b->c_int = 1.0f;
    uint64_t S = 0, T = 0;
    const char* p = (const char*)rays;
    for (uint64_t i = 0; i < n; i++, p += stride) {
        Ray r; load(r, p);
        int32_t c = layout == L_BVH ? s->bvh.Intersect(r) : layout == L_BVH_GPU ? s->gpu2->Intersect(r) : s->gpu4->Intersect(r);
        S += (uint32_t)c >> 16; T += (uint32_t)c & 65535;
    }
    b->c_trav = ct; b->c_int = ci;
    *steps = S; *tris = T;
    return 0;
}

// Timed multi-threaded baseline with the speedtest's dynamic batch scheme: 10 000-ray
// batches handed out through an atomic counter (tiny_bvh_speedtest.cpp:392-401,
// 1077-1083).  layout 11 = BVH8_CPU (AVX2), 1 = BVH::Intersect (anything else: returns -1).  shadow != 0 times
// IsOccluded.  Rays are 64-byte records; each thread expands them to host Rays in
// batches *outside* nothing — the expansion is part of what a caller holding packed rays
// would pay, but it is small (64-byte copy) next to traversal.  Returns seconds.
double ref_time_mt(void* h, int layout, const void* rays, uint64_t n, uint32_t stride, int threads, int shadow, uint64_t* hits) {
    RefScene* s = (RefScene*)h;
    if ((layout != L_BVH && layout != L_BVH8_CPU) || !ensureLayout(s, layout)) return -1.0;
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    // expand once, untimed (the speedtest times traversal over pre-built Ray arrays)
    Ray* R = (Ray*)malloc64(n * sizeof(Ray));
    const char* p = (const char*)rays;
    for (uint64_t i = 0; i < n; i++, p += stride) load(R[i], p);
    std::atomic<uint64_t> next{0}, hitCount{0};
    auto work = [&]() {
        uint64_t local = 0;
        for (;;) {
            const uint64_t b = next.fetch_add(10000);
            if (b >= n) break;
            const uint64_t e = b + 10000 < n ? b + 10000 : n;
            if (shadow) {
                if (layout == L_BVH8_CPU) for (uint64_t i = b; i < e; i++) local += s->cpu8->IsOccluded(R[i]);
                else for (uint64_t i = b; i < e; i++) local += s->bvh.IsOccluded(R[i]);
            } else {
                if (layout == L_BVH8_CPU) for (uint64_t i = b; i < e; i++) s->cpu8->Intersect(R[i]);
                else for (uint64_t i = b; i < e; i++) s->bvh.Intersect(R[i]);
                for (uint64_t i = b; i < e; i++) local += R[i].hit.t < BVH_FAR;
            }
        }
        hitCount += local;
    };
    const auto t0 = std::chrono::high_resolution_clock::now();
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    const double sec = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    if (hits) *hits = hitCount.load();
    free64(R);
    return sec;
}

// ---- TLAS (BVH::Build(BLASInstance*...), tiny_bvh.h:2221-2259; IntersectTLAS 3306-3380) ----
// instances192: BLASInstance records with transform[] and blasIdx set; blasScenes[i] are
// RefScene handles (their BVH::bvh is the BLAS, LAYOUT_BVH as the CPU TLAS requires).
void* ref_tlas_build(void* instances192, uint32_t nInst, void** blasScenes, uint32_t nBlas) {
    RefTlas* t = new RefTlas;
    t->inst.resize(nInst);
    std::memcpy((void*)t->inst.data(), instances192, (size_t)nInst * 192);
    for (uint32_t i = 0; i < nBlas; i++) t->blas.push_back(&((RefScene*)blasScenes[i])->bvh);
    t->tlas.Build(t->inst.data(), nInst, t->blas.data(), nBlas);
    std::memcpy(instances192, (void*)t->inst.data(), (size_t)nInst * 192);  // hand back updated records
    return t;
}
void ref_tlas_free(void* h) { RefTlas* t = (RefTlas*)h; delete t->tlasGpu; delete t; }
int ref_tlas_intersect(void* h, void* rays, uint64_t n, uint32_t stride) {
    RefTlas* t = (RefTlas*)h;
    forRays(rays, n, stride, [&](Ray& r) { t->tlas.Intersect(r); });
    return 0;
}
// TLAS blobs in BVH_GPU format (BVH_GPU::Build(BLASInstance*...), tiny_bvh.h:4575-4581).
uint64_t ref_tlas_blob(void* h, int which, const void** out) {
    RefTlas* t = (RefTlas*)h;
    if (!t->tlasGpu) { t->tlasGpu = new BVH_GPU(); t->tlasGpu->ConvertFrom(t->tlas, false); }
    if (which == 0) { *out = t->tlasGpu->bvhNode; return t->tlasGpu->usedNodes; }
    if (which == 1) { *out = t->tlas.primIdx; return t->tlas.idxCount; }
    if (which == 2) { *out = t->tlas.bvhNode; return t->tlas.usedNodes; }
    *out = nullptr; return 0;
}

// ---- BVH8_CWBVH::Save / Load (tiny_bvh.h:5786-5820): the file a blob cache has to read and write -----------------
// The file embeds a raw dump of the C++ object, so a reader needs the object's size and the offsets of the fields Load
// uses afterwards.  out[0..15]: sizeof(BVH8_CWBVH), then the byte offsets of layout, triCount, idxCount, aabbMin, aabbMax,
// opmapN, opmap, bvh8Data, bvh8Tris, allocatedBlocks, usedBlocks, bvh8.idxCount, ownBVH8, c_trav, hqbvhbins.
void ref_cwbvh_object_layout(uint32_t out[16]) {
    BVH8_CWBVH o;
    const char* b = (const char*)&o;
#define OFF(m) (uint32_t)((const char*)&(o.m) - b)
    out[0] = (uint32_t)sizeof(BVH8_CWBVH);
    out[1] = OFF(layout); out[2] = OFF(triCount); out[3] = OFF(idxCount); out[4] = OFF(aabbMin); out[5] = OFF(aabbMax);
    out[6] = OFF(opmapN); out[7] = OFF(opmap); out[8] = OFF(bvh8Data); out[9] = OFF(bvh8Tris);
    out[10] = OFF(allocatedBlocks); out[11] = OFF(usedBlocks); out[12] = OFF(bvh8.idxCount); out[13] = OFF(ownBVH8);
    out[14] = OFF(c_trav); out[15] = OFF(hqbvhbins);
#undef OFF
}
// the 560-byte image of a default-constructed object (flags, cost constants, ... as the reference initialises them)
void ref_cwbvh_default_image(void* out, uint32_t bytes) {
    BVH8_CWBVH o;
    std::memcpy(out, (const void*)&o, bytes < sizeof o ? bytes : sizeof o);
}
int ref_cwbvh_save(void* h, const char* path) {
    RefScene* s = (RefScene*)h;
    ensureLayout(s, L_CWBVH);
    s->cw->Save(path);
    return 0;
}
// BVH8_CWBVH::Load on a fresh object, then BVH8_CWBVH::Intersect over the rays.  Returns 0, or 1 if Load refused the file.
int ref_cwbvh_load_and_intersect(const char* path, uint32_t expectedTris, void* rays, uint64_t n, uint32_t stride) {
    BVH8_CWBVH cw;
    if (!cw.Load(path, expectedTris)) return 1;
    forRays(rays, n, stride, [&](Ray& r) { cw.Intersect(r); });
    return 0;
}

}  // extern "C"
