// speedtest_gpu_section.cpp — the GPU section of the reference's tiny_bvh_speedtest.cpp
// (:1092-1241) re-hosted on the HIP engine: the REAL tiny_bvh.h builds BVH_GPU / BVH4_GPU /
// BVH8_CWBVH with BuildHQ exactly as the speedtest does (:1098-1099, :1149-1150, :1196-1197),
// the blobs go through the C ABI verbatim, the host tinybvh::Ray[] array is traced IN PLACE
// (stride 128), and the result is validated ray by ray against tinybvh::BVH::Intersect.
// This file needs the reference header at compile time (it is how a tinybvh user would adopt
// the engine); it is therefore built only where /root/reference exists:
//
//   g++ -std=c++20 -O3 -mavx2 -mfma -I/root/reference -Iinclude examples/speedtest_gpu_section.cpp \
//       -Ltinybvh_amd -ltinybvh_amd -Wl,-rpath,'$ORIGIN/../../tinybvh_amd' -lpthread -o examples/_build/speedtest_gpu_section
//   examples/_build/speedtest_gpu_section [mesh.bin]
#define TINYBVH_IMPLEMENTATION
#include "tiny_bvh.h"

#include <cstdio>
#include <fstream>
#include <vector>

#include "tinybvh_amd.h"

using namespace tinybvh;

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, tbvh_last_error()); return 1; } } while (0)

static int validate(const char* name, const Ray* got, const Ray* ref, unsigned N) {
    unsigned hitmiss = 0, prim = 0, tie = 0, bits = 0, hits = 0;
    for (unsigned i = 0; i < N; i++) {
        const bool a = got[i].hit.t < BVH_FAR, b = ref[i].hit.t < BVH_FAR;
        if (a != b) { hitmiss++; continue; }
        if (!b) continue;
        hits++;
        if (got[i].hit.prim != ref[i].hit.prim) { prim++; if (got[i].hit.t == ref[i].hit.t) tie++; continue; }
        if (memcmp(&got[i].hit.t, &ref[i].hit.t, 12) != 0) bits++;
    }
    printf("  %-12s hits %u  hit/miss mismatches %u  prim mismatches %u (exact-t ties %u)  t/u/v not bit-identical %u\n", name, hits, hitmiss, prim, tie, bits);
    return (hitmiss > 2 || prim - tie > 2 || bits) ? 1 : 0;
}

int main(int argc, char** argv) {
    // geometry: a .bin mesh (int32 count, then count*3 float4, tiny_bvh_speedtest.cpp:490-495) or a procedural blob
    std::vector<bvhvec4> tris;
    if (argc > 1) {
        std::fstream s{argv[1], s.binary | s.in};
        int n = 0; s.read((char*)&n, 4);
        tris.resize((size_t)n * 3); s.read((char*)tris.data(), (size_t)n * 48);
    } else {
        for (int i = 0; i < 40000; i++) {
            const float a = i * 0.37f, b = i * 0.011f;
            const bvhvec3 p(5 * sinf(a) * cosf(b), 5 * sinf(b), 5 * cosf(a) * cosf(b));
            tris.push_back(bvhvec4(p, 0)); tris.push_back(bvhvec4(p + bvhvec3(0.2f * cosf(a * 3), 0.1f, 0.15f), 0)); tris.push_back(bvhvec4(p + bvhvec3(0.05f, 0.2f * sinf(b * 7), -0.1f), 0));
        }
    }
    const unsigned triCount = (unsigned)tris.size() / 3;
    bvhvec3 lo(1e30f), hi(-1e30f);
    for (auto& v : tris) lo = tinybvh_min(lo, bvhvec3(v)), hi = tinybvh_max(hi, bvhvec3(v));
    const bvhvec3 c = (lo + hi) * 0.5f; const float ext = tinybvh_max(tinybvh_max(hi.x - lo.x, hi.y - lo.y), hi.z - lo.z);
    // rays: 640 x 480 pinhole in the speedtest's 4x4-tile order
    const unsigned W = 640, H = 480, N = W * H;
    Ray* rays = (Ray*)malloc64(N * sizeof(Ray));
    Ray* ref = (Ray*)malloc64(N * sizeof(Ray));
    const bvhvec3 eye = c + bvhvec3(0.2f * ext, 0.3f * ext, 1.5f * ext), view = tinybvh_normalize(c - eye);
    const bvhvec3 right = tinybvh_normalize(tinybvh_cross(bvhvec3(0, 1, 0), view)), up = 0.75f * tinybvh_cross(view, right), C = eye + 2 * view;
    const bvhvec3 p1 = C - right + up, p2 = C + right + up, p3 = C - right - up;
    unsigned k = 0;
    for (unsigned ty = 0; ty < H / 4; ty++) for (unsigned tx = 0; tx < W / 4; tx++) for (unsigned y = 0; y < 4; y++) for (unsigned x = 0; x < 4; x++) {
        const float u = (float)(tx * 4 + x) / W, v = (float)(ty * 4 + y) / H;
        rays[k++] = Ray(eye, tinybvh_normalize(p1 + u * (p2 - p1) + v * (p3 - p1) - eye));
    }
    // reference: BVH::Intersect (the speedtest's refDistFull, :1076-1090)
    BVH refbvh; refbvh.Build(tris.data(), triCount);
    memcpy((void*)ref, (void*)rays, N * sizeof(Ray));
    for (unsigned i = 0; i < N; i++) refbvh.Intersect(ref[i]);

    tbvh_context* ctx = nullptr;
    CHECK(tbvh_init(0, &ctx));
    int bad = 0;
    Ray* work = (Ray*)malloc64(N * sizeof(Ray));
    printf("%u triangles, %u rays, layouts built by tiny_bvh.h %d.%d.%d BuildHQ, traced by the HIP engine:\n", triCount, N, TINY_BVH_VERSION_MAJOR, TINY_BVH_VERSION_MINOR, TINY_BVH_VERSION_SUB);
    {   // BVH_GPU (:1098-1141)
        BVH_GPU bvh; bvh.BuildHQ(tris.data(), triCount);
        tbvh_scene* s = nullptr;
        CHECK(tbvh_upload_bvh_gpu(ctx, bvh.bvhNode, bvh.usedNodes, bvh.bvh.primIdx, bvh.bvh.idxCount, tris.data(), triCount, &s));
        memcpy((void*)work, (void*)rays, N * sizeof(Ray));
        CHECK(tbvh_intersect(s, work, N, sizeof(Ray)));
        printf("  BVH_GPU      %.1f MRays/s (kernel %.3f ms)\n", N / (tbvh_time_last_ms(ctx) * 1e3), tbvh_time_last_ms(ctx));
        bad += validate("BVH_GPU", work, ref, N);
        tbvh_free_scene(s);
    }
    {   // BVH4_GPU (:1149-1188)
        BVH4_GPU bvh; bvh.BuildHQ(tris.data(), triCount);
        tbvh_scene* s = nullptr;
        CHECK(tbvh_upload_bvh4_gpu(ctx, bvh.bvh4Data, bvh.usedBlocks, &s));
        memcpy((void*)work, (void*)rays, N * sizeof(Ray));
        CHECK(tbvh_intersect(s, work, N, sizeof(Ray)));
        printf("  BVH4_GPU     %.1f MRays/s (kernel %.3f ms)\n", N / (tbvh_time_last_ms(ctx) * 1e3), tbvh_time_last_ms(ctx));
        bad += validate("BVH4_GPU", work, ref, N);
        tbvh_free_scene(s);
    }
    {   // BVH8_CWBVH (:1196-1241)
        BVH8_CWBVH bvh; bvh.BuildHQ(tris.data(), triCount);
        tbvh_scene* s = nullptr;
        CHECK(tbvh_upload_cwbvh(ctx, bvh.bvh8Data, bvh.usedBlocks, bvh.bvh8Tris, (uint64_t)bvh.bvh8.idxCount * 3, &s));
        memcpy((void*)work, (void*)rays, N * sizeof(Ray));
        CHECK(tbvh_intersect(s, work, N, sizeof(Ray)));
        printf("  BVH8_CWBVH   %.1f MRays/s (kernel %.3f ms)\n", N / (tbvh_time_last_ms(ctx) * 1e3), tbvh_time_last_ms(ctx));
        bad += validate("BVH8_CWBVH", work, ref, N);
        tbvh_free_scene(s);
    }
    tbvh_shutdown(ctx);
    printf(bad ? "VALIDATION FAILED\n" : "all layouts agree with BVH::Intersect\n");
    return bad ? 1 : 0;
}
