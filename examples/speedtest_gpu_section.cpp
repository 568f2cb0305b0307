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
#include <thread>
#include <vector>

#include "tiny_hip.h"   // the binding a tinybvh maintainer adds (include/tiny_hip.h over the C ABI of tinybvh_amd.h)

using namespace tinybvh;

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, tbvh_last_error()); return 1; } } while (0)

// tol = 0: t,u,v must be bit-identical; tol > 0: relative tolerance on t, absolute on u,v (the BASELINE contract, 1e-5)
static int validate(const char* name, const Ray* got, const Ray* ref, unsigned N, float tol = 0.f) {
    unsigned hitmiss = 0, prim = 0, tie = 0, bits = 0, hits = 0, outOfTol = 0;
    for (unsigned i = 0; i < N; i++) {
        const bool a = got[i].hit.t < BVH_FAR, b = ref[i].hit.t < BVH_FAR;
        if (a != b) { hitmiss++; continue; }
        if (!b) continue;
        hits++;
        if (got[i].hit.prim != ref[i].hit.prim) { prim++; if (fabsf(got[i].hit.t - ref[i].hit.t) <= tol * fabsf(ref[i].hit.t)) tie++; continue; }
        if (memcmp(&got[i].hit.t, &ref[i].hit.t, 12) != 0) bits++;
        if (fabsf(got[i].hit.t - ref[i].hit.t) > tol * fabsf(ref[i].hit.t) || fabsf(got[i].hit.u - ref[i].hit.u) > 10 * tol || fabsf(got[i].hit.v - ref[i].hit.v) > 10 * tol) outOfTol++;
    }
    if (tol == 0.f) {
        printf("  %-12s hits %u  hit/miss mismatches %u  prim mismatches %u (exact-t ties %u)  t/u/v not bit-identical %u\n", name, hits, hitmiss, prim, tie, bits);
        return (hitmiss > 2 || prim - tie > 2 || bits) ? 1 : 0;
    }
    printf("  %-12s hits %u  hit/miss mismatches %u  prim mismatches %u (t within %.0e: %u)  t/u/v beyond %.0e: %u  (bit-identical: %u)\n", name, hits, hitmiss, prim, tol, tie, tol,
           outOfTol, hits - prim - bits);
    return (hitmiss > 4 || prim - tie > 4 || outOfTol) ? 1 : 0;
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

    tbvh_context* ctx = tinyhip::Context(0);   // the binding's context of device 0 (the blocks below that use the C ABI directly share it)
    int bad = 0;
    Ray* work = (Ray*)malloc64(N * sizeof(Ray));
    printf("%u triangles, %u rays, layouts built by tiny_bvh.h %d.%d.%d BuildHQ, traced by the HIP engine:\n", triCount, N, TINY_BVH_VERSION_MAJOR, TINY_BVH_VERSION_MINOR, TINY_BVH_VERSION_SUB);
    // the three GPU blocks of the speedtest through tinyhip::Scene (tiny_hip.h): what was two or three tinyocl::Buffers + CopyToDevice + SetArguments +
    // Run + clWaitForEvents + CopyFromDevice per layout is a constructor and a call
    {   // BVH_GPU (:1098-1141)
        BVH_GPU bvh; bvh.BuildHQ(tris.data(), triCount);
        tinyhip::Scene gpu(bvh, tris.data());
        memcpy((void*)work, (void*)rays, N * sizeof(Ray));
        gpu.Intersect(work, N);
        printf("  BVH_GPU      %.1f MRays/s (kernel %.3f ms)\n", N / (gpu.LastKernelMs() * 1e3), gpu.LastKernelMs());
        bad += validate("BVH_GPU", work, ref, N);
    }
    {   // BVH4_GPU (:1149-1188)
        BVH4_GPU bvh; bvh.BuildHQ(tris.data(), triCount);
        tinyhip::Scene gpu(bvh);
        memcpy((void*)work, (void*)rays, N * sizeof(Ray));
        gpu.Intersect(work, N);
        printf("  BVH4_GPU     %.1f MRays/s (kernel %.3f ms)\n", N / (gpu.LastKernelMs() * 1e3), gpu.LastKernelMs());
        bad += validate("BVH4_GPU", work, ref, N);
    }
    {   // BVH8_CWBVH (:1196-1241)
        BVH8_CWBVH bvh; bvh.BuildHQ(tris.data(), triCount);
        tinyhip::Scene gpu(bvh);
        memcpy((void*)work, (void*)rays, N * sizeof(Ray));
        gpu.Intersect(work, N);
        printf("  BVH8_CWBVH   %.1f MRays/s (kernel %.3f ms)\n", N / (gpu.LastKernelMs() * 1e3), gpu.LastKernelMs());
        bad += validate("BVH8_CWBVH", work, ref, N);
        // any-hit through the binding: a ray is occluded iff BVH::IsOccluded says so
        std::vector<uint8_t> occ(N);
        gpu.IsOccluded(rays, N, occ.data());
        unsigned occBad = 0;
        for (unsigned i = 0; i < N; i++) occBad += (occ[i] != 0) != refbvh.IsOccluded(rays[i]);
        printf("  %-12s IsOccluded mismatches %u\n", "BVH8_CWBVH", occBad);
        bad += occBad > 2;
    }
    {   // the reference's own flow for animated geometry — move the vertices, BVH::Refit on the host (tiny_bvh.h:3055-3093), BVH_GPU::ConvertFrom
        // again — handed to the engine IN PLACE (tinyhip::Scene::Update -> tbvh_update_bvh_gpu), and the speedtest's thread loop
        // (tiny_bvh_speedtest.cpp:1077-1083: 8 threads on one BVH) kept as it is: 4 host threads share the one Scene
        std::vector<bvhvec4> anim(tris);
        BVH_GPU g; g.Build(anim.data(), triCount);
        tinyhip::Scene gpu(g, anim.data());
        for (auto& v : anim) v.y += 0.04f * ext * sinf(v.x * (5.0f / ext));
        g.bvh.Refit();
        g.ConvertFrom(g.bvh, false);
        gpu.Update(g, anim.data());
        BVH movedBvh; movedBvh.Build(anim.data(), triCount);
        Ray* ref2 = (Ray*)malloc64(N * sizeof(Ray));
        memcpy((void*)ref2, (void*)rays, N * sizeof(Ray));
        for (unsigned i = 0; i < N; i++) movedBvh.Intersect(ref2[i]);
        const int T = 4;
        std::vector<Ray*> mine(T);
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) { mine[t] = (Ray*)malloc64(N * sizeof(Ray)); memcpy((void*)mine[t], (void*)rays, N * sizeof(Ray)); }
        for (int t = 0; t < T; t++) th.emplace_back([&, t] { gpu.Intersect(mine[t], N); });
        for (auto& x : th) x.join();
        for (int t = 0; t < T; t++) { bad += validate(t ? "  (thread)" : "host Refit + Update, 4 threads", mine[t], ref2, N); free64(mine[t]); }
        free64(ref2);
    }
    // ---- beyond the speedtest: the per-frame / build-time host work of a tinybvh user, moved to the GPU -------------
    {   // BVH8_CWBVH::ConvertFrom on the device: tinybvh builds the BVH2 (Build + Compact + SplitLeafs(3), what
        // BVH8_CWBVH::Build does before converting, tiny_bvh.h:5829-5835), the GPU collapses and encodes it
        BVH bvh2; bvh2.Build(tris.data(), triCount); bvh2.Compact(); bvh2.SplitLeafs(3);
        tbvh_scene* s = nullptr;
        CHECK(tbvh_convert_bvh2_device(ctx, bvh2.bvhNode, bvh2.usedNodes, bvh2.primIdx, bvh2.idxCount, tris.data(), triCount, 0, TBVH_LAYOUT_CWBVH, &s));
        const float msConv = tbvh_time_last_ms(ctx);
        memcpy((void*)work, (void*)rays, N * sizeof(Ray));
        CHECK(tbvh_intersect(s, work, N, sizeof(Ray)));
        printf("  ConvertFrom on the device (tinybvh BVH2 -> CWBVH) %.3f ms, then %.1f MRays/s\n", msConv, N / (tbvh_time_last_ms(ctx) * 1e3));
        bad += validate("converted", work, ref, N);
        // animate: move the vertices, BVH::Refit on the host gives the reference answer, tbvh_refit the device one
        std::vector<bvhvec4> moved(tris);
        for (auto& v : moved) v.y += 0.05f * ext * sinf(v.x * (6.0f / ext));
        CHECK(tbvh_refit(s, moved.data(), triCount, 0));
        const float msRefit = tbvh_time_last_ms(ctx);
        BVH movedBvh; movedBvh.Build(moved.data(), triCount);
        Ray* ref2 = (Ray*)malloc64(N * sizeof(Ray));
        memcpy((void*)ref2, (void*)rays, N * sizeof(Ray));
        for (unsigned i = 0; i < N; i++) movedBvh.Intersect(ref2[i]);
        memcpy((void*)work, (void*)rays, N * sizeof(Ray));
        CHECK(tbvh_intersect(s, work, N, sizeof(Ray)));
        printf("  tbvh_refit to moved vertices %.3f ms\n", msRefit);
        bad += validate("refitted", work, ref2, N);
        tbvh_free_scene(s);
        // build from scratch on the device (LBVH): same answers from a different tree
        CHECK(tbvh_build_device(ctx, moved.data(), triCount, 0, TBVH_LAYOUT_CWBVH, 3, &s));
        const float msBuild = tbvh_time_last_ms(ctx);
        memcpy((void*)work, (void*)rays, N * sizeof(Ray));
        CHECK(tbvh_intersect(s, work, N, sizeof(Ray)));
        printf("  tbvh_build_device %.3f ms, then %.1f MRays/s\n", msBuild, N / (tbvh_time_last_ms(ctx) * 1e3));
        bad += validate("device-built", work, ref2, N);
        tbvh_free_scene(s);
        free64(ref2);
    }
    {   // TLAS of tinybvh BLASInstances (tiny_bvh_gpu2.cpp:108-136): tinybvh builds frame 0, the device rebuilds frame 1
        BVH8_CWBVH blas; blas.Build(tris.data(), triCount);
        tbvh_scene* bs = nullptr;
        CHECK(tbvh_upload_cwbvh(ctx, blas.bvh8Data, blas.usedBlocks, blas.bvh8Tris, (uint64_t)blas.bvh8.idxCount * 3, &bs));
        BVH blasRef; blasRef.Build(tris.data(), triCount);       // the same mesh as a plain BVH for the reference TLAS query
        const int NI = 27;
        std::vector<BLASInstance> inst(NI);
        BVHBase* blasList[] = {&blasRef};
        auto place = [&](float t) {
            for (int i = 0; i < NI; i++) {
                inst[i] = BLASInstance(0);
                const float a = t + i * 0.7f, sc = 0.3f;
                bvhmat4& T = inst[i].transform;
                T[0] = sc * cosf(a); T[2] = sc * sinf(a); T[5] = sc; T[8] = -sc * sinf(a); T[10] = sc * cosf(a);
                T[3] = c.x + (i % 3 - 1) * 0.6f * ext; T[7] = c.y + (i / 3 % 3 - 1) * 0.6f * ext; T[11] = c.z + (i / 9 - 1) * 0.6f * ext;
            }
        };
        place(0.f);
        BVH_GPU tlas; tlas.Build(inst.data(), NI, blasList, 1);   // fills invTransform + world bounds, builds the TLAS
        tbvh_scene* ts = nullptr;
        CHECK(tbvh_upload_tlas(ctx, tlas.bvhNode, tlas.usedNodes, tlas.bvh.primIdx, tlas.bvh.idxCount, inst.data(), NI, &bs, 1, &ts));
        // frame 1: only the transforms change; the device updates the instances and rebuilds the TLAS
        place(0.9f);
        std::vector<float> xf((size_t)NI * 16);
        for (int i = 0; i < NI; i++) memcpy(&xf[(size_t)i * 16], &inst[i].transform, 64);
        const float bounds[6] = {blasRef.bvhNode[0].aabbMin.x, blasRef.bvhNode[0].aabbMin.y, blasRef.bvhNode[0].aabbMin.z,
                                 blasRef.bvhNode[0].aabbMax.x, blasRef.bvhNode[0].aabbMax.y, blasRef.bvhNode[0].aabbMax.z};
        CHECK(tbvh_rebuild_tlas_device(ts, xf.data(), 0, bounds, 1));
        memcpy((void*)work, (void*)rays, N * sizeof(Ray));
        CHECK(tbvh_intersect(ts, work, N, sizeof(Ray)));
        printf("  TLAS of %d tinybvh instances rebuilt on the device: %.1f MRays/s\n", NI, N / (tbvh_time_last_ms(ctx) * 1e3));
        // reference for frame 1: tinybvh's own TLAS build + BVH::Intersect through it
        BVH tlasRef; tlasRef.Build(inst.data(), NI, blasList, 1);
        Ray* ref3 = (Ray*)malloc64(N * sizeof(Ray));
        memcpy((void*)ref3, (void*)rays, N * sizeof(Ray));
        for (unsigned i = 0; i < N; i++) tlasRef.Intersect(ref3[i]);
        // the device update restates BLASInstance::Update operation for operation (records bit-identical to tinybvh's own),
        // and the traversal transforms the ray like IntersectTLAS does: bit-identical hits
        bad += validate("TLAS", work, ref3, N);
        unsigned instBad = 0;
        for (unsigned i = 0; i < N; i++) if (ref3[i].hit.t < BVH_FAR && work[i].hit.prim == ref3[i].hit.prim && work[i].hit.t == ref3[i].hit.t && work[i].hit.inst != ref3[i].hit.inst) instBad++;
        printf("  %-12s instance index mismatches %u\n", "TLAS", instBad);
        bad += instBad > 2;
        free64(ref3);
        tbvh_free_scene(ts); tbvh_free_scene(bs);
    }
    tbvh_shutdown(ctx);   // (every tinyhip::Scene above is out of scope)
    printf(bad ? "VALIDATION FAILED\n" : "all layouts agree with BVH::Intersect\n");
    return bad ? 1 : 0;
}
