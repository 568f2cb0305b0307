// wavefront_demos.cpp — the frame loops of the reference's two GPU path-tracing demos re-hosted on the HIP engine, with the
// REAL tiny_bvh.h building everything the demos build:
//
//   part 1  tiny_bvh_gpu.cpp   Init (:68-107) + Tick (:128-158): one BVH8_CWBVH over the scene + the light quad,
//           Generate, { Extend, Shade } x 3, Connect, Finalize per frame, accumulating samples
//   part 2  tiny_bvh_gpu2.cpp  Init (:82-143) + Tick (:187-198): a BVH8_CWBVH BLAS instanced in a grid under a TLAS
//           (BVH_GPU over BLASInstance records), rebuilt every frame as instances move (the animation of :171-186)
//
// What the demos do with tinyocl::Kernel / Buffer (nine kernels of wavefront.cl / wavefront2.cl, atomic counters, one
// Run() per stage) is ONE call here: tbvh_wavefront_render enqueues the whole frame with all queues on the device.
// No window: the program renders a few frames, checks that the image converges (frame N + 1 differs less from frame N
// than frame 1 did from frame 0, no NaNs, the light is visible where it should be) and prints the frame time.
// Needs the reference header at compile time (as a tinybvh user has it); built by __graft_entry__.build() where
// /root/reference exists:
//   g++ -std=c++20 -O3 -mavx2 -mfma -I/root/reference -Iinclude examples/wavefront_demos.cpp -Ltinybvh_amd -ltinybvh_amd \
//       -Wl,-rpath,'$ORIGIN/../../tinybvh_amd' -lpthread -o examples/_build/wavefront_demos
#define TINYBVH_IMPLEMENTATION
#include "tiny_bvh.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "tiny_hip.h"   // tinyhip::Scene / tinyhip::PathTracer (include/tiny_hip.h) over the C ABI of tinybvh_amd.h

using namespace tinybvh;

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, tbvh_last_error()); return 1; } } while (0)

static const unsigned W = 640, H = 320;   // the demos run 1280 x 720 / 1600 x 800; multiples of 4 either way

static float asFloat(unsigned u) { float f; memcpy(&f, &u, 4); return f; }

// scene geometry in the demos' vertex format: 3 x bvhvec4 per triangle, material in w of the first vertex (type << 24 | RGB8)
static void addQuad(std::vector<bvhvec4>& v, bvhvec3 a, bvhvec3 b, bvhvec3 c, bvhvec3 d, unsigned mat) {
    const float m = asFloat(mat);
    for (bvhvec3 p : {a, b, c, a, c, d}) v.push_back(bvhvec4(p, m));
}
static void addBox(std::vector<bvhvec4>& v, bvhvec3 lo, bvhvec3 hi, unsigned mat) {
    const bvhvec3 p[8] = {{lo.x, lo.y, lo.z}, {hi.x, lo.y, lo.z}, {hi.x, hi.y, lo.z}, {lo.x, hi.y, lo.z}, {lo.x, lo.y, hi.z}, {hi.x, lo.y, hi.z}, {hi.x, hi.y, hi.z}, {lo.x, hi.y, hi.z}};
    const int f[6][4] = {{0, 1, 2, 3}, {5, 4, 7, 6}, {4, 0, 3, 7}, {1, 5, 6, 2}, {3, 2, 6, 7}, {4, 5, 1, 0}};
    for (auto& q : f) addQuad(v, p[q[0]], p[q[1]], p[q[2]], p[q[3]], mat);
}
static void addBlob(std::vector<bvhvec4>& v, bvhvec3 c, float r, int nu, int nv, unsigned mat) {
    auto P = [&](int i, int j) {
        const float th = 3.14159265f * j / nv, ph = 6.2831853f * i / nu, rr = r * (1 + 0.15f * sinf(5 * ph) * sinf(3 * th));
        return c + bvhvec3(rr * sinf(th) * cosf(ph), rr * cosf(th), rr * sinf(th) * sinf(ph));
    };
    for (int i = 0; i < nu; i++) for (int j = 0; j < nv; j++) addQuad(v, P(i, j), P(i + 1, j), P(i + 1, j + 1), P(i, j + 1), mat);
}

struct Camera { bvhvec3 eye, view; };
static tbvh_camera makeCamera(const Camera& c) {   // UpdateCamera of the demos (tiny_bvh_gpu.cpp:110-125)
    const bvhvec3 right = tinybvh_normalize(tinybvh_cross(bvhvec3(0, 1, 0), c.view)), up = 0.8f * tinybvh_cross(c.view, right), C = c.eye + 1.2f * c.view;
    const bvhvec3 p0 = C - right + up, p1 = C + right + up, p2 = C - right - up;
    tbvh_camera cam;
    memcpy(cam.eye, &c.eye, 12); memcpy(cam.p1, &p0, 12); memcpy(cam.p2, &p1, 12); memcpy(cam.p3, &p2, 12);
    cam.width = W; cam.height = H; cam.spp_x = cam.spp_y = 1;
    return cam;
}

// the demos' light: a 9 x 5 quad at (-22, 12, 2), emission (25, 25, 22) (tiny_bvh_gpu.cpp:94, wavefront.cl:21-22)
static tbvh_wf_params demoParams(unsigned frame) {
    tbvh_wf_params p;
    memset(&p, 0, sizeof p);
    p.light_pos[0] = -22; p.light_pos[1] = 12; p.light_pos[2] = 2;
    p.light_color[0] = 25; p.light_color[1] = 25; p.light_color[2] = 22;
    for (int i = 0; i < 3; i++) p.sky_lo[i] = p.sky_hi[i] = i == 2 ? 1.2f : 0.7f;   // wavefront.cl:153
    p.eps = 1e-4f; p.max_depth = 3; p.seed = (frame + 1) * 19191u; p.clear = frame == 0;
    p.light_size[0] = 9; p.light_size[1] = 5;
    p.flags = TBVH_WF_ONE_DIFFUSE_BOUNCE;
    p.sample_index = frame;   // spp - 1
    return p;
}

static double meanAbsDiff(const std::vector<float>& a, float sa, const std::vector<float>& b, float sb) {
    double s = 0;
    for (size_t i = 0; i < a.size(); i += 4) for (int k = 0; k < 3; k++) s += fabs((double)a[i + k] * sa - (double)b[i + k] * sb);
    return s / (a.size() / 4 * 3);
}

int main() {
    tbvh_context* ctx = nullptr;
    CHECK(tbvh_init(0, &ctx));
    tbvh_wavefront* wf = nullptr;
    CHECK(tbvh_wavefront_create(ctx, W, H, &wf));
    std::vector<float> img((size_t)W * H * 4), prev, first;
    std::vector<uint32_t> pixels((size_t)W * H);
    int bad = 0;

    // ---- part 1: tiny_bvh_gpu.cpp -------------------------------------------------------------------------------------------
    {
        std::vector<bvhvec4> tris;
        addQuad(tris, bvhvec3(-26.5f, 12, -0.5f), bvhvec3(-17.5f, 12, -0.5f), bvhvec3(-17.5f, 12, 4.5f), bvhvec3(-26.5f, 12, 4.5f), 0x1ffffff);   // AddQuad( (-22,12,2), 9, 5, 0x1ffffff )
        addQuad(tris, bvhvec3(-40, 0, -16), bvhvec3(36, 0, -16), bvhvec3(36, 0, 16), bvhvec3(-40, 0, 16), 0xc0c0c0);                                 // floor
        addBox(tris, bvhvec3(-40, 0, -17), bvhvec3(36, 20, -16), 0xc08040); addBox(tris, bvhvec3(-40, 0, 16), bvhvec3(36, 20, 17), 0x4080c0);        // two walls
        for (int i = 0; i < 8; i++) addBox(tris, bvhvec3(-34.f + 8 * i, 0, -9), bvhvec3(-32.5f + 8 * i, 14, -7.5f), 0xe0e0e0);                       // columns
        addBlob(tris, bvhvec3(-20, 3, 3), 2.5f, 48, 32, 0xd0d0ff);
        addBox(tris, bvhvec3(-16, 0, -2), bvhvec3(-12, 6, 2), 0x2ffffff);                                                                             // a mirror block (MATERIAL_SPECULAR)
        const unsigned triCount = (unsigned)tris.size() / 3;
        BVH8_CWBVH bvh;                                   // "build bvh (here: 'compressed wide bvh', for efficient GPU rendering)" (:95-96)
        bvh.Build(tris.data(), triCount);
        tbvh_scene* scene = nullptr;
        CHECK(tbvh_upload_cwbvh(ctx, bvh.bvh8Data, bvh.usedBlocks, bvh.bvh8Tris, (uint64_t)bvh.bvh8.idxCount * 3, &scene));   // cwbvhNodes / cwbvhTris (:98-101)
        void* dVerts = nullptr;                           // triData (:102-103)
        CHECK(tbvh_device_malloc(ctx, (uint64_t)triCount * 48, &dVerts));
        CHECK(tbvh_copy_to_device(ctx, dVerts, tris.data(), (uint64_t)triCount * 48));
        const tbvh_camera cam = makeCamera({bvhvec3(-2, 9, 11), tinybvh_normalize(bvhvec3(-0.86f, -0.22f, -0.46f))});
        const unsigned frames = 24;
        double d01 = 0, dLast = 0; float ms = 0;
        for (unsigned f = 0; f < frames; f++) {           // Tick (:128-158)
            const tbvh_wf_params p = demoParams(f);
            tbvh_wf_stats st;
            CHECK(tbvh_wavefront_render(wf, scene, dVerts, &cam, &p, &st));
            ms = st.frame_ms;
            CHECK(tbvh_wavefront_read(wf, img.data()));
            if (f == 1) d01 = meanAbsDiff(img, 1.f / 2, prev, 1.f);
            if (f == frames - 1) dLast = meanAbsDiff(img, 1.f / frames, prev, 1.f / (frames - 1));
            prev = img;
            if (f == 0) first = img;
        }
        CHECK(tbvh_wavefront_finalize(wf, 1.0f / frames, pixels.data()));   // finalize->Run2D, pixels->CopyFromDevice (:155-157)
        double lum = 0; unsigned nan = 0, white = 0;
        for (size_t i = 0; i < img.size(); i += 4) { for (int k = 0; k < 3; k++) { if (!std::isfinite(img[i + k])) nan++; lum += img[i + k] / frames; } }
        for (uint32_t px : pixels) if ((px & 0xffffff) == 0xffffff) white++;
        lum /= (double)W * H * 3;
        printf("tiny_bvh_gpu frame loop: %u triangles (tiny_bvh.h %d.%d.%d BVH8_CWBVH::Build), %u x %u, %u frames, %.3f ms per frame; mean radiance %.4f, "
               "frame-to-frame change %.5f -> %.5f, %u saturated pixels\n", triCount, TINY_BVH_VERSION_MAJOR, TINY_BVH_VERSION_MINOR, TINY_BVH_VERSION_SUB, W, H, frames, ms, lum, d01, dLast, white);
        if (nan || !(lum > 0.02) || !(dLast < 0.25 * d01) || white == 0) { printf("  part 1: image check FAILED (nan %u)\n", nan); bad++; }
        // the centre ray through BVH::Intersect of the real library hits what the engine's first Extend hit: same t through tbvh_intersect
        Ray r(bvhvec3(cam.eye[0], cam.eye[1], cam.eye[2]), tinybvh_normalize(bvhvec3(-0.86f, -0.22f, -0.46f)));
        Ray viaEngine = r;
        bvh.bvh8.bvh.Intersect(r);                        // the demo's mouse-pick ray (:163-165)
        CHECK(tbvh_intersect(scene, &viaEngine, 1, sizeof(Ray)));
        if (r.hit.prim != viaEngine.hit.prim || r.hit.t != viaEngine.hit.t) { printf("  part 1: pick ray differs: t %f prim %u vs t %f prim %u\n", r.hit.t, r.hit.prim, viaEngine.hit.t, viaEngine.hit.prim); bad++; }
        // ---- part 1b: the same frame loop over SEVERAL devices (the reference has one process-global OpenCL device, tiny_ocl.h:362-364) ----
        // tinyhip::PathTracer cuts the image into bands of rows, one per device — every device of the box, and at least two contexts (on a
        // 1-GPU box both bands live on device 0: the band arithmetic, the per-band queues and the gather run as they do on 8 GPUs) —, each band
        // generated, traced, shaded and accumulated where it lives.  The bands draw the random numbers the full frame draws for their pixels:
        // the gathered image must be the single-device image up to the order of the float accumulations.
        {
            const int nDev = tinyhip::DeviceCount();
            const int nBands = nDev > 2 ? nDev : 2;
            std::vector<tbvh_context*> bandCtx(nBands, nullptr);
            std::vector<tbvh_scene*> bandScene(nBands, nullptr);
            std::vector<tbvh_wavefront*> bandWf(nBands, nullptr);
            std::vector<void*> bandVerts(nBands, nullptr);
            uint32_t row = 0;
            for (int i = 0; i < nBands; i++) {
                CHECK(tbvh_init(i % nDev, &bandCtx[i]));                 // a context of its own per band (device i, or device 0 again)
                CHECK(tbvh_upload_cwbvh(bandCtx[i], bvh.bvh8Data, bvh.usedBlocks, bvh.bvh8Tris, (uint64_t)bvh.bvh8.idxCount * 3, &bandScene[i]));
                const uint32_t next = (uint32_t)((uint64_t)(H / 4) * (i + 1) / nBands) * 4;
                CHECK(tbvh_wavefront_create(bandCtx[i], W, next - row, &bandWf[i]));
                CHECK(tbvh_wavefront_set_band(bandWf[i], row, H));
                CHECK(tbvh_device_malloc(bandCtx[i], (uint64_t)triCount * 48, &bandVerts[i]));
                CHECK(tbvh_copy_to_device(bandCtx[i], bandVerts[i], tris.data(), (uint64_t)triCount * 48));
                row = next;
            }
            std::vector<float> banded((size_t)W * H * 4), dispatch(nBands);
            std::vector<tbvh_wf_stats> bst(nBands);
            float slowest = 0;
            for (unsigned f = 0; f < frames; f++) {
                const tbvh_wf_params p = demoParams(f);
                CHECK(tbvh_wavefront_render_sharded(bandWf.data(), bandScene.data(), (const void* const*)bandVerts.data(), (uint32_t)nBands, &cam, &p, bst.data(), dispatch.data()));
                slowest = 0;
                for (auto& st : bst) slowest = st.frame_ms > slowest ? st.frame_ms : slowest;
            }
            CHECK(tbvh_wavefront_read_sharded(bandWf.data(), (uint32_t)nBands, banded.data()));
            double diff = 0, ref = 0;
            for (size_t i = 0; i < img.size(); i++) { diff += fabs((double)banded[i] - (double)img[i]); ref += fabs((double)img[i]); }
            float gap = 0;
            for (float d : dispatch) gap += d;
            printf("  the same %u frames over %d context(s) on %d device(s), %d bands of rows: slowest band %.3f ms per frame, host dispatch %.3f ms per frame; "
                   "relative difference of the gathered image from the single-device one %.2e\n", frames, nBands, nDev, nBands, slowest, gap, diff / ref);
            if (!(diff <= 1e-5 * ref)) { printf("  part 1b: the banded image differs\n"); bad++; }
            for (int i = 0; i < nBands; i++) { tbvh_wavefront_destroy(bandWf[i]); tbvh_device_free(bandCtx[i], bandVerts[i]); tbvh_free_scene(bandScene[i]); tbvh_shutdown(bandCtx[i]); }
        }
        // ... and through the binding (include/tiny_hip.h): one tinyhip::Scene per device of the box, tinyhip::PathTracer over them
        {
            const int nDev = tinyhip::DeviceCount();
            std::vector<tinyhip::Scene*> perDevice;
            for (int d = 0; d < nDev; d++) perDevice.push_back(new tinyhip::Scene(bvh, d));
            {
                tinyhip::PathTracer pt(perDevice, tris.data(), tris.size(), W, H);
                for (unsigned f = 0; f < frames; f++) pt.Render(cam, demoParams(f));
                std::vector<float> viaBinding((size_t)W * H * 4);
                pt.Read(viaBinding.data());
                double diff = 0, ref = 0;
                for (size_t i = 0; i < img.size(); i++) { diff += fabs((double)viaBinding[i] - (double)img[i]); ref += fabs((double)img[i]); }
                printf("  tinyhip::PathTracer over the box's %d device(s), %u band(s): relative difference %.2e\n", nDev, pt.Bands(), diff / ref);
                if (!(diff <= 1e-5 * ref)) { printf("  part 1b: the image rendered through tiny_hip.h differs\n"); bad++; }
            }
            for (auto* sPtr : perDevice) delete sPtr;
        }
        tbvh_device_free(ctx, dVerts);
        tbvh_free_scene(scene);
    }

    // ---- part 2: tiny_bvh_gpu2.cpp ------------------------------------------------------------------------------------------
    {
        const int GRID = 6, COUNT = GRID * GRID * GRID;
        std::vector<bvhvec4> verts;                       // the instanced mesh ("dragon" there)
        addBlob(verts, bvhvec3(0), 4.5f, 64, 48, 0xc8c8c8);
        const unsigned triCount = (unsigned)verts.size() / 3;
        std::vector<bvhvec4> env;                         // a second BLAS: ground + light quad (the demo's BLAS 0 slot)
        addQuad(env, bvhvec3(-26.5f, 12, -0.5f), bvhvec3(-17.5f, 12, -0.5f), bvhvec3(-17.5f, 12, 4.5f), bvhvec3(-26.5f, 12, 4.5f), 0x1ffffff);
        addQuad(env, bvhvec3(-60, -1, -60), bvhvec3(60, -1, -60), bvhvec3(60, -1, 60), bvhvec3(-60, -1, 60), 0xb0b0b0);
        BVH8_CWBVH dragon, ground;
        dragon.BuildHQ(verts.data(), triCount);           // dragon.BuildHQ( verts, triCount ) (:104-105)
        ground.Build(env.data(), (unsigned)env.size() / 3);
        tbvh_scene *blasDragon = nullptr, *blasGround = nullptr, *tlasScene = nullptr;
        CHECK(tbvh_upload_cwbvh(ctx, ground.bvh8Data, ground.usedBlocks, ground.bvh8Tris, (uint64_t)ground.bvh8.idxCount * 3, &blasGround));
        CHECK(tbvh_upload_cwbvh(ctx, dragon.bvh8Data, dragon.usedBlocks, dragon.bvh8Tris, (uint64_t)dragon.bvh8.idxCount * 3, &blasDragon));
        void *dDragon = nullptr, *dGround = nullptr;
        CHECK(tbvh_device_malloc(ctx, verts.size() * 16, &dDragon)); CHECK(tbvh_copy_to_device(ctx, dDragon, verts.data(), verts.size() * 16));
        CHECK(tbvh_device_malloc(ctx, env.size() * 16, &dGround)); CHECK(tbvh_copy_to_device(ctx, dGround, env.data(), env.size() * 16));
        const void* blasVerts[2] = {dGround, dDragon};
        CHECK(tbvh_wavefront_set_blas_vertices(wf, blasVerts, 2));
        BVHBase* blasList[2] = {&ground, &dragon};        // blasList (:44)
        std::vector<BLASInstance> instance(COUNT + 1);
        BVH_GPU tlas;
        tbvh_scene* blasScenes[2] = {blasGround, blasDragon};
        const tbvh_camera cam = makeCamera({bvhvec3(-14, 9, -12), tinybvh_normalize(bvhvec3(0.55f, -0.35f, 0.6f))});
        const unsigned frames = 6;
        float msTrace = 0; double lum = 0; unsigned nan = 0;
        for (unsigned f = 0; f < frames; f++) {
            // the dragon grid (:108-115), animated: every frame the instances move, the TLAS is rebuilt on the host by tiny_bvh.h and re-uploaded
            instance[0] = BLASInstance(0);
            for (int b = 1, x = 0; x < GRID; x++) for (int y = 0; y < GRID; y++) for (int z = 0; z < GRID; z++, b++) {
                instance[b] = BLASInstance(1);
                BLASInstance& i = instance[b];
                const float s = 0.07f * (1.0f + 0.2f * sinf(0.7f * f + b));
                i.transform[0] = i.transform[5] = i.transform[10] = s;
                i.transform[3] = (float)x - 2.5f, i.transform[7] = (float)y + 0.3f * sinf(0.5f * f + x), i.transform[11] = (float)z - 2.5f;
            }
            tlas.Build(instance.data(), COUNT + 1, blasList, 2);   // tlas.Build( instance, DRAGONS, blasList, 2 ) (:118)
            // tlasNodes / tlasIndices / blasInstances buffers + CopyToDevice (:121-127)
            if (!tlasScene) CHECK(tbvh_upload_tlas(ctx, tlas.bvhNode, tlas.usedNodes, tlas.bvh.primIdx, tlas.bvh.idxCount, instance.data(), COUNT + 1, blasScenes, 2, &tlasScene));
            else CHECK(tbvh_update_tlas(tlasScene, tlas.bvhNode, tlas.usedNodes, tlas.bvh.primIdx, tlas.bvh.idxCount, instance.data(), COUNT + 1));
            tbvh_wf_params p = demoParams(0);             // the scene moves: every frame starts a new accumulation, as the demo does when the camera moves
            p.seed = (f + 1) * 19191u;
            tbvh_wf_stats st;
            CHECK(tbvh_wavefront_render(wf, tlasScene, nullptr, &cam, &p, &st));
            msTrace = st.frame_ms;
        }
        CHECK(tbvh_wavefront_read(wf, img.data()));
        unsigned lit = 0;
        for (size_t i = 0; i < img.size(); i += 4) { for (int k = 0; k < 3; k++) { if (!std::isfinite(img[i + k])) nan++; lum += img[i + k]; } if (img[i] + img[i + 1] + img[i + 2] > 0.05f) lit++; }
        lum /= (double)W * H * 3;
        // the engine's TLAS traversal against the real library's: the centre ray through BVH::IntersectTLAS of tiny_bvh.h
        BVH cpuTlas;
        BVH dragonBvh, groundBvh;
        dragonBvh.Build(verts.data(), triCount); groundBvh.Build(env.data(), (unsigned)env.size() / 3);
        BVHBase* cpuList[2] = {&groundBvh, &dragonBvh};
        cpuTlas.Build(instance.data(), COUNT + 1, cpuList, 2);
        unsigned agree = 0, total = 0;
        std::vector<Ray> rays, viaEngine;
        for (unsigned y = 8; y < H; y += 16) for (unsigned x = 8; x < W; x += 16) {
            const float u = (float)x / W, v = (float)y / H;
            const bvhvec3 e(cam.eye[0], cam.eye[1], cam.eye[2]), a(cam.p1[0], cam.p1[1], cam.p1[2]), b(cam.p2[0], cam.p2[1], cam.p2[2]), c(cam.p3[0], cam.p3[1], cam.p3[2]);
            rays.push_back(Ray(e, tinybvh_normalize(a + u * (b - a) + v * (c - a) - e)));
        }
        viaEngine = rays;
        for (auto& r : rays) cpuTlas.Intersect(r);
        CHECK(tbvh_intersect(tlasScene, viaEngine.data(), viaEngine.size(), sizeof(Ray)));
        for (size_t i = 0; i < rays.size(); i++) {
            total++;
            const bool h1 = rays[i].hit.t < BVH_FAR, h2 = viaEngine[i].hit.t < BVH_FAR;
            if (h1 == h2 && (!h1 || (rays[i].hit.prim == viaEngine[i].hit.prim && rays[i].hit.inst == viaEngine[i].hit.inst && fabsf(rays[i].hit.t - viaEngine[i].hit.t) <= 1e-5f * rays[i].hit.t))) agree++;
        }
        printf("tiny_bvh_gpu2 frame loop: %d instances of a %u-triangle BLAS (BuildHQ) + ground under a tiny_bvh.h TLAS rebuilt per frame, %u frames, %.3f ms per frame; "
               "mean radiance %.4f, %u lit pixels; %u of %u probe rays agree with BVH::Intersect over the CPU TLAS\n", COUNT, triCount, frames, msTrace, lum, lit, agree, total);
        if (nan || !(lum > 0.02) || lit < W * H / 4 || agree + 2 < total) { printf("  part 2: check FAILED (nan %u)\n", nan); bad++; }
        tbvh_free_scene(tlasScene); tbvh_free_scene(blasDragon); tbvh_free_scene(blasGround);
        tbvh_device_free(ctx, dDragon); tbvh_device_free(ctx, dGround);
    }
    tbvh_wavefront_destroy(wf);
    tbvh_shutdown(ctx);
    printf(bad ? "WAVEFRONT DEMOS FAILED\n" : "wavefront demos ok\n");
    return bad ? 1 : 0;
}
