// minimal_gpu.cpp — the shape of the reference's tiny_bvh_minimal_gpu.cpp (8192 random
// triangles, BVH_GPU layout, 1024 rays, print the nearest hit of the centre ray) on the HIP
// engine, using only the C ABI.  Builds without the reference: the layout comes from the
// library's own host builder.  With tiny_bvh.h available the three tbvh_host_* lines become
// `tinybvh::BVH_GPU bvh; bvh.Build(tris, N);` + tbvh_upload_bvh_gpu(...) — see INTEGRATION.md.
//
//   g++ -O2 -Iinclude examples/minimal_gpu.cpp -Ltinybvh_amd -ltinybvh_amd -Wl,-rpath,$PWD/tinybvh_amd -o examples/_build/minimal_gpu
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "tinybvh_amd.h"

struct Vec4 { float x, y, z, w; };
struct Ray64 { float O[3]; uint32_t mask; float D[3]; uint32_t instIdx; float rD[3]; uint32_t inst; float t, u, v; uint32_t prim; };
static_assert(sizeof(Ray64) == 64, "64-byte ray record");

static uint32_t rng = 0x12345;
static float uniform_rand() { rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5; return (float)(rng >> 8) * (1.0f / 16777216.0f); }
static float safercp(float x) { return (x > 1e-12f || x < -1e-12f) ? 1.0f / x : (x >= 0 ? 1e30f : -1e30f); }

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, tbvh_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    const int N = 8192;
    std::vector<Vec4> tris(N * 3);
    for (int i = 0; i < N; i++) {
        const float x = uniform_rand(), y = uniform_rand(), z = uniform_rand();
        for (int k = 0; k < 3; k++) tris[i * 3 + k] = Vec4{x + 0.1f * uniform_rand(), y + 0.1f * uniform_rand(), z + 0.1f * uniform_rand(), 0};
    }
    const int layout = argc > 1 ? atoi(argv[1]) : TBVH_LAYOUT_BVH_GPU;
    tbvh_hostbvh* host = nullptr;
    CHECK(tbvh_host_build(tris.data(), N, layout, nullptr, &host));
    tbvh_context* ctx = nullptr;
    CHECK(tbvh_init(0, &ctx));
    tbvh_scene* scene = nullptr;
    CHECK(tbvh_upload_host(ctx, host, tris.data(), N, &scene));
    // 32 x 32 rays from (0.5, 0.5, -1) through a unit square at z = 0
    std::vector<Ray64> rays(1024);
    for (int i = 0; i < 1024; i++) {
        Ray64& r = rays[i];
        memset(&r, 0, sizeof r);
        const float px = (float)(i & 31) / 32.0f, py = (float)(i >> 5) / 32.0f;
        float d[3] = {px - 0.5f, py - 0.5f, 1.0f};
        const float l = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        r.O[0] = 0.5f; r.O[1] = 0.5f; r.O[2] = -1.0f; r.mask = 0xFFFF;
        for (int a = 0; a < 3; a++) { r.D[a] = d[a] * l; r.rD[a] = safercp(r.D[a]); }
        r.t = 1e30f;
    }
    CHECK(tbvh_intersect(scene, rays.data(), rays.size(), sizeof(Ray64)));
    int hits = 0; double sum = 0;
    for (const Ray64& r : rays) if (r.t < 1e30f) hits++, sum += r.t;
    const Ray64& c = rays[16 * 32 + 16];
    printf("layout %d: %d of 1024 rays hit, mean t %.6f, centre ray t %.6f prim %u, kernel %.3f ms\n", layout, hits, hits ? sum / hits : 0.0, c.t, c.prim,
           tbvh_time_last_ms(ctx));
    tbvh_free_scene(scene);
    tbvh_shutdown(ctx);
    tbvh_host_free(host);
    return hits > 0 ? 0 : 2;
}
