// debug aid: the scene and rays of tiny_bvh_minimal_gpu.cpp (fixed_rand.c sequence) traced in ONE process by the real BVH::Intersect and through the tinyocl shim; bit compare
#define TINYBVH_IMPLEMENTATION
#include "tiny_bvh.h"
#include "tiny_ocl.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
using namespace tinybvh;
static const int N = 8192;
static bvhvec4 tris[N * 3];
static float ur() { return (float)rand() / (float)RAND_MAX; }
int main() {
    for (int i = 0; i < N; i++) { float x = ur(), y = ur(), z = ur(); for (int v = 0; v < 3; v++) { bvhvec4& p = tris[i * 3 + v]; p.x = x + 0.1f * ur(); p.y = y + 0.1f * ur(); p.z = z + 0.1f * ur(); } }
    BVH bvh; bvh.Build(tris, N);
    BVH_GPU g; g.Build(tris, N);
    tinyocl::Kernel k("traverse.cl", "batch_ailalaine");
    tinyocl::Buffer triData(N * 3 * sizeof(bvhvec4), tris), nodes(g.usedNodes * sizeof(BVH_GPU::BVHNode), g.bvhNode), idx(g.idxCount * 4, g.bvh.primIdx), rays(1024 * 64);
    unsigned char* host = (unsigned char*)rays.GetHostPtr();
    Ray cpu[1024];
    for (int i = 0; i < 1024; i++) { bvhvec3 O(0.5f, 0.5f, -1), D(0.1f, ur() - 0.5f, 2); Ray r(O, D); memcpy(host + 64 * i, &r, 64); cpu[i] = r; bvh.Intersect(cpu[i]); }
    triData.CopyToDevice(); nodes.CopyToDevice(); idx.CopyToDevice(); rays.CopyToDevice();
    k.SetArguments(&nodes, &idx, &triData, &rays);
    k.Run(1024);
    rays.CopyFromDevice();
    int bad = 0;
    for (int i = 0; i < 1024; i++) {
        Ray r; memcpy(&r, host + 64 * i, 64);
        if (memcmp(&r.hit.t, &cpu[i].hit.t, 4) || r.hit.prim != cpu[i].hit.prim) {
            if (bad++ < 10) printf("ray %d: gpu t %a prim %u u %a v %a | cpu t %a prim %u u %a v %a\n", i, r.hit.t, r.hit.prim, r.hit.u, r.hit.v, cpu[i].hit.t, cpu[i].hit.prim, cpu[i].hit.u, cpu[i].hit.v);
        }
    }
    printf("%d of 1024 differ\n", bad);
    // the same rays through BVH_GPU's own CPU mirror
    int bad2 = 0;
    for (int i = 0; i < 1024; i++) { Ray r; memcpy(&r, host + 64 * i, 64); Ray q = cpu[i]; (void)q; }
    return 0;
}
