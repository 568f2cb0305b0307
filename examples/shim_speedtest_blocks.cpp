// shim_speedtest_blocks.cpp — the three GPU blocks of the reference's speedtest (tiny_bvh_speedtest.cpp:1092-1241: BVH_GPU, BVH4_GPU, BVH8_CWBVH;
// the reference #undefs them on Linux, :86-92) written with tinyocl's OWN names and statement order — Buffer, CopyToDevice, Kernel( "traverse.cl",
// entry ), SetArguments, Run( N, 64, 0, &event ), clWaitForEvents, clGetEventProfilingInfo, CopyFromDevice — against include/shim/tiny_ocl.h, i.e.
// on the HIP engine.  The layouts are built by the REAL tiny_bvh.h (BuildHQ, as the speedtest does); every block's hit distances are validated
// against tinybvh::BVH::Intersect the way ValidateTraceResult does (:352-377), but exactly (bit-equal t), not to 1 %.
//
//   g++ -std=c++20 -O3 -mavx2 -mfma -Iinclude/shim -Iinclude -I<tinybvh checkout> examples/shim_speedtest_blocks.cpp -Ltinybvh_amd -ltinybvh_amd
#define TINYBVH_IMPLEMENTATION
#include "tiny_bvh.h"
#include "tiny_ocl.h"   // include/shim/tiny_ocl.h: put include/shim before the tinybvh checkout on the include path

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace tinybvh;

static uint32_t rng = 0x2468ace1u;
static float rnd() { rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5; return (float)(rng >> 8) * (1.0f / 16777216.0f); }

static int validate(const char* what, const std::vector<float>& ref, tinyocl::Buffer& rayData, unsigned N) {
    int bad = 0;
    for (unsigned i = 0; i < N; i++) {
        Ray r;
        memcpy(&r, (unsigned char*)rayData.GetHostPtr() + 64 * i, 64);
        if (r.hit.t != ref[i]) bad++;
    }
    printf("%s: %d of %u rays differ from BVH::Intersect\n", what, bad, N);
    return bad;
}

int main() {
    const int tricount = 60000;
    bvhvec4* triangles = (bvhvec4*)malloc64(tricount * 3 * sizeof(bvhvec4));
    for (int i = 0; i < tricount; i++) {
        const float x = rnd() * 10, y = rnd() * 4, z = rnd() * 10;
        for (int v = 0; v < 3; v++) triangles[i * 3 + v] = bvhvec4(x + 0.3f * rnd(), y + 0.3f * rnd(), z + 0.3f * rnd(), 0);
    }
    const unsigned side = 512, Nfull = side * side;
    Ray* fullBatch = (Ray*)malloc64(Nfull * sizeof(Ray));
    for (unsigned y = 0; y < side; y++) for (unsigned x = 0; x < side; x++) {
        const bvhvec3 O(5.0f, 2.0f, -6.0f), D((x + 0.5f) / side - 0.5f, (y + 0.5f) / side - 0.5f, 1.0f);
        new (&fullBatch[y * side + x]) Ray(O, D);
    }
    // the CPU reference pass (tiny_bvh_speedtest.cpp:1077-1090)
    BVH bvh;
    bvh.BuildHQ(triangles, tricount);
    std::vector<float> refDist(Nfull);
    for (unsigned i = 0; i < Nfull; i++) { Ray r = fullBatch[i]; bvh.Intersect(r); refDist[i] = r.hit.t; }

    tinyocl::Kernel ailalaine_kernel("traverse.cl", "batch_ailalaine");
    tinyocl::Kernel gpu4way_kernel("traverse.cl", "batch_gpu4way");
    tinyocl::Kernel cwbvh_kernel("traverse.cl", "batch_cwbvh");
    int bad = 0;
    cl_event event;
    cl_ulong startTime, endTime;
    {   // ---- GPU_2WAY (:1094-1141)
        BVH_GPU* bvh_gpu = new BVH_GPU();
        bvh_gpu->BuildHQ(triangles, tricount);
        tinyocl::Buffer gpuNodes(bvh_gpu->usedNodes * sizeof(BVH_GPU::BVHNode), bvh_gpu->bvhNode);
        tinyocl::Buffer idxData(bvh_gpu->idxCount * sizeof(unsigned), bvh_gpu->bvh.primIdx);
        tinyocl::Buffer triData(bvh_gpu->triCount * 3 * sizeof(tinybvh::bvhvec4), triangles);
        gpuNodes.CopyToDevice(); idxData.CopyToDevice(); triData.CopyToDevice();
        tinyocl::Buffer rayData(Nfull * 64);
        for (unsigned i = 0; i < Nfull; i++) memcpy((unsigned char*)rayData.GetHostPtr() + 64 * i, &fullBatch[i], 64);
        rayData.CopyToDevice();
        float traceTime = 0;
        ailalaine_kernel.SetArguments(&gpuNodes, &idxData, &triData, &rayData);
        for (int pass = 0; pass < 9; pass++) {
            ailalaine_kernel.Run(Nfull, 64, 0, &event);
            clWaitForEvents(1, &event);
            clGetEventProfilingInfo(event, CL_PROFILING_COMMAND_START, sizeof(cl_ulong), &startTime, 0);
            clGetEventProfilingInfo(event, CL_PROFILING_COMMAND_END, sizeof(cl_ulong), &endTime, 0);
            if (pass == 0) continue;
            traceTime += (endTime - startTime) * 1e-9f;
        }
        rayData.CopyFromDevice();
        traceTime /= 8.0f;
        printf("- BVH_GPU     - primary: %7.2fMRays/s\n", (float)Nfull / traceTime * 1e-6f);
        if (!(traceTime > 0)) bad++;
        bad += validate("BVH_GPU", refDist, rayData, Nfull);
        delete bvh_gpu;
    }
    {   // ---- GPU_4WAY (:1145-1190)
        BVH4_GPU* bvh4_gpu = new BVH4_GPU();
        bvh4_gpu->BuildHQ(triangles, tricount);
        tinyocl::Buffer gpu4Nodes(bvh4_gpu->usedBlocks * sizeof(tinybvh::bvhvec4), bvh4_gpu->bvh4Data);
        gpu4Nodes.CopyToDevice();
        tinyocl::Buffer rayData(Nfull * 64, 0);
        for (unsigned i = 0; i < Nfull; i++) memcpy((unsigned char*)rayData.GetHostPtr() + 64 * i, &fullBatch[i], 64);
        rayData.CopyToDevice();
        float traceTime = 0;
        gpu4way_kernel.SetArguments(&gpu4Nodes, &rayData);
        for (int pass = 0; pass < 9; pass++) {
            gpu4way_kernel.Run(Nfull, 64, 0, &event);
            clWaitForEvents(1, &event);
            clGetEventProfilingInfo(event, CL_PROFILING_COMMAND_START, sizeof(cl_ulong), &startTime, 0);
            clGetEventProfilingInfo(event, CL_PROFILING_COMMAND_END, sizeof(cl_ulong), &endTime, 0);
            if (pass == 0) continue;
            traceTime += (endTime - startTime) * 1e-9f;
        }
        rayData.CopyFromDevice();
        traceTime /= 8.0f;
        printf("- BVH4_GPU    - primary: %7.2fMRays/s\n", (float)Nfull / traceTime * 1e-6f);
        bad += validate("BVH4_GPU", refDist, rayData, Nfull);
        delete bvh4_gpu;
    }
    {   // ---- GPU_CWBVH (:1192-1241)
        BVH8_CWBVH* cwbvh = new BVH8_CWBVH();
        cwbvh->BuildHQ(triangles, tricount);
        tinyocl::Buffer cwbvhNodes(cwbvh->usedBlocks * sizeof(tinybvh::bvhvec4), cwbvh->bvh8Data);
        tinyocl::Buffer cwbvhTris(cwbvh->idxCount * 3 * sizeof(tinybvh::bvhvec4), cwbvh->bvh8Tris);
        cwbvhNodes.CopyToDevice(); cwbvhTris.CopyToDevice();
        tinyocl::Buffer rayData(Nfull * 64, 0);
        for (unsigned i = 0; i < Nfull; i++) memcpy((unsigned char*)rayData.GetHostPtr() + 64 * i, &fullBatch[i], 64);
        rayData.CopyToDevice();
        float traceTime = 0;
        cwbvh_kernel.SetArguments(&cwbvhNodes, &cwbvhTris, &rayData);
        for (int pass = 0; pass < 9; pass++) {
            cwbvh_kernel.Run(Nfull, 64, 0, &event);
            clWaitForEvents(1, &event);
            clGetEventProfilingInfo(event, CL_PROFILING_COMMAND_START, sizeof(cl_ulong), &startTime, 0);
            clGetEventProfilingInfo(event, CL_PROFILING_COMMAND_END, sizeof(cl_ulong), &endTime, 0);
            if (pass == 0) continue;
            traceTime += (endTime - startTime) * 1e-9f;
        }
        rayData.CopyFromDevice();
        traceTime /= 8.0f;
        printf("- BVH8/CWBVH  - primary: %7.2fMRays/s\n", (float)Nfull / traceTime * 1e-6f);
        bad += validate("BVH8_CWBVH", refDist, rayData, Nfull);
        delete cwbvh;
    }
    if (bad) { printf("FAILED: %d\n", bad); return 1; }
    printf("shim speedtest blocks ok\n");
    return 0;
}
