// shim_speedtest_blocks.cpp — what the three GPU blocks of the reference's speedtest do (tiny_bvh_speedtest.cpp:1092-1241: BVH_GPU, BVH4_GPU, BVH8_CWBVH;
// the reference #undefs them on Linux, :86-92), written against tinyocl's OWN names — Buffer, CopyToDevice, Kernel( "traverse.cl", entry ),
// SetArguments, Run( N, 64, 0, &event ), clWaitForEvents, clGetEventProfilingInfo, CopyFromDevice — as include/shim/tiny_ocl.h provides them, i.e.
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
    // what every block of the speedtest does once its scene buffers are on the device: rays into a tinyocl::Buffer (the first 64 bytes of each tinybvh::Ray),
    // nine timed launches through a cl_event (the first warms up), results back, validation
    auto trace_block = [&](const char* name, tinyocl::Kernel& kernel) {
        tinyocl::Buffer rayData(Nfull * 64);
        unsigned char* dst = (unsigned char*)rayData.GetHostPtr();
        for (unsigned i = 0; i < Nfull; i++) memcpy(dst + 64 * i, &fullBatch[i], 64);
        rayData.CopyToDevice();
        // (the scene buffers were bound by the caller; the ray buffer is the kernel's last argument — rebind it for this block)
        kernel.SetRayBuffer(&rayData);
        double seconds = 0;
        for (int pass = 0; pass < 9; pass++) {
            cl_event event;
            cl_ulong t0 = 0, t1 = 0;
            kernel.Run(Nfull, 64, 0, &event);
            clWaitForEvents(1, &event);
            clGetEventProfilingInfo(event, CL_PROFILING_COMMAND_START, sizeof(cl_ulong), &t0, 0);
            clGetEventProfilingInfo(event, CL_PROFILING_COMMAND_END, sizeof(cl_ulong), &t1, 0);
            if (pass) seconds += (double)(t1 - t0) * 1e-9;
        }
        rayData.CopyFromDevice();
        printf("- %-11s - primary: %7.2fMRays/s\n", name, (double)Nfull / (seconds / 8.0) * 1e-6);
        if (!(seconds > 0)) bad++;
        bad += validate(name, refDist, rayData, Nfull);
    };
    tinyocl::Buffer noRays(64);   // placeholder last argument until trace_block binds the real ray buffer
    {   // GPU_2WAY (tiny_bvh_speedtest.cpp:1094-1141): Aila-Laine nodes + primIdx + vertices
        BVH_GPU layout;
        layout.BuildHQ(triangles, tricount);
        tinyocl::Buffer gpuNodes(layout.usedNodes * sizeof(BVH_GPU::BVHNode), layout.bvhNode), idxData(layout.idxCount * sizeof(unsigned), layout.bvh.primIdx),
            triData(layout.triCount * 3 * sizeof(bvhvec4), triangles);
        gpuNodes.CopyToDevice(); idxData.CopyToDevice(); triData.CopyToDevice();
        ailalaine_kernel.SetArguments(&gpuNodes, &idxData, &triData, &noRays);
        trace_block("BVH_GPU", ailalaine_kernel);
    }
    {   // GPU_4WAY (:1145-1190): one stream
        BVH4_GPU layout;
        layout.BuildHQ(triangles, tricount);
        tinyocl::Buffer gpu4Nodes(layout.usedBlocks * sizeof(bvhvec4), layout.bvh4Data);
        gpu4Nodes.CopyToDevice();
        gpu4way_kernel.SetArguments(&gpu4Nodes, &noRays);
        trace_block("BVH4_GPU", gpu4way_kernel);
    }
    {   // GPU_CWBVH (:1192-1241): nodes + triangles
        BVH8_CWBVH layout;
        layout.BuildHQ(triangles, tricount);
        tinyocl::Buffer cwbvhNodes(layout.usedBlocks * sizeof(bvhvec4), layout.bvh8Data), cwbvhTris(layout.idxCount * 3 * sizeof(bvhvec4), layout.bvh8Tris);
        cwbvhNodes.CopyToDevice(); cwbvhTris.CopyToDevice();
        cwbvh_kernel.SetArguments(&cwbvhNodes, &cwbvhTris, &noRays);
        trace_block("BVH8_CWBVH", cwbvh_kernel);
    }
    if (bad) { printf("FAILED: %d\n", bad); return 1; }
    printf("shim speedtest blocks ok\n");
    return 0;
}
