// ref_minimal_check.cpp — what the reference's tiny_bvh_minimal_gpu.cpp must print, computed on the CPU by the REAL tiny_bvh.h.
// tests/test_examples.py compiles the reference's own main UNMODIFIED against include/shim/tiny_ocl.h (the HIP engine behind tinyocl's names) and
// compares its stdout with this program's, line for line.  The scene and the rays are the same by construction: both programs draw from the rand()
// of examples/fixed_rand.c (linked into both, hidden from the GPU runtime's own calls) in the order tiny_bvh_minimal_gpu.cpp:25-73 does (3 + 9 draws per triangle, then one per ray; no srand), and `%f` prints a float's value.
// Tracing: tinybvh::BVH::Intersect (tiny_bvh.h:3222-3304) on a BVH::Build tree of the same triangles — the parity oracle of SURVEY par. 8(c).
//
//   g++ -std=c++20 -O2 -mavx2 -mfma -I<tinybvh checkout> examples/ref_minimal_check.cpp -o examples/_build/ref_minimal_check
#define TINYBVH_IMPLEMENTATION
#include "tiny_bvh.h"

#include <cstdio>
#include <cstdlib>

static const int kTriangles = 8192, kRays = 1024;
static tinybvh::bvhvec4 tris[kTriangles * 3];
static float uniform_rand() { return (float)rand() / (float)RAND_MAX; }

int main() {
    for (int i = 0; i < kTriangles; i++) {
        const float x = uniform_rand(), y = uniform_rand(), z = uniform_rand();
        for (int v = 0; v < 3; v++) {
            tinybvh::bvhvec4& p = tris[i * 3 + v];
            p.x = x + 0.1f * uniform_rand();
            p.y = y + 0.1f * uniform_rand();
            p.z = z + 0.1f * uniform_rand();
        }
    }
    tinybvh::BVH bvh;
    bvh.Build(tris, kTriangles);
    for (int i = 0; i < kRays; i++) {
        const tinybvh::bvhvec3 O(0.5f, 0.5f, -1), D(0.1f, uniform_rand() - 0.5f, 2);
        tinybvh::Ray ray(O, D);
        bvh.Intersect(ray);
        printf("ray %i, nearest intersection: %f\n", i, ray.hit.t);
    }
    return 0;
}
