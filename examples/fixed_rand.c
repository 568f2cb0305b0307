/* fixed_rand.c — rand() / srand() for the two programs of tests/test_examples.py::test_reference_minimal_gpu_main_unmodified.
 * The reference's tiny_bvh_minimal_gpu.cpp draws its triangles, constructs its tinyocl::Kernel (which brings the GPU runtime up) and THEN draws its
 * rays from rand(); a runtime that itself calls libc's rand() during start-up shifts the sequence, and the CPU checker (ref_minimal_check.cpp), which
 * starts no runtime, would trace other rays (observed on the MI355X box: the HIP runtime draws from rand() while it starts, from its own threads, so the
 * program's sequence even changed from run to run).  Linked into both executables with HIDDEN visibility — the link editor would otherwise export
 * the executable's rand() to the shared libraries that reference the symbol — this definition serves the program's own calls only, so both programs
 * see the same sequence whatever the runtime does.  The reference source stays unmodified. */
#include <stdlib.h>
static unsigned long long state = 0x2545F4914F6CDD1Dull;
__attribute__((visibility("hidden"))) int rand(void) {
    state = state * 6364136223846793005ull + 1442695040888963407ull;
    return (int)((state >> 33) & 0x7fffffff);   /* 0 .. RAND_MAX (2^31 - 1) */
}
__attribute__((visibility("hidden"))) void srand(unsigned seed) { state = 0x2545F4914F6CDD1Dull ^ ((unsigned long long)seed << 17); }
