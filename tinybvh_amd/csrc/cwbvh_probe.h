// cwbvh_probe.h — the per-launch coherence probe shared by the BVH8_CWBVH traversal kernels (kernels_cwbvh.hip, kernels_cwbvh_packet.hip).
#pragma once
#include "device_common.h"

namespace tbvh {

// Coherence of a batch: of kProbePairs pairs of neighbouring rays spread over the batch, how many agree in direction (camera rays and shadow
// rays towards one light: almost all; bounce rays: almost none) AND start close to each other measured by how far they reach (shadow rays of a
// path tracer's later depths all point at the light but start all over the scene: they walk different subtrees — the incoherent flavor traces
// them 6-9 % faster; a ray that may reach infinitely far has no such measure and counts by its direction alone).  Wave-uniform result; the same
// in every wave of a launch whose rays' t does not change under it (fresh and any-hit launches; see the note at the call).
constexpr uint32_t kProbePairs = 256;
__device__ __forceinline__ void coherence_sample(const RayRec* __restrict__ rays, uint64_t n, bool fresh, float freshTmax, uint32_t& agree, uint32_t& pairs) {
    const uint64_t stride = n / kProbePairs > 2 ? n / kProbePairs : 2;
    agree = 0; pairs = 0;
#pragma unroll
    for (uint32_t j = 0; j < kProbePairs / 64u; j++) {
        const uint64_t i = (uint64_t)(j * 64u + threadIdx.x) * stride;
        const bool valid = i + 1 < n;
        bool ok = false;
        if (valid) {
            const float4 a = rays[i].D, b = rays[i + 1].D;
            ok = a.x * b.x + a.y * b.y + a.z * b.z > 0.98f;
            const float ta = fresh ? freshTmax : rays[i].hit.x, tb = fresh ? freshTmax : rays[i + 1].hit.x;
            const float reach = ta < tb ? ta : tb;
            if (ok && reach < 1e29f) {
                const float4 oa = rays[i].O, ob = rays[i + 1].O;
                const float dx = oa.x - ob.x, dy = oa.y - ob.y, dz = oa.z - ob.z;
                ok = dx * dx + dy * dy + dz * dz <= 0.0025f * reach * reach;   // origins within 5 % of the reach
            }
        }
        agree += (uint32_t)__popcll(__ballot(ok)); pairs += (uint32_t)__popcll(__ballot(valid));
    }
}

}  // namespace tbvh
