// kernels.h — host-visible launchers of the device kernels (defined in kernels_*.hip).
#pragma once
#include <vector>
#include "device_common.h"

namespace tbvh {

// layout codes as in include/tinybvh_amd.h (= BVHBase::BVHType, tiny_bvh.h:773-791); capi.hip checks they agree
constexpr int kLayoutBvhGpu = 5, kLayoutBvh4Gpu = 8, kLayoutCwbvh = 10;

void launch_bvh2(bool anyhit, int variant, const float4* nodes, const float4* tris, const QueryArgs& q, uint32_t* status,
                 uint32_t blocks, hipStream_t s);
void launch_bvh4(bool anyhit, int variant, const float4* data, const QueryArgs& q, uint32_t* status, uint32_t blocks, hipStream_t s);
// nodeStride: 5 = `nodes` is the packed array; 8 = the copy with one node per 128-byte line (scenes whose nodes outgrow the Infinity
// Cache); 13 (cwbvh_node.h: kNodeHybrid) = the priority-ordered copy whose first q.hybridK nodes are packed and the others one per line
// the coherent flavor of a probed launch as ONE traversal per wave of 64 consecutive rays (kernels_cwbvh_packet.hip)
void launch_cwbvh_packet(bool anyhit, const float4* nodes, const float4* tris, const QueryArgs& q, uint32_t* status, uint32_t blocks, hipStream_t s);
void launch_cwbvh(bool anyhit, int variant, const float4* nodes, const float4* tris, const QueryArgs& q, uint32_t* status,
                  uint32_t blocks, hipStream_t s, int nodeStride = 5, bool shallow = false, uint32_t blocks7 = 0xFFFFFFFFu);   // blocks7: grid of the kernels built for 7 waves per SIMD
void launch_cwbvh_derive_hybrid(const float4* src, const uint32_t* perm, float4* dst, uint32_t nNodes, uint32_t hybridK, const float4* tris, hipStream_t s);   // tris: the packed 48-byte records (one per node is embedded in its line), or nullptr
bool cwbvh_variant_valid(int variant);     // diagnostic variants of the BVH8_CWBVH kernel (tbvh_set_variant); the other layouts have none
void launch_cwbvh_pad(const float4* src, float4* dst, uint32_t nNodes, hipStream_t s);
void launch_cwbvh_pad_tris(const float4* src, float4* dst, uint64_t nTris, hipStream_t s);
struct BlasDesc { const float4* nodes; const float4* tris; const uint32_t* opmap; uint32_t opmapN; uint32_t layout; };  // one per BLAS of a TLAS (layout: TBVH_LAYOUT_*)
void launch_tlas(bool anyhit, int blasLayout, const float4* tlasNodes, const uint32_t* tlasIdx, const float4* instances,
                 const BlasDesc* blas, const QueryArgs& q, uint32_t* status, uint32_t blocks, hipStream_t s);
// 4-wide TLAS in the BVH4_GPU node format + the unified two-level kernel for BVH4_GPU BLASes (kernels_tlas4.hip)
size_t tlas_wide_scratch_bytes(uint64_t nAL, uint64_t nInst);   // kernels_tlaswide.hip builds both wide TLAS formats
uint64_t tlas4_cap_blocks(uint64_t nAL, uint64_t nInst);
void launch_tlas4_build(const float4* al, uint32_t nAL, const uint32_t* idx, uint32_t nIdx, const float4* inst, uint32_t nInst, float4* blocks, uint32_t capBlocks,
                        void* scratch, uint32_t* status, hipStream_t s);   // status |= 4 when the capacity does not hold the tree
void launch_tlas4(bool anyhit, int variant, const float4* tlas4, const float4* instances, const BlasDesc* blas, const QueryArgs& q, uint32_t* status, uint32_t blocks,
                  hipStream_t s, uint32_t blocks7);
// 8-wide TLAS in the BVH8_CWBVH node format + the unified two-level kernel for BVH8_CWBVH BLASes (kernels_tlas8.hip)
uint64_t tlas8_cap_nodes(uint64_t nAL, uint64_t nInst);
void launch_tlas8_build(const float4* al, uint32_t nAL, const uint32_t* idx, uint32_t nIdx, const float4* inst, uint32_t nInst, float4* nodes, uint32_t capNodes,
                        uint32_t* instRef, uint32_t capRefs, void* scratch, uint32_t* status, hipStream_t s);
void launch_tlas8(bool anyhit, int variant, const float4* tlasNodes, const uint32_t* instRef, const float4* instances, const BlasDesc* blas, const QueryArgs& q,
                  uint32_t* status, uint32_t blocks, hipStream_t s, uint32_t blocks7, bool mixed = false);
void launch_tlas2(bool anyhit, int variant, const float4* tlasNodes, const uint32_t* tlasIdx, const float4* instances, const BlasDesc* blas, const QueryArgs& q,
                  uint32_t* status, uint32_t blocks, hipStream_t s, uint32_t blocks7);   // BVH_GPU BLASes (kernels_tlas2.hip)
// device TLAS rebuild (kernels_tlasbuild.hip)
size_t tlas_build_scratch_bytes(uint32_t n, size_t* sortTempBytes);
hipError_t launch_tlas_rebuild(float4* tlasNodes, uint32_t* tlasIdx, float4* instances, const float* transformsDev, const float* blasBoundsDev,
                               uint32_t n, uint32_t nBlas, void* scratch, size_t sortTempBytes, hipStream_t s);
// LBVH build on the device (kernels_build.hip)
size_t lbvh_scratch_bytes(uint32_t n, size_t* sortTempBytes);
size_t ploc_scratch_bytes(uint32_t n, size_t* sortTempBytes, size_t* scanTempBytes);
hipError_t launch_ploc_build(const float4* verts, uint32_t n, uint32_t radius, float4* nodes32, uint32_t* primIdx, void* scratch, size_t sortTempBytes,
                             size_t scanTempBytes, hipStream_t s, uint32_t* steps);
hipError_t launch_lbvh_build(const float4* verts, uint32_t n, uint32_t maxLeaf, float4* nodes32, uint32_t* primIdx, void* scratch, size_t sortTempBytes,
                             hipStream_t s);
// BVH2 -> CWBVH conversion on the device (kernels_convert.hip)
hipError_t run_convert_cwbvh(const float4* nodes2, uint32_t nNodes2, const uint32_t* primIdx, uint64_t nIdx, const float4* verts, uint64_t nTris,
                             float4* cwNodes, uint32_t capNodes, float4* cwTris, uint64_t capTris, uint2* itemsA, uint2* itemsB, uint32_t* counters,
                             uint32_t* status, hipStream_t s, uint32_t* nNodesOut, uint64_t* nTrisOut, uint32_t* levelsOut);
hipError_t run_convert_bvh4(const float4* nodes2, uint32_t nNodes2, const uint32_t* primIdx, uint64_t nIdx, const float4* verts, uint64_t nTris,
                            float4* blocks, uint64_t capBlocks, uint2* itemsA, uint2* itemsB, uint32_t* counters, uint32_t* status, hipStream_t s,
                            uint64_t* nBlocksOut, uint32_t* levelsOut);
// device BLAS refit (kernels_refit.hip)
hipError_t run_refit_bvh4(float4* blocks, uint64_t nBlocks, const float4* verts, uint64_t nTris, void* itemsDev, uint32_t capNodes, uint32_t* counterDev,
                          float4* childBox, std::vector<uint32_t>& levelFirst, uint32_t* status, hipStream_t s);
size_t refit_scratch_bytes(int layout, uint32_t nNodes);
hipError_t launch_refit(int layout, float4* nodes, uint32_t nNodes, float4* tris, uint64_t nTriRecords, const float4* verts, uint64_t nTris,
                        void* scratch, uint32_t* status, hipStream_t s);
void launch_stream_copy(const float4* src, float4* dst, uint64_t n16, hipStream_t s);
void launch_stream_read(const float4* src, float* sink, uint64_t n16, uint32_t blocks, hipStream_t s);
void launch_valu_mix(float* out, int iters, uint32_t blocks, hipStream_t s);   // 32 VALU instructions per iteration and wave
void launch_pack_hits(const RayRec* rays, uint32_t* out, uint64_t n, hipStream_t s);
void launch_gather_tris(const uint32_t* primIdx, const float4* verts, float4* out, uint64_t nIdx, uint64_t nTris,
                        hipStream_t s);

// ray generators (kernels_raygen.hip)
struct CameraArgs {
    float eye[3], p1[3], p2[3], p3[3];
    uint32_t width, height, sppX, sppY;
};
void launch_gen_primary(const CameraArgs& cam, RayRec* rays, uint64_t first, uint64_t n, hipStream_t s);
// triangle fetch mode for the bounce generator: how to find the geometric normal of prim p
struct TriSource {
    int mode;              // 0: verts (3 float4 per prim, original order); 1: none
    const float4* verts;
};
void launch_gen_bounce(const TriSource& src, const RayRec* in, RayRec* out, uint64_t n, uint32_t seed, hipStream_t s);
void launch_reset_hits(RayRec* rays, uint64_t n, float tmax, hipStream_t s);
void launch_gen_shadow(const RayRec* in, RayRec* out, uint64_t n, float lx, float ly, float lz, float eps, hipStream_t s);

// ray binning (kernels_raybin.hip): counting sort of a batch by (Morton code of the origin's cell, direction octant)
struct RayBinArgs {
    float lo[3], scale[3];   // cell = (O - lo) * scale, clamped to [0, 2^cellBits)
    uint32_t cellBits;       // 0..6 bits per axis
    uint32_t flags;          // 1: octant as the minor part of the key, 2: as the major part, 0: cell only
};
uint32_t ray_bin_count(uint32_t cellBits, uint32_t flags);
size_t ray_bin_scratch_bytes(uint64_t n, uint32_t cellBits, uint32_t flags, size_t* scanTempBytes);
hipError_t launch_ray_bin(const RayRec* in, RayRec* out, uint32_t* perm, uint64_t n, const unsigned long long* nDev, const RayBinArgs& a, void* scratch,
                          size_t scanTempBytes, uint32_t blocks, hipStream_t s);

// wavefront path tracer stages (kernels_wavefront.hip)
struct PathAux { float T[3]; uint32_t pixel; };   // throughput + pixel << 8 | depth << 4 | path flags (paths), or pending contribution + pixel (shadow rays); 16 bytes
struct ShadeArgs {
    const RayRec* in; const PathAux* auxIn; const unsigned long long* nIn;
    RayRec* out; PathAux* auxOut; unsigned long long* nOut;
    RayRec* shadow; PathAux* shadowAux; unsigned long long* nShadow;
    const float4* verts; float* accum;
    const float4* const* blasVerts; const float4* instances;   // TLAS scenes: vertex array per BLAS, BLASInstance records (else nullptr)
    float lightPos[3], lightColor[3], skyLo[3], skyHi[3];
    float lightSize[2];   // extent of the rectangular light along x and z (0, 0 = point light)
    float eps; uint32_t depth, maxDepth, seed, flags;   // flags bit 0: at most one diffuse bounce per path (wavefront.cl:233); bit 1: wavefront.cl to the letter
    const uint32_t* blueNoise; uint32_t sampleIdx, width, height;   // 128 x 128 x 8 table (or nullptr), the frame's sample index, size of the FULL image
    uint32_t pixelOffset;   // a band of a larger image (tbvh_wavefront_set_band): index of the band's first pixel in the full image (else 0)
};
void launch_wf_generate(const CameraArgs& cam, RayRec* rays, PathAux* aux, uint64_t n, uint32_t seed, uint32_t firstRow, uint32_t bandRows, unsigned long long* queueCounters,
                        uint32_t nCounterWords, hipStream_t s);
void launch_wf_shade(const ShadeArgs& a, uint64_t capacity, hipStream_t s);
void launch_wf_finalize(const float* accum, float scale, uint32_t* pixels, uint64_t n, hipStream_t s);
void launch_wf_connect(const uint8_t* occ, const PathAux* aux, const unsigned long long* nShadow, float* accum, uint64_t capacity, hipStream_t s);

}  // namespace tbvh
