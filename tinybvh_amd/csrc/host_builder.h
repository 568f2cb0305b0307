// host_builder.h — host-side BVH construction and encoding into the blob formats the
// traversal kernels consume (formats: SURVEY.md Appendix A; reference writers cited per
// function in host_builder.cpp).  Independent implementation: binned SAH BVH2 builder with
// deterministic task-parallel subtree construction, surface-area-greedy wide collapse, and
// encoders for BVH_GPU (Aila-Laine), BVH4_GPU and BVH8_CWBVH.
#pragma once
#include <cstdint>
#include <vector>

namespace tbvh {

struct Vec4 { float x, y, z, w; };

// 32-byte node, same field order as tinybvh::BVH::BVHNode (tiny_bvh.h:857-866):
// interior: leftFirst = index of left child, right child = leftFirst + 1, triCount = 0;
// leaf: leftFirst = first entry in primIdx, triCount > 0.
struct Node2 {
    float mn[3]; uint32_t leftFirst;
    float mx[3]; uint32_t triCount;
    bool leaf() const { return triCount > 0; }
};
static_assert(sizeof(Node2) == 32, "Wald node is 32 bytes");

// 64-byte Aila-Laine node (tiny_bvh.h:1095-1105).
struct NodeAL {
    float lmin[3]; uint32_t left;
    float lmax[3]; uint32_t right;
    float rmin[3]; uint32_t triCount;
    float rmax[3]; uint32_t firstTri;
};
static_assert(sizeof(NodeAL) == 64, "Aila-Laine node is 64 bytes");

// Host threads this process can really run at once: hardware_concurrency() cut by the affinity mask and by the cgroup
// CPU quota (inside a container limited to 16 cores of a 256-thread machine hardware_concurrency() still says 256, and
// 256 builder threads on 16 cores are slower than 16).
uint32_t usable_host_threads();

struct BuildParams {
    uint32_t bins = 8;
    uint32_t maxLeafTris = 4;
    uint32_t threads = 0;
    bool greedyCollapse = true;   // wide layouts: surface-area-greedy collapse (default; measured faster on the GPU) or the SAH-optimal one
    float cPrim = 0.3f;           // cost of one triangle test relative to one wide-node visit (optimal collapse)
    float splitBudget = 0.f;      // > 0: triangle splitting ahead of the build, up to splitBudget * triCount extra references (host_builder.cpp: presplit)
};

struct BVH2 {
    std::vector<Node2> nodes;       // root = 0, siblings adjacent
    std::vector<uint32_t> primIdx;  // permutation of [0, triCount); with BuildParams::splitBudget a triangle may be named by several leaves
    uint32_t triCount = 0;
};

// Build a BVH2 over triangles given as 3 x Vec4 per triangle.
void build_bvh2(const Vec4* verts, uint32_t triCount, const BuildParams& p, BVH2& out);

// Build a BVH2 over arbitrary boxes (used for the TLAS): box i = {mn[3], mx[3]}.
void build_bvh2_boxes(const float* boxes6, uint32_t count, const BuildParams& p, BVH2& out);

// Encoders.
void encode_bvh_gpu(const BVH2& bvh, std::vector<NodeAL>& out);
void encode_bvh4_gpu(const BVH2& bvh, const Vec4* verts, const BuildParams& p, std::vector<Vec4>& blocks);
void encode_cwbvh(const BVH2& bvh, const Vec4* verts, const BuildParams& p, std::vector<Vec4>& nodeBlocks,
                  std::vector<Vec4>& triBlocks);

// An uploaded BVH_GPU blob (Aila-Laine nodes: every node carries its CHILDREN's boxes, tiny_bvh.h:1095-1105) as a BVH2 in the Wald layout the wide
// converters take (kernels_convert.hip), leaves cut down to maxLeafTris entries (the SplitLeafs( 3 ) step of BVH8_CWBVH::Build, tiny_bvh.h:5831,
// here by halving a leaf's primIdx range: the order of the entries is kept, primIdx is shared with the blob).  A half's box = the bounds of its
// triangles CLIPPED to the leaf's box, so the leaves of an SBVH (BuildHQ: spatial splits, boxes smaller than their triangles) stay as tight as
// they were.  false: the root is a leaf (nothing to collapse).
// recs != nullptr: the leaves' triangles as gathered records {v0|prim, e1, e2}, one per primIdx entry (what the library keeps of a BVH_GPU blob on the
// device: kernels_query.hip: k_gather_tris) instead of primIdx + verts — the halves' boxes then come from the records' corners, padded by ulps.
bool bvh_gpu_to_bvh2(const NodeAL* al, uint64_t nNodes, const uint32_t* primIdx, uint64_t nIdx, const Vec4* verts, uint64_t nTris, uint32_t maxLeafTris,
                     std::vector<Node2>& out, const Vec4* recs = nullptr);

// The same for an uploaded BVH4_GPU stream (tiny_bvh.h:1248-1266, 5115-5244): every 4-wide node becomes one to three binary nodes over its children's
// DEQUANTISED boxes (padded outward by two ulps: they only cull), leaves of at most maxLeafTris entries index `recs`, the blob's inline triangle
// records {v0|prim, e1, e2} gathered in depth-first order (the converter copies them as they are: kernels_convert.hip, record mode).
bool bvh4_gpu_to_bvh2(const Vec4* blocks, uint64_t nBlocks, uint32_t maxLeafTris, std::vector<Node2>& out, std::vector<Vec4>& recs);
// ... and of a BVH8_CWBVH blob (nodes: 5 float4 each; tris: 3 float4 per record {e2, e1, v0|prim}); records come out as {v0|prim, e1, e2}
bool cwbvh_to_bvh2(const Vec4* nodes, uint64_t nNodes, const Vec4* tris, uint64_t nTriBlocks, std::vector<Node2>& out, std::vector<Vec4>& recs);

// CWBVH nodes (5 x Vec4 each) in surface-area priority order: newIdx[old] = new; see host_builder.cpp.
bool cwbvh_priority_order(const Vec4* in, uint32_t nNodes, std::vector<uint32_t>& newIdx);   // false: not a strict tree, no numbering

// Structural validation of caller-supplied blobs (returns nullptr when fine, else a message).
const char* validate_bvh_gpu(const NodeAL* nodes, uint64_t nNodes, uint64_t nIdx);
const char* validate_bvh4_gpu(const Vec4* blocks, uint64_t nBlocks);
const char* validate_cwbvh(const Vec4* nodes, uint64_t nNodes, uint64_t nTriBlocks);
extern const char* const kValidateNoMemory;   // what the three return (by address) when the walk itself ran out of host memory

// BLASInstance record, 192 bytes (tiny_bvh.h:1443-1457).
struct Instance192 {
    float transform[16];
    float invTransform[16];
    float aabbMin[3]; uint32_t blasIdx;
    float aabbMax[3]; uint32_t mask;
    uint32_t pad[8];
};
static_assert(sizeof(Instance192) == 192, "BLASInstance is 192 bytes");

// Fill invTransform and world-space bounds of one instance from the bounds of its BLAS.
void update_instance(Instance192& inst, const float* blasBounds6);

}  // namespace tbvh
