/*
 * tinybvh_amd.h — C ABI of the MI355X (gfx950) batched ray-traversal engine.
 *
 * This is the drop-in boundary for the ONE hot path of jbikker/tinybvh that this
 * repository accelerates: batched Intersect / IsOccluded over packed 64-byte rays on
 * the GPU layouts that tiny_bvh.h builds on the host (BVH_GPU, BVH4_GPU, BVH8_CWBVH)
 * and the TLAS wrapper around them.  Every entry point cites the reference interface
 * it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - plain pointers and 64-bit sizes only; no C++/torch types; all sizes are element
 *     counts unless the name says "bytes".
 *   - every function returns 0 on success, a negative TBVH_E_* code on failure, and
 *     never calls exit() (the reference's BVH_FATAL_ERROR_IF, tiny_bvh.h:1611-1620,
 *     and tinyocl's FatalError, tiny_ocl.h:409-507, terminate the process; a library
 *     behind an FFI must not).  tbvh_last_error() returns the message for the calling
 *     thread's most recent failure.
 *   - "blocks16" means units of 16 bytes (one bvhvec4), which is how the reference
 *     counts BVH4_GPU::usedBlocks and BVH8_CWBVH::usedBlocks.
 *   - a ray record is the first 64 bytes of tinybvh::Ray (tiny_bvh.h:689-709), equal to
 *     the device struct Ray of traverse.cl:11-17:
 *         off  0  O.xyz     12  mask(u32)
 *         off 16  D.xyz     28  instIdx(u32)
 *         off 32  rD.xyz    44  hit.inst (when INST_IDX_BITS==32) or pad
 *         off 48  hit.t  52 hit.u  56 hit.v  60 hit.prim(u32)
 *     Intersect reads O, D, rD and hit.t (= tmax) and writes bytes 48..63 only, and only
 *     for rays that hit (a miss leaves the record untouched, like BVH::Intersect,
 *     tiny_bvh.h:3247-3304 / 8524-8529).  TLAS queries also write hit.inst at byte 44.
 *   - one tbvh_context per HIP device — or several: contexts are independent (the reference
 *     has a single process-global OpenCL device, tiny_ocl.h:362-364).  Every entry point takes
 *     its context's lock, so host threads may share a context and its scenes the way the
 *     reference's callers share a const BVH (tiny_bvh_speedtest.cpp:1077-1083: Intersect from
 *     8 threads): concurrent calls on ONE context are safe and serialise; threads whose
 *     queries should overlap on the device use one context (and one upload of the scene)
 *     each.  tbvh_shutdown / tbvh_free_scene must not race with calls on what they destroy.
 *     tbvh_time_last_ms reports the calling context's most recent operation — with several
 *     threads on one context that may be another thread's.
 */
#ifndef TINYBVH_AMD_H_
#define TINYBVH_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TBVH_ABI_VERSION 5   /* 5: tbvh_pinned_malloc / _free, tbvh_scene_get / _set_schedule_hint, tbvh_measure_link_bandwidth, TBVH_BUILD_SPLIT_TRIANGLES / _WHOLE_TRIANGLES, development aids moved to tinybvh_amd_debug.h (nothing removed from the library); 4: tbvh_update_bvh_gpu / _bvh4_gpu / _cwbvh, tbvh_time_history, contexts are thread-safe; 3: deterministic ties, tbvh_bin_rays_device, tbvh_cwbvh_set_hybrid, device-resident multi-device calls */

/* error codes */
#define TBVH_OK            0
#define TBVH_E_INVALID    -1   /* bad argument (null pointer, bad stride, bad layout)   */
#define TBVH_E_NODEVICE   -2   /* no usable HIP device / device index out of range      */
#define TBVH_E_HIP        -3   /* a HIP runtime call failed; see tbvh_last_error()      */
#define TBVH_E_NOMEM      -4   /* host or device allocation failed                      */
#define TBVH_E_FORMAT     -5   /* blob failed validation (e.g. CWBVH root is a leaf)    */

/* layouts; the values ARE BVHBase::BVHType (tiny_bvh.h:773-791), so a caller can pass bvh.layout
 * (ABI version 1 used 4 / 6 / 9 here, which are LAYOUT_BVH_SOA / LAYOUT_MBVH / LAYOUT_MBVH8 in the reference) */
#define TBVH_LAYOUT_BVH2_WALD  1  /* LAYOUT_BVH       32-byte nodes (host/oracle only)   */
#define TBVH_LAYOUT_BVH_GPU    5  /* LAYOUT_BVH_GPU   Aila-Laine 64-byte nodes           */
#define TBVH_LAYOUT_BVH4_GPU   8  /* LAYOUT_BVH4_GPU  quantized 4-wide + inline tris     */
#define TBVH_LAYOUT_CWBVH     10  /* LAYOUT_CWBVH     compressed wide BVH8               */

typedef struct tbvh_context tbvh_context;  /* one HIP device + stream + scratch          */
typedef struct tbvh_scene   tbvh_scene;    /* one uploaded layout (BLAS or TLAS)         */
typedef struct tbvh_hostbvh tbvh_hostbvh;  /* host-built blobs (see "host builder")      */

/* ------------------------------------------------------------------------------------
 * context — replaces tinyocl::Kernel::InitCL (tiny_ocl.h:945-1139)
 * ---------------------------------------------------------------------------------- */
int         tbvh_abi_version(void);
const char* tbvh_last_error(void);
int         tbvh_device_count(void);                     /* >=0, or TBVH_E_HIP           */
int         tbvh_init(int device, tbvh_context** out);   /* explicit context per device  */
void        tbvh_shutdown(tbvh_context* ctx);            /* frees scenes' device memory   */
int         tbvh_synchronize(tbvh_context* ctx);
/* Make subsequent launches of this context go to an external hipStream_t (e.g. the
 * current torch stream); NULL restores the context's own stream. */
int         tbvh_set_stream(tbvh_context* ctx, void* hip_stream);
/* Per-operation device timing on (default) or off.  On: every query / refit / rebuild is bracketed by a HIP event pair — what
 * tbvh_time_last_ms and tbvh_time_history read, the CL_PROFILING_COMMAND_START / END of tiny_bvh_speedtest.cpp:1126-1131.  Off: nothing but the
 * kernels is enqueued (two event records fewer per query: a few microseconds each, which a renderer issuing many small queries per frame
 * notices); the time calls then keep reporting the last operation that was timed.  (A scene whose coherent-batch schedule is still being
 * measured keeps measuring: the tuner brackets those few launches with events of its own, tbvh_scene_get_schedule_hint.) */
int         tbvh_set_timing(tbvh_context* ctx, int enabled);

/* ------------------------------------------------------------------------------------
 * uploads — replace tinyocl::Buffer(bytes, hostPtr) + CopyToDevice()
 * (tiny_ocl.h:571-691) as used by tiny_bvh_speedtest.cpp:1102-1108, 1153-1155,
 * 1200-1206 and tiny_bvh_minimal_gpu.cpp:55-70.  Blobs are consumed verbatim in the
 * reference's formats; the library may keep a re-laid-out copy internally.
 * ---------------------------------------------------------------------------------- */

/* BVH_GPU (Aila-Laine).  nodes64 = BVH_GPU::bvhNode (tiny_bvh.h:1095-1105), n_nodes =
 * usedNodes; prim_idx = BVH_GPU::bvh.primIdx, n_idx = bvh.idxCount; verts = the caller's
 * bvhvec4 vertex array (3 per triangle), n_tris = triCount. */
int tbvh_upload_bvh_gpu(tbvh_context* ctx, const void* nodes64, uint64_t n_nodes,
                        const uint32_t* prim_idx, uint64_t n_idx,
                        const void* verts16, uint64_t n_tris, tbvh_scene** out);

/* BVH4_GPU.  blocks16 = BVH4_GPU::bvh4Data, n_blocks = usedBlocks (tiny_bvh.h:1284-1286). */
int tbvh_upload_bvh4_gpu(tbvh_context* ctx, const void* blocks16, uint64_t n_blocks,
                         tbvh_scene** out);

/* BVH8_CWBVH.  nodes16 = bvh8Data, n_node_blocks = usedBlocks (5 per node);
 * tris16 = bvh8Tris, n_tri_blocks = 3 * idxCount (tiny_bvh.h:1356-1359;
 * tiny_bvh_speedtest.cpp:1200-1204).  The default triangle records only (48 bytes: e2, e1, v0 | prim): blobs built with the reference's
 * experimental CWBVH_COMPRESSED_TRIS switch (tiny_bvh.h:170-171, off by default; 64-byte records) are refused with TBVH_E_FORMAT. */
int tbvh_upload_cwbvh(tbvh_context* ctx, const void* nodes16, uint64_t n_node_blocks,
                      const void* tris16, uint64_t n_tri_blocks, tbvh_scene** out);

/* TLAS in BVH_GPU format over BLASInstance records (tiny_bvh.h:1443-1457, 192 bytes
 * each) — replaces the uploads of tiny_bvh_gpu2.cpp:122-130.  blas[i] is the scene for
 * BLASInstance::blasIdx == i.  tlas_idx = tlas.bvh.primIdx (instance indices).
 * The BLASes may be BVH8_CWBVH, BVH4_GPU or BVH_GPU scenes, also mixed within one TLAS (as traverse_tlas.cl:50-72
 * selects the BLAS traversal per instance through blasDesc[].blasType).
 * The TLAS keeps references to its BLAS scenes: tbvh_free_scene on a BLAS that a live TLAS still uses is deferred until the
 * last such TLAS is freed, and tbvh_set_opacity_micromaps on a BLAS updates every TLAS built over it.
 * The TLAS blobs are validated (child / primIdx / blasIdx ranges; TBVH_E_FORMAT), here and in tbvh_update_tlas. */
int tbvh_upload_tlas(tbvh_context* ctx, const void* tlas_nodes64, uint64_t n_nodes,
                     const uint32_t* tlas_idx, uint64_t n_idx,
                     const void* instances192, uint64_t n_instances,
                     tbvh_scene* const* blas, uint64_t n_blas, tbvh_scene** out);
/* Per-frame TLAS refresh (same blobs, rebuilt on the host; tiny_bvh_gpu2.cpp:122-130). */
int tbvh_update_tlas(tbvh_scene* tlas, const void* tlas_nodes64, uint64_t n_nodes,
                     const uint32_t* tlas_idx, uint64_t n_idx,
                     const void* instances192, uint64_t n_instances);

/* In-place re-upload of a BLAS whose blob the caller refitted and re-converted on the host — BVH::Refit (tiny_bvh.h:3055-3093) followed by
 * BVH_GPU / BVH4_GPU / BVH8_CWBVH::ConvertFrom again, the reference's own flow for animated geometry — without freeing the scene: the handle,
 * the device allocations and the pointers every TLAS over this BLAS holds stay valid (tbvh_update_tlas is the same for the top level;
 * tbvh_refit does the whole refit on the device instead).  Arguments as for the matching tbvh_upload_*; the blob may be SMALLER than the
 * one uploaded (ConvertFrom of a refitted tree can collapse differently), a larger one is TBVH_E_INVALID: free the scene and upload.
 * Validated like an upload (TBVH_E_FORMAT).  The library's derived copies follow: re-derived on the device when the tree kept its shape,
 * dropped and rebuilt as after an upload when it did not; the copies in ANOTHER layout (the 8-wide copy of a BVH_GPU / BVH4_GPU scene, the 4- and
 * 8-wide forms TLASes enter their BLASes through) are dropped and come back after four queries without another update (a blob re-uploaded every
 * frame is traced as uploaded: making them again would cost more than a frame's queries gain; tbvh_refit refits them in place, unless fewer than 8 M rays were traced
 * since the previous refit: then it drops them the same way).  Synchronous (the caller's arrays may be reused on return).  A TLAS over the BLAS
 * sees the new contents at once, but its instance boxes are its own: when the geometry left its old bounds, update or rebuild the TLAS as the
 * reference's frame loop does (tbvh_update_tlas, tbvh_rebuild_tlas_device with the new BLAS bounds). */
int tbvh_update_bvh_gpu(tbvh_scene* scene, const void* nodes64, uint64_t n_nodes, const uint32_t* prim_idx, uint64_t n_idx, const void* verts16, uint64_t n_tris);
int tbvh_update_bvh4_gpu(tbvh_scene* scene, const void* blocks16, uint64_t n_blocks);
int tbvh_update_cwbvh(tbvh_scene* scene, const void* nodes16, uint64_t n_node_blocks, const void* tris16, uint64_t n_tri_blocks);

/* BVH2 -> wide layout conversion ON THE DEVICE: replaces BVH8_CWBVH::ConvertFrom (tiny_bvh.h:5884-6018, with
 * the MBVH<8> collapse of 4975-5048) for callers that have a plain BVH2.  nodes32 = BVH::bvhNode (32-byte
 * BVHNode: aabbMin, leftFirst, aabbMax, triCount; children adjacent; tiny_bvh.h:1050-1062), n_nodes =
 * usedNodes, prim_idx = BVH::primIdx (n_idx = idxCount), verts16 = the vertex array; all three in host memory
 * (on_device = 0) or device memory (1).  Leaves must hold at most 3 triangles (BVH::SplitLeafs(3), which the
 * reference's ConvertFrom also requires) for TBVH_LAYOUT_CWBVH; otherwise TBVH_E_FORMAT.  layout: TBVH_LAYOUT_CWBVH, or
 * TBVH_LAYOUT_BVH4_GPU (BVH4_GPU::ConvertFrom, tiny_bvh.h:5115-5244: 4-wide collapse, quantised child boxes,
 * triangles inline after their node).
 * Synchronous (one small read-back per level of the wide tree); tbvh_time_last_ms() = time spent converting. */
int tbvh_convert_bvh2_device(tbvh_context* ctx, const void* nodes32, uint64_t n_nodes, const uint32_t* prim_idx, uint64_t n_idx,
                             const void* verts16, uint64_t n_tris, int on_device, int layout, tbvh_scene** out);

/* BVH BUILD on the device: triangles -> LBVH (Morton order, Karras topology) -> BVH8_CWBVH, nothing on the host.
 * The fast path for content whose topology changes every frame or whose host build (BVH::Build,
 * tiny_bvh.h:2124-2461) is the bottleneck; the tree is of lower quality than the binned-SAH build (more node
 * visits per ray), so it is not what the host builder or the bench use.  verts16: bvhvec4 vertices, 3 per
 * triangle, host (on_device = 0) or device memory (1); layout: TBVH_LAYOUT_CWBVH (max_leaf_tris 1..3, 0 = the
 * default 1: Morton ranges make poor multi-triangle leaves) or TBVH_LAYOUT_BVH4_GPU (1..4, 0 = 4).
 * prim indices in the hit records are the triangle's index in verts16, as with every other builder. */
int tbvh_build_device(tbvh_context* ctx, const void* verts16, uint64_t n_tris, int on_device, int layout,
                      uint32_t max_leaf_tris, tbvh_scene** out);

/* The same with the tree built by PLOC (parallel locally-ordered clustering, Meister & Bittner 2018) instead of the LBVH's Morton splits:
 * bottom-up agglomeration of the Morton-ordered triangles — every cluster merges with the neighbour within `radius` positions whose union
 * has the smallest surface area, once both agree; one triangle per leaf; about 1.5 x the LBVH's build time.  Lower surface-area cost
 * than the LBVH (the reference's bar for GPU traversal is BVH::BuildHQ, tiny_bvh.h:2623-3040), but NOT faster to trace on the scenes
 * this library is measured on (profiles/r03_device_builders.txt: axis-aligned procedural geometry suits Morton splits),
 * hence an entry point of its own rather than the default.
 * radius: 1..32 positions to each side, 0 = the default 16.  Everything else as tbvh_build_device. */
int tbvh_build_device_ploc(tbvh_context* ctx, const void* verts16, uint64_t n_tris, int on_device, int layout, uint32_t radius, tbvh_scene** out);

/* Read a BLAS scene's device blobs back (tests, caching a refitted blob): which = 0 nodes, 1 triangle
 * records (BVH_GPU: the gathered {v0|prim, e1, e2} form; BVH4_GPU has none).  dst = NULL only reports the size. */
int tbvh_scene_download(tbvh_scene* scene, int which, void* dst, uint64_t cap_bytes, uint64_t* bytes_out);

/* Opacity micromaps — BVHBase::SetOpacityMicroMaps (tiny_bvh.h:823-826): N x N bits per triangle (N * N bits rounded up to
 * whole 32-bit words per triangle, triangle i's words first word at i * ((N * N + 31) / 32)); a ray / triangle hit whose
 * barycentrics land on a clear bit is not a hit, exactly as in IntersectTri / TriOccludes (tiny_bvh.h:8514-8522,
 * 8562-8570; traverse_bvh2.cl:112-117).  Honoured by Intersect and IsOccluded of every layout; for instanced scenes set
 * the maps on the BLASes before uploading their TLAS.  map_data = NULL or N = 0 removes the maps.  The data is copied. */
int tbvh_set_opacity_micromaps(tbvh_scene* blas, const uint32_t* map_data, uint32_t N, uint64_t n_tris, int on_device);

/* BLAS refit ON THE DEVICE for animated meshes: same topology and triangle order, every box recomputed
 * bottom-up from the new vertex positions, CWBVH nodes re-quantised, triangle records re-gathered.
 * Replaces "BVH::Refit (tiny_bvh.h:3055-3093) / MBVH::Refit (4925-4961) on the host, ConvertFrom again
 * (4612-4655, 5884-6018), upload" of the reference flow.  verts16: the caller's bvhvec4 vertex array
 * (3 per triangle, n_tris triangles, same indexing as at build time); host memory (on_device = 0, staged
 * asynchronously) or device memory (1).  Works on all three layouts, including reference-built blobs
 * (BVH4_GPU: the first call walks the stream once to list its nodes level by level).  Returns when the refit is done (the bottom-up passes are
 * launched in batches with one 4-byte read-back each); tbvh_time_last_ms() = time spent refitting.
 * A vertex array shorter than the blob's primitive indices is reported (TBVH_E_FORMAT) by the next
 * synchronising call. */
int tbvh_refit(tbvh_scene* scene, const void* verts16, uint64_t n_tris, int on_device);

/* Per-frame TLAS rebuild ON THE DEVICE (no host build, no node upload): does BLASInstance::Update
 * (tiny_bvh.h:8386-8427) for every instance and builds a new BVH_GPU-format TLAS over them, replacing
 * BVH::Build(BLASInstance*, ...) + BVH_GPU::ConvertFrom of the reference frame loop
 * (tiny_bvh.h:2221-2259, 4612-4655; tiny_bvh_gpu2.cpp:113-130).  The instance records on the device
 * keep their blasIdx / mask; transform, invTransform, aabbMin, aabbMax are rewritten.
 *   transforms    n_instances x 16 floats (BLASInstance::transform, row-major); host memory
 *                 (on_device = 0, staged asynchronously) or device memory (on_device = 1); NULL keeps
 *                 the transforms already in the records (e.g. written there by the caller's kernel)
 *   blas_bounds6  per BLAS min.xyz, max.xyz (the BLAS root box: bvhNode[0].aabbMin/aabbMax); needed on
 *                 the first call, NULL afterwards
 * Asynchronous on the context's stream; tbvh_time_last_ms() reports the device time of the rebuild.
 * The instance update follows BLASInstance::Update / InvertTransform operation for operation (including the FMA
 * contraction of the reference build), so the records equal the ones tinybvh computes bit for bit.
 * The tree is an LBVH, not the reference's binned-SAH TLAS: same hit records, different node order. */
int tbvh_rebuild_tlas_device(tbvh_scene* tlas, const void* transforms, int on_device,
                             const float* blas_bounds6, uint64_t n_blas);
/* Read the TLAS back (tests, inspection): any of the three buffers may be NULL. */
int tbvh_tlas_download(tbvh_scene* tlas, void* tlas_nodes64, uint64_t cap_nodes, uint32_t* tlas_idx, uint64_t cap_idx,
                       void* instances192, uint64_t cap_instances, uint64_t* n_nodes_out);
void     tbvh_free_scene(tbvh_scene* scene);
int      tbvh_scene_layout(const tbvh_scene* scene);
uint64_t tbvh_scene_device_bytes(const tbvh_scene* scene);

/* ------------------------------------------------------------------------------------
 * queries — replace Kernel::SetArguments + Kernel::Run(count, 64) on batch_ailalaine /
 * batch_gpu4way / batch_cwbvh (traverse_bvh2.cl:209-219, traverse_bvh4.cl:277-286,
 * traverse_cwbvh.cl:554-570; tiny_bvh_speedtest.cpp:1117-1133) and the per-ray host API
 * X::Intersect(Ray&) / X::IsOccluded(const Ray&).
 * ---------------------------------------------------------------------------------- */

/* Host ray arrays.  stride_bytes is 64 (packed) or 128 (a tinybvh::Ray[] passed in
 * place); the first 64 bytes of each record are uploaded, traversed, and bytes 44..63
 * copied back. */
int tbvh_intersect(tbvh_scene* scene, void* rays, uint64_t n_rays, uint32_t stride_bytes);
/* occluded[i] = 1 if anything lies in [0, hit.t] along ray i, else 0
 * (BVH::IsOccluded, tiny_bvh.h:3382-3453; isoccluded_* in the .cl files). */
int tbvh_occluded(tbvh_scene* scene, const void* rays, uint64_t n_rays,
                  uint32_t stride_bytes, uint8_t* occluded);
/* Host arrays of more than 16 k rays are pipelined in groups of ~1 M rays: host threads pack group g + 1 into pinned buffers while the link
 * carries it up, the device traces group g, a second stream carries its 20 result bytes per ray down (full duplex) and the host scatters
 * group g - 1's results into the caller's records (smaller batches: two strided copies around one launch, no threads).  TBVH_HOST_THREADS =
 * host threads used (default: every core the process may use, up to 16).  tbvh_time_last_ms then reports the sum of the groups' kernel times.
 * WHAT TO EXPECT: a host array is bound by the HOST, not by the GPU — 16.7 M pageable tinybvh::Ray records cost ~6 GB of host memory traffic
 * per call (pack + scatter) next to 1.4 GB on the link, and run at 0.3-0.75 of the link rate depending on the box's cores and memory
 * (220-500 MRays/s measured on 16-core MI355X hosts) against 5-7 GRays/s for device-resident rays.  A caller that traces a batch more than once, or
 * can generate its rays where it likes, should use tbvh_pinned_malloc below (packed rays: 0.85 of the link rate) or the device entry points.
 *
 * The tinyocl::Buffer( bytes ) of this boundary (a Buffer made without a host pointer owns its host side, tiny_ocl.h; tiny_bvh_speedtest.cpp:1101-1108
 * wraps its ray array in one before every GPU block): tbvh_pinned_malloc hands out page-locked host memory.  A PACKED (64-byte stride) ray array that lives
 * there goes up by DMA straight from it, without the packing pass (a 128-byte-stride array is packed by the host threads wherever it lives: letting the
 * device read it in place costs a 128-byte read per 64 useful bytes, profiles/r05_link_rate.txt).  The library never page-locks memory it did not allocate:
 * registering caller memory (hipHostRegister) made later, unrelated pageable copies fault the GPU on this stack (DESIGN.md par. 0, "found on the way").
 * Memory not given back by tbvh_pinned_free goes with the context. */
int tbvh_pinned_malloc(tbvh_context* ctx, uint64_t bytes, void** out);
int tbvh_pinned_free(tbvh_context* ctx, void* ptr);

/* Device-resident packed rays (64-byte stride, 16-byte aligned).  Asynchronous on the
 * context's stream; no host copies.  This is the timed path.
 * Records are the reference's (BVH::Intersect: prim exact, t / u / v bit-identical) with ONE deliberate difference: among
 * triangles a ray hits at exactly the same t the reference reports whichever it tested last (tiny_bvh.h:1656 accepts
 * t <= hit.t), so its answer depends on the traversal order and its own layouts disagree with each other there; this
 * library reports the one with the smaller primitive index (TLAS: then the smaller instance index), whatever the layout,
 * the schedule or the batch size, among the triangles a traversal tests.  (Residual, shared with every BVH traversal including
 * the reference's: WHICH triangles are tested is decided by box tests, and where such a test decides at rounding level — a hit
 * grazing a triangle's edge, a ray that starts in or skims the plane of an axis-aligned triangle, coplanar triangles within ulps
 * of one t — two traversals of one scene can part: the 8-wide copy a BVH_GPU / BVH4_GPU scene is traced through and the
 * wave-packet kernel test more candidates than the native / per-lane kernels and report more of these (real) hits.  DESIGN.md
 * par. 4 has the classes and measured rates; tbvh_set_variant(scene, 1) and TBVH_COHERENT_TUNER=2 pin one traversal.)
 * Rays with a zero-length direction (rD = 1e30, tinybvh_safercp) or an infinite origin get the reference's answer; rays with NaN components
 * or an infinite direction component are traced without fault and without disturbing other rays, but what they report is unspecified
 * (the reference's own answer for them depends on the order of its comparisons).  Degenerate and duplicate triangles are fine
 * (tests/test_gpu_parity.py: test_hostile_geometry_and_rays). */
int tbvh_intersect_device(tbvh_scene* scene, void* d_rays64, uint64_t n_rays);
int tbvh_occluded_device(tbvh_scene* scene, const void* d_rays64, uint64_t n_rays,
                         uint8_t* d_occluded);
/* tbvh_reset_hits_device(rays, n, tmax) fused into tbvh_intersect_device: every ray starts from
 * hit = {tmax, 0, 0, 0} whatever its record holds (what constructing the Ray again would give,
 * tiny_bvh.h:695-703) and its record is always written (a miss stores {tmax, 0, 0, 0}).  For
 * callers that re-trace a resident batch, e.g. one frame after another. */
int tbvh_intersect_device_fresh(tbvh_scene* scene, void* d_rays64, uint64_t n_rays, float tmax);

/* ONE ray array over SEVERAL devices (the reference has a single process-global OpenCL device, tiny_ocl.h:362-364;
 * SURVEY.md §8(e)): scenes[i] is the same BVH uploaded through the context of device i (the BVH is replicated), the
 * array is cut into n_devices contiguous shards whose boundaries are multiples of 64 rays (tbvh_shard_range), one host
 * thread per device stages, traces and reads back its shard, and the results land in the caller's array in place —
 * bytes 44..63 of each record, or occluded[i], exactly as tbvh_intersect / tbvh_occluded on one device would write
 * them.  No exchange between devices.  Every scene must belong to a different context.  With n_devices = 1 this is
 * tbvh_intersect / tbvh_occluded. */
int tbvh_intersect_sharded(tbvh_scene* const* scenes, uint32_t n_devices, void* rays, uint64_t n_rays, uint32_t stride_bytes);
int tbvh_occluded_sharded(tbvh_scene* const* scenes, uint32_t n_devices, const void* rays, uint64_t n_rays,
                          uint32_t stride_bytes, uint8_t* occluded);
/* The same with DEVICE-RESIDENT rays — nothing crosses the host: d_rays64[i] / n_rays[i] is the batch resident on the device of
 * scenes[i] (produced there: a wavefront path tracer per device, or a kernel of the caller's; tbvh_shard_range cuts a global batch).
 * One host thread enqueues every device's launch (each is asynchronous on its context's stream), then waits for all; records are
 * written in place as by tbvh_intersect_device (fresh != 0: as by tbvh_intersect_device_fresh with tmax), flags as by
 * tbvh_occluded_device.  kernel_ms[i] (optional) = device time of device i's launch, dispatch_ms[i] (optional) = host time spent
 * enqueueing it — the per-device dispatch gap of SURVEY.md par. 8(e). */
int tbvh_intersect_sharded_device(tbvh_scene* const* scenes, uint32_t n_devices, void* const* d_rays64, const uint64_t* n_rays,
                                  int fresh, float tmax, float* kernel_ms, float* dispatch_ms);
int tbvh_occluded_sharded_device(tbvh_scene* const* scenes, uint32_t n_devices, const void* const* d_rays64, const uint64_t* n_rays,
                                 uint8_t* const* d_occluded, float* kernel_ms, float* dispatch_ms);
/* shard `rank` of `world` covers rays [*begin, *end) */
void tbvh_shard_range(uint64_t n_rays, uint32_t rank, uint32_t world, uint64_t* begin, uint64_t* end);

/* Re-arm a device ray batch for another Intersect: hit = {tmax, 0, 0, 0} for every record
 * (what re-running the tinybvh::Ray constructor's hit.t = t would do, tiny_bvh.h:700). */
int tbvh_reset_hits_device(tbvh_context* ctx, void* d_rays64, uint64_t n_rays, float tmax);

/* HIP-event time of the most recent query kernel on this context, in milliseconds
 * (mirrors the CL_PROFILING_COMMAND_START/END read of tiny_bvh_speedtest.cpp:1126-1131).
 * Synchronizes the stream. */
float tbvh_time_last_ms(tbvh_context* ctx);

/* The same without a synchronisation per call: the HIP-event durations (ms) of the most recent timed operations on this
 * context (queries, refits, builds, rebuilds: everything tbvh_time_last_ms reports), oldest first, at most `cap` and at most
 * the 256 the context remembers; *count = how many were written.  A renderer (or bench.py's timed loop) enqueues its
 * launches back to back and reads the durations once at the end — per-launch tbvh_time_last_ms() calls expose every
 * launch's latency (0.4 ms per two-query step on the round-3 driver box).  Waits for the operations it reports.
 * An entry is -1 when that operation failed before its end was recorded. */
int tbvh_time_history(tbvh_context* ctx, float* ms, uint32_t cap, uint32_t* count);

/* The machine's own ceilings, measured where the kernels run (bench.py's roofline denominators; best of `reps` launches, 0 = 3):
 *   copy   GB/s read + written by a streaming copy over `bytes` (one float4 per thread, non-temporal; MI355X: 8 TB/s HBM3E peak on
 *          the data sheet, 6.3-6.5 measured: tools/ubench/copy_rate.hip);
 *   read   GB/s of a read-only sweep over `bytes` (the traversal kernels' traffic is almost all reads);
 *   valu   1e9 wave64 VALU instructions per second over the whole chip for the instruction mix of the CWBVH node test at 8 waves per
 *          SIMD, clock throttling included (tools/ubench/valu_issue.hip). */
int tbvh_measure_copy_bandwidth(tbvh_context* ctx, uint64_t bytes, uint32_t reps, double* gbps);
int tbvh_measure_read_bandwidth(tbvh_context* ctx, uint64_t bytes, uint32_t reps, double* gbps);
int tbvh_measure_valu_issue(tbvh_context* ctx, uint32_t reps, double* ginstr_per_s);
/* the host link: GB/s of a pinned hipMemcpyAsync of `bytes` up and down, best of `reps` (bench.py: detail.host_rays) */
int tbvh_measure_link_bandwidth(tbvh_context* ctx, uint64_t bytes, uint32_t reps, double* h2d_gbps, double* d2h_gbps);

/* BVH8_CWBVH placement for INCOHERENT batches (the blob the caller uploaded is unchanged; "the library may keep a re-laid-out copy"):
 * a copy of the nodes in surface-area priority order — the nodes a ray is most likely to visit first —, the first packed_nodes of them
 * (rounded down to a multiple of 8) packed 80 bytes apart, all later ones one per 128-byte line, plus the triangle records padded to 64
 * bytes.  The top of the tree is served by the L2s, where the packed form moves 40 % more nodes per second; deep nodes and triangles come
 * from beyond them, where a cache line is the unit and a packed node straddles 1.6 lines, a 48-byte record 1.4 (tools/ubench/
 * gather_lanes.hip).  Scenes of 48 - 384 MB get these copies from their first launch of 2 M rays or more (packed_nodes =
 * 8192; every node on a line of its own also carries ONE of its triangles in the line's spare 48 bytes, and the traversal reads that triangle
 * from there), and batches of 2 M rays and more whose coherence probe says "incoherent" are traced on them (DESIGN.md par. 5; TBVH_INCOHERENT_COPIES=0 turns that off); this call (re)builds
 * them with another split for any BVH8_CWBVH scene — tbvh_set_variant(scene, 90) then traces every batch on them.  packed_nodes >= the
 * node count: priority order, all packed; 0: all padded; < 0: drop the node copy.  Hit records do not depend on the placement.  Kept
 * current by tbvh_refit. */
int tbvh_cwbvh_set_hybrid(tbvh_scene* scene, int64_t packed_nodes);

/* Which schedule serves COHERENT batches of 2 M rays and more on a BVH8_CWBVH scene — (1) deferred triangles + a gated triangle phase on a third more
 * waves, (2) the strict per-lane schedule, or (3) ONE traversal per wave of 64 consecutive rays (kernels_cwbvh_packet.hip: +24 % on camera rays of the 2.83 M-
 * triangle street at 16.7 M rays, +46 % on the same street off the axes, slower on shadow rays and on batches of a few M rays) — is measured by the library
 * during the scene's first such launches: no static property of a blob tells which is faster (profiles/r04_sensitivity.txt, profiles/r05_packet.txt).
 * The decision as something a caller can READ, KEEP and GIVE BACK (a renderer that wants its first frame at full speed and the same schedule from run to
 * run): one entry per batch-size class (fewer than 6 M rays, fewer than 12 M, more) and query kind; 0 = not decided yet (the library is still alternating
 * and timing — with its own events, so tbvh_set_timing(0) does not stop it; batches whose size only the device knows are never sampled and run schedule 1),
 * 1, 2, 3 as above.  tbvh_scene_set_schedule_hint pins the non-zero entries (no measuring launches at all for those classes; kept across
 * tbvh_update_cwbvh) and sends the zero ones back to measuring.  The struct is 8 plain bytes: store it next to the scene's blob cache
 * (tbvh_cwbvh_file_write) if it should outlive the process.  Measured decisions are taken from the best of 3 device-timed launches per schedule, 3 %
 * apart at least; other work on the GPU during those launches can tip a close call, which is what pinning is for.  Hit records do not depend on the
 * schedule: the same bytes.
 * Round 6: scenes under 48 MB are measured too, from 768 k rays on, between the per-lane kernel (entry 2: afterwards ONE unprobed kernel, as before) and the
 * packet kernel (entry 3: a probed two-kernel launch; the Sponza stand-in's camera rays x 1.2 - 1.6, a finely tessellated mesh seen from afar x 0.15 - that is
 * why it is measured); `reserved[0]` / `reserved[1]` carry their extra class of 768 k .. 1.5 M-ray batches (closest-hit / any-hit).  On a BVH_GPU / BVH4_GPU scene
 * both calls address the scene's 8-wide copy, which is what its queries run on. */
typedef struct tbvh_schedule_hint { uint8_t closest_hit[3]; uint8_t any_hit[3]; uint8_t reserved[2]; } tbvh_schedule_hint;
int tbvh_scene_get_schedule_hint(tbvh_scene* scene, tbvh_schedule_hint* out);
int tbvh_scene_set_schedule_hint(tbvh_scene* scene, const tbvh_schedule_hint* hint);

/* ------------------------------------------------------------------------------------
 * wavefront path-tracing helpers (device) — the ray generators of
 * wavefront.cl:52-287 ("Generate" / "Shade" bounce) reduced to what the benchmark
 * configurations need: primary rays from a pinhole camera in the speedtest's 4x4-tile
 * order (tiny_bvh_speedtest.cpp:517-551), and diffuse-bounce / shadow rays derived from
 * the hit records of a previous Intersect.
 * ---------------------------------------------------------------------------------- */
typedef struct tbvh_camera {
    float eye[3];  float p1[3]; float p2[3]; float p3[3];  /* view pyramid corners      */
    uint32_t width, height, spp_x, spp_y;                 /* pixels and samples/pixel   */
} tbvh_camera;
int tbvh_generate_primary_device(tbvh_context* ctx, const tbvh_camera* cam,
                                 void* d_rays64, uint64_t first, uint64_t n_rays);
/* For every ray i: if it hit (hit.t < 1e30), spawn a uniform random bounce in the hemisphere
 * about the geometric normal of hit.prim (tiny_bvh_speedtest.cpp:567-586; RNG = xorshift32
 * seeded by WangHash, tools.cl:9-11), origin I + 1e-3*R; missed rays spawn from O + 20*D.
 * d_verts16 = device copy of the original vertex array (3 x 16 bytes per triangle).
 * d_out may alias d_in. */
int tbvh_generate_bounce_device(tbvh_context* ctx, const void* d_verts16,
                                const void* d_in_rays64, void* d_out_rays64,
                                uint64_t n_rays, uint32_t seed);
/* Shadow rays from hit points toward light_pos with origin offset eps and
 * tmax = dist - eps (tiny_bvh_speedtest.cpp:851-865). */
int tbvh_generate_shadow_device(tbvh_context* ctx, const void* d_in_rays64,
                                void* d_out_rays64, uint64_t n_rays,
                                const float light_pos[3], float eps);

/* Reorder a resident ray batch into bins of (origin cell, direction octant): d_out[slot] = d_in[i], all rays of a bin adjacent, bins
 * in Morton order of the cell (2^cell_bits cells per axis over bounds6 = min.xyz, max.xyz; cell_bits 0..6).  flags: 0 = cell only,
 * 1 = the direction's sign octant as the minor part of the key, 2 = as the major part.  For incoherent batches — bounce rays from
 * depth 2 on — ahead of tbvh_intersect_device: the launch consumes a batch front to back, so the whole GPU then works on one
 * region of space at a time (DESIGN.md par. 5).  d_perm (optional, n x u32): d_perm[slot] = i, for callers that need the results
 * back in the original order.  A counting sort: three launches, asynchronous on the context's stream; tbvh_time_last_ms() reports
 * its device time.  No counterpart in the reference (wavefront.cl:236-245 appends extension rays in completion order). */
int tbvh_bin_rays_device(tbvh_context* ctx, const void* d_in_rays64, void* d_out_rays64, uint64_t n_rays, const float bounds6[6],
                         uint32_t cell_bits, uint32_t flags, uint32_t* d_perm);

/* ------------------------------------------------------------------------------------
 * device-resident wavefront path tracer — the frame loop of tiny_bvh_gpu.cpp:128-158 over
 * wavefront.cl:52-287 (Generate, { Extend, Shade } x depth, Connect, accumulate) with every
 * queue and counter on the device: one host call enqueues a whole frame, nothing is read back
 * unless `stats` is requested.  Extend / Connect are the traversal kernels above; Shade follows
 * wavefront.cl:127-246: the material of a triangle is v0.w of its first vertex = type << 24 | RGB8 (type 0 diffuse,
 * 1 = MATERIAL_LIGHT, 2 = MATERIAL_SPECULAR; RGB 0 = 70 % grey), next-event estimation towards a rectangular light
 * with multiple importance sampling, postponed BRDF pdf, the reference's path flags.  Extensions: light_size 0 = point
 * light, a two-colour sky, any number of diffuse bounces unless TBVH_WF_ONE_DIFFUSE_BOUNCE.
 * ---------------------------------------------------------------------------------- */
typedef struct tbvh_wavefront tbvh_wavefront;
typedef struct tbvh_wf_params {
    float light_pos[3], light_color[3], sky_lo[3], sky_hi[3];
    float eps;            /* ray origin offset along the new direction                      */
    uint32_t max_depth;   /* path segments per pixel (0 = 3, the reference's bounce count)  */
    uint32_t seed;        /* per-frame RNG seed                                             */
    uint32_t clear;       /* non-zero: zero the accumulator first                           */
    float light_size[2];  /* extent of the rectangular light along x and z, centred at light_pos, facing down
                             (wavefront.cl:208: 9 x 5); 0, 0 = point light                   */
    uint32_t flags;       /* TBVH_WF_*                                                      */
    uint32_t sample_index; /* the demo's spp - 1 (tiny_bvh_gpu.cpp:147): frames 0..3 take the random numbers of a path's first
                             vertex from the blue-noise table, if one is set (wavefront.cl:183-189)  */
} tbvh_wf_params;
#define TBVH_WF_ONE_DIFFUSE_BOUNCE 1u /* a path ends at its second diffuse vertex, as in wavefront.cl:233 */
#define TBVH_WF_REFERENCE_LETTER   2u /* follow wavefront.cl to the letter in the three places where Shade otherwise follows its
                                         intent: a path leaving the scene adds T * sky BEFORE the postponed pdf is divided out
                                         (wavefront.cl:151-156 vs :180), a light reached by a BSDF sample is weighted with
                                         LightPDF( D.w = 1e30 ) (:174: the weight vanishes), and bounces use tools.cl:34-39's
                                         CosWeightedDiffReflection (a world-space half sphere added to N).  For comparing images
                                         with the reference's own kernels (tests/test_wavefront_reference.py). */
#define TBVH_MATERIAL_DIFFUSE  0u     /* v0.w of a triangle's first vertex: type << 24 | 0xRRGGBB (wavefront.cl:12-13, 160) */
#define TBVH_MATERIAL_LIGHT    1u
#define TBVH_MATERIAL_SPECULAR 2u
typedef struct tbvh_wf_stats {
    uint64_t extend_rays[8];  /* nearest-hit rays traced at depth d                         */
    uint64_t shadow_rays[8];  /* any-hit rays traced after depth d                          */
    float frame_ms;           /* HIP-event time of the whole frame                          */
} tbvh_wf_stats;
int  tbvh_wavefront_create(tbvh_context* ctx, uint32_t width, uint32_t height, tbvh_wavefront** out);
void tbvh_wavefront_destroy(tbvh_wavefront* wf);
/* d_verts16: device copy of the scene's original vertex array.  stats may be NULL (fully
 * asynchronous); when given, the call synchronizes and fills it.  The frame is bracketed by ONE event pair
 * (frame_ms; tbvh_time_last_ms() after the call = the frame): its queries do not enter tbvh_time_history. */
int  tbvh_wavefront_render(tbvh_wavefront* wf, tbvh_scene* scene, const void* d_verts16, const tbvh_camera* cam,
                           const tbvh_wf_params* params, tbvh_wf_stats* stats);
/* Several devices, one image (BASELINE config 4: "wavefront path tracer, 3 bounces, ray batch sharded across 8 x MI355X"; the frame loop
 * of tiny_bvh_gpu.cpp:128-158 with the reference's single device, tiny_ocl.h:362-364, lifted): a wavefront object can stand for a BAND
 * of rows of a larger image — created with the band's height, then told where the band lies; it is then rendered with the FULL image's
 * camera and draws exactly the random numbers the full image's frame would draw for its pixels.  tbvh_wavefront_render_sharded renders
 * one frame with band i on the device of wfs[i] / scenes[i] (the scene uploaded once per device; d_verts16[i] that device's vertex
 * array, or NULL for TLAS scenes): the frames are enqueued one after the other by the calling thread and run concurrently, rays are
 * generated, traced, shaded and accumulated on their device, nothing is exchanged.  stats[i] / dispatch_ms[i] (optional, n_devices
 * entries): per-band ray counts and device time, host time spent enqueueing band i.  tbvh_wavefront_read_sharded gathers the bands'
 * accumulators into one width x full_height x 4 float image.  The bands must tile the image in order; first_row and the heights are
 * multiples of 4. */
int  tbvh_wavefront_set_band(tbvh_wavefront* wf, uint32_t first_row, uint32_t full_height);   /* full_height 0: the whole image again */
int  tbvh_wavefront_render_sharded(tbvh_wavefront* const* wfs, tbvh_scene* const* scenes, const void* const* d_verts16, uint32_t n_devices,
                                   const tbvh_camera* cam, const tbvh_wf_params* params, tbvh_wf_stats* stats, float* dispatch_ms);
int  tbvh_wavefront_read_sharded(tbvh_wavefront* const* wfs, uint32_t n_devices, float* rgba);
/* TLAS scenes (the path tracer of tiny_bvh_gpu2.cpp / wavefront2.cl): one device vertex array per BLAS, in blasIdx order
 * (wavefront2.cl:183 picks bistroVerts / dragonVerts by instance); tbvh_wavefront_render then takes the TLAS scene and
 * ignores d_verts16.  Normals go to world space through the instance's inverse transform. */
int  tbvh_wavefront_set_blas_vertices(tbvh_wavefront* wf, const void* const* d_verts16_per_blas, uint64_t n_blas);
/* The blue-noise table of the demos (tiny_bvh_gpu.cpp:61-66: testdata/blue_noise_128x128x8_2d.raw, 128 x 128 x 8 32-bit words,
 * two 8-bit channels per word; wavefront.cl:24-31).  With a table set, frames with sample_index < 4 draw the four random numbers of
 * every path's first vertex from it.  table = NULL removes it.  The data is copied. */
int  tbvh_wavefront_set_blue_noise(tbvh_wavefront* wf, const uint32_t* table, uint64_t n_words);
/* copy the float RGBA accumulator (width * height * 4 floats, row-major) to the host */
int  tbvh_wavefront_read(tbvh_wavefront* wf, float* rgba);
/* Finalize (wavefront.cl:275-286): accumulator * scale, square root, 8 bits per channel: width * height x 0x00RRGGBB */
int  tbvh_wavefront_finalize(tbvh_wavefront* wf, float scale, uint32_t* pixels);

/* ------------------------------------------------------------------------------------
 * device buffers — replace tinyocl::Buffer for callers that keep rays resident
 * (tiny_ocl.h:130-154, 571-708).  64-bit sizes (tinyocl::Buffer::size is 32-bit,
 * tiny_ocl.h:152: 64 M rays x 64 B wraps to 0 there).
 * ---------------------------------------------------------------------------------- */
int tbvh_device_malloc(tbvh_context* ctx, uint64_t bytes, void** d_out);
int tbvh_device_free(tbvh_context* ctx, void* d_ptr);
int tbvh_copy_to_device(tbvh_context* ctx, void* d_dst, const void* src, uint64_t bytes);
int tbvh_copy_from_device(tbvh_context* ctx, void* dst, const void* d_src, uint64_t bytes);

/* ------------------------------------------------------------------------------------
 * host builder — the blobs above normally come from the caller's own tiny_bvh.h
 * (BVH_GPU::Build, BVH4_GPU::Build, BVH8_CWBVH::Build, tiny_bvh.h:4551-4560,
 * 5059-5070, 5822-5835).  For callers without it (benchmarks on a box that only has
 * this library) the same blob formats are produced by an independent binned-SAH
 * builder + wide-BVH collapse + encoders.
 * ---------------------------------------------------------------------------------- */
typedef struct tbvh_build_params {
    uint32_t bins;          /* SAH bins per axis; 0 = default (8, BVHBINS)               */
    uint32_t max_leaf_tris; /* 0 = layout default (CWBVH 1 with the optimal collapse, 3 with the greedy one; others 4) */
    uint32_t threads;       /* 0 = hardware concurrency                                  */
    uint32_t flags;         /* TBVH_BUILD_* | (triangle cost in 1/100 of a node visit, 16 bits) << 8 | (split budget in percent) << 24, 0 = defaults */
} tbvh_build_params;
#define TBVH_BUILD_OPTIMAL_COLLAPSE 2u /* wide layouts: SAH-optimal collapse (Ylitie et al. 2017 dynamic program: merges
                                          <= 3-triangle subtrees into leaves, fills nodes) instead of the surface-area-greedy
                                          collapse (the strategy of MBVH::ConvertFrom, tiny_bvh.h:4975-5048).  DEFAULT for
                                          BVH8_CWBVH and BVH4_GPU, with a triangle test priced like a node visit
                                          (flags >> 8 = 100). */
#define TBVH_BUILD_GREEDY_COLLAPSE 4u  /* force the greedy collapse (the reference's strategy) */
#define TBVH_BUILD_SPLIT_TRIANGLES 8u  /* what BVH::BuildHQ's spatial splits are for (tiny_bvh.h:2623-3040): triangles whose boxes are
                                          mostly empty (large, off the coordinate axes) are cut into pieces BEFORE the binned-SAH build,
                                          up to (flags >> 24) per cent extra references (0 = 30), shared out by wasted box area x tree
                                          level (Karras & Aila 2013 section 4.3).  A triangle may then sit in several leaves, as in a
                                          BuildHQ tree (bvh8Tris / primIdx grow by the budget at most).  DEFAULT for BVH8_CWBVH at 30 %
                                          (measured +1 ... +8 % MRays/s, profiles/r05_rotated.txt); off for the other layouts. */
#define TBVH_BUILD_WHOLE_TRIANGLES 16u /* never split: every triangle in exactly one leaf, primIdx a permutation (as BVH::Build) */

int tbvh_host_build(const void* verts16, uint64_t n_tris, int layout,
                    const tbvh_build_params* params, tbvh_hostbvh** out);
/* TLAS over instances192 (BLASInstance records whose transform[] and blasIdx are set);
 * fills invTransform / aabbMin / aabbMax like BLASInstance::Update (tiny_bvh.h:8386-8427)
 * from blas_bounds (6 floats per BLAS: min.xyz, max.xyz) and builds a BVH_GPU TLAS. */
int tbvh_host_build_tlas(void* instances192, uint64_t n_instances,
                         const float* blas_bounds6, uint64_t n_blas, tbvh_hostbvh** out);
void     tbvh_host_free(tbvh_hostbvh* h);

/* ---- blob cache: files of BVH8_CWBVH::Save / Load (tiny_bvh.h:5786-5820) ------------------------------------------
 * The reference's file is: u32 header (sub | minor<<8 | major<<16 | layout<<24), u32 triCount, a raw dump of the C++
 * object, the node blocks, the triangle blocks.  The object dump ties a file to one tinybvh version and C++ ABI; these
 * two functions read and write the files of tinybvh 1.6.7 built for x86-64 (LP64), the version this library mirrors,
 * and refuse anything else with TBVH_E_FORMAT (same checks as Load: version, layout, triangle count; plus the file
 * length against the counts in the object dump, plus the structural validation every uploaded blob gets).
 * A file written here loads in BVH8_CWBVH::Load and traces there; a file written by BVH8_CWBVH::Save reads here. */
/* nodes16 / tris16: the blobs of tbvh_upload_cwbvh (bvh8Data, bvh8Tris); n_tris: BVH8_CWBVH::triCount (primitives);
 * bounds6: min.xyz, max.xyz of the scene for the object's aabbMin / aabbMax, or NULL to use the root node's box. */
int tbvh_cwbvh_file_write(const char* path, const void* nodes16, uint64_t n_node_blocks, const void* tris16,
                          uint64_t n_tri_blocks, uint64_t n_tris, const float* bounds6);
/* expected_tris: as BVH8_CWBVH::Load's expectedTris (mismatch = TBVH_E_FORMAT); 0 accepts any count.  On success *out
 * is a host BVH of layout TBVH_LAYOUT_CWBVH holding blobs 0 (nodes) and 1 (triangles) — pass it to tbvh_upload_host —
 * and *n_tris_out (optional) the file's triCount. */
int tbvh_cwbvh_file_read(const char* path, uint64_t expected_tris, tbvh_hostbvh** out, uint64_t* n_tris_out);
int      tbvh_host_layout(const tbvh_hostbvh* h);
/* blob accessors; which = 0 nodes, 1 prim indices / triangles (layout dependent):
 *   BVH2_WALD: 0 = 32-byte nodes, 1 = primIdx (u32)
 *   BVH_GPU  : 0 = 64-byte nodes, 1 = primIdx (u32)
 *   BVH4_GPU : 0 = 16-byte blocks
 *   CWBVH    : 0 = node blocks (16 B, 5 per node), 1 = triangle blocks (16 B, 3 per tri) */
const void* tbvh_host_blob(const tbvh_hostbvh* h, int which);
uint64_t    tbvh_host_blob_count(const tbvh_hostbvh* h, int which); /* elements, see above */
/* Convenience: upload a host-built BVH (verts16 needed for BVH_GPU only). */
int tbvh_upload_host(tbvh_context* ctx, const tbvh_hostbvh* h, const void* verts16,
                     uint64_t n_tris, tbvh_scene** out);

#ifdef __cplusplus
}
#endif
#endif /* TINYBVH_AMD_H_ */
