/* tinybvh_amd_debug.h — development aids of the MI355X engine: instrumentation counters, experiment switches, forced kernel schedules, the
 * coherence probe's last verdict.  NOT part of the boundary a tinybvh maintainer binds (include/tinybvh_amd.h is); used by tests/ and tools/.
 * Same library, same ABI version. */
#ifndef TINYBVH_AMD_DEBUG_H_
#define TINYBVH_AMD_DEBUG_H_
#include "tinybvh_amd.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Lane-utilisation counters of the instrumented kernel variants (development aid):
 * out[0] wave iterations, [1] sum of active lanes, [2] sum of lanes in the node step,
 * [3] triangle-loop iterations, [4] sum of lanes in them, [5] refill events, [6] rays handed out. */
int tbvh_debug_stats(tbvh_context* ctx, uint64_t out[8], int reset);

/* Experiment switches for the BVH8_CWBVH kernel of the next launches on this context (development aid; 0 = as shipped):
 * 1 = non-temporal ray loads / hit stores, 2 = the 64-byte triangle records in the ordinary kernels too (only where the scene has them:
 * after tbvh_cwbvh_set_hybrid or the first large launch; ignored otherwise) — both only in the experiment build (make EXPERIMENTS=1: the
 * shipped kernels do not carry the two code paths), 4 = a probed launch as ONE kernel even where the copies exist,
 * 8 = the next tbvh_cwbvh_set_hybrid / lazy build derives the node copy WITHOUT a triangle in each node's line, 16 = the coherent flavor of a two-flavor
 * launch takes the batch whatever its probe finds (tests put incoherent rays through the coherent schedules), 64 = no coherence probe (every
 * launch takes the unprobed path), bits 8..15 = waves per CU of the incoherent flavor (clamped to the 32 per CU the stack spill area is sized for). */
int tbvh_debug_set_flags(tbvh_context* ctx, uint32_t flags);

/* Which schedule the library has settled on for COHERENT batches of 2 M rays and more on this BVH8_CWBVH scene (closest-hit: anyhit = 0,
 * any-hit: 1) — the deferred-triangles + gated schedule on a third more waves, or the strict one.  No static property of a blob tells which
 * is faster (profiles/r04_sensitivity.txt: +1 ... +8 % for the first on most scenes, +10 % for the second on large-occluder scenes), so the
 * first few such launches alternate and are timed on the device (no synchronisation), then the faster stays (TBVH_COHERENT_TUNER=0 / 2 in
 * the environment pins the first / the second, 3 one traversal per wave).  out[0] = 0 still measuring, 1 deferred + gated, 2 strict, 3 one traversal per wave (kernels_cwbvh_packet.hip); out[1], out[2] = coherent samples
 * taken of each; out[3] = 1000 x best time per ray of the strict schedule / of the deferred one (0 until both have samples).  The choice is kept
 * per batch-size class (below 6 M rays, below 12 M, more: the end of a launch weighs differently — the atrium generator's camera rays are 15 % faster
 * strict at 16.7 M rays and even at 4.2 M); the call reports the class of the most recent launch. */
int tbvh_debug_coherent_schedule(tbvh_scene* scene, int anyhit, uint32_t out[4]);

/* The per-launch coherence probe of the most recent query on this context (development aid; DESIGN.md par. 3): out[0] = sampled
 * neighbouring ray pairs whose directions agree (and, for rays of finite reach, whose origins lie within 5 % of that reach), out[1] = pairs sampled, out[2] = 0 no probe ran (small batches, small or very
 * large scenes, other layouts), 1 the batch was classified incoherent (strict schedule), 2 coherent (deferred triangles, gated
 * triangle phase, a third more waves).  Synchronizes the stream. */
int tbvh_debug_last_probe(tbvh_context* ctx, uint32_t out[3]);

/* Diagnostic kernel variants of BVH8_CWBVH scenes (0 = default): 72 / 52 force the strict / the coherent schedule whatever
 * the probe says, 90 the incoherent flavor on the copies of tbvh_cwbvh_set_hybrid, 75 / 88 split the last rays whatever the batch size, 59 / 61 / 73 / 78 / 82 / 83 are the instrumented
 * kernels behind tbvh_debug_stats.  Returns TBVH_E_INVALID for anything else. */
int tbvh_set_variant(tbvh_scene* scene, int variant);

/* The BVH2 (tinybvh::BVH::BVHNode layout, 32 bytes per node, leaves of at most max_leaf_tris entries) the library derives on the HOST from an uploaded
 * BVH_GPU blob (layout 5: nodes64 + prim_idx + verts16) or BVH4_GPU stream (layout 8: blocks16 in `blob`, n_blob 16-byte blocks; prim_idx / verts16 unused)
 * before it collapses it into the scene's 8-wide copy (tinybvh_amd/csrc/capi_scene.hip: makeWideCopy).  No device involved: the tests walk the result
 * with the oracle.  Layout 5 with prim_idx == NULL = RECORD MODE, the form the library runs (the blob is read back from the device, where the triangles
 * live as gathered records): verts16 then holds n_idx records {v0|prim, e1 = v1 - v0, e2 = v2 - v0} of 48 bytes.  nodes32_out: cap_nodes x 32 bytes; recs_out (layout 8 only): the stream's triangle records {v0|prim, e1, e2} in the order the leaves
 * index them, cap_recs x 48 bytes.  Counts are returned also when a capacity is too small (TBVH_E_INVALID then): call twice.  TBVH_E_FORMAT: the root is
 * a leaf / the stream is malformed (no copy is made for such a blob). */
int tbvh_debug_wide_copy_bvh2(int layout, const void* blob, uint64_t n_blob, const uint32_t* prim_idx, uint64_t n_idx, const void* verts16, uint64_t n_tris,
                              uint32_t max_leaf_tris, void* nodes32_out, uint64_t cap_nodes, uint64_t* n_nodes_out, void* recs_out, uint64_t cap_recs,
                              uint64_t* n_recs_out);

#ifdef __cplusplus
}
#endif
#endif /* TINYBVH_AMD_DEBUG_H_ */
