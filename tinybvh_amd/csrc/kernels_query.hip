// kernels_query.hip — BVH_GPU (Aila-Laine 2-wide) and BVH4_GPU Intersect / IsOccluded kernels
// for gfx950 (MI355X), plus the upload-time triangle gather.
//
// Replace (from scratch, not a port) the reference's OpenCL entry points
//   batch_ailalaine / isoccluded_ailalaine   traverse_bvh2.cl:80-219
//   batch_gpu4way   / isoccluded_gpu4way     traverse_bvh4.cl:74-286
// Result semantics follow the CPU oracle BVH::Intersect / IsOccluded (tiny_bvh.h:3247-3304,
// 3408-3453, 1644-1656), not the .cl files (strict comparisons, native_recip): a candidate is
// rejected only if t < 0 || t > tmax, |det| < 1e-6 misses, nodes whose entry distance equals the
// current hit distance are still visited, a miss leaves the ray record untouched.
//
// Structure (same as kernels_cwbvh.hip): persistent one-wave workgroups, one lane = one ray,
// per-lane ray replacement from a wave-local pool (ray_pool.h), per-lane traversal stack with
// its top in LDS and a global spill area, and an interleaved schedule in which a lane does at
// most one triangle test and one node visit per iteration.  For the 2-wide layout the
// triangle phase additionally waits until TRI_MIN lanes have a leaf pending (leaves are rare
// events there: ~6 % of the steps), which is Aila & Laine's "postpone the leaf" idea expressed
// with wave64 ballots.
#include "device_common.h"
#include "lane_stack.h"
#include "ray_pool.h"
#include "ray_split.h"
#include "kernels.h"

namespace tbvh {

namespace {

constexpr int WG = 64;

__device__ __forceinline__ float fmin3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
__device__ __forceinline__ float fmax3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

// per-lane stack of 32-bit entries: [entry][lane] in LDS, [entry][globalLane] in the spill area (lane_stack.h)
template <int LDS_N> using Stack32 = LaneStack<uint32_t, LDS_N, WG>;

// ---------------------------------------------------------------------------------------
// BVH_GPU (Aila-Laine 2-wide).  nodes: 4 x float4 per node, verbatim BVH_GPU::bvhNode.
// tris: 3 x float4 per primIdx entry {v0.xyz|prim, e1, e2}, gathered at upload so a leaf
// reads one contiguous run instead of primIdx -> verts (two dependent gathers).
// ---------------------------------------------------------------------------------------
// STEAL > 0 (idle lanes needed): once the ray pool is dry, idle lanes take the nearest pending subtree (the top of the stack) off a lane
// that is still traversing (ray_split.h)
template <bool ANYHIT, int LDS_N, int REFILL_MIN, int TRI_MIN, bool ADAPT = false, int NODE_REPS = 1, uint32_t GOV_KEEP = kLockstepKeep, bool HAS_OMM = true, int STEAL = 0>
__global__ __launch_bounds__(WG) void k_bvh2(const float4* __restrict__ nodes, const float4* __restrict__ tris, QueryArgs q,
                                             uint32_t* __restrict__ status) {
    __shared__ uint32_t stk[LDS_N][WG];
    Stack32<LDS_N> st;
    st.init(&stk[0][threadIdx.x], q.spill + (blockIdx.x * WG + threadIdx.x), (size_t)gridDim.x * WG, q.spillStride);
    RayPool<64> pool;
    const uint64_t nRaysTotal = q.nRaysDev ? *q.nRaysDev : q.nRays;   // batch size may live on the device (wavefront queues)
    pool.init(q.poolParts, q.counterNext);

    bool active = false;
    uint64_t ri = 0;
    float3 O = make_float3(0, 0, 0), D = O, rD = O, ro = O;
    float4 hit = make_float4(0, 0, 0, 0);
    bool found = false;
    uint32_t node = 0, triLeft = 0, triPtr = 0;   // triLeft > 0: a leaf's triangles are pending
    __shared__ SplitLds<STEAL ? WG : 1> split;
    int grp = -1;

    LockstepGovernorT<GOV_KEEP, GOV_KEEP - 5u> gov;   // ADAPT only
    gov.init();
    for (;;) {
        const uint32_t nIdle = (uint32_t)__popcll(__ballot(!active));
        if ((ADAPT ? gov.want_refill(nIdle, (uint32_t)REFILL_MIN) : nIdle >= (uint32_t)REFILL_MIN) || nIdle == (uint32_t)WG) {
            if (!pool.dry()) {
                uint64_t nri = 0;
                if (pool.acquire(!active, q.counter, nRaysTotal, nri)) {
                    ri = nri;
                    const RayRec* rp = q.rays + ri;
                    O = xyz(rp->O); D = xyz(rp->D); rD = xyz(rp->rD);
                    hit = q.fresh ? make_float4(q.freshTmax, 0.f, 0.f, 0.f) : rp->hit;
                    ro = make_float3(O.x * rD.x, O.y * rD.y, O.z * rD.z);
                    found = false; node = 0; triLeft = 0; st.sp = 0;
                    active = true;
                }
            }
            if (__ballot(active) == 0) break;
        }
        const bool tail = STEAL && pool.dry();   // wave-uniform: nothing of the split-ray code costs a vector instruction before the pool is dry
        if (tail && nIdle >= (uint32_t)STEAL) {
            SplitMatch m;
            if (split_match(active && st.sp != 0, !active, m)) {
                uint32_t part = 0;
                if (m.gives) { part = st.pop(); split_give<ANYHIT>(split, m, grp, found, hit); }
                const int src = split_take_ray(split, m, O, D, rD, hit, ri, grp);
                part = __shfl(part, src);
                if (m.takes) {
                    ro = make_float3(O.x * rD.x, O.y * rD.y, O.z * rD.z);
                    found = false; node = part; triLeft = 0; st.sp = 0;
                    active = true;
                }
            }
        }
        if (!active) continue;

        bool done = false;
        if (tail && grp >= 0) split_poll<ANYHIT>(split, grp, hit, done);   // a split ray: bounded by its group's closest hit
        // ---- triangle phase -------------------------------------------------------------------
        const uint32_t nPend = (uint32_t)__popcll(__ballot(triLeft != 0));
        const bool triPhase = nPend >= (uint32_t)TRI_MIN || nPend == (uint32_t)__popcll(__ballot(true));
        if (triPhase && triLeft != 0 && !done) {
            const float4 v0 = tris[triPtr], e1 = tris[triPtr + 1], e2 = tris[triPtr + 2];
            triPtr += 3; triLeft--;
            TriHit h;
            tri_loads_together(v0);   // (BVH_GPU, A/B of two builds: Bistro stand-in camera / bounce / shadow rays +1.4 / +2.2 / +1 %, Sponza stand-in bounce rays +3 %; k_bvh4 and the two-level kernels measured neutral or slower with it)
            if (tri_test(O, D, xyz(v0), xyz(e1), xyz(e2), hit.x, h, HAS_OMM ? q.omm : Omm{nullptr, 0}, as_u32(v0.w)) &&
                (ANYHIT || (tail && grp >= 0) || hit_wins(h.t, as_u32(v0.w), found, hit))) {   // (a split ray's group arbitrates: split_publish)
                found = true;
                if (ANYHIT) done = true;
                else hit = make_float4(h.t, h.u, h.v, v0.w);
                if (tail && grp >= 0) split_publish<ANYHIT>(split, grp, hit);
            }
            if (!done && triLeft == 0) {  // leaf finished: continue with the stack
                if (st.sp == 0) done = true;
                else node = st.pop();
            }
        }
        // ---- node phase: NODE_REPS visits per iteration (the per-iteration bookkeeping — idle / pending ballots, the refill
        // check — is a sizeable part of a 2-wide step) ---------------------------------------------
#pragma unroll
        for (int rep = 0; rep < NODE_REPS; rep++)
        if (!done && triLeft == 0) {
            const float4 n0 = nodes[node * 4], n1 = nodes[node * 4 + 1], n2 = nodes[node * 4 + 2], n3 = nodes[node * 4 + 3];
            const uint32_t triCount = as_u32(n2.w);
            if (triCount) {
                triLeft = triCount; triPtr = as_u32(n3.w) * 3u;
            } else {
                // slab test of both children, oracle form: t = plane * rD - O*rD, inclusive
                // visit rule tmax >= tmin (SLAB_TEST_TWO_NODES, tiny_bvh.h:3202-3220)
                const float lx1 = __builtin_fmaf(n0.x, rD.x, -ro.x), lx2 = __builtin_fmaf(n1.x, rD.x, -ro.x);
                const float ly1 = __builtin_fmaf(n0.y, rD.y, -ro.y), ly2 = __builtin_fmaf(n1.y, rD.y, -ro.y);
                const float lz1 = __builtin_fmaf(n0.z, rD.z, -ro.z), lz2 = __builtin_fmaf(n1.z, rD.z, -ro.z);
                const float rx1 = __builtin_fmaf(n2.x, rD.x, -ro.x), rx2 = __builtin_fmaf(n3.x, rD.x, -ro.x);
                const float ry1 = __builtin_fmaf(n2.y, rD.y, -ro.y), ry2 = __builtin_fmaf(n3.y, rD.y, -ro.y);
                const float rz1 = __builtin_fmaf(n2.z, rD.z, -ro.z), rz2 = __builtin_fmaf(n3.z, rD.z, -ro.z);
                const float tminL = __builtin_fmaxf(fmax3(__builtin_fminf(lx1, lx2), __builtin_fminf(ly1, ly2), __builtin_fminf(lz1, lz2)), 0.0f);
                const float tmaxL = __builtin_fminf(fmin3(__builtin_fmaxf(lx1, lx2), __builtin_fmaxf(ly1, ly2), __builtin_fmaxf(lz1, lz2)), cull_bound(hit.x));
                const float tminR = __builtin_fmaxf(fmax3(__builtin_fminf(rx1, rx2), __builtin_fminf(ry1, ry2), __builtin_fminf(rz1, rz2)), 0.0f);
                const float tmaxR = __builtin_fminf(fmin3(__builtin_fmaxf(rx1, rx2), __builtin_fmaxf(ry1, ry2), __builtin_fmaxf(rz1, rz2)), cull_bound(hit.x));
                const bool hL = tmaxL >= tminL, hR = tmaxR >= tminR;
                uint32_t l = as_u32(n0.w), r = as_u32(n1.w);
                if (hL && hR) {
                    if (tminL > tminR) { const uint32_t t = l; l = r; r = t; }
                    st.push(r);
                    node = l;
                } else if (hL) node = l;
                else if (hR) node = r;
                else {
                    if (st.sp == 0) done = true;
                    else node = st.pop();
                }
            }
        }
        if (done) {
            if (tail && grp >= 0) split_finish<ANYHIT>(split, grp, q, ri);
            else if (ANYHIT) q.occluded[ri] = found ? 1 : 0;
            else if (found || q.fresh) q.rays[ri].hit = hit;
            active = false;
        }
    }
    if (st.overflow) atomicOr(status, 1u);
}

// ---------------------------------------------------------------------------------------
// BVH4_GPU.  One float4 stream, verbatim BVH4_GPU::bvh4Data.
// Per-ray order follows the CPU mirror (tiny_bvh.h:5252-5343): children sorted far to near,
// interior children pushed far first (nearest on top), hit leaves processed in the same
// sorted order.  The hit leaves of a node (at most 4) are queued in registers and their
// triangles tested one per iteration.
// ---------------------------------------------------------------------------------------
// TIMELINE (experiment build): when the waves start, see the ray pool run dry and end, in 10 ns ticks of the constant
// clock, folded into q.stats as {~min start, max start, sum start, ~min end, max end, sum end, sum dry, waves}
template <bool ANYHIT, int LDS_N, int REFILL_MIN, int TRI_MIN, bool ADAPT, int NODE_REPS, bool SIGNSEL, uint32_t GOV_KEEP, bool HAS_OMM, bool TIMELINE = false, int STEAL = 0>
__device__ __forceinline__ void bvh4_body(const float4* __restrict__ data, const QueryArgs& q, uint32_t* __restrict__ status) {
    __shared__ uint32_t stk[LDS_N][WG];
    Stack32<LDS_N> st;
    st.init(&stk[0][threadIdx.x], q.spill + (blockIdx.x * WG + threadIdx.x), (size_t)gridDim.x * WG, q.spillStride);
    RayPool<64> pool;
    const uint64_t nRaysTotal = q.nRaysDev ? *q.nRaysDev : q.nRays;   // batch size may live on the device (wavefront queues)
    pool.init(q.poolParts, q.counterNext);
    const unsigned long long tStart = TIMELINE ? wall_clock64() : 0ull;
    unsigned long long tDry = 0ull;

    bool active = false;
    uint64_t ri = 0;
    float3 O = make_float3(0, 0, 0), D = O, rD = O;
    float4 hit = make_float4(0, 0, 0, 0);
    bool found = false;
    uint32_t offset = 0;
    // pending leaves of the current node in processing order: leafQ0 first.  leafQn = absolute
    // block offset of the leaf's next triangle; remaining triangle counts in 16-bit fields (the
    // format's count is 15 bits): slots 0, 1 in leafCnt (slot 0 low), slots 2, 3 in leafCntB.
    // The queue is kept compacted towards slot 0, so leafCnt == 0 means no leaf is pending.
    uint32_t leafQ0 = 0, leafQ1 = 0, leafQ2 = 0, leafQ3 = 0, leafCnt = 0, leafCntB = 0;
    __shared__ SplitLds<STEAL ? WG : 1> split;   // STEAL as in k_bvh2
    int grp = -1;

    LockstepGovernorT<GOV_KEEP, GOV_KEEP - 5u> gov;   // ADAPT only
    gov.init();
    for (;;) {
        const uint32_t nIdle = (uint32_t)__popcll(__ballot(!active));
        if ((ADAPT ? gov.want_refill(nIdle, (uint32_t)REFILL_MIN) : nIdle >= (uint32_t)REFILL_MIN) || nIdle == (uint32_t)WG) {
            if (!pool.dry()) {
                uint64_t nri = 0;
                if (pool.acquire(!active, q.counter, nRaysTotal, nri)) {
                    ri = nri;
                    const RayRec* rp = q.rays + ri;
                    O = xyz(rp->O); D = xyz(rp->D); rD = xyz(rp->rD);
                    hit = q.fresh ? make_float4(q.freshTmax, 0.f, 0.f, 0.f) : rp->hit;
                    found = false; offset = 0; leafCnt = 0; leafCntB = 0; st.sp = 0;
                    active = true;
                }
            }
            if (TIMELINE && !tDry && pool.dry()) tDry = wall_clock64();
            if (__ballot(active) == 0) break;
        }
        const bool tail = STEAL && pool.dry();   // wave-uniform: nothing of the split-ray code costs a vector instruction before the pool is dry
        if (tail && nIdle >= (uint32_t)STEAL) {
            SplitMatch m;
            if (split_match(active && st.sp != 0, !active, m)) {
                uint32_t part = 0;
                if (m.gives) { part = st.pop(); split_give<ANYHIT>(split, m, grp, found, hit); }
                const int src = split_take_ray(split, m, O, D, rD, hit, ri, grp);
                part = __shfl(part, src);
                if (m.takes) {
                    found = false; offset = part; leafCnt = 0; leafCntB = 0; st.sp = 0;
                    active = true;
                }
            }
        }
        if (!active) continue;

        bool done = false;
        if (tail && grp >= 0) split_poll<ANYHIT>(split, grp, hit, done);   // a split ray: bounded by its group's closest hit
        const uint32_t nPend = (uint32_t)__popcll(__ballot(leafCnt != 0));
        const bool triPhase = nPend >= (uint32_t)TRI_MIN || nPend == (uint32_t)__popcll(__ballot(true));
        if (triPhase && leafCnt != 0 && !done) {
            const uint32_t ta = leafQ0;
            const float4 v0 = data[ta], e1 = data[ta + 1], e2 = data[ta + 2];
            leafQ0 += 3u; leafCnt -= 1u;                      // next triangle, one fewer left in slot 0
            if ((leafCnt & 0xffffu) == 0) {
                leafQ0 = leafQ1; leafQ1 = leafQ2; leafQ2 = leafQ3;
                leafCnt = __builtin_amdgcn_alignbit(leafCntB, leafCnt, 16); leafCntB >>= 16;
            }
            TriHit h;
            if (tri_test(O, D, xyz(v0), xyz(e1), xyz(e2), hit.x, h, HAS_OMM ? q.omm : Omm{nullptr, 0}, as_u32(v0.w)) &&
                (ANYHIT || (tail && grp >= 0) || hit_wins(h.t, as_u32(v0.w), found, hit))) {   // (a split ray's group arbitrates: split_publish)
                found = true;
                if (ANYHIT) done = true;
                else hit = make_float4(h.t, h.u, h.v, v0.w);
                if (tail && grp >= 0) split_publish<ANYHIT>(split, grp, hit);
            }
            if (!done && leafCnt == 0) {
                if (st.sp == 0) done = true;
                else offset = st.pop();
            }
        }
#pragma unroll
        for (int rep = 0; rep < NODE_REPS; rep++)   // several node visits per iteration, as in k_bvh2
        if (!done && leafCnt == 0) {
            const float4 d0 = data[offset], d1 = data[offset + 1], d2 = data[offset + 2], d3 = data[offset + 3];
            // per-axis: t(q) = (bmin + ext*q - O) * rD = q * (ext*rD) + (bmin - O)*rD
            const float sx = d1.x * rD.x, sy = d1.y * rD.y, sz = d1.z * rD.z;
            const float bx = (d0.x - O.x) * rD.x, by = (d0.y - O.y) * rD.y, bz = (d0.z - O.z) * rD.z;
            const uint32_t qx0 = as_u32(d0.w), qx1 = as_u32(d1.w);
            const uint32_t qy0 = as_u32(d2.x), qy1 = as_u32(d2.y), qz0 = as_u32(d2.z), qz1 = as_u32(d2.w);
            float dist[4];
            uint32_t info[4] = { as_u32(d3.x), as_u32(d3.y), as_u32(d3.z), as_u32(d3.w) };
            // SIGNSEL: the sign of the ray direction says which of the two quantised planes of an axis is the near one
            // (q0 <= q1 and t is monotonic in q), so the words are swapped once per node instead of a min and a max per
            // child and axis.  NaNs (0 * inf on degenerate axes) drop out of max3 / min3 as they do out of the min / max pairs.
            const bool ngx = sx < 0.f, ngy = sy < 0.f, ngz = sz < 0.f;
            const uint32_t nx = ngx ? qx1 : qx0, fx = ngx ? qx0 : qx1;
            const uint32_t ny = ngy ? qy1 : qy0, fy = ngy ? qy0 : qy1;
            const uint32_t nz = ngz ? qz1 : qz0, fz = ngz ? qz0 : qz1;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int sh = 8 * i;
                if (SIGNSEL) {
                    const float x1 = __builtin_fmaf((float)((nx >> sh) & 255), sx, bx), x2 = __builtin_fmaf((float)((fx >> sh) & 255), sx, bx);
                    const float y1 = __builtin_fmaf((float)((ny >> sh) & 255), sy, by), y2 = __builtin_fmaf((float)((fy >> sh) & 255), sy, by);
                    const float z1 = __builtin_fmaf((float)((nz >> sh) & 255), sz, bz), z2 = __builtin_fmaf((float)((fz >> sh) & 255), sz, bz);
                    const float tmin = __builtin_fmaxf(fmax3(x1, y1, z1), 0.0f);
                    const float tmax = __builtin_fminf(fmin3(x2, y2, z2), cull_bound(hit.x));
                    dist[i] = (tmin > tmax || info[i] == 0) ? kFar : tmin;
                    continue;
                }
                const float x1 = __builtin_fmaf((float)((qx0 >> sh) & 255), sx, bx), x2 = __builtin_fmaf((float)((qx1 >> sh) & 255), sx, bx);
                const float y1 = __builtin_fmaf((float)((qy0 >> sh) & 255), sy, by), y2 = __builtin_fmaf((float)((qy1 >> sh) & 255), sy, by);
                const float z1 = __builtin_fmaf((float)((qz0 >> sh) & 255), sz, bz), z2 = __builtin_fmaf((float)((qz1 >> sh) & 255), sz, bz);
                const float tmin = __builtin_fmaxf(fmax3(__builtin_fminf(x1, x2), __builtin_fminf(y1, y2), __builtin_fminf(z1, z2)), 0.0f);
                const float tmax = __builtin_fminf(fmin3(__builtin_fmaxf(x1, x2), __builtin_fmaxf(y1, y2), __builtin_fmaxf(z1, z2)), cull_bound(hit.x));
                dist[i] = (tmin > tmax || info[i] == 0) ? kFar : tmin;
            }
            // 5-comparator network, descending (farthest first)
#define TBVH_CSWAP(a, b) if (dist[a] < dist[b]) { const float tf = dist[a]; dist[a] = dist[b]; dist[b] = tf; const uint32_t tu = info[a]; info[a] = info[b]; info[b] = tu; }
            TBVH_CSWAP(0, 2) TBVH_CSWAP(1, 3) TBVH_CSWAP(0, 1) TBVH_CSWAP(2, 3) TBVH_CSWAP(1, 2)
#undef TBVH_CSWAP
            uint32_t nq = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (!(dist[i] < kFar)) continue;
                if (!(info[i] & 0x80000000u)) { st.push(info[i]); continue; }
                const uint32_t cnt = (info[i] >> 16) & 0x7fffu;
                if (cnt == 0) continue;   // an empty leaf would break the compaction of the queue
                const uint32_t ta = offset + (info[i] & 0xffffu);
                if (nq == 0) leafQ0 = ta; else if (nq == 1) leafQ1 = ta; else if (nq == 2) leafQ2 = ta; else leafQ3 = ta;
                if (nq < 2) leafCnt |= cnt << (16 * nq); else leafCntB |= cnt << (16 * (nq - 2));
                nq++;
            }
            if (!done && leafCnt == 0) {
                if (st.sp == 0) done = true;
                else offset = st.pop();
            }
        }
        if (done) {
            if (tail && grp >= 0) split_finish<ANYHIT>(split, grp, q, ri);
            else if (ANYHIT) q.occluded[ri] = found ? 1 : 0;
            else if (found || q.fresh) q.rays[ri].hit = hit;
            active = false;
        }
    }
    if (st.overflow) atomicOr(status, 1u);
    if (TIMELINE && threadIdx.x == 0 && (blockIdx.x % 31u) == 0) {   // every 31st wave (all XCDs): 8192 waves on eight addresses would serialise for longer than the kernel runs
        const unsigned long long tEnd = wall_clock64();
        atomicMax(q.stats + 0, ~tStart); atomicMax(q.stats + 1, tStart); atomicAdd(q.stats + 2, tStart);
        atomicMax(q.stats + 3, ~tEnd); atomicMax(q.stats + 4, tEnd); atomicAdd(q.stats + 5, tEnd);
        atomicAdd(q.stats + 6, tDry ? tDry : tEnd); atomicAdd(q.stats + 7, 1ull);
    }
}

template <bool ANYHIT, int LDS_N, int REFILL_MIN, int TRI_MIN, bool ADAPT = false, int NODE_REPS = 1, bool SIGNSEL = false, uint32_t GOV_KEEP = kLockstepKeep, bool HAS_OMM = true>
__global__ __launch_bounds__(WG) void k_bvh4(const float4* __restrict__ data, QueryArgs q, uint32_t* __restrict__ status) {
    bvh4_body<ANYHIT, LDS_N, REFILL_MIN, TRI_MIN, ADAPT, NODE_REPS, SIGNSEL, GOV_KEEP, HAS_OMM>(data, q, status);
}
// the same with the register budget of 8 waves per SIMD (<= 64 VGPRs; left alone the compiler takes 65-68)
template <bool ANYHIT, int LDS_N, int REFILL_MIN, int TRI_MIN, bool ADAPT = false, int NODE_REPS = 1, bool SIGNSEL = false, uint32_t GOV_KEEP = kLockstepKeep, bool HAS_OMM = true, bool TIMELINE = false, int STEAL = 0>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_bvh4_w8(const float4* __restrict__ data, QueryArgs q, uint32_t* __restrict__ status) {
    bvh4_body<ANYHIT, LDS_N, REFILL_MIN, TRI_MIN, ADAPT, NODE_REPS, SIGNSEL, GOV_KEEP, HAS_OMM, TIMELINE, STEAL>(data, q, status);
}

// ---------------------------------------------------------------------------------------
// upload helper: gather {v0|prim, e1, e2} per primIdx entry for the BVH_GPU layout.
// e1 = v1 - v0, e2 = v2 - v0 are the same single IEEE subtractions IntersectTri performs
// per test (tiny_bvh.h:8510-8511), so pre-computing them changes no result bit.
// ---------------------------------------------------------------------------------------
__global__ void k_gather_tris(const uint32_t* __restrict__ primIdx, const float4* __restrict__ verts,
                              float4* __restrict__ out, uint64_t nIdx, uint64_t nTris) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nIdx) return;
    const uint32_t p = primIdx[i];
    if (p >= nTris) {  // slack entries of SBVH primIdx arrays (idxCount = 1.5 * triCount)
        out[i * 3] = make_float4(0, 0, 0, 0); out[i * 3 + 1] = make_float4(0, 0, 0, 0); out[i * 3 + 2] = make_float4(0, 0, 0, 0);
        return;
    }
    const float4 a = verts[(uint64_t)p * 3], b = verts[(uint64_t)p * 3 + 1], c = verts[(uint64_t)p * 3 + 2];
    out[i * 3] = make_float4(a.x, a.y, a.z, as_f32(p));
    out[i * 3 + 1] = make_float4(b.x - a.x, b.y - a.y, b.z - a.z, 0.f);
    out[i * 3 + 2] = make_float4(c.x - a.x, c.y - a.y, c.z - a.z, 0.f);
}

}  // namespace

// ---- launchers (called from capi.hip) ----------------------------------------------------

void launch_bvh2(bool anyhit, int variant, const float4* nodes, const float4* tris, const QueryArgs& q, uint32_t* status,
                 uint32_t blocks, hipStream_t s) {
#define TBVH_L2(...)                                                                                                    \
    do {                                                                                                                \
        if (anyhit) hipLaunchKernelGGL((k_bvh2<true, 16, 16, __VA_ARGS__>), dim3(blocks), dim3(WG), 0, s, nodes, tris, q, status); \
        else hipLaunchKernelGGL((k_bvh2<false, 16, 16, __VA_ARGS__>), dim3(blocks), dim3(WG), 0, s, nodes, tris, q, status);       \
    } while (0)
    switch (variant) {
    default:   // + the lockstep governor (ray_pool.h): Sponza camera rays +9 %, shadow +6 %; small incoherent batches -3..5 %
        // batches below 12 M rays, and the wavefront stages (ray count known to the device only), split their last rays over idle lanes (ray_split.h)
        if (split_rays_wanted(q)) {
            if (q.omm.map) TBVH_L2(16, true, 3, kLockstepKeep, true, 16);
            else TBVH_L2(16, true, 3, kLockstepKeep, false, 16);
        } else if (q.omm.map) TBVH_L2(16, true, 3);
        else TBVH_L2(16, true, 3, kLockstepKeep, false);   // no opacity micromaps: the check is compiled out
        break;
    }
#undef TBVH_L2
}

void launch_bvh4(bool anyhit, int variant, const float4* data, const QueryArgs& q, uint32_t* status, uint32_t blocks, hipStream_t s) {
#define TBVH_L4(...)                                                                                                \
    do {                                                                                                            \
        if (anyhit) hipLaunchKernelGGL((k_bvh4<true, 12, 16, __VA_ARGS__>), dim3(blocks), dim3(WG), 0, s, data, q, status);  \
        else hipLaunchKernelGGL((k_bvh4<false, 12, 16, __VA_ARGS__>), dim3(blocks), dim3(WG), 0, s, data, q, status);        \
    } while (0)
#define TBVH_L4W(...)                                                                                               \
    do {                                                                                                            \
        if (anyhit) hipLaunchKernelGGL((k_bvh4_w8<true, 12, 16, __VA_ARGS__>), dim3(blocks), dim3(WG), 0, s, data, q, status);  \
        else hipLaunchKernelGGL((k_bvh4_w8<false, 12, 16, __VA_ARGS__>), dim3(blocks), dim3(WG), 0, s, data, q, status);        \
    } while (0)
    switch (variant) {
    default:   // per-lane replacement throughout, sign-selected planes, 8 waves per SIMD
        if (split_rays_wanted(q)) {   // split rays, as in launch_bvh2
            if (q.omm.map) TBVH_L4W(8, false, 1, true, kLockstepKeep, true, false, 16);
            else TBVH_L4W(8, false, 1, true, kLockstepKeep, false, false, 16);
        } else if (q.omm.map) TBVH_L4W(8, false, 1, true);
        else TBVH_L4W(8, false, 1, true, kLockstepKeep, false);   // no opacity micromaps: the check is compiled out
        break;
    }
#undef TBVH_L4
#undef TBVH_L4W
}

void launch_gather_tris(const uint32_t* primIdx, const float4* verts, float4* out, uint64_t nIdx, uint64_t nTris, hipStream_t s) {
    const uint32_t bs = 256;
    hipLaunchKernelGGL(k_gather_tris, dim3((uint32_t)((nIdx + bs - 1) / bs)), dim3(bs), 0, s, primIdx, verts, out, nIdx, nTris);
}

}  // namespace tbvh
