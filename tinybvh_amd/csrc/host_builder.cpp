// host_builder.cpp — see host_builder.h.
//
// What the reference does for the same job (for parity of the *formats*, not the code):
//   BVH::Build binned SAH, 8 bins           tiny_bvh.h:2124-2461
//   BVH::SplitLeafs (<=3 tris for CWBVH)    tiny_bvh.h:1988-2017
//   MBVH<M>::ConvertFrom (wide collapse)    tiny_bvh.h:4975-5048
//   BVH_GPU::ConvertFrom                    tiny_bvh.h:4612-4655
//   BVH4_GPU::ConvertFrom                   tiny_bvh.h:5115-5244
//   BVH8_CWBVH::ConvertFrom                 tiny_bvh.h:5884-6018
// The encoders here emit byte-compatible blobs (same field meaning, same quantisation
// rules: conservative floor/ceil) but the tree that goes in is built by this file.
#include "host_builder.h"

#if defined(__linux__)
#include <sched.h>
#endif
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <atomic>
#include <cassert>
#include <cmath>
#include <cstring>
#include <functional>
#include <thread>

namespace tbvh {

namespace {

constexpr float kFar = 1e30f;

struct Box {
    float mn[3], mx[3];
    void reset() { mn[0] = mn[1] = mn[2] = kFar; mx[0] = mx[1] = mx[2] = -kFar; }
    void grow(const float* p) {
        for (int a = 0; a < 3; a++) { mn[a] = std::min(mn[a], p[a]); mx[a] = std::max(mx[a], p[a]); }
    }
    void grow(const Box& b) {
        for (int a = 0; a < 3; a++) { mn[a] = std::min(mn[a], b.mn[a]); mx[a] = std::max(mx[a], b.mx[a]); }
    }
    float halfArea() const {
        const float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
        return ex * ey + ey * ez + ez * ex;
    }
};

// One primitive as the builder sees it: bounds + centroid.
struct Prim {
    Box box;
    float c[3];
};

struct Builder {
    const Prim* prims;
    uint32_t* idx;  // global permutation, partitioned in place
    uint32_t bins;
    uint32_t maxLeaf;

    static constexpr uint32_t kMaxBins = 32;

    // Returns true and the split position if [first, first+count) was partitioned.
    bool split(const Box& nodeBox, uint32_t first, uint32_t count, uint32_t& mid) const {
        Box cb; cb.reset();
        for (uint32_t i = 0; i < count; i++) cb.grow(prims[idx[first + i]].c);
        float bestCost = kFar; int bestAxis = -1; uint32_t bestPos = 0;
        float bestScale = 0, bestMin = 0;
        for (int a = 0; a < 3; a++) {
            const float ext = cb.mx[a] - cb.mn[a];
            if (!(ext > 0)) continue;
            Box bb[kMaxBins]; uint32_t bc[kMaxBins];
            for (uint32_t b = 0; b < bins; b++) { bb[b].reset(); bc[b] = 0; }
            const float scale = (float)bins / ext;
            for (uint32_t i = 0; i < count; i++) {
                const Prim& p = prims[idx[first + i]];
                uint32_t b = (uint32_t)std::min((float)(bins - 1), (p.c[a] - cb.mn[a]) * scale);
                bb[b].grow(p.box); bc[b]++;
            }
            // sweep: right-to-left accumulates, then left-to-right evaluates
            float rArea[kMaxBins]; uint32_t rCnt[kMaxBins];
            Box acc; acc.reset(); uint32_t n = 0;
            for (uint32_t b = bins - 1; b >= 1; b--) {
                if (bc[b]) acc.grow(bb[b]);
                n += bc[b]; rCnt[b] = n; rArea[b] = n ? acc.halfArea() : 0.f;
            }
            acc.reset(); n = 0;
            for (uint32_t b = 0; b + 1 < bins; b++) {
                if (bc[b]) acc.grow(bb[b]);
                n += bc[b];
                if (n == 0 || rCnt[b + 1] == 0) continue;
                const float cost = acc.halfArea() * (float)n + rArea[b + 1] * (float)rCnt[b + 1];
                if (cost < bestCost) { bestCost = cost; bestAxis = a; bestPos = b + 1; bestScale = scale; bestMin = cb.mn[a]; }
            }
        }
        const float leafCost = nodeBox.halfArea() * (float)count;
        const bool mustSplit = count > maxLeaf;
        if (bestAxis < 0) {
            // all centroids coincide: nothing to gain from a spatial split
            if (!mustSplit) return false;
            mid = first + count / 2;
            return true;
        }
        // SAH termination (C_TRAV = C_INT = 1, tiny_bvh.h:125-130)
        if (!mustSplit && bestCost + nodeBox.halfArea() >= leafCost) return false;
        uint32_t i = first, j = first + count;
        while (i < j) {
            const Prim& p = prims[idx[i]];
            uint32_t b = (uint32_t)std::min((float)(bins - 1), (p.c[bestAxis] - bestMin) * bestScale);
            if (b < bestPos) i++; else std::swap(idx[i], idx[--j]);
        }
        if (i == first || i == first + count) i = first + count / 2;  // numerical corner
        mid = i;
        return true;
    }

    Box bounds(uint32_t first, uint32_t count) const {
        Box b; b.reset();
        for (uint32_t i = 0; i < count; i++) b.grow(prims[idx[first + i]].box);
        return b;
    }

    static void setNode(Node2& n, const Box& b, uint32_t leftFirst, uint32_t triCount) {
        for (int a = 0; a < 3; a++) n.mn[a] = b.mn[a], n.mx[a] = b.mx[a];
        n.leftFirst = leftFirst; n.triCount = triCount;
    }

    // Build the subtree for [first, first+count) into `nodes`, root at nodes[rootIdx]
    // (already allocated).  Depth-first with an explicit stack; children are appended as
    // adjacent pairs.
    void buildSubtree(std::vector<Node2>& nodes, uint32_t rootIdx, uint32_t first, uint32_t count) const {
        struct Item { uint32_t node, first, count; };
        std::vector<Item> stack;
        stack.push_back({rootIdx, first, count});
        while (!stack.empty()) {
            const Item it = stack.back(); stack.pop_back();
            const Box b = bounds(it.first, it.count);
            uint32_t mid;
            if (it.count == 1 || !split(b, it.first, it.count, mid)) {
                setNode(nodes[it.node], b, it.first, it.count);
                continue;
            }
            const uint32_t l = (uint32_t)nodes.size();
            nodes.push_back(Node2{}); nodes.push_back(Node2{});
            setNode(nodes[it.node], b, l, 0);
            stack.push_back({l + 1, mid, it.first + it.count - mid});
            stack.push_back({l, it.first, mid - it.first});
        }
    }
};

void buildFromPrims(const std::vector<Prim>& prims, const BuildParams& p, BVH2& out) {
    const uint32_t n = (uint32_t)prims.size();
    out.triCount = n;
    out.primIdx.resize(n);
    for (uint32_t i = 0; i < n; i++) out.primIdx[i] = i;
    out.nodes.clear();
    if (n == 0) { out.nodes.push_back(Node2{}); return; }

    Builder B;
    B.prims = prims.data(); B.idx = out.primIdx.data();
    B.bins = std::min<uint32_t>(std::max<uint32_t>(p.bins ? p.bins : 8, 2), Builder::kMaxBins);
    B.maxLeaf = std::max<uint32_t>(p.maxLeafTris ? p.maxLeafTris : 4, 1);
    uint32_t threads = p.threads ? p.threads : usable_host_threads();
    if (n < 65536) threads = 1;

    out.nodes.reserve((size_t)n * 2);
    out.nodes.push_back(Node2{});
    if (threads == 1) { B.buildSubtree(out.nodes, 0, 0, n); return; }

    // Phase A: expand the top of the tree serially (breadth-first) until there are enough
    // open subtrees to keep every thread busy.  Phase B: open subtrees are built
    // concurrently into private vectors.  Phase C: stitched back in a fixed order, so the
    // result does not depend on thread timing (the reference's threaded builder is
    // numbering-nondeterministic, tiny_bvh.h:2427; this one is not).
    struct Open { uint32_t node, first, count; };
    std::vector<Open> open{{0, 0, n}}, done;
    const uint32_t target = threads * 8;
    const uint32_t grain = std::max<uint32_t>(n / (threads * 16), 4096);
    while (!open.empty() && open.size() + done.size() < target) {
        // expand the largest open range
        size_t big = 0;
        for (size_t i = 1; i < open.size(); i++) if (open[i].count > open[big].count) big = i;
        Open o = open[big];
        if (o.count <= grain) break;
        open.erase(open.begin() + big);
        const Box b = B.bounds(o.first, o.count);
        uint32_t mid;
        if (!B.split(b, o.first, o.count, mid)) { done.push_back(o); continue; }
        const uint32_t l = (uint32_t)out.nodes.size();
        out.nodes.push_back(Node2{}); out.nodes.push_back(Node2{});
        Builder::setNode(out.nodes[o.node], b, l, 0);
        open.push_back({l, o.first, mid - o.first});
        open.push_back({l + 1, mid, o.first + o.count - mid});
    }
    for (auto& o : done) open.push_back(o);
    std::vector<std::vector<Node2>> local(open.size());
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= open.size()) break;
            local[i].reserve((size_t)open[i].count * 2);
            local[i].push_back(Node2{});
            B.buildSubtree(local[i], 0, open[i].first, open[i].count);
        }
    };
    std::vector<std::thread> pool;
    for (uint32_t t = 1; t < threads; t++) pool.emplace_back(worker);
    worker();
    for (auto& t : pool) t.join();
    for (size_t i = 0; i < open.size(); i++) {
        const std::vector<Node2>& L = local[i];
        const uint32_t base = (uint32_t)out.nodes.size();  // local index k>=1 -> base + k - 1
        Node2 r = L[0];
        if (!r.leaf()) r.leftFirst += base - 1;
        out.nodes[open[i].node] = r;
        for (size_t k = 1; k < L.size(); k++) {
            Node2 c = L[k];
            if (!c.leaf()) c.leftFirst += base - 1;
            out.nodes.push_back(c);
        }
    }
}

// ---- wide collapse --------------------------------------------------------------------

template <int M> struct WideNode {
    Box box;
    uint32_t child[M];   // index into the wide node array; 0 = empty slot
    uint32_t childCount;
    uint32_t firstTri, triCount;  // leaf when triCount > 0
};

// Collapse a BVH2 into an M-wide tree: starting from a node's two children, repeatedly
// open the interior child with the largest surface area until M children are reached.
// Wide node 0 is the root and is always interior (a single-leaf BVH2 gets an extra level,
// like tiny_bvh.h:5036-5044, because CWBVH and BVH4_GPU need an interior root).
template <int M> void collapse(const BVH2& bvh, std::vector<WideNode<M>>& W) {
    W.clear();
    W.reserve(bvh.nodes.size());
    auto boxOf = [&](uint32_t n2) { Box b; for (int a = 0; a < 3; a++) b.mn[a] = bvh.nodes[n2].mn[a], b.mx[a] = bvh.nodes[n2].mx[a]; return b; };
    auto makeLeaf = [&](uint32_t n2) {
        WideNode<M> w{}; w.box = boxOf(n2); w.firstTri = bvh.nodes[n2].leftFirst; w.triCount = bvh.nodes[n2].triCount;
        W.push_back(w); return (uint32_t)W.size() - 1;
    };
    struct Item { uint32_t wide, n2; };
    std::vector<Item> stack;
    {
        WideNode<M> root{}; root.box = boxOf(0); W.push_back(root);
        if (bvh.nodes[0].leaf()) {
            const uint32_t l = makeLeaf(0);
            W[0].child[0] = l; W[0].childCount = 1;
            return;
        }
        stack.push_back({0, 0});
    }
    while (!stack.empty()) {
        const Item it = stack.back(); stack.pop_back();
        uint32_t kids[M]; uint32_t nk = 2;
        kids[0] = bvh.nodes[it.n2].leftFirst; kids[1] = kids[0] + 1;
        while (nk < (uint32_t)M) {
            int best = -1; float bestSA = -1.f;
            for (uint32_t i = 0; i < nk; i++) {
                const Node2& c = bvh.nodes[kids[i]];
                if (c.leaf()) continue;
                const float sa = boxOf(kids[i]).halfArea();
                if (sa > bestSA) bestSA = sa, best = (int)i;
            }
            if (best < 0) break;
            const uint32_t l = bvh.nodes[kids[best]].leftFirst;
            kids[best] = l; kids[nk++] = l + 1;
        }
        // allocate children contiguously
        uint32_t slots[M];
        for (uint32_t i = 0; i < nk; i++) {
            if (bvh.nodes[kids[i]].leaf()) slots[i] = makeLeaf(kids[i]);
            else {
                WideNode<M> w{}; w.box = boxOf(kids[i]); W.push_back(w);
                slots[i] = (uint32_t)W.size() - 1;
            }
        }
        for (uint32_t i = 0; i < nk; i++) W[it.wide].child[i] = slots[i];
        W[it.wide].childCount = nk;
        for (uint32_t i = nk; i-- > 0;)
            if (!bvh.nodes[kids[i]].leaf()) stack.push_back({slots[i], kids[i]});
    }
}

// Cost-optimal collapse (the SAH dynamic program of Ylitie, Karras & Laine 2017, §3.1):
// for every BVH2 node n and every i in 1..M-1, C(n,i) is the cheapest way to represent n's
// subtree as a forest of at most i wide-BVH roots:
//   C(n,1) = min( leaf:      A_n * P_n * cPrim           if the subtree has P_n <= maxLeaf tris,
//                 internal:  A_n * cNode + min_k C(l,k) + C(r,M-k) )
//   C(n,i) = min( min_k C(l,k) + C(r,i-k),  C(n,i-1) )   for i >= 2
// Compared with the greedy collapse this (a) merges subtrees of <= maxLeaf triangles into one
// leaf instead of spending a child slot per triangle, and (b) fills nodes, so a ray visits
// fewer nodes.  Relies on two properties of build_bvh2: children have larger indices than
// their parent, and a subtree's triangles are one contiguous primIdx range.
template <int M> void collapse_optimal(const BVH2& bvh, uint32_t maxLeaf, float cNode, float cPrim,
                                       std::vector<WideNode<M>>& W) {
    const uint32_t n2 = (uint32_t)bvh.nodes.size();
    constexpr int K = M - 1;                       // forest sizes 1..M-1
    std::vector<float> C((size_t)n2 * K);          // C[n*K + (i-1)]
    std::vector<uint8_t> dec((size_t)n2 * K);      // i==1: 0 = leaf, k = internal with split k (left gets k of M)
                                                   // i>=2 : 0 = same as i-1, k = distribute with left k
    std::vector<uint32_t> cnt(n2), first(n2);
    auto area = [&](uint32_t n) { Box b; for (int a = 0; a < 3; a++) b.mn[a] = bvh.nodes[n].mn[a], b.mx[a] = bvh.nodes[n].mx[a]; return b.halfArea(); };
    for (uint32_t n = n2; n-- > 0;) {
        const Node2& nd = bvh.nodes[n];
        float* c = &C[(size_t)n * K]; uint8_t* d = &dec[(size_t)n * K];
        if (nd.leaf()) {
            cnt[n] = nd.triCount; first[n] = nd.leftFirst;
            const float leafCost = area(n) * (float)nd.triCount * cPrim;
            for (int i = 0; i < K; i++) c[i] = leafCost, d[i] = 0;
            continue;
        }
        const uint32_t l = nd.leftFirst, r = l + 1;
        cnt[n] = cnt[l] + cnt[r]; first[n] = std::min(first[l], first[r]);
        const float* cl = &C[(size_t)l * K]; const float* cr = &C[(size_t)r * K];
        auto distribute = [&](int j, int& bestK) {   // min over k of C(l,k) + C(r,j-k), 1 <= k, j-k <= K
            float best = kFar; bestK = 1;
            for (int k = std::max(1, j - K); k <= std::min(K, j - 1); k++) {
                const float v = cl[k - 1] + cr[j - k - 1];
                if (v < best) best = v, bestK = k;
            }
            return best;
        };
        int kInt; const float internal = area(n) * cNode + distribute(M, kInt);
        const float leaf = cnt[n] <= maxLeaf ? area(n) * (float)cnt[n] * cPrim : kFar;
        if (leaf <= internal) c[0] = leaf, d[0] = 0; else c[0] = internal, d[0] = (uint8_t)kInt;
        for (int i = 2; i <= K; i++) {
            int k; const float v = distribute(i, k);
            if (v < c[i - 2]) c[i - 1] = v, d[i - 1] = (uint8_t)k; else c[i - 1] = c[i - 2], d[i - 1] = 0;
        }
    }
    // top-down reconstruction
    W.clear();
    W.reserve(n2 / 4 + 16);
    auto boxOf = [&](uint32_t n) { Box b; for (int a = 0; a < 3; a++) b.mn[a] = bvh.nodes[n].mn[a], b.mx[a] = bvh.nodes[n].mx[a]; return b; };
    auto makeLeaf = [&](uint32_t n) { WideNode<M> w{}; w.box = boxOf(n); w.firstTri = first[n]; w.triCount = cnt[n]; W.push_back(w); return (uint32_t)W.size() - 1; };
    // collect the roots of the forest that represents node n with budget i
    struct Root { uint32_t n2; };
    std::vector<Root> roots;
    struct Frame { uint32_t n; int i; };
    std::vector<Frame> st;                          // reused by every gather call (one heap allocation, not one per wide node)
    auto gather = [&](uint32_t n, int budget) {   // expands "n as a forest of <= budget roots" into `roots`
        st.clear(); st.push_back({n, budget});
        while (!st.empty()) {
            Frame f = st.back(); st.pop_back();
            int i = f.i;
            while (i >= 2 && dec[(size_t)f.n * K + i - 1] == 0) i--;   // "same as i-1"
            if (i == 1) { roots.push_back({f.n}); continue; }
            const int k = dec[(size_t)f.n * K + i - 1];
            const uint32_t l = bvh.nodes[f.n].leftFirst;
            st.push_back({l + 1, i - k});
            st.push_back({l, k});
        }
    };
    struct Item { uint32_t wide, n2; };
    std::vector<Item> stack;
    {
        WideNode<M> root{}; root.box = boxOf(0); W.push_back(root);
        if (dec[0] == 0) {  // whole scene is one leaf: interior root above it (tiny_bvh.h:5036-5044)
            const uint32_t lf = makeLeaf(0);
            W[0].child[0] = lf; W[0].childCount = 1;
            return;
        }
        stack.push_back({0, 0});
    }
    while (!stack.empty()) {
        const Item it = stack.back(); stack.pop_back();
        const int k = dec[(size_t)it.n2 * K];      // internal: left gets k, right M-k
        const uint32_t l = bvh.nodes[it.n2].leftFirst;
        roots.clear();
        gather(l, k); gather(l + 1, M - k);
        uint32_t slots[M]; const uint32_t nk = (uint32_t)roots.size();
        for (uint32_t i = 0; i < nk; i++) {
            const uint32_t c = roots[i].n2;
            if (dec[(size_t)c * K] == 0) slots[i] = makeLeaf(c);
            else { WideNode<M> w{}; w.box = boxOf(c); W.push_back(w); slots[i] = (uint32_t)W.size() - 1; }
        }
        for (uint32_t i = 0; i < nk; i++) W[it.wide].child[i] = slots[i];
        W[it.wide].childCount = nk;
        for (uint32_t i = nk; i-- > 0;)
            if (dec[(size_t)roots[i].n2 * K] != 0) stack.push_back({slots[i], roots[i].n2});
    }
}

inline uint32_t asU32(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float asF32(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

}  // namespace

// ---- public: builders -------------------------------------------------------------------

uint32_t usable_host_threads() {
    uint32_t n = std::max(1u, std::thread::hardware_concurrency());
#if defined(__linux__)
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) {
        const int c = CPU_COUNT(&set);
        if (c > 0) n = std::min<uint32_t>(n, (uint32_t)c);
    }
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2: "<quota> <period>" or "max <period>"
        char q[32] = {0};
        long long period = 0;
        if (std::fscanf(f, "%31s %lld", q, &period) == 2 && period > 0 && std::strcmp(q, "max") != 0) {
            const long long quota = std::atoll(q);
            if (quota > 0) n = std::min<uint32_t>(n, (uint32_t)std::max<long long>(1, (quota + period / 2) / period));
        }
        std::fclose(f);
    }
#endif
    return std::max(1u, n);
}

// ---- triangle splitting ahead of the build (BuildParams::splitBudget) --------------------------------------------------
// What BVH::BuildHQ's spatial splits are for (tiny_bvh.h:2623-3040, 8614-8795: Stich et al. 2009, a reference-duplicating
// split evaluated per node and only where the children overlap by more than 1e-4 of the root's area): triangles whose boxes
// are mostly empty — large ones off the coordinate axes — make every box above them loose.  Here the duplication happens BEFORE
// the build and the builder stays the in-place binned-SAH sweep over boxes: every triangle gets a share of a global budget of
// extra references by how much box it wastes and how high up in the tree the waste sits (after Karras & Aila 2013, §4.3:
// priority = cbrt(2^-level * (A_box - A_ideal)), level = the most important spatial-median plane of the scene box that cuts
// the triangle's box), and is cut along such median planes into that many pieces, each reference carrying the box of its
// clipped piece.  primIdx then names a triangle once per leaf it ended up in (the format allows that; BuildHQ emits the same).
namespace {

struct SplitGrid {   // the scene box and its binary subdivision: plane j * 2^-level along each axis
    float mn[3], ext[3];
    static constexpr int kBits = 22;
    uint32_t cell(int a, float x) const {
        const float r = ext[a] > 0 ? (x - mn[a]) / ext[a] : 0.f;
        const float c = std::min(std::max(r, 0.f), 1.f) * (float)(1u << kBits);
        return std::min((uint32_t)c, (1u << kBits) - 1u);
    }
    // the most important plane cutting [lo, hi] on axis a: level (1 = the scene's middle, kBits + 1 = none) and position
    int plane(int a, float lo, float hi, float& pos) const {
        const uint32_t cl = cell(a, lo), ch = cell(a, hi);
        if (cl == ch) return kBits + 1;
        const int msb = 31 - __builtin_clz(cl ^ ch);
        const uint32_t k = (ch >> msb) << msb;            // first cell right of the plane
        pos = mn[a] + ext[a] * ((float)k / (float)(1u << kBits));
        if (!(pos > lo && pos < hi)) return kBits + 1;    // rounding put the plane on or outside the box: nothing to cut
        return kBits - msb;
    }
};

struct Poly { float v[10][3]; int n; };

// Sutherland-Hodgman against the plane x[a] = pos: `in` -> the part on the low side and the part on the high side
void clip_poly(const Poly& in, int a, float pos, Poly& lo, Poly& hi) {
    lo.n = hi.n = 0;
    for (int i = 0; i < in.n; i++) {
        const float* p = in.v[i]; const float* q = in.v[(i + 1) % in.n];
        const bool pl = p[a] <= pos, ph = p[a] >= pos;
        if (pl && lo.n < 10) std::memcpy(lo.v[lo.n++], p, 12);
        if (ph && hi.n < 10) std::memcpy(hi.v[hi.n++], p, 12);
        if ((p[a] < pos && q[a] > pos) || (p[a] > pos && q[a] < pos)) {
            const float t = (pos - p[a]) / (q[a] - p[a]);
            float x[3];
            for (int k = 0; k < 3; k++) x[k] = p[k] + t * (q[k] - p[k]);
            x[a] = pos;
            if (lo.n < 10) std::memcpy(lo.v[lo.n++], x, 12);
            if (hi.n < 10) std::memcpy(hi.v[hi.n++], x, 12);
        }
    }
}

// box of a clipped piece: its vertices, widened by a few ulps of the coordinates involved (the cut points are rounded results; a box must never
// be smaller than the exact piece — and along the cut axis the two halves overlap by that much: split_piece), and never beyond the box of the
// piece it was cut from.  The polygon a piece is
// clipped from was itself clipped `depth` times before — every level interpolates between rounded points, so the error of the off-axis
// coordinates grows with the depth — and the pad grows with it (round 6; a fixed pad was only safe under BVH8_CWBVH's outward quantisation,
// and the float-box layouts BVH_GPU / BVH4_GPU take split references too).
Box piece_box(const Poly& p, const Box& parent, const float* pad, uint32_t depth) {
    Box b; b.reset();
    for (int i = 0; i < p.n; i++) b.grow(p.v[i]);
    const float k = (float)(depth + 1u);
    for (int a = 0; a < 3; a++) {
        b.mn[a] = std::max(b.mn[a] - k * pad[a], parent.mn[a]);
        b.mx[a] = std::min(b.mx[a] + k * pad[a], parent.mx[a]);
    }
    return b;
}

// cutFaces: which faces of `box` are cut planes (bit a: its low face on axis a, bit 3 + a: its high face).
void split_piece(const SplitGrid& g, const Poly& poly, const Box& box, uint32_t splits, uint32_t tri, const float* pad,
                 std::vector<Prim>& prims, std::vector<uint32_t>& refTri, uint32_t depth = 0, uint32_t cutFaces = 0) {
    if (splits > 0 && poly.n >= 3) {
        int bestA = -1, bestLevel = SplitGrid::kBits + 1; float bestPos = 0, bestExt = -1;
        for (int a = 0; a < 3; a++) {
            float pos; const int lv = g.plane(a, box.mn[a], box.mx[a], pos);
            const float e = box.mx[a] - box.mn[a];
            if (lv < bestLevel || (lv == bestLevel && lv <= SplitGrid::kBits && e > bestExt)) bestLevel = lv, bestA = a, bestPos = pos, bestExt = e;
        }
        if (bestA >= 0 && bestLevel <= SplitGrid::kBits) {
            Poly lo, hi;
            clip_poly(poly, bestA, bestPos, lo, hi);
            if (lo.n >= 3 && hi.n >= 3) {
                Box pl = box, ph = box; pl.mx[bestA] = bestPos; ph.mn[bestA] = bestPos;
                const Box bl = piece_box(lo, pl, pad, depth + 1u), bh = piece_box(hi, ph, pad, depth + 1u);
                auto longest = [](const Box& b) { return std::max(std::max(b.mx[0] - b.mn[0], b.mx[1] - b.mn[1]), b.mx[2] - b.mn[2]); };
                const float wl = longest(bl), wh = longest(bh);
                const uint32_t rest = splits - 1;
                uint32_t sl = wl + wh > 0 ? (uint32_t)((float)rest * wl / (wl + wh) + 0.5f) : rest / 2;
                if (sl > rest) sl = rest;
                split_piece(g, lo, bl, sl, tri, pad, prims, refTri, depth + 1u, cutFaces | (8u << bestA));
                split_piece(g, hi, bh, rest - sl, tri, pad, prims, refTri, depth + 1u, cutFaces | (1u << bestA));
                return;
            }
        }
    }
    // The pieces of a triangle tile it exactly: neighbours share the cut plane.  A hit within a rounding error of that plane can then fall between
    // them — the piece that holds it in exact arithmetic rejects the ray by one ulp of its slab test, its neighbour rightly does not contain it: a
    // hole in the INTERIOR of a triangle (found by tests/test_random_large.py, seed 222: one camera ray of 1.6 M went through a wall; the two box
    // tests: tools/debug/cw_trace.py).  So the reference a piece becomes reaches a pad's width across each of its cut faces.
    Prim pr; pr.box = box;
    for (int a = 0; a < 3; a++) {
        pr.c[a] = 0.5f * (box.mn[a] + box.mx[a]);
        const float over = (float)(depth + 1u) * pad[a];
        if (cutFaces & (1u << a)) pr.box.mn[a] -= over;
        if (cutFaces & (8u << a)) pr.box.mx[a] += over;
    }
    prims.push_back(pr); refTri.push_back(tri);
}

// prims / refTri: one entry per reference (>= one per triangle), in triangle order
void presplit(const Vec4* verts, uint32_t triCount, float budgetFrac, std::vector<Prim>& prims, std::vector<uint32_t>& refTri) {
    Box scene; scene.reset();
    for (size_t i = 0; i < (size_t)triCount * 3; i++) scene.grow(&verts[i].x);
    SplitGrid g;
    float pad[3];
    for (int a = 0; a < 3; a++) {
        g.mn[a] = scene.mn[a]; g.ext[a] = scene.mx[a] - scene.mn[a];
        pad[a] = 4e-7f * std::max(std::max(std::fabs(scene.mn[a]), std::fabs(scene.mx[a])), g.ext[a]);
    }
    std::vector<float> prio(triCount);
    std::vector<Box> boxes(triCount);
    for (uint32_t i = 0; i < triCount; i++) {
        const Vec4* v = verts + 3 * (size_t)i;
        Box b; b.reset();
        for (int k = 0; k < 3; k++) b.grow(&v[k].x);
        boxes[i] = b;
        int level = SplitGrid::kBits + 1;
        for (int a = 0; a < 3; a++) { float pos; level = std::min(level, g.plane(a, b.mn[a], b.mx[a], pos)); }
        float p = 0.f;
        if (level <= SplitGrid::kBits) {
            const float e1[3] = {v[1].x - v[0].x, v[1].y - v[0].y, v[1].z - v[0].z}, e2[3] = {v[2].x - v[0].x, v[2].y - v[0].y, v[2].z - v[0].z};
            const float cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
            const float ideal = std::fabs(cx) + std::fabs(cy) + std::fabs(cz);    // twice the triangle's three axis projections: the box area a flat triangle needs
            const float waste = 2.f * b.halfArea() - ideal;
            // (exponents 1/2, 2/3 and 1, with and without the level term, measured on the rotated street: node visits + triangle tests per ray within
            // 1 % of each other, profiles/r05_rotated.txt — the published rule stays)
            if (waste > 0 && std::isfinite(waste)) p = std::cbrt(std::ldexp(waste, -level));
        }
        prio[i] = p;
    }
    // the scale D with sum floor(D * prio) <= budget, by bisection
    const double budget = std::max(0.0, (double)budgetFrac) * (double)triCount;
    double sum = 0; for (uint32_t i = 0; i < triCount; i++) sum += prio[i];
    auto total = [&](double D) { double t = 0; for (uint32_t i = 0; i < triCount; i++) t += std::floor(D * (double)prio[i]); return t; };
    double lo = 0, hi = sum > 0 ? 2.0 * budget / sum + 1.0 : 0;
    while (hi > 0 && total(hi) <= budget && hi < 1e30) hi *= 2;
    for (int it = 0; it < 40 && hi > 0; it++) { const double mid = 0.5 * (lo + hi); if (total(mid) <= budget) lo = mid; else hi = mid; }
    prims.clear(); refTri.clear();
    prims.reserve((size_t)triCount + (size_t)budget + 16); refTri.reserve(prims.capacity());
    for (uint32_t i = 0; i < triCount; i++) {
        const uint32_t s = (uint32_t)std::min(std::floor(lo * (double)prio[i]), 4096.0);
        if (s == 0) {
            Prim pr; pr.box = boxes[i];
            for (int a = 0; a < 3; a++) pr.c[a] = 0.5f * (pr.box.mn[a] + pr.box.mx[a]);
            prims.push_back(pr); refTri.push_back(i);
            continue;
        }
        Poly poly; poly.n = 3;
        for (int k = 0; k < 3; k++) { poly.v[k][0] = verts[3 * (size_t)i + k].x; poly.v[k][1] = verts[3 * (size_t)i + k].y; poly.v[k][2] = verts[3 * (size_t)i + k].z; }
        split_piece(g, poly, boxes[i], s, i, pad, prims, refTri);
    }
}

// after a build over references: primIdx names triangles again, a leaf names each of its triangles once, no holes in primIdx
void refs_to_triangles(BVH2& bvh, const std::vector<uint32_t>& refTri) {
    std::vector<uint32_t> leaves;
    for (uint32_t n = 0; n < (uint32_t)bvh.nodes.size(); n++) if (bvh.nodes[n].leaf()) leaves.push_back(n);
    std::sort(leaves.begin(), leaves.end(), [&](uint32_t a, uint32_t b) { return bvh.nodes[a].leftFirst < bvh.nodes[b].leftFirst; });
    std::vector<uint32_t> out; out.reserve(bvh.primIdx.size());
    for (const uint32_t n : leaves) {
        Node2& nd = bvh.nodes[n];
        const uint32_t first = (uint32_t)out.size();
        for (uint32_t j = 0; j < nd.triCount; j++) {
            const uint32_t t = refTri[bvh.primIdx[nd.leftFirst + j]];
            bool dup = false;
            for (uint32_t k = first; k < (uint32_t)out.size(); k++) if (out[k] == t) { dup = true; break; }
            if (!dup) out.push_back(t);
        }
        nd.leftFirst = first; nd.triCount = (uint32_t)out.size() - first;
    }
    bvh.primIdx.swap(out);
}

}  // namespace

void build_bvh2(const Vec4* verts, uint32_t triCount, const BuildParams& p, BVH2& out) {
    if (p.splitBudget > 0 && triCount > 0) {
        std::vector<Prim> prims; std::vector<uint32_t> refTri;
        presplit(verts, triCount, p.splitBudget, prims, refTri);
        buildFromPrims(prims, p, out);
        refs_to_triangles(out, refTri);
        out.triCount = triCount;
        return;
    }
    std::vector<Prim> prims(triCount);
    for (uint32_t i = 0; i < triCount; i++) {
        Prim& pr = prims[i];
        pr.box.reset();
        for (int k = 0; k < 3; k++) pr.box.grow(&verts[3 * (size_t)i + k].x);
        for (int a = 0; a < 3; a++) pr.c[a] = 0.5f * (pr.box.mn[a] + pr.box.mx[a]);
    }
    buildFromPrims(prims, p, out);
}

void build_bvh2_boxes(const float* boxes6, uint32_t count, const BuildParams& p, BVH2& out) {
    std::vector<Prim> prims(count);
    for (uint32_t i = 0; i < count; i++) {
        Prim& pr = prims[i];
        for (int a = 0; a < 3; a++) {
            pr.box.mn[a] = boxes6[6 * (size_t)i + a];
            pr.box.mx[a] = boxes6[6 * (size_t)i + 3 + a];
            pr.c[a] = 0.5f * (pr.box.mn[a] + pr.box.mx[a]);
        }
    }
    buildFromPrims(prims, p, out);
}

// ---- public: encoders -------------------------------------------------------------------

// Aila-Laine layout (format: tiny_bvh.h:1095-1105): depth-first pre-order, the left child
// of node k is node k+1, interior nodes carry both children's boxes, leaves are all-zero
// except triCount / firstTri.
void encode_bvh_gpu(const BVH2& bvh, std::vector<NodeAL>& out) {
    out.assign(bvh.nodes.size(), NodeAL{});
    constexpr uint32_t kNone = 0xffffffffu;
    struct Item { uint32_t src, parent; };  // a deferred right child and the node to patch
    std::vector<Item> stack{{0, kNone}};
    uint32_t next = 0;
    while (!stack.empty()) {
        const Item it = stack.back(); stack.pop_back();
        uint32_t src = it.src, dst = next++;
        if (it.parent != kNone) out[it.parent].right = dst;
        for (;;) {  // walk down the left spine; numbers are consecutive along it
            const Node2& s = bvh.nodes[src];
            NodeAL& d = out[dst];
            if (s.leaf()) { d.triCount = s.triCount; d.firstTri = s.leftFirst; break; }
            const Node2& l = bvh.nodes[s.leftFirst];
            const Node2& r = bvh.nodes[s.leftFirst + 1];
            for (int a = 0; a < 3; a++) d.lmin[a] = l.mn[a], d.lmax[a] = l.mx[a], d.rmin[a] = r.mn[a], d.rmax[a] = r.mx[a];
            d.left = next;
            stack.push_back({s.leftFirst + 1, dst});  // numbered after the whole left subtree
            src = s.leftFirst; dst = next++;
        }
    }
    out.resize(next);
}

// Runs body(first, last) over [0, n) on `threads` threads (contiguous ranges; the calling thread takes one).
template <class F> static void parallel_ranges(size_t n, uint32_t threads, F body) {
    threads = (uint32_t)std::min<size_t>(std::max<uint32_t>(threads, 1), std::max<size_t>(n / 4096, 1));
    if (threads <= 1) { body((size_t)0, n); return; }
    std::vector<std::thread> pool;
    for (uint32_t t = 1; t < threads; t++) pool.emplace_back([=] { body(n * t / threads, n * (t + 1) / threads); });
    body((size_t)0, n / threads);
    for (auto& t : pool) t.join();
}

// BVH4_GPU stream (format: tiny_bvh.h:1248-1266, 5120-5127).
void encode_bvh4_gpu(const BVH2& bvh, const Vec4* verts, const BuildParams& p, std::vector<Vec4>& blocks) {
    std::vector<WideNode<4>> W;
    if (p.greedyCollapse) collapse<4>(bvh, W);
    else collapse_optimal<4>(bvh, std::max<uint32_t>(p.maxLeafTris, 1), 1.0f, p.cPrim, W);
    const uint32_t threads = p.threads ? p.threads : usable_host_threads();
    // Two passes: a serial depth-first walk hands out the block offset of every interior node (node = 4 blocks, followed
    // by the triangles of its leaf children in child order), then the nodes are quantised and written on all threads —
    // byte for byte what one walk that does both would write.
    constexpr uint32_t kNone = 0xffffffffu;
    std::vector<uint32_t> baseOf(W.size(), kNone);   // block offset of interior wide node w
    std::vector<uint32_t> order;                      // interior wide nodes in emission order
    order.reserve(W.size());
    uint64_t total = 0;
    {
        std::vector<uint32_t> stack{0};
        while (!stack.empty()) {
            const uint32_t wi = stack.back(); stack.pop_back();
            const WideNode<4>& n = W[wi];
            baseOf[wi] = (uint32_t)total;
            order.push_back(wi);
            total += 4;
            for (uint32_t i = 0; i < n.childCount; i++) if (W[n.child[i]].triCount) total += 3ull * W[n.child[i]].triCount;
            for (uint32_t i = n.childCount; i-- > 0;) if (!W[n.child[i]].triCount) stack.push_back(n.child[i]);
        }
    }
    assert(total <= 0xffffffffull);
    blocks.assign((size_t)total, Vec4{0, 0, 0, 0});
    parallel_ranges(order.size(), threads, [&](size_t olo, size_t ohi) {
        for (size_t oi = olo; oi < ohi; oi++) {
            const WideNode<4>& n = W[order[oi]];
            const uint32_t base = baseOf[order[oi]];
            uint32_t out = base + 4;   // next free block: the inline triangles follow the node
            uint32_t info[4] = {0, 0, 0, 0};
            uint8_t q[6][4] = {};  // xmin, xmax, ymin, ymax, zmin, zmax per child
            const float ext[3] = {n.box.mx[0] - n.box.mn[0], n.box.mx[1] - n.box.mn[1], n.box.mx[2] - n.box.mn[2]};
            float scale[3], e255[3];
            for (int a = 0; a < 3; a++) {
                scale[a] = ext[a] > 1e-10f ? 254.999f / ext[a] : 0.f;
                e255[a] = ext[a] * (1.0f / 255.0f);
                const float guard = 4e-7f * std::max(std::max(std::fabs(n.box.mn[a]), std::fabs(n.box.mx[a])), ext[a]);
                if (ext[a] > 0) {   // the decode step must carry 255 steps past the far face: jump there, then settle ulp by ulp
                    const float need = ((n.box.mx[a] + guard) - n.box.mn[a]) * (1.0f / 255.0f);
                    if (need > e255[a]) e255[a] = need;
                    while (n.box.mn[a] + e255[a] * 255.0f < n.box.mx[a] + guard) e255[a] = std::nextafter(e255[a], kFar);
                }
            }
            for (uint32_t i = 0; i < n.childCount; i++) {
                const WideNode<4>& c = W[n.child[i]];
                for (int a = 0; a < 3; a++) {
                    // The reference quantises with floor/ceil(rel * 254.999 / extent)
                    // (tiny_bvh.h:5196-5231) and decodes with bmin + (extent/255) * q; the 254.999
                    // makes the decoded maximum fall short of the true one by up to 4e-6 * rel, so a
                    // reference-encoded BVH4_GPU can cull a box a ray grazes.  Same format here, but
                    // the quantised box is verified against the decode and widened until it really
                    // contains the child (with a few-ulp guard for decoder rounding).
                    const float guard = 4e-7f * std::max(std::max(std::fabs(n.box.mn[a]), std::fabs(n.box.mx[a])), ext[a]);
                    int lo = (int)std::floor((c.box.mn[a] - n.box.mn[a]) * scale[a]);
                    int hi = (int)std::ceil((c.box.mx[a] - n.box.mn[a]) * scale[a]);
                    lo = std::min(std::max(lo, 0), 255); hi = std::min(std::max(hi, 0), 255);
                    while (lo > 0 && n.box.mn[a] + e255[a] * (float)lo > c.box.mn[a] - guard) lo--;
                    while (hi < 255 && n.box.mn[a] + e255[a] * (float)hi < c.box.mx[a] + guard) hi++;
                    q[2 * a][i] = (uint8_t)lo;
                    q[2 * a + 1][i] = (uint8_t)hi;
                }
                if (c.triCount) {
                    const uint32_t rel = out - base;
                    assert(rel < 65536 && c.triCount < 32768);
                    info[i] = 0x80000000u | (c.triCount << 16) | rel;
                    for (uint32_t j = 0; j < c.triCount; j++) {
                        const uint32_t prim = bvh.primIdx[c.firstTri + j];
                        const Vec4 v0 = verts[3 * (size_t)prim], v1 = verts[3 * (size_t)prim + 1], v2 = verts[3 * (size_t)prim + 2];
                        blocks[out++] = Vec4{v0.x, v0.y, v0.z, asF32(prim)};
                        blocks[out++] = Vec4{v1.x - v0.x, v1.y - v0.y, v1.z - v0.z, v1.w - v0.w};
                        blocks[out++] = Vec4{v2.x - v0.x, v2.y - v0.y, v2.z - v0.z, v2.w - v0.w};
                    }
                } else
                    info[i] = baseOf[n.child[i]];   // interior child: its absolute block offset
            }
            Vec4* nb = blocks.data() + base;
            uint32_t w0, w1, w2[4];
            std::memcpy(&w0, q[0], 4); std::memcpy(&w1, q[1], 4);
            std::memcpy(&w2[0], q[2], 4); std::memcpy(&w2[1], q[3], 4); std::memcpy(&w2[2], q[4], 4); std::memcpy(&w2[3], q[5], 4);
            nb[0] = Vec4{n.box.mn[0], n.box.mn[1], n.box.mn[2], asF32(w0)};
            nb[1] = Vec4{e255[0], e255[1], e255[2], asF32(w1)};
            nb[2] = Vec4{asF32(w2[0]), asF32(w2[1]), asF32(w2[2]), asF32(w2[3])};
            nb[3] = Vec4{asF32(info[0]), asF32(info[1]), asF32(info[2]), asF32(info[3])};
        }
    });
}

// CWBVH (format: Ylitie et al. 2017 as laid out by tiny_bvh.h:5884-6018).
void encode_cwbvh(const BVH2& bvh, const Vec4* verts, const BuildParams& p, std::vector<Vec4>& nodeBlocks,
                  std::vector<Vec4>& triBlocks) {
    std::vector<WideNode<8>> W;
    if (p.greedyCollapse) collapse<8>(bvh, W);
    else collapse_optimal<8>(bvh, std::min<uint32_t>(std::max<uint32_t>(p.maxLeafTris, 1), 3), 1.0f, p.cPrim, W);
    const uint32_t threads = p.threads ? p.threads : usable_host_threads();
    // Three passes, so that the expensive parts run on all threads while the output stays byte for byte what a single
    // depth-first walk writes: (1) octant slot assignment per node, parallel; (2) the depth-first walk itself, which only
    // hands out node and triangle addresses, serial; (3) quantisation and triangle records per node, parallel.
    // --- pass 1: children go to the octant slot their centroid offset points at (slot bit 2 = -x, bit 1 = -y, bit 0 =
    // -z side), solved greedily on the cost matrix cost[s][i] = dot(centroid_i - centroid_node, dir_s).
    struct Slots { int8_t childIn[8]; };
    std::vector<Slots> slots(W.size());
    parallel_ranges(W.size(), threads, [&](size_t lo, size_t hi) {
        for (size_t wi = lo; wi < hi; wi++) {
            const WideNode<8>& n = W[wi];
            Slots& so = slots[wi];
            for (int s = 0; s < 8; s++) so.childIn[s] = -1;
            if (n.triCount) continue;   // leaves have no children
            float cost[8][8];
            int slotOf[8];
            for (int i = 0; i < 8; i++) slotOf[i] = -1;
            float nc[3];
            for (int a = 0; a < 3; a++) nc[a] = 0.5f * (n.box.mn[a] + n.box.mx[a]);
            for (uint32_t i = 0; i < n.childCount; i++) {
                const Box& cb = W[n.child[i]].box;
                float d[3];
                for (int a = 0; a < 3; a++) d[a] = 0.5f * (cb.mn[a] + cb.mx[a]) - nc[a];
                for (int s = 0; s < 8; s++)
                    cost[s][i] = ((s & 4) ? -d[0] : d[0]) + ((s & 2) ? -d[1] : d[1]) + ((s & 1) ? -d[2] : d[2]);
            }
            for (uint32_t k = 0; k < n.childCount; k++) {
                float best = kFar; int bs = -1, bi = -1;
                for (int s = 0; s < 8; s++) if (so.childIn[s] < 0)
                    for (uint32_t i = 0; i < n.childCount; i++) if (slotOf[i] < 0 && cost[s][i] < best)
                        best = cost[s][i], bs = s, bi = (int)i;
                slotOf[bi] = bs; so.childIn[bs] = (int8_t)bi;
            }
        }
    });
    // --- pass 2: depth-first walk: output node index per interior wide node, first triangle block per node
    struct Placed { uint32_t wide, addr, childBase, triBase; };
    std::vector<Placed> placed;
    placed.reserve(W.size());
    {
        struct Item { uint32_t wide; uint32_t addr; };
        std::vector<Item> stack{{0, 0}};
        uint32_t nNodes = 1;
        uint64_t nTriBlocks = 0;
        while (!stack.empty()) {
            const Item it = stack.back(); stack.pop_back();
            const WideNode<8>& n = W[it.wide];
            Placed pl{it.wide, it.addr, 0, 0};
            uint32_t nInner = 0, nTris = 0;
            for (int s = 0; s < 8; s++) {
                const int k = slots[it.wide].childIn[s];
                if (k < 0) continue;
                const uint32_t ci = n.child[k];
                const WideNode<8>& c = W[ci];
                if (!c.triCount) {
                    const uint32_t addr = nNodes++;
                    if (nInner++ == 0) pl.childBase = addr;
                    stack.push_back({ci, addr});
                } else {
                    if (nTris == 0) pl.triBase = (uint32_t)nTriBlocks;
                    nTris += c.triCount;
                    nTriBlocks += 3ull * c.triCount;
                }
            }
            assert(nTris <= 24);
            placed.push_back(pl);
        }
        nodeBlocks.assign((size_t)nNodes * 5, Vec4{0, 0, 0, 0});
        triBlocks.assign((size_t)nTriBlocks, Vec4{0, 0, 0, 0});
    }
    // --- pass 3: one output node (and its triangle records) per placed entry
    parallel_ranges(placed.size(), threads, [&](size_t plo, size_t phi) {
        for (size_t pi = plo; pi < phi; pi++) {
            const Placed& pl = placed[pi];
            const WideNode<8>& n = W[pl.wide];
            const int8_t* childIn = slots[pl.wide].childIn;
            // per-axis exponent: smallest e with extent / 2^e <= 255
            int e[3];
            float inv[3];
            for (int a = 0; a < 3; a++) {
                const float ext = n.box.mx[a] - n.box.mn[a];
                int ea = ext > 0 ? (int)std::ceil(std::log2(ext / 255.0f)) : -126;
                ea = std::max(ea, -126);
                // guard the ceil() of every child against 255 overflow from rounding
                for (;;) {
                    const float sc = std::ldexp(1.0f, -ea);
                    bool ok = true;
                    for (uint32_t i = 0; i < n.childCount; i++)
                        if (std::ceil((W[n.child[i]].box.mx[a] - n.box.mn[a]) * sc) > 255.f) ok = false;
                    if (n.box.mn[a] + std::ldexp(255.0f, ea) < n.box.mx[a]) ok = false;
                    if (ok) break;
                    ea++;
                }
                e[a] = ea; inv[a] = std::ldexp(1.0f, -ea);
            }
            uint8_t meta[8] = {}, q[6][8] = {};
            uint32_t imask = 0, nTris = 0;
            size_t triOut = pl.triBase;
            for (int s = 0; s < 8; s++) {
                if (childIn[s] < 0) continue;
                const uint32_t ci = n.child[childIn[s]];
                const WideNode<8>& c = W[ci];
                for (int a = 0; a < 3; a++) {
                    // floor / ceil in units of 2^e (tiny_bvh.h:5952-5957), then checked against the
                    // decode lo + q * 2^e so float rounding of the subtraction cannot shrink the box
                    const float sc = std::ldexp(1.0f, e[a]);
                    int lo = (int)std::floor((c.box.mn[a] - n.box.mn[a]) * inv[a]);
                    int hi = (int)std::ceil((c.box.mx[a] - n.box.mn[a]) * inv[a]);
                    lo = std::min(std::max(lo, 0), 255); hi = std::min(std::max(hi, 0), 255);
                    while (lo > 0 && n.box.mn[a] + sc * (float)lo > c.box.mn[a]) lo--;
                    while (hi < 255 && n.box.mn[a] + sc * (float)hi < c.box.mx[a]) hi++;
                    q[a][s] = (uint8_t)lo;
                    q[3 + a][s] = (uint8_t)hi;
                }
                if (!c.triCount) {
                    imask |= 1u << s;
                    meta[s] = (uint8_t)((1u << 5) | (24 + s));
                } else {
                    assert(c.triCount <= 3);
                    const uint32_t unary = c.triCount == 1 ? 1u : c.triCount == 2 ? 3u : 7u;
                    meta[s] = (uint8_t)((unary << 5) | nTris);
                    nTris += c.triCount;
                    for (uint32_t j = 0; j < c.triCount; j++) {
                        const uint32_t prim = bvh.primIdx[c.firstTri + j];
                        const Vec4 v0 = verts[3 * (size_t)prim], v1 = verts[3 * (size_t)prim + 1], v2 = verts[3 * (size_t)prim + 2];
                        triBlocks[triOut++] = Vec4{v2.x - v0.x, v2.y - v0.y, v2.z - v0.z, v2.w - v0.w};
                        triBlocks[triOut++] = Vec4{v1.x - v0.x, v1.y - v0.y, v1.z - v0.z, v1.w - v0.w};
                        triBlocks[triOut++] = Vec4{v0.x, v0.y, v0.z, asF32(prim)};
                    }
                }
            }
            Vec4* nb = nodeBlocks.data() + (size_t)pl.addr * 5;
            const uint32_t eim = ((uint32_t)(uint8_t)(int8_t)e[0]) | ((uint32_t)(uint8_t)(int8_t)e[1] << 8) |
                                 ((uint32_t)(uint8_t)(int8_t)e[2] << 16) | (imask << 24);
            uint32_t m0, m1; std::memcpy(&m0, meta, 4); std::memcpy(&m1, meta + 4, 4);
            nb[0] = Vec4{n.box.mn[0], n.box.mn[1], n.box.mn[2], asF32(eim)};
            nb[1] = Vec4{asF32(pl.childBase), asF32(pl.triBase), asF32(m0), asF32(m1)};
            uint32_t w[12];
            std::memcpy(w, q, 48);  // qlo_x[8] qlo_y[8] qlo_z[8] qhi_x[8] qhi_y[8] qhi_z[8]
            nb[2] = Vec4{asF32(w[0]), asF32(w[1]), asF32(w[2]), asF32(w[3])};
            nb[3] = Vec4{asF32(w[4]), asF32(w[5]), asF32(w[6]), asF32(w[7])};
            nb[4] = Vec4{asF32(w[8]), asF32(w[9]), asF32(w[10]), asF32(w[11])};
        }
    });
}

// ---- instances ---------------------------------------------------------------------------

// BLASInstance::InvertTransform (tiny_bvh.h:8402-8427): the MESA cofactor formula in float.  To hand a tinybvh
// user the very records tinybvh itself would compute, the evaluation follows the reference build (g++ 11.4 -O3
// -mavx2 -mfma, -ffp-contract=fast) operation for operation.  Every six-term cofactor sum is ONE rounded triple
// product plus five terms fused into the running sum (fma((x*y), z, sum), the inner product rounded); which term is
// the rounded one differs between the two halves of the matrix because the compiler SLP-vectorised rows 0-7 (term 0
// rounded, terms 1..5 fused in order) and left rows 8-15 scalar (term 1 rounded, then term 0, 2, 3, 4, 5 fused).
// Found by exhaustive search over the evaluation orders against the real reference (64 000 of 64 000 elements of
// 4 000 random matrices bit-identical) and pinned in tests/test_oracle_vs_reference.py.  The same table and code run
// on the device (kernels_tlasbuild.hip).
namespace {
struct Term { int s, a, b, c; };   // s * T[a] * T[b] * T[c]
static const Term kCof[16][6] = {
    {{+1, 5, 10, 15}, {-1, 5, 11, 14}, {-1, 9, 6, 15}, {+1, 9, 7, 14}, {+1, 13, 6, 11}, {-1, 13, 7, 10}},
    {{-1, 1, 10, 15}, {+1, 1, 11, 14}, {+1, 9, 2, 15}, {-1, 9, 3, 14}, {-1, 13, 2, 11}, {+1, 13, 3, 10}},
    {{+1, 1, 6, 15}, {-1, 1, 7, 14}, {-1, 5, 2, 15}, {+1, 5, 3, 14}, {+1, 13, 2, 7}, {-1, 13, 3, 6}},
    {{-1, 1, 6, 11}, {+1, 1, 7, 10}, {+1, 5, 2, 11}, {-1, 5, 3, 10}, {-1, 9, 2, 7}, {+1, 9, 3, 6}},
    {{-1, 4, 10, 15}, {+1, 4, 11, 14}, {+1, 8, 6, 15}, {-1, 8, 7, 14}, {-1, 12, 6, 11}, {+1, 12, 7, 10}},
    {{+1, 0, 10, 15}, {-1, 0, 11, 14}, {-1, 8, 2, 15}, {+1, 8, 3, 14}, {+1, 12, 2, 11}, {-1, 12, 3, 10}},
    {{-1, 0, 6, 15}, {+1, 0, 7, 14}, {+1, 4, 2, 15}, {-1, 4, 3, 14}, {-1, 12, 2, 7}, {+1, 12, 3, 6}},
    {{+1, 0, 6, 11}, {-1, 0, 7, 10}, {-1, 4, 2, 11}, {+1, 4, 3, 10}, {+1, 8, 2, 7}, {-1, 8, 3, 6}},
    {{+1, 4, 9, 15}, {-1, 4, 11, 13}, {-1, 8, 5, 15}, {+1, 8, 7, 13}, {+1, 12, 5, 11}, {-1, 12, 7, 9}},
    {{-1, 0, 9, 15}, {+1, 0, 11, 13}, {+1, 8, 1, 15}, {-1, 8, 3, 13}, {-1, 12, 1, 11}, {+1, 12, 3, 9}},
    {{+1, 0, 5, 15}, {-1, 0, 7, 13}, {-1, 4, 1, 15}, {+1, 4, 3, 13}, {+1, 12, 1, 7}, {-1, 12, 3, 5}},
    {{-1, 0, 5, 11}, {+1, 0, 7, 9}, {+1, 4, 1, 11}, {-1, 4, 3, 9}, {-1, 8, 1, 7}, {+1, 8, 3, 5}},
    {{-1, 4, 9, 14}, {+1, 4, 10, 13}, {+1, 8, 5, 14}, {-1, 8, 6, 13}, {-1, 12, 5, 10}, {+1, 12, 6, 9}},
    {{+1, 0, 9, 14}, {-1, 0, 10, 13}, {-1, 8, 1, 14}, {+1, 8, 2, 13}, {+1, 12, 1, 10}, {-1, 12, 2, 9}},
    {{-1, 0, 5, 14}, {+1, 0, 6, 13}, {+1, 4, 1, 14}, {-1, 4, 2, 13}, {-1, 12, 1, 6}, {+1, 12, 2, 5}},
    {{+1, 0, 5, 10}, {-1, 0, 6, 9}, {-1, 4, 1, 10}, {+1, 4, 2, 9}, {+1, 8, 1, 6}, {-1, 8, 2, 5}},
};
}  // namespace
static bool invert4x4(const float* T, float* iT) {
    for (int k = 0; k < 16; k++) {
        const Term* t = kCof[k];
        // inner products (rounded); a leading minus in the source negates the first factor, which is exact
        float m[6];
        for (int j = 0; j < 6; j++) m[j] = T[t[j].a] * T[t[j].b];
        const int r = k < 8 ? 0 : 1;                                    // the term that is a plain (rounded) product
        const float tr = m[r] * T[t[r].c];
        float s = t[r].s > 0 ? tr : -tr;
        for (int j = 0; j < 6; j++) if (j != r) s = std::fma(t[j].s > 0 ? m[j] : -m[j], T[t[j].c], s);
        iT[k] = s;
    }
    const float p1 = T[1] * iT[4];
    float det = std::fma(T[0], iT[0], p1);
    det = std::fma(T[2], iT[8], det);
    det = std::fma(T[3], iT[12], det);
    if (det == 0) return false;
    const float invdet = 1.0f / det;
    for (int i = 0; i < 16; i++) iT[i] *= invdet;
    return true;
}

// Same job as BLASInstance::Update (tiny_bvh.h:8386-8400): invert the transform, then take
// the world-space box of the 8 transformed corners of the BLAS root box.
void update_instance(Instance192& inst, const float* bb) {
    invert4x4(inst.transform, inst.invTransform);   // a singular transform leaves the unscaled cofactors behind, as in the reference
    Box w; w.reset();
    const float* T = inst.transform;
    for (int j = 0; j < 8; j++) {
        const float p[3] = {(j & 1) ? bb[3] : bb[0], (j & 2) ? bb[4] : bb[1], (j & 4) ? bb[5] : bb[2]};
        float t[3];
        // tinybvh_transform_point (tiny_bvh.h:512-522) as the reference build contracts it: the first product is fused
        // into the first addition, the second is a rounded product, the third is fused, the translation is added last
        for (int r = 0; r < 3; r++) t[r] = std::fma(T[r * 4 + 2], p[2], std::fma(T[r * 4], p[0], T[r * 4 + 1] * p[1])) + T[r * 4 + 3];
        const float ww = std::fma(T[14], p[2], std::fma(T[12], p[0], T[13] * p[1])) + T[15];
        if (ww != 1.0f) { const float r = 1.0f / ww; t[0] *= r; t[1] *= r; t[2] *= r; }
        w.grow(t);
    }
    for (int a = 0; a < 3; a++) inst.aabbMin[a] = w.mn[a], inst.aabbMax[a] = w.mx[a];
}

}  // namespace tbvh

// ---- CWBVH node renumbering: the nodes a ray is most likely to visit first -----------------------
#include <queue>
namespace tbvh {

// Best-first renumbering: repeatedly take the not-yet-expanded node with the largest surface
// area and give its interior children the next consecutive indices (slot order, so
// childBaseIndex + popc(...) addressing still works).  The first K nodes of the result are
// (to first order) the K nodes a random ray is most likely to visit.  Works on any valid
// CWBVH blob, reference-built or ours.  newIdx[old] = new; unreachable nodes keep 0xffffffff.
// Returns false — and the caller must not renumber — when the blob is not a strict tree (a child range that leaves the array, or a node
// reachable from two parents): the numbering would then collide or run past nNodes.
bool cwbvh_priority_order(const Vec4* in, uint32_t nNodes, std::vector<uint32_t>& newIdx) {
    auto u32 = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
    auto area = [&](uint32_t n) {
        const Vec4* p = in + (size_t)n * 5;
        const uint32_t ew = u32(p[0].w);
        const uint8_t* q = (const uint8_t*)(p + 2);
        float ext[3];
        for (int a = 0; a < 3; a++) {
            int mx = 0;
            for (int s = 0; s < 8; s++) mx = std::max(mx, (int)q[24 + 8 * a + s]);
            ext[a] = std::ldexp((float)mx, (int)(int8_t)((ew >> (8 * a)) & 255));
        }
        return ext[0] * ext[1] + ext[1] * ext[2] + ext[2] * ext[0];
    };
    newIdx.assign(nNodes, 0xffffffffu);
    std::priority_queue<std::pair<float, uint32_t>> pq;
    newIdx[0] = 0;
    uint32_t next = 1;
    pq.push({area(0), 0u});
    while (!pq.empty()) {
        const uint32_t n = pq.top().second; pq.pop();
        const Vec4* p = in + (size_t)n * 5;
        const uint32_t imask = u32(p[0].w) >> 24, base = u32(p[1].x);
        const uint32_t cnt = (uint32_t)__builtin_popcount(imask);
        if ((uint64_t)next + cnt > nNodes) return false;
        for (uint32_t j = 0; j < cnt; j++) {
            const uint32_t c = base + j;
            if (c >= nNodes || newIdx[c] != 0xffffffffu) return false;   // child out of range / shared by two parents: not a tree
            newIdx[c] = next + j;
            pq.push({area(c), c});
        }
        next += cnt;
    }
    for (uint32_t i = 0; i < nNodes; i++) if (newIdx[i] == 0xffffffffu) newIdx[i] = next++;   // unreachable nodes go last
    return next == nNodes;
}

}  // namespace tbvh

// ---- blob validation at upload: out-of-range indices must become an error code on the host,
// ---- not a wild read on the device ----------------------------------------------------------
namespace tbvh {

bool bvh_gpu_to_bvh2(const NodeAL* al, uint64_t nNodes, const uint32_t* primIdx, uint64_t nIdx, const Vec4* verts, uint64_t nTris, uint32_t maxLeafTris,
                     std::vector<Node2>& out, const Vec4* recs) {
    out.clear();
    if (!nNodes || al[0].triCount) return false;
    out.reserve(nNodes + nNodes / 4 + 2);
    out.resize(2);                                  // root = 0; slot 1 stays unused so that siblings sit at (even, odd) like the reference's array (tiny_bvh.h:2277)
    std::memset(out.data(), 0, 2 * sizeof(Node2));
    for (int k = 0; k < 3; k++) {
        out[0].mn[k] = std::min(al[0].lmin[k], al[0].rmin[k]);
        out[0].mx[k] = std::max(al[0].lmax[k], al[0].rmax[k]);
    }
    struct Item { uint32_t src, dst; bool leafRange; uint32_t first, count; };
    std::vector<Item> stack;
    stack.push_back(Item{0u, 0u, false, 0u, 0u});
    auto range_box = [&](uint32_t first, uint32_t count, const Node2& clip, Node2& n) {
        float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
        for (uint32_t i = first; i < first + count; i++) {
            if (recs) {                             // record mode: corners v0, v0 + e1, v0 + e2, each padded by two ulps of the larger operand
                if (i >= nIdx) continue;
                const Vec4 &v0 = recs[3 * (size_t)i], &e1 = recs[3 * (size_t)i + 1], &e2 = recs[3 * (size_t)i + 2];
                const float A[3] = {v0.x, v0.y, v0.z}, E1[3] = {e1.x, e1.y, e1.z}, E2[3] = {e2.x, e2.y, e2.z};
                for (int k = 0; k < 3; k++) {
                    const float m = std::max(std::fabs(A[k]), std::max(std::fabs(E1[k]), std::fabs(E2[k]))) * 2.4e-7f;
                    const float c0 = A[k], c1 = A[k] + E1[k], c2 = A[k] + E2[k];
                    mn[k] = std::min(mn[k], std::min(c0, std::min(c1, c2)) - m);
                    mx[k] = std::max(mx[k], std::max(c0, std::max(c1, c2)) + m);
                }
                continue;
            }
            const uint32_t p = i < nIdx ? primIdx[i] : 0xffffffffu;
            if (p >= nTris) continue;               // slack entries of an SBVH primIdx array
            for (int v = 0; v < 3; v++) {
                const Vec4& q = verts[(uint64_t)p * 3 + v];
                const float c[3] = {q.x, q.y, q.z};
                for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], c[k]); mx[k] = std::max(mx[k], c[k]); }
            }
        }
        for (int k = 0; k < 3; k++) {
            n.mn[k] = std::max(mn[k], clip.mn[k]); n.mx[k] = std::min(mx[k], clip.mx[k]);
            if (!(n.mn[k] <= n.mx[k])) { n.mn[k] = clip.mn[k]; n.mx[k] = clip.mx[k]; }   // (nothing valid inside: keep the leaf's box, conservative)
        }
    };
    while (!stack.empty()) {
        const Item it = stack.back(); stack.pop_back();
        uint32_t first = it.first, count = it.count;
        if (!it.leafRange) {
            const NodeAL& a = al[it.src];
            if (!a.triCount) {
                const uint32_t c = (uint32_t)out.size();
                out.resize(out.size() + 2);
                out[it.dst].leftFirst = c; out[it.dst].triCount = 0;
                for (int k = 0; k < 3; k++) {
                    out[c].mn[k] = a.lmin[k]; out[c].mx[k] = a.lmax[k];
                    out[c + 1].mn[k] = a.rmin[k]; out[c + 1].mx[k] = a.rmax[k];
                }
                out[c].leftFirst = out[c].triCount = out[c + 1].leftFirst = out[c + 1].triCount = 0;
                stack.push_back(Item{a.right, c + 1, false, 0u, 0u});
                stack.push_back(Item{a.left, c, false, 0u, 0u});
                continue;
            }
            first = a.firstTri; count = a.triCount;
        }
        if (count <= maxLeafTris) { out[it.dst].leftFirst = first; out[it.dst].triCount = count; continue; }
        const uint32_t c = (uint32_t)out.size(), half = count / 2;
        out.resize(out.size() + 2);
        const Node2 clip = out[it.dst];
        out[it.dst].leftFirst = c; out[it.dst].triCount = 0;
        out[c].leftFirst = out[c].triCount = out[c + 1].leftFirst = out[c + 1].triCount = 0;
        range_box(first, half, clip, out[c]);
        range_box(first + half, count - half, clip, out[c + 1]);
        stack.push_back(Item{0u, c + 1, true, first + half, count - half});
        stack.push_back(Item{0u, c, true, first, half});
    }
    return true;
}

bool bvh4_gpu_to_bvh2(const Vec4* b, uint64_t nBlocks, uint32_t maxLeafTris, std::vector<Node2>& out, std::vector<Vec4>& recs) {
    out.clear(); recs.clear();
    if (nBlocks < 4) return false;
    auto u32 = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
    auto down = [](float x) { return std::nextafter(std::nextafter(x, -INFINITY), -INFINITY); };
    auto up = [](float x) { return std::nextafter(std::nextafter(x, INFINITY), INFINITY); };
    struct Child { float mn[3], mx[3]; uint32_t info; };
    // the children of the 4-wide node at block offset o, with their boxes as the kernels evaluate them: bmin + q * (ext / 255)
    auto children = [&](uint32_t o, Child* c) -> int {
        const Vec4 d0 = b[o], d1 = b[o + 1], d2 = b[o + 2], d3 = b[o + 3];
        const uint32_t q[6] = {u32(d0.w), u32(d2.x), u32(d2.z), u32(d1.w), u32(d2.y), u32(d2.w)};   // xmin, ymin, zmin, xmax, ymax, zmax: four bytes each
        const float bmin[3] = {d0.x, d0.y, d0.z}, sc[3] = {d1.x, d1.y, d1.z};
        const uint32_t info[4] = {u32(d3.x), u32(d3.y), u32(d3.z), u32(d3.w)};
        int n = 0;
        for (int i = 0; i < 4; i++) {
            if (!info[i] || ((info[i] & 0x80000000u) && ((info[i] >> 16) & 0x7fffu) == 0u)) continue;   // empty slot / empty leaf
            Child& k = c[n++];
            k.info = info[i];
            for (int a = 0; a < 3; a++) {
                // the plane bmin + q * scale in float: its rounding error is an ulp of the LARGER operand (the sum may cancel to nearly zero), so the pad is two
                // ulps of that magnitude, not of the result
                const float pl = (float)((q[a] >> (8 * i)) & 255u) * sc[a], ph = (float)((q[3 + a] >> (8 * i)) & 255u) * sc[a];
                const float lo = bmin[a] + pl, hi = bmin[a] + ph;
                const float ml = std::max(std::fabs(bmin[a]), std::max(std::fabs(pl), std::fabs(lo))), mh = std::max(std::fabs(bmin[a]), std::max(std::fabs(ph), std::fabs(hi)));
                k.mn[a] = down(lo - ml * 2.4e-7f);
                k.mx[a] = up(hi + mh * 2.4e-7f);
            }
        }
        return n;
    };
    out.resize(2);
    std::memset(out.data(), 0, 2 * sizeof(Node2));
    struct Item { uint32_t dst; int kind; uint32_t a, b_; };   // kind 0: 4-wide node at block a; 1: leaf run of b_ triangles at block a; 2: record range [a, a + b_)
    std::vector<Item> stack;
    auto set_box = [&](uint32_t i, const float* mn, const float* mx) { for (int k = 0; k < 3; k++) { out[i].mn[k] = mn[k]; out[i].mx[k] = mx[k]; } };
    auto pair = [&]() { const uint32_t c = (uint32_t)out.size(); out.resize(out.size() + 2); std::memset(&out[c], 0, 2 * sizeof(Node2)); return c; };
    auto push_child = [&](uint32_t dst, const Child& k, uint32_t nodeOff) {
        set_box(dst, k.mn, k.mx);
        if (k.info & 0x80000000u) stack.push_back(Item{dst, 1, nodeOff + (k.info & 0xffffu), (k.info >> 16) & 0x7fffu});
        else stack.push_back(Item{dst, 0, k.info, 0u});
    };
    auto unite = [&](const Child& x, const Child& y, Child& r) { for (int k = 0; k < 3; k++) { r.mn[k] = std::min(x.mn[k], y.mn[k]); r.mx[k] = std::max(x.mx[k], y.mx[k]); } r.info = 0; };
    {   // the root's own box: the union of its children
        Child c[4];
        if ((uint64_t)0 + 4 > nBlocks) return false;
        const int n = children(0, c);
        if (n == 0) return false;
        Child all = c[0];
        for (int i = 1; i < n; i++) unite(all, c[i], all);
        set_box(0, all.mn, all.mx);
        if (n == 1 && (c[0].info & 0x80000000u)) return false;   // a single leaf: nothing to collapse
    }
    stack.push_back(Item{0u, 0, 0u, 0u});
    uint64_t guard = 0;
    while (!stack.empty()) {
        const Item it = stack.back(); stack.pop_back();
        if (++guard > nBlocks * 4 + 16) return false;           // (validated streams are trees; belt and braces)
        if (it.kind == 0) {
            if ((uint64_t)it.a + 4 > nBlocks) return false;
            Child c[4];
            const int n = children(it.a, c);
            if (n == 0) { out[it.dst].leftFirst = 0; out[it.dst].triCount = 0; return false; }
            if (n == 1) {   // one child: this binary node IS that child (its box stays the tighter of the two: the child's)
                Child k = c[0];
                for (int a = 0; a < 3; a++) { k.mn[a] = std::max(k.mn[a], out[it.dst].mn[a]); k.mx[a] = std::min(k.mx[a], out[it.dst].mx[a]); if (!(k.mn[a] <= k.mx[a])) { k.mn[a] = c[0].mn[a]; k.mx[a] = c[0].mx[a]; } }
                push_child(it.dst, k, it.a);
                continue;
            }
            const uint32_t p = pair();
            out[it.dst].leftFirst = p; out[it.dst].triCount = 0;
            if (n == 2) { push_child(p, c[0], it.a); push_child(p + 1, c[1], it.a); }
            else {
                // ((a, b), c) or ((a, b), (c, d)): the pairing whose unions have the smallest surface area (three candidates either way)
                {
                    auto area2 = [&](const Child& x, const Child& y) { Child u; unite(x, y, u); const float dx = u.mx[0] - u.mn[0], dy = u.mx[1] - u.mn[1], dz = u.mx[2] - u.mn[2]; return dx * dy + dy * dz + dz * dx; };
                    if (n == 3) {
                        const float a01 = area2(c[0], c[1]), a02 = area2(c[0], c[2]), a12 = area2(c[1], c[2]);
                        if (a02 < a01 && a02 <= a12) std::swap(c[1], c[2]);        // (0, 2) pair, 1 alone
                        else if (a12 < a01 && a12 < a02) std::swap(c[0], c[2]);    // (2, 1) pair, 0 alone
                    } else {
                        const float p0 = area2(c[0], c[1]) + area2(c[2], c[3]), p1 = area2(c[0], c[2]) + area2(c[1], c[3]), p2 = area2(c[0], c[3]) + area2(c[1], c[2]);
                        if (p1 < p0 && p1 <= p2) std::swap(c[1], c[2]);            // (0, 2) (1, 3)
                        else if (p2 < p0 && p2 < p1) std::swap(c[1], c[3]);        // (0, 3) (2, 1)
                    }
                }
                Child l; unite(c[0], c[1], l);
                set_box(p, l.mn, l.mx);
                const uint32_t pl = pair();
                out[p].leftFirst = pl; out[p].triCount = 0;
                push_child(pl, c[0], it.a); push_child(pl + 1, c[1], it.a);
                if (n == 3) push_child(p + 1, c[2], it.a);
                else {
                    Child r; unite(c[2], c[3], r);
                    set_box(p + 1, r.mn, r.mx);
                    const uint32_t pr = pair();
                    out[p + 1].leftFirst = pr; out[p + 1].triCount = 0;
                    push_child(pr, c[2], it.a); push_child(pr + 1, c[3], it.a);
                }
            }
            continue;
        }
        uint32_t first = it.a, count = it.b_;
        if (it.kind == 1) {   // gather the run's records; from here on the leaf is a range of `recs`
            if ((uint64_t)it.a + 3ull * count > nBlocks) return false;
            first = (uint32_t)(recs.size() / 3);
            recs.insert(recs.end(), b + it.a, b + it.a + 3ull * count);
        }
        if (count <= maxLeafTris) { out[it.dst].leftFirst = first; out[it.dst].triCount = count; continue; }
        const uint32_t p = pair(), half = count / 2;
        const Node2 clip = out[it.dst];
        out[it.dst].leftFirst = p; out[it.dst].triCount = 0;
        for (int h = 0; h < 2; h++) {
            const uint32_t f = h ? first + half : first, n = h ? count - half : half;
            float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
            for (uint32_t i = f; i < f + n; i++) {
                const Vec4 &v0 = recs[3 * (size_t)i], &e1 = recs[3 * (size_t)i + 1], &e2 = recs[3 * (size_t)i + 2];
                const float A[3] = {v0.x, v0.y, v0.z}, E1[3] = {e1.x, e1.y, e1.z}, E2[3] = {e2.x, e2.y, e2.z};
                for (int k = 0; k < 3; k++) {   // the corners v0, v0 + e1, v0 + e2, each padded by two ulps of the larger operand (as above)
                    const float m = std::max(std::fabs(A[k]), std::max(std::fabs(E1[k]), std::fabs(E2[k]))) * 2.4e-7f;
                    const float c0 = A[k], c1 = A[k] + E1[k], c2 = A[k] + E2[k];
                    mn[k] = std::min(mn[k], down(std::min(c0, std::min(c1, c2)) - m));
                    mx[k] = std::max(mx[k], up(std::max(c0, std::max(c1, c2)) + m));
                }
            }
            Node2& c = out[p + h];
            for (int k = 0; k < 3; k++) {
                c.mn[k] = std::max(mn[k], clip.mn[k]); c.mx[k] = std::min(mx[k], clip.mx[k]);
                if (!(c.mn[k] <= c.mx[k])) { c.mn[k] = clip.mn[k]; c.mx[k] = clip.mx[k]; }
            }
            stack.push_back(Item{p + (uint32_t)h, 2, f, n});
        }
    }
    // every level of the source rounded its children on a grid of its own, so a child may reach an ulp beyond the box its parent stored for it: make the
    // boxes nest (children are allocated after their parent, so one sweep from the back sees every child before its parent)
    for (size_t k = out.size(); k-- > 0;) {
        Node2& n = out[k];
        if (n.triCount || k == 1) continue;
        const Node2 &l = out[n.leftFirst], &r = out[n.leftFirst + 1];
        for (int a = 0; a < 3; a++) { n.mn[a] = std::min(n.mn[a], std::min(l.mn[a], r.mn[a])); n.mx[a] = std::max(n.mx[a], std::max(l.mx[a], r.mx[a])); }
    }
    return true;
}

// BVH8_CWBVH blob -> Wald-layout BVH2 + triangle records {v0|prim, e1, e2} in leaf order (the 4-wide copy a TLAS enters a BVH8_CWBVH BLAS through for
// closest-hit queries: capi_scene.hip: makeWide4Copy).  Child boxes as the kernels evaluate them (origin + q * 2^e, padded by ulps of the larger operand);
// the up to eight children of a node become a small binary tree: sorted along the node's widest axis, halved recursively.  Leaves keep their 1-3 triangles.
bool cwbvh_to_bvh2(const Vec4* nodes, uint64_t nNodes, const Vec4* tris, uint64_t nTriBlocks, std::vector<Node2>& out, std::vector<Vec4>& recs) {
    out.clear(); recs.clear();
    if (nNodes == 0) return false;
    auto u32 = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
    auto down = [](float x) { return std::nextafter(std::nextafter(x, -INFINITY), -INFINITY); };
    auto up = [](float x) { return std::nextafter(std::nextafter(x, INFINITY), INFINITY); };
    struct Child { float mn[3], mx[3]; uint32_t inner; uint32_t a, n; };   // inner: node index a; else n triangles from record a
    auto children = [&](uint64_t ni, Child* c) -> int {
        const Vec4* nd = nodes + 5 * ni;
        const uint32_t ew = u32(nd[0].w), imask = ew >> 24;
        const float org[3] = {nd[0].x, nd[0].y, nd[0].z};
        const float sc[3] = {std::ldexp(1.0f, (int)(int8_t)(ew & 255u)), std::ldexp(1.0f, (int)(int8_t)((ew >> 8) & 255u)), std::ldexp(1.0f, (int)(int8_t)((ew >> 16) & 255u))};
        const uint32_t childBase = u32(nd[1].x), triBase = u32(nd[1].y);
        const uint32_t meta[2] = {u32(nd[1].z), u32(nd[1].w)};
        const uint32_t q[6][2] = {{u32(nd[2].x), u32(nd[2].y)}, {u32(nd[2].z), u32(nd[2].w)}, {u32(nd[3].x), u32(nd[3].y)},    // qlo x, y, z
                                  {u32(nd[3].z), u32(nd[3].w)}, {u32(nd[4].x), u32(nd[4].y)}, {u32(nd[4].z), u32(nd[4].w)}};   // qhi x, y, z
        int n = 0;
        for (int s = 0; s < 8; s++) {
            const uint32_t m = (meta[s >> 2] >> (8 * (s & 3))) & 255u;
            if (!m) continue;
            Child& k = c[n++];
            if ((m & 0x18u) == 0x18u && (m >> 5) == 1u) {
                const uint32_t slot = (m & 31u) - 24u;
                k.inner = 1; k.a = childBase + (uint32_t)__builtin_popcount(imask & ((1u << slot) - 1u)); k.n = 0;
            } else {
                k.inner = 0; k.a = triBase / 3u + (m & 31u); k.n = (uint32_t)__builtin_popcount(m >> 5);
            }
            for (int a = 0; a < 3; a++) {
                const float pl = (float)((q[a][s >> 2] >> (8 * (s & 3))) & 255u) * sc[a], ph = (float)((q[3 + a][s >> 2] >> (8 * (s & 3))) & 255u) * sc[a];
                const float lo = org[a] + pl, hi = org[a] + ph;
                const float ml = std::max(std::fabs(org[a]), std::max(std::fabs(pl), std::fabs(lo))), mh = std::max(std::fabs(org[a]), std::max(std::fabs(ph), std::fabs(hi)));
                k.mn[a] = down(lo - ml * 2.4e-7f);
                k.mx[a] = up(hi + mh * 2.4e-7f);
            }
        }
        return n;
    };
    struct Item { uint32_t dst; uint32_t node; };
    std::vector<Item> stack;
    auto set_box = [&](uint32_t i, const float* mn, const float* mx) { for (int k = 0; k < 3; k++) { out[i].mn[k] = mn[k]; out[i].mx[k] = mx[k]; } };
    auto pair = [&]() { const uint32_t c = (uint32_t)out.size(); out.resize(out.size() + 2); std::memset(&out[c], 0, 2 * sizeof(Node2)); return c; };
    bool ok = true;
    auto place = [&](uint32_t dst, const Child& k) {   // binary node dst IS child k
        set_box(dst, k.mn, k.mx);
        if (k.inner) { if (k.a >= nNodes) { ok = false; return; } stack.push_back(Item{dst, k.a}); return; }
        if (k.n == 0 || 3ull * ((uint64_t)k.a + k.n) > nTriBlocks) { ok = false; return; }
        out[dst].leftFirst = (uint32_t)(recs.size() / 3); out[dst].triCount = k.n;
        for (uint32_t j = 0; j < k.n; j++) {   // {e2, e1, v0|prim} -> {v0|prim, e1, e2}
            const Vec4* r = tris + 3ull * (k.a + j);
            recs.push_back(r[2]); recs.push_back(r[1]); recs.push_back(r[0]);
        }
    };
    // children c[0..n) under binary node dst (n >= 2): agglomerative — the two clusters whose union has the smallest surface area are merged until two are
    // left (n <= 8: a few hundred box unions) —, so that the binary tree the 4-wide collapse starts from is as good as the wide node allows
    struct Cluster { float mn[3], mx[3]; int left, right, child; };   // a leaf cluster names a child; an inner one two clusters
    std::function<void(uint32_t, const Cluster*, int, const Child*)> emit = [&](uint32_t dst, const Cluster* cl, int ci, const Child* c) {
        const Cluster& k = cl[ci];
        if (k.child >= 0) { place(dst, c[k.child]); return; }
        set_box(dst, k.mn, k.mx);
        const uint32_t p = pair();
        out[dst].leftFirst = p; out[dst].triCount = 0;
        emit(p, cl, k.left, c); emit(p + 1, cl, k.right, c);
    };
    auto group = [&](uint32_t dst, Child* c, int n) {
        Cluster cl[16];
        int live[8], nLive = n, nCl = n;
        for (int i = 0; i < n; i++) { for (int a = 0; a < 3; a++) { cl[i].mn[a] = c[i].mn[a]; cl[i].mx[a] = c[i].mx[a]; } cl[i].left = cl[i].right = -1; cl[i].child = i; live[i] = i; }
        auto area = [](const float* mn, const float* mx) { const float x = mx[0] - mn[0], y = mx[1] - mn[1], z = mx[2] - mn[2]; return x * y + y * z + z * x; };
        while (nLive > 1) {
            int bi = 0, bj = 1; float best = 1e38f;
            for (int i = 0; i < nLive; i++) for (int j = i + 1; j < nLive; j++) {
                float mn[3], mx[3];
                for (int a = 0; a < 3; a++) { mn[a] = std::min(cl[live[i]].mn[a], cl[live[j]].mn[a]); mx[a] = std::max(cl[live[i]].mx[a], cl[live[j]].mx[a]); }
                const float ar = area(mn, mx);
                if (ar < best) { best = ar; bi = i; bj = j; }
            }
            Cluster& m = cl[nCl];
            for (int a = 0; a < 3; a++) { m.mn[a] = std::min(cl[live[bi]].mn[a], cl[live[bj]].mn[a]); m.mx[a] = std::max(cl[live[bi]].mx[a], cl[live[bj]].mx[a]); }
            m.left = live[bi]; m.right = live[bj]; m.child = -1;
            live[bi] = nCl++; live[bj] = live[--nLive];
        }
        // the root cluster IS binary node dst (its box was set by the caller, and is at least as tight)
        const Cluster& root = cl[live[0]];
        const uint32_t p = pair();
        out[dst].leftFirst = p; out[dst].triCount = 0;
        emit(p, cl, root.left, c); emit(p + 1, cl, root.right, c);
    };
    out.resize(2);
    std::memset(out.data(), 0, 2 * sizeof(Node2));
    {
        Child c[8];
        const int n = children(0, c);
        if (n == 0 || (n == 1 && !c[0].inner)) return false;   // a single leaf: nothing to collapse
        float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
        for (int i = 0; i < n; i++) for (int a = 0; a < 3; a++) { mn[a] = std::min(mn[a], c[i].mn[a]); mx[a] = std::max(mx[a], c[i].mx[a]); }
        set_box(0, mn, mx);
    }
    stack.push_back(Item{0u, 0u});
    uint64_t guard = 0;
    while (!stack.empty() && ok) {
        const Item it = stack.back(); stack.pop_back();
        if (++guard > nNodes + 16) return false;           // (validated blobs are trees; belt and braces)
        Child c[8];
        const int n = children(it.node, c);
        if (n == 0) return false;
        if (n == 1) {   // one child: this binary node IS that child (its box the tighter of the two)
            Child k = c[0];
            for (int a = 0; a < 3; a++) { k.mn[a] = std::max(k.mn[a], out[it.dst].mn[a]); k.mx[a] = std::min(k.mx[a], out[it.dst].mx[a]); if (!(k.mn[a] <= k.mx[a])) { k.mn[a] = c[0].mn[a]; k.mx[a] = c[0].mx[a]; } }
            place(it.dst, k);
            continue;
        }
        group(it.dst, c, n);
    }
    if (!ok) return false;
    // every level of the source rounded its children on a grid of its own: make the boxes nest (children are allocated after their parent)
    for (size_t k = out.size(); k-- > 0;) {
        Node2& n = out[k];
        if (n.triCount || k == 1) continue;
        const Node2 &l = out[n.leftFirst], &r = out[n.leftFirst + 1];
        for (int a = 0; a < 3; a++) { n.mn[a] = std::min(n.mn[a], std::min(l.mn[a], r.mn[a])); n.mx[a] = std::max(n.mx[a], std::max(l.mx[a], r.mx[a])); }
    }
    return true;
}

static const char* validate_bvh_gpu_impl(const NodeAL* n, uint64_t nNodes, uint64_t nIdx) {
    if (nNodes == 0) return "BVH_GPU: empty node array";
    for (uint64_t i = 0; i < nNodes; i++) {
        if (n[i].triCount) {
            if ((uint64_t)n[i].firstTri + n[i].triCount > nIdx) return "BVH_GPU leaf: firstTri + triCount exceeds the primIdx array";
        } else if (n[i].left >= nNodes || n[i].right >= nNodes) return "BVH_GPU interior node: child index out of range";
    }
    // in-range indices keep every read in bounds; only a TREE keeps the traversal finite (a cycle would hang the GPU): walk from the root,
    // no node may be reached twice
    std::vector<uint8_t> seen(nNodes, 0);
    std::vector<uint32_t> stack{0};
    seen[0] = 1;
    while (!stack.empty()) {
        const uint32_t i = stack.back(); stack.pop_back();
        if (n[i].triCount) continue;
        for (const uint32_t c : {n[i].left, n[i].right}) {
            if (seen[c]) return "BVH_GPU: a node is reachable along two paths (the node array is not a tree)";
            seen[c] = 1; stack.push_back(c);
        }
    }
    return nullptr;
}

static const char* validate_bvh4_gpu_impl(const Vec4* b, uint64_t nBlocks) {
    // walk the stream from the root; every reachable node and triangle run must lie inside it
    std::vector<uint32_t> stack{0};
    uint64_t visited = 0;
    while (!stack.empty()) {
        const uint32_t o = stack.back(); stack.pop_back();
        if ((uint64_t)o + 4 > nBlocks) return "BVH4_GPU node offset out of range";
        if (++visited > nBlocks) return "BVH4_GPU stream contains a cycle";
        uint32_t info[4]; std::memcpy(info, &b[o + 3], 16);
        for (int i = 0; i < 4; i++) {
            if (!info[i]) continue;
            if (info[i] & 0x80000000u) {
                const uint64_t cnt = (info[i] >> 16) & 0x7fff, rel = info[i] & 0xffff;
                if ((uint64_t)o + rel + 3 * cnt > nBlocks) return "BVH4_GPU leaf: triangle run exceeds the stream";
            } else stack.push_back(info[i]);
        }
    }
    return nullptr;
}

static const char* validate_cwbvh_impl(const Vec4* nodes, uint64_t nNodes, uint64_t nTriBlocks) {
    auto u32 = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
    if (nNodes == 0) return "CWBVH: empty node array";
    const uint32_t rootImask = u32(nodes[0].w) >> 24;
    uint32_t rootMeta[2]; std::memcpy(rootMeta, &nodes[1].z, 8);
    if (rootImask == 0 && rootMeta[0] == 0 && rootMeta[1] == 0) return "CWBVH root node is empty";
    for (uint64_t k = 0; k < nNodes; k++) {
        const Vec4* p = nodes + k * 5;
        const uint32_t imask = u32(p[0].w) >> 24, base = u32(p[1].x), triBase = u32(p[1].y);
        const uint32_t cnt = (uint32_t)__builtin_popcount(imask);
        if (cnt && (uint64_t)base + cnt > nNodes) return "CWBVH node: childBaseIndex + interior children exceeds the node array";
        uint8_t meta[8]; std::memcpy(meta, &p[1].z, 8);
        uint32_t maxTri = 0;
        for (int s = 0; s < 8; s++) {
            // the kernels take a slot for an interior child by its meta byte (0b001sssss, sssss = 24 + slot: bits 3 and 4 set) and index the
            // children through imask: the two must agree, or a child index runs past the range validated above
            if ((((uint32_t)meta[s] & 0x18u) == 0x18u) != (((imask >> s) & 1u) != 0u)) return "CWBVH node: a slot's meta byte and the interior mask disagree";
            if ((imask >> s) & 1) continue;
            const uint32_t m = meta[s];
            if (!m) continue;
            const uint32_t c = (m >> 5) == 1 ? 1 : (m >> 5) == 3 ? 2 : (m >> 5) == 7 ? 3 : 0;
            if (!c) return "CWBVH leaf slot: triangle count is not unary-encoded 1..3";
            maxTri = std::max(maxTri, (m & 31) + c);
        }
        if (maxTri > 24) return "CWBVH node: more than 24 triangles";
        // 48-byte records {e2, e1, v0 | prim}: three float4 each.  A blob made with CWBVH_COMPRESSED_TRIS (tiny_bvh.h:170-171, an experimental
        // switch: 64-byte Baldwin-Weber records, four float4 each) counts in fours
        if (maxTri && triBase % 3u != 0u) return "CWBVH node: triangle base is not a multiple of 3 float4 (a CWBVH_COMPRESSED_TRIS blob? that experimental format is not supported)";
        if (maxTri && (uint64_t)triBase + 3ull * maxTri > nTriBlocks) return "CWBVH node: triangle range exceeds the triangle array";
    }
    // ... and the node array must be a TREE under the root: a child range shared by two parents may close a cycle, and a cyclic blob is a
    // traversal that never ends (a hung GPU, not a wrong answer).  Nodes the root does not reach are ignored, as the traversal ignores them.
    std::vector<uint8_t> seen(nNodes, 0);
    std::vector<uint32_t> stack{0};
    seen[0] = 1;
    while (!stack.empty()) {
        const Vec4* p = nodes + (size_t)stack.back() * 5; stack.pop_back();
        const uint32_t imask = u32(p[0].w) >> 24, base = u32(p[1].x);
        const uint32_t cnt = (uint32_t)__builtin_popcount(imask);
        for (uint32_t j = 0; j < cnt; j++) {
            if (seen[base + j]) return "CWBVH: a node is reachable along two paths (the node array is not a strict tree)";
            seen[base + j] = 1; stack.push_back(base + j);
        }
    }
    return nullptr;
}

// The walks above allocate (a visited flag per node, a stack): running out of host memory on a very large blob must come back through the C ABI
// as TBVH_E_NOMEM, not unwind through extern "C" (kValidateNoMemory is compared by address in capi_internal.h: validate_failed).
const char* const kValidateNoMemory = "out of host memory while validating the blob";
const char* validate_bvh_gpu(const NodeAL* n, uint64_t nNodes, uint64_t nIdx) {
    try { return validate_bvh_gpu_impl(n, nNodes, nIdx); } catch (const std::bad_alloc&) { return kValidateNoMemory; }
}
const char* validate_bvh4_gpu(const Vec4* b, uint64_t nBlocks) {
    try { return validate_bvh4_gpu_impl(b, nBlocks); } catch (const std::bad_alloc&) { return kValidateNoMemory; }
}
const char* validate_cwbvh(const Vec4* nodes, uint64_t nNodes, uint64_t nTriBlocks) {
    try { return validate_cwbvh_impl(nodes, nNodes, nTriBlocks); } catch (const std::bad_alloc&) { return kValidateNoMemory; }
}

}  // namespace tbvh
