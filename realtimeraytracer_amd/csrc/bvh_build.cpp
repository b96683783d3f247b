/* bvh_build.cpp — the host tree builder: deterministic binned-SAH BVH2 (an intermediate pointer tree), insertion-based
 * optimisation of that tree, linearisation to children-in-parent nodes with fp32 planes, outward quantisation to the 32-B
 * RtrBvhNode, and the cost-driven collapse into the 4-wide view with its host restatement.  See bvh_build.h for what it
 * replaces in the reference. */
#include "bvh_build.h"
#include "../../include/rtr_math.h"

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <queue>

namespace rtr {
namespace {

struct Box {
    float mn[3], mx[3];
    void reset() {
        for (int k = 0; k < 3; ++k) { mn[k] = std::numeric_limits<float>::max(); mx[k] = -std::numeric_limits<float>::max(); }
    }
    void grow(const float* p) {
        for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], p[k]); mx[k] = std::max(mx[k], p[k]); }
    }
    void grow(const Box& b) {
        for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], b.mn[k]); mx[k] = std::max(mx[k], b.mx[k]); }
    }
    float half_area() const {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (dx < 0.f || dy < 0.f || dz < 0.f) return 0.f;
        return dx * dy + dy * dz + dz * dx;
    }
    bool same(const Box& b) const { return memcmp(this, &b, sizeof(Box)) == 0; }
};
inline Box unite(const Box& a, const Box& b) { Box r = a; r.grow(b); return r; }

struct Prim { Box box; float c[3]; uint32_t index; };

/* the tree while it is being built and optimised: leaves are nodes too (a range of `prims`) */
struct TNode {
    Box box;
    int32_t parent = -1, child[2] = {-1, -1};
    uint32_t lo = 0, hi = 0;            /* leaf: prims[lo, hi) */
    bool leaf() const { return child[0] < 0; }
};

constexpr uint32_t kMedianDepth = 48;      /* beyond this depth fall back to median splits: bounds the stack */
constexpr int kMaxBins = 64;

struct Builder {
    std::vector<Prim> prims;
    std::vector<TNode> t;
    BuildOptions opt;

    int32_t new_node(const Box& b, int32_t parent) { TNode n; n.box = b; n.parent = parent; t.push_back(n); return (int32_t)t.size() - 1; }

    /* Find the split of prims[lo,hi); returns mid (lo<mid<hi) or lo when a leaf is cheaper. */
    uint32_t split(uint32_t lo, uint32_t hi, const Box& box, uint32_t depth) {
        const int kBins = (int)std::min<uint32_t>(std::max<uint32_t>(opt.bins, 4u), (uint32_t)kMaxBins);
        uint32_t n = hi - lo;
        Box cb; cb.reset();
        for (uint32_t i = lo; i < hi; ++i) cb.grow(prims[i].c);
        float ext[3] = {cb.mx[0] - cb.mn[0], cb.mx[1] - cb.mn[1], cb.mx[2] - cb.mn[2]};
        int longest = ext[0] >= ext[1] ? (ext[0] >= ext[2] ? 0 : 2) : (ext[1] >= ext[2] ? 1 : 2);

        auto median_split = [&](int axis) {
            uint32_t mid = lo + n / 2;
            std::stable_sort(prims.begin() + lo, prims.begin() + hi, [axis](const Prim& a, const Prim& b) {
                return a.c[axis] < b.c[axis];
            });
            return mid;
        };
        if (depth >= kMedianDepth) return n <= opt.leafTarget ? lo : median_split(longest);

        float bestCost = std::numeric_limits<float>::max();
        int bestAxis = -1, bestBin = -1;
        float parentArea = box.half_area();
        for (int axis = 0; axis < 3; ++axis) {
            if (!(ext[axis] > 0.f)) continue;
            Box bb[kMaxBins]; uint32_t cnt[kMaxBins];
            for (int b = 0; b < kBins; ++b) { bb[b].reset(); cnt[b] = 0; }
            float scale = (float)kBins / ext[axis];
            for (uint32_t i = lo; i < hi; ++i) {
                int b = (int)((prims[i].c[axis] - cb.mn[axis]) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                bb[b].grow(prims[i].box); cnt[b]++;
            }
            float rightArea[kMaxBins]; uint32_t rightCnt[kMaxBins];
            Box acc; acc.reset(); uint32_t c = 0;
            for (int b = kBins - 1; b > 0; --b) { acc.grow(bb[b]); c += cnt[b]; rightArea[b] = acc.half_area(); rightCnt[b] = c; }
            acc.reset(); c = 0;
            for (int b = 0; b < kBins - 1; ++b) {
                acc.grow(bb[b]); c += cnt[b];
                if (c == 0 || rightCnt[b + 1] == 0) continue;
                float cost = opt.costTraverse + opt.costIntersect * (acc.half_area() * c + rightArea[b + 1] * rightCnt[b + 1]) /
                                                    (parentArea > 0.f ? parentArea : 1.f);
                if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestBin = b; }
            }
        }
        float leafCost = opt.costIntersect * n;
        if (bestAxis < 0) {                       /* all centroids coincide */
            return n <= opt.maxLeaf ? lo : median_split(longest);
        }
        if (n <= opt.leafTarget && leafCost <= bestCost) return lo;
        if (n <= opt.maxLeaf && leafCost <= bestCost) return lo;
        float scale = (float)kBins / ext[bestAxis];
        float cmn = cb.mn[bestAxis];
        auto it = std::stable_partition(prims.begin() + lo, prims.begin() + hi, [=](const Prim& p) {
            int b = (int)((p.c[bestAxis] - cmn) * scale);
            b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
            return b <= bestBin;
        });
        uint32_t mid = (uint32_t)(it - prims.begin());
        if (mid == lo || mid == hi) return median_split(longest);
        return mid;
    }

    Box range_box(uint32_t lo, uint32_t hi) const {
        Box b; b.reset();
        for (uint32_t i = lo; i < hi; ++i) b.grow(prims[i].box);
        return b;
    }

    /* Builds the subtree over [lo,hi) whose box is `box`; returns its node. */
    int32_t build(uint32_t lo, uint32_t hi, const Box& box, uint32_t depth, int32_t parent) {
        uint32_t n = hi - lo;
        uint32_t mid = lo;
        if (n > 1) mid = split(lo, hi, box, depth);
        const int32_t idx = new_node(box, parent);
        if (n == 1 || mid == lo) { t[idx].lo = lo; t[idx].hi = hi; return idx; }
        Box lb = range_box(lo, mid), rb = range_box(mid, hi);
        const int32_t l = build(lo, mid, lb, depth + 1, idx);
        const int32_t r = build(mid, hi, rb, depth + 1, idx);
        t[idx].child[0] = l; t[idx].child[1] = r;
        return idx;
    }
};

/* SAH cost of the subtree at `root`, relative to the root's area */
double tree_cost(const std::vector<TNode>& t, int32_t root, const BuildOptions& opt) {
    const double ra = std::max(t[root].box.half_area(), 1e-30f);
    double c = 0.0;
    std::vector<int32_t> st{root};
    while (!st.empty()) {
        const int32_t i = st.back(); st.pop_back();
        const TNode& n = t[i];
        if (n.leaf()) c += (double)n.box.half_area() / ra * (n.hi - n.lo) * opt.costIntersect;
        else { c += (double)n.box.half_area() / ra * opt.costTraverse; st.push_back(n.child[0]); st.push_back(n.child[1]); }
    }
    return c;
}

uint32_t tree_depth(const std::vector<TNode>& t, int32_t root) {       /* inner nodes on the longest root -> leaf path */
    uint32_t best = 0;
    std::vector<std::pair<int32_t, uint32_t>> st{{root, 0u}};
    while (!st.empty()) {
        auto [i, d] = st.back(); st.pop_back();
        if (t[i].leaf()) { best = std::max(best, d); continue; }
        st.push_back({t[i].child[0], d + 1}); st.push_back({t[i].child[1], d + 1});
    }
    return best;
}

/* Insertion-based optimisation (Bittner, Hapala, Havran: "Fast insertion-based optimization of bounding volume hierarchies",
 * CGF 2013).  An inner node N whose box is badly used is dissolved: its two subtrees are taken out (N and its parent become free
 * nodes, N's sibling moves up) and each is put back at the place in the tree where it raises the SAH cost least, found by a
 * branch-and-bound search over (cost so far of enlarging the ancestors) + (area of the new common parent).  Leaves keep their
 * triangles, so the triangle order changes only at linearisation.  Deterministic: one thread, fixed visiting order. */
struct Optimizer {
    std::vector<TNode>& t;
    int32_t root;
    explicit Optimizer(std::vector<TNode>& tt, int32_t r) : t(tt), root(r) {}

    void refit_up(int32_t i) {
        while (i >= 0) {
            const Box nb = unite(t[t[i].child[0]].box, t[t[i].child[1]].box);
            if (nb.same(t[i].box)) break;
            t[i].box = nb;
            i = t[i].parent;
        }
    }

    struct Cand { float induced; int32_t node; bool operator<(const Cand& o) const { return induced > o.induced; } };   /* min-heap */

    int32_t find_place(const Box& nb) {
        const float na = nb.half_area();
        float best = std::numeric_limits<float>::max(); int32_t bestNode = root;
        std::priority_queue<Cand> q;
        q.push({0.f, root});
        while (!q.empty()) {
            const Cand c = q.top(); q.pop();
            if (c.induced + na >= best) break;                       /* nothing below can beat the best: every later entry has more induced cost */
            const TNode& x = t[c.node];
            const float direct = unite(x.box, nb).half_area();
            const float total = c.induced + direct;
            if (total < best) { best = total; bestNode = c.node; }
            if (!x.leaf()) {
                const float ind = c.induced + (direct - x.box.half_area());
                if (ind + na < best) { q.push({ind, x.child[0]}); q.push({ind, x.child[1]}); }
            }
        }
        return bestNode;
    }

    /* `sub` (detached) goes next to x under the free node `fresh` */
    void insert(int32_t sub, int32_t fresh) {
        const int32_t x = find_place(t[sub].box);
        const int32_t px = t[x].parent;
        t[fresh].child[0] = x; t[fresh].child[1] = sub; t[fresh].parent = px;
        t[fresh].box = unite(t[x].box, t[sub].box);
        t[x].parent = fresh; t[sub].parent = fresh;
        if (px < 0) root = fresh;
        else { t[px].child[t[px].child[0] == x ? 0 : 1] = fresh; refit_up(px); }
    }

    bool dissolve(int32_t n) {
        if (t[n].leaf() || n == root) return false;
        const int32_t p = t[n].parent;
        if (p == root) return false;                                   /* keeps the root where it is */
        const int32_t g = t[p].parent;
        const int32_t s = t[p].child[0] == n ? t[p].child[1] : t[p].child[0];
        const int32_t l = t[n].child[0], r = t[n].child[1];
        t[g].child[t[g].child[0] == p ? 0 : 1] = s; t[s].parent = g;
        refit_up(g);
        t[l].parent = t[r].parent = -1;
        const bool lFirst = t[l].box.half_area() >= t[r].box.half_area();     /* the larger subtree first */
        insert(lFirst ? l : r, n);
        insert(lFirst ? r : l, p);
        return true;
    }

    void pass(float fraction) {
        /* how badly a node uses its box: large, with children much smaller than itself, one of them especially */
        std::vector<std::pair<float, int32_t>> order;
        order.reserve(t.size());
        for (int32_t i = 0; i < (int32_t)t.size(); ++i) {
            const TNode& n = t[i];
            if (n.leaf() || i == root || n.parent == root || n.parent < 0) continue;
            const float a = n.box.half_area(), a0 = t[n.child[0]].box.half_area(), a1 = t[n.child[1]].box.half_area();
            const float msum = a / std::max(0.5f * (a0 + a1), 1e-30f), mmin = a / std::max(std::min(a0, a1), 1e-30f);
            order.push_back({msum * mmin * a, i});
        }
        std::stable_sort(order.begin(), order.end(), [](const auto& x, const auto& y) { return x.first > y.first; });
        size_t take = (size_t)((double)order.size() * fraction);
        if (take > order.size()) take = order.size();
        for (size_t k = 0; k < take; ++k) {
            const int32_t n = order[k].second;
            /* an earlier step of this pass may have made n a child of the root, or the root */
            if (n == root || t[n].parent < 0 || t[n].parent == root) continue;
            dissolve(n);
        }
    }
};

}  // namespace

constexpr size_t kReinsertAutoMinTriangles = 4096;    /* below this a frame's rays cost nothing to trace whatever the tree; the decision is left alone */
constexpr float kReinsertProbeGain = 0.02f;      /* the probe must take 2 % off the SAH cost for the passes to run (measured gains: build_options_from_env) */

BuildOptions build_options_from_env() {
    BuildOptions o;
    /* Insertion-based optimisation is decided PER SCENE (BuildOptions::kReinsertAuto, build_bvh): on uniformly tessellated geometry
     * (the bench scene) the binned-SAH tree is already within 1.2 % of what three passes reach in SAH cost and the passes buy nothing
     * (profiles/r03/tree_lab_sponza_480x270.log); where large triangles sit beside fine detail (walls and column shafts beside cloth
     * and ornaments: scenes.sponza_mixed, the reference's Bistro) they take 10 % off the SAH cost, 8 % off the shadow rays' visits and
     * a quarter off the camera rays' (profiles/r04/tree_lab_sponza_mixed_480x270.log).  RTR_BVH_REINSERT_PASSES=n forces n passes (0: none). */
    o.reinsertPasses = BuildOptions::kReinsertAuto; o.reinsertFraction = 1.0f;
    auto u = [](const char* name, uint32_t& v) { if (const char* e = getenv(name)) if (*e) v = (uint32_t)strtoul(e, nullptr, 10); };
    auto f = [](const char* name, float& v) { if (const char* e = getenv(name)) if (*e) v = strtof(e, nullptr); };
    u("RTR_BVH_BINS", o.bins); u("RTR_BVH_LEAF_TARGET", o.leafTarget); u("RTR_BVH_MAX_LEAF", o.maxLeaf);
    u("RTR_BVH_WIDE_GREEDY", o.wideGreedy); u("RTR_BVH_REINSERT_PASSES", o.reinsertPasses); f("RTR_BVH_REINSERT_FRACTION", o.reinsertFraction);
    f("RTR_BVH_COST_TRAVERSE", o.costTraverse); f("RTR_BVH_COST_INTERSECT", o.costIntersect);
    if (o.maxLeaf < 1) o.maxLeaf = 1;
    if (o.maxLeaf > RTR_BVH_MAX_LEAF) o.maxLeaf = RTR_BVH_MAX_LEAF;
    if (o.leafTarget > o.maxLeaf) o.leafTarget = o.maxLeaf;
    return o;
}

bool build_bvh(const std::vector<WorldTriangle>& tris, BvhResult& out, std::string* err, const BuildOptions& opt) {
    auto t0 = std::chrono::steady_clock::now();
    out = BvhResult();
    Builder b;
    b.opt = opt;
    static const WorldTriangle kDummy = {{{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, 0xffffffffu, 0xffffffffu, 0};
    std::vector<WorldTriangle> dummy;
    const std::vector<WorldTriangle>* in = &tris;
    if (tris.empty()) {           /* a degenerate triangle never passes Moeller-Trumbore (a == 0) */
        dummy.push_back(kDummy); in = &dummy;
    }
    if (in->size() >= (1u << 28)) { if (err) *err = "too many triangles for the leaf encoding (2^28)"; return false; }
    b.prims.resize(in->size());
    Box all; all.reset();
    float maxAbs = 0.f;
    for (size_t i = 0; i < in->size(); ++i) {
        Prim& p = b.prims[i];
        p.box.reset(); p.index = (uint32_t)i;
        for (int c = 0; c < 3; ++c) {
            for (int k = 0; k < 3; ++k) {
                float x = (*in)[i].v[c][k];
                if (!std::isfinite(x)) { if (err) *err = "non-finite vertex position in triangle " + std::to_string(i); return false; }
                maxAbs = std::max(maxAbs, std::fabs(x));
            }
            p.box.grow((*in)[i].v[c]);
        }
        for (int k = 0; k < 3; ++k) p.c[k] = 0.5f * (p.box.mn[k] + p.box.mx[k]);
        all.grow(p.box);
    }
    /* outward padding of every stored box: 2^-18 of the largest coordinate magnitude (see rtr_slab) */
    const float pad = std::max(maxAbs, 1e-6f) * 3.814697265625e-06f;
    b.t.reserve(2 * in->size());
    int32_t root = b.build(0, (uint32_t)in->size(), all, 0, -1);
    out.sahCostBeforeOpt = (float)tree_cost(b.t, root, opt);

    /* a tree built for trace speed: take badly placed subtrees out and put them back where they cost least.  Kept only while the
     * tree stays within the 64-entry depth bound of the traversal stacks. */
    if (opt.reinsertPasses > 0 && b.t.size() > 7 && (opt.reinsertPasses != BuildOptions::kReinsertAuto || in->size() >= kReinsertAutoMinTriangles)) {
        auto o0 = std::chrono::steady_clock::now();
        std::vector<TNode> keep = b.t;
        Optimizer op(b.t, root);
        if (opt.reinsertPasses == BuildOptions::kReinsertAuto) {
            /* the probe: re-insert the worst tenth of the nodes (a tenth of a pass's time).  A top-down SAH tree over uniformly sized
             * triangles gains well under a per cent from it and is kept AS BUILT; one whose big triangles were binned beside small ones
             * gains several per cent, and then whole passes follow while each still takes a per cent off the cost (at most four). */
            const double c0 = out.sahCostBeforeOpt;
            op.pass(0.1f);
            double c = tree_cost(b.t, op.root, opt);
            out.optProbeGain = c0 > 0 ? (float)((c0 - c) / c0) : 0.f;
            if (out.optProbeGain >= kReinsertProbeGain) {
                for (uint32_t p = 0; p < 4; ++p) {
                    op.pass(1.0f);
                    ++out.optPasses;
                    const double cn = tree_cost(b.t, op.root, opt);
                    const bool worthIt = (c - cn) / c >= 0.01;
                    c = cn;
                    if (!worthIt) break;
                }
                if (tree_depth(b.t, op.root) > 64) { b.t.swap(keep); out.optPasses = 0; }
                else root = op.root;
            } else b.t.swap(keep);                                      /* as built: the probe's own moves are not kept */
        } else {
            for (uint32_t p = 0; p < opt.reinsertPasses; ++p) op.pass(opt.reinsertFraction);
            if (tree_depth(b.t, op.root) > 64) b.t.swap(keep);
            else { root = op.root; out.optPasses = opt.reinsertPasses; }
        }
        out.optMs = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - o0).count();
    }
    out.sahCost = (float)tree_cost(b.t, root, opt);
    if (const char* v = getenv("RTR_BVH_VERBOSE")) if (*v == '1')
        fprintf(stderr, "build_bvh: %zu triangles, SAH cost %.3f as built, re-insertion probe gain %.2f %%, %u passes kept, cost %.3f, %.0f ms\n", in->size(), out.sahCostBeforeOpt,
                100.0 * out.optProbeGain, out.optPasses, out.sahCost, out.optMs);

    /* linearise: inner nodes in DFS pre-order (nodes[0] = root), leaf triangles in the order the walk meets them */
    std::vector<BvhNodeF> nodesF;
    nodesF.reserve(b.t.size() / 2 + 1);
    out.tris.reserve(in->size());
    auto write_box = [&](float* dst, const Box& bx) { for (int k = 0; k < 3; ++k) { dst[k] = bx.mn[k] - pad; dst[3 + k] = bx.mx[k] + pad; } };
    auto emit_leaf = [&](const TNode& n) -> int32_t {
        const uint32_t first = (uint32_t)out.tris.size();
        for (uint32_t i = n.lo; i < n.hi; ++i) {
            const WorldTriangle& w = (*in)[b.prims[i].index];
            RtrBvhTri tr;
            for (int k = 0; k < 3; ++k) {
                tr.v0[k] = w.v[0][k];
                tr.e1[k] = w.v[1][k] - w.v[0][k];
                tr.e2[k] = w.v[2][k] - w.v[0][k];
            }
            tr.customIndex = w.customIndex; tr.primitiveId = w.primitiveId; tr.flags = w.flags;
            out.tris.push_back(tr);
        }
        const uint32_t cnt = n.hi - n.lo;
        out.maxLeafSize = std::max(out.maxLeafSize, cnt);
        return (int32_t)~((first << 3) | (cnt - 1u));
    };
    if (b.t[root].leaf()) {
        /* the root must be an inner node: both children point at the same leaf (testing a triangle twice cannot change the
         * (t,id)-minimal hit) */
        nodesF.emplace_back();
        memset(&nodesF[0], 0, sizeof(BvhNodeF));
        const int32_t leaf = emit_leaf(b.t[root]);
        write_box(&nodesF[0].f[0], b.t[root].box); write_box(&nodesF[0].f[6], b.t[root].box);
        nodesF[0].child[0] = leaf; nodesF[0].child[1] = leaf;
        out.maxDepth = 1;
    } else {
        struct Item { int32_t node; int32_t slotOwner; int side; uint32_t depth; };
        std::vector<Item> st{{root, -1, 0, 0}};
        while (!st.empty()) {
            const Item it = st.back(); st.pop_back();
            const TNode& n = b.t[it.node];
            int32_t code;
            if (n.leaf()) code = emit_leaf(n);
            else {
                code = (int32_t)nodesF.size();
                nodesF.emplace_back();
                BvhNodeF& nd = nodesF.back();
                memset(&nd, 0, sizeof nd);
                write_box(&nd.f[0], b.t[n.child[0]].box); write_box(&nd.f[6], b.t[n.child[1]].box);
                out.maxDepth = std::max(out.maxDepth, it.depth + 1);
                st.push_back({n.child[1], code, 1, it.depth + 1});      /* left subtree first */
                st.push_back({n.child[0], code, 0, it.depth + 1});
            }
            if (it.slotOwner >= 0) nodesF[(size_t)it.slotOwner].child[it.side] = code;
        }
    }
    out.nodes.resize(nodesF.size());
    quantize_nodes(nodesF.data(), nodesF.size(), out.grid, out.nodes.data());
    collapse_wide(nodesF.data(), nodesF.size(), opt.costTraverse, opt.costIntersect, out.wideShape, &out.wideCost, &out.wideCostGreedy);
    if (opt.wideGreedy) out.wideShape.clear();
    for (int k = 0; k < 3; ++k) { out.boundsMin[k] = all.mn[k]; out.boundsMax[k] = all.mx[k]; }
    out.boxPad = pad;
    out.buildMs = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return true;
}

void quantize_nodes(const BvhNodeF* in, size_t count, RtrBvhGrid& grid, RtrBvhNode* out) {
    float mn[3], mx[3];
    for (int k = 0; k < 3; ++k) {                       /* root box = union of the root's two child boxes (already padded) */
        mn[k] = std::min(in[0].f[k], in[0].f[6 + k]);
        mx[k] = std::max(in[0].f[3 + k], in[0].f[9 + k]);
    }
    grid = RtrBvhGrid{};
    rtr_grid_from_bounds(mn, mx, grid.origin, grid.scale);
    for (size_t i = 0; i < count; ++i) {
        const BvhNodeF& s = in[i];
        RtrBvhNode& d = out[i];
        for (int side = 0; side < 2; ++side)
            for (int k = 0; k < 3; ++k) {
                d.q[RTR_BVH_QSLOT(side, 0, k)] = (uint16_t)rtr_quant_lo(s.f[6 * side + k], grid.origin[k], grid.scale[k]);
                d.q[RTR_BVH_QSLOT(side, 1, k)] = (uint16_t)rtr_quant_hi(s.f[6 * side + 3 + k], grid.origin[k], grid.scale[k]);
            }
        d.child[0] = s.child[0]; d.child[1] = s.child[1];
    }
}

/* ---- the 4-wide view ---------------------------------------------------------------------------------------------------------------- */
namespace {

inline float area_of(const float* f6) {                      /* f6 = min xyz, max xyz */
    const float dx = f6[3] - f6[0], dy = f6[4] - f6[1], dz = f6[5] - f6[2];
    if (dx < 0.f || dy < 0.f || dz < 0.f) return 0.f;
    return dx * dy + dy * dz + dz * dx;
}
inline uint32_t leaf_count(int32_t code) { return (((uint32_t)~code) & 7u) + 1u; }

/* shape code = open1 | open2 << 2: open1 0 none, 1 slot 0, 2 slot 1; open2 0 none, 1..3 slot 0..2.  An "open" replaces the entry
 * in that slot by its left child and appends its right child in the next free slot — the in-place rule of k_wide_nodes. */
constexpr uint8_t kShapeOf[8] = {0u, 1u, 2u, 1u | (2u << 2), 1u | (1u << 2), 1u | (3u << 2), 2u | (2u << 2), 2u | (3u << 2)};

}  // namespace

void collapse_wide(const BvhNodeF* nodes, size_t count, float ct, float ci, std::vector<uint8_t>& shape, float* costOpt, float* costGreedy) {
    shape.assign(count, 0);
    const double inf = std::numeric_limits<double>::infinity();
    /* per node: cost as the root of a wide record (best shape / greedy shape), and the cheapest covers of its subtree with exactly
     * two entries; children have larger indices than their parent (DFS pre-order), so one backward sweep sees children first */
    std::vector<double> croot(count, 0.0), cgreedy(count, 0.0);
    auto one = [&](const std::vector<double>& c, const BvhNodeF& n, int side) -> double {       /* the child as ONE entry */
        const int32_t code = n.child[side];
        return code >= 0 ? c[(size_t)code] : (double)area_of(&n.f[6 * side]) * leaf_count(code) * ci;
    };
    auto two = [&](const std::vector<double>& c, const BvhNodeF& n, int side) -> double {       /* the child opened into its two children */
        const int32_t code = n.child[side];
        if (code < 0) return inf;
        const BvhNodeF& m = nodes[(size_t)code];
        return one(c, m, 0) + one(c, m, 1);
    };
    for (size_t ii = count; ii-- > 0;) {
        const BvhNodeF& n = nodes[ii];
        float own[6];
        for (int k = 0; k < 3; ++k) { own[k] = std::min(n.f[k], n.f[6 + k]); own[3 + k] = std::max(n.f[3 + k], n.f[9 + k]); }
        const double visit = (double)area_of(own) * ct;
        /* ---- cost-optimal shape ---- */
        {
            const double l1 = one(croot, n, 0), r1 = one(croot, n, 1), l2 = two(croot, n, 0), r2 = two(croot, n, 1);
            double c[8] = {l1 + r1, l2 + r1, l1 + r2, l2 + r2, inf, inf, inf, inf};
            if (n.child[0] >= 0) {                      /* three entries from the left child: open it, then one of its children */
                const BvhNodeF& m = nodes[(size_t)n.child[0]];
                c[4] = two(croot, m, 0) + one(croot, m, 1) + r1;
                c[5] = one(croot, m, 0) + two(croot, m, 1) + r1;
            }
            if (n.child[1] >= 0) {
                const BvhNodeF& m = nodes[(size_t)n.child[1]];
                c[6] = l1 + two(croot, m, 0) + one(croot, m, 1);
                c[7] = l1 + one(croot, m, 0) + two(croot, m, 1);
            }
            int best = 0;
            for (int s = 1; s < 8; ++s) if (c[s] < c[best]) best = s;      /* ties: the lower shape number */
            shape[ii] = kShapeOf[best];
            croot[ii] = visit + c[best];
        }
        /* ---- greedy shape (open the inner entry with the largest box while a slot is free), for comparison ---- */
        {
            struct E { const float* box; int32_t code; };
            E e[4] = {{&n.f[0], n.child[0]}, {&n.f[6], n.child[1]}, {nullptr, 0}, {nullptr, 0}};
            int k = 2;
            while (k < 4) {
                int best = -1; float ba = -1.f;
                for (int j = 0; j < k; ++j) if (e[j].code >= 0) { const float a = area_of(e[j].box); if (a > ba) { ba = a; best = j; } }
                if (best < 0) break;
                const BvhNodeF& m = nodes[(size_t)e[best].code];
                e[best] = E{&m.f[0], m.child[0]}; e[k++] = E{&m.f[6], m.child[1]};
            }
            double c = visit;
            for (int j = 0; j < k; ++j) c += e[j].code >= 0 ? cgreedy[(size_t)e[j].code] : (double)area_of(e[j].box) * leaf_count(e[j].code) * ci;
            cgreedy[ii] = c;
        }
    }
    float own[6];
    for (int k = 0; k < 3; ++k) { own[k] = std::min(nodes[0].f[k], nodes[0].f[6 + k]); own[3 + k] = std::max(nodes[0].f[3 + k], nodes[0].f[9 + k]); }
    const double ra = std::max(area_of(own), 1e-30f);
    if (costOpt) *costOpt = (float)(croot[0] / ra);
    if (costGreedy) *costGreedy = (float)(cgreedy[0] / ra);
}

/* ---- host restatement of the device's wide-view build (kernels/rtr_bvh.hip: k_wide_centre_*, k_wide_nodes; rtr_api.cpp: the
 * breadth-first order).  Integer arithmetic except the greedy rule's box areas, which are the device's float expression. ---- */
namespace {

inline uint32_t f16_bits_of_int(uint32_t a) {
    if (a == 0) return 0u;
    if (a > 65504u) return 0x7c00u;
    const int e = 31 - __builtin_clz(a);
    const uint32_t m = (e >= 10) ? (a >> (e - 10)) : (a << (10 - e));
    return ((uint32_t)(e + 15) << 10) | (m & 0x3ffu);
}
inline uint32_t f16_mag_down(uint32_t a) { if (a <= 2048u) return a; if (a > 65504u) return 65504u; const int sh = (31 - __builtin_clz(a)) - 10; return (a >> sh) << sh; }
inline uint32_t f16_mag_up(uint32_t a) { if (a <= 2048u) return a; const int sh = (31 - __builtin_clz(a)) - 10; return ((a + (1u << sh) - 1u) >> sh) << sh; }
inline uint32_t f16_floor_bits(int32_t v) { return v >= 0 ? f16_bits_of_int(f16_mag_down((uint32_t)v)) : (0x8000u | f16_bits_of_int(f16_mag_up((uint32_t)(-v)))); }
inline uint32_t f16_ceil_bits(int32_t v) { return v >= 0 ? f16_bits_of_int(f16_mag_up((uint32_t)v)) : (0x8000u | f16_bits_of_int(f16_mag_down((uint32_t)(-v)))); }
inline uint32_t slack_lo(uint32_t q, uint32_t c) { const int32_t v = (int32_t)q - (int32_t)c; return v >= 0 ? (uint32_t)v - f16_mag_down((uint32_t)v) : f16_mag_up((uint32_t)(-v)) - (uint32_t)(-v); }
inline uint32_t slack_hi(uint32_t q, uint32_t c) { const int32_t v = (int32_t)q - (int32_t)c; return v >= 0 ? f16_mag_up((uint32_t)v) - (uint32_t)v : (uint32_t)(-v) - f16_mag_down((uint32_t)(-v)); }

struct QBox { uint32_t lo[3], hi[3]; };
inline QBox side_box(const RtrBvhNode& n, int side) {
    QBox b;
    for (int k = 0; k < 3; ++k) { b.lo[k] = n.q[RTR_BVH_QSLOT(side, 0, k)]; b.hi[k] = n.q[RTR_BVH_QSLOT(side, 1, k)]; }
    return b;
}

}  // namespace

void make_wide_host(const RtrBvhNode* nodes, size_t count, const uint8_t* shape, RtrBvhGrid& grid, std::vector<RtrWideNode>& wide) {
    /* the wide centre (k_wide_centre_sum / candidates / cost / set) */
    constexpr uint32_t kBins = 512, kWindow = 32;
    unsigned long long sum[4] = {0, 0, 0, 0};
    std::vector<unsigned long long> hist(3 * kBins, 0ull);
    auto for_leaves = [&](auto&& fn) {
        for (size_t i = 0; i < count; ++i)
            for (int sd = 0; sd < 2; ++sd) if (nodes[i].child[sd] < 0) fn(side_box(nodes[i], sd));
    };
    for_leaves([&](const QBox& b) {
        for (int k = 0; k < 3; ++k) {
            sum[k] += b.lo[k] + b.hi[k];
            if (b.hi[k] - b.lo[k] <= 2u) {
                const int a = (k + 1) % 3, c = (k + 2) % 3;
                const unsigned long long area = (((unsigned long long)(b.hi[a] - b.lo[a]) * (b.hi[c] - b.lo[c])) >> 8) + 1ull;
                hist[k * kBins + (b.lo[k] >> 7)] += area;
                hist[k * kBins + (b.hi[k] >> 7)] += area;
            }
        }
        sum[3] += 2;
    });
    unsigned long long cand[6];
    for (int k = 0; k < 3; ++k) {
        unsigned long long mean = 32768ull;
        if (sum[3]) { mean = (sum[k] + sum[3] / 2) / sum[3]; if (mean > 65535ull) mean = 65535ull; }
        unsigned long long run = 0, best = 0; uint32_t bestStart = 0;
        for (uint32_t b = 0; b < kBins; ++b) {
            run += hist[k * kBins + b];
            if (b >= kWindow) run -= hist[k * kBins + b - kWindow];
            if (b + 1 >= kWindow && run > best) { best = run; bestStart = b + 1 - kWindow; }
        }
        cand[2 * k] = mean;
        cand[2 * k + 1] = best ? (unsigned long long)(bestStart * 128u + 2048u) : mean;
    }
    unsigned long long cost[6] = {0, 0, 0, 0, 0, 0};
    for_leaves([&](const QBox& b) {
        for (int k = 0; k < 3; ++k) {
            const int a = (k + 1) % 3, c = (k + 2) % 3;
            const unsigned long long area = (((unsigned long long)(b.hi[a] - b.lo[a]) * (b.hi[c] - b.lo[c])) >> 8) + 1ull;
            const uint32_t ext = b.hi[k] - b.lo[k] + 1u;
            for (int q = 0; q < 2; ++q) {
                const uint32_t cc = (uint32_t)cand[2 * k + q];
                uint32_t rel = ((slack_lo(b.lo[k], cc) + slack_hi(b.hi[k], cc)) << 8) / ext;
                if (rel > 256u) rel = 256u;
                cost[2 * k + q] += area * rel;
            }
        }
    });
    uint32_t c[3];
    for (int k = 0; k < 3; ++k) c[k] = (uint32_t)((cost[2 * k + 1] * 4ull <= cost[2 * k]) ? cand[2 * k + 1] : cand[2 * k]);
    grid.wideCentreXY = c[0] | (c[1] << 16);
    grid.wideCentreZ = c[2];

    /* one record per BVH2 node (k_wide_nodes), then the breadth-first order from the root */
    std::vector<RtrWideNode> all(count);
    const float sx = grid.scale[0], sy = grid.scale[1], sz = grid.scale[2];
    for (size_t i = 0; i < count; ++i) {
        RtrWideNode& o = all[i];
        for (int k = 0; k < 4; ++k) { o.plane[k][0] = 0x7c007c00u; o.plane[k][1] = 0xfc00fc00u; o.plane[k][2] = 0xfc007c00u; o.child[k] = RTR_WIDE_EMPTY; }
        uint32_t own[4]; int side[4]; int k = 2;
        own[0] = own[1] = (uint32_t)i; side[0] = 0; side[1] = 1;
        auto code_of = [&](int j) { return nodes[own[j]].child[side[j]]; };
        auto open = [&](int j) {
            const int32_t code = code_of(j);
            own[j] = (uint32_t)code; side[j] = 0; own[k] = (uint32_t)code; side[k] = 1; ++k;
        };
        if (shape) {
            const uint32_t s = shape[i], o1 = s & 3u, o2 = (s >> 2) & 3u;
            if (o1 && code_of((int)o1 - 1) >= 0) {
                open((int)o1 - 1);
                if (o2 && code_of((int)o2 - 1) >= 0) open((int)o2 - 1);
            }
        } else {
            while (k < 4) {
                int best = -1; float bestA = -1.0f;
                for (int j = 0; j < k; ++j) {
                    if (code_of(j) < 0) continue;
                    const QBox b = side_box(nodes[own[j]], side[j]);
                    const float dx = (float)(b.hi[0] - b.lo[0]) * sx, dy = (float)(b.hi[1] - b.lo[1]) * sy, dz = (float)(b.hi[2] - b.lo[2]) * sz;
                    const float a = dx * dy + dy * dz + dz * dx;
                    if (a > bestA) { bestA = a; best = j; }
                }
                if (best < 0) break;
                open(best);
            }
        }
        for (int j = 0; j < k; ++j) {
            const QBox b = side_box(nodes[own[j]], side[j]);
            const int32_t xmin = (int32_t)b.lo[0] - (int32_t)c[0], ymin = (int32_t)b.lo[1] - (int32_t)c[1], zmin = (int32_t)b.lo[2] - (int32_t)c[2];
            const int32_t xmax = (int32_t)b.hi[0] - (int32_t)c[0], ymax = (int32_t)b.hi[1] - (int32_t)c[1], zmax = (int32_t)b.hi[2] - (int32_t)c[2];
            o.plane[j][0] = f16_floor_bits(xmin) | (f16_floor_bits(ymin) << 16);
            o.plane[j][1] = f16_ceil_bits(xmax) | (f16_ceil_bits(ymax) << 16);
            o.plane[j][2] = f16_floor_bits(zmin) | (f16_ceil_bits(zmax) << 16);
            o.child[j] = code_of(j);
        }
    }
    std::vector<uint32_t> remap(count, 0xffffffffu), order;
    order.reserve(count);
    order.push_back(0); remap[0] = 0;
    for (size_t head = 0; head < order.size(); ++head)
        for (int k = 0; k < 4; ++k) {
            const int32_t cd = all[order[head]].child[k];
            if (cd >= 0 && (size_t)cd < count && remap[(size_t)cd] == 0xffffffffu) { remap[(size_t)cd] = (uint32_t)order.size(); order.push_back((uint32_t)cd); }
        }
    wide.resize(order.size());
    for (size_t j = 0; j < order.size(); ++j) {
        wide[j] = all[order[j]];
        for (int k = 0; k < 4; ++k) if (wide[j].child[k] >= 0) wide[j].child[k] = (int32_t)remap[(size_t)wide[j].child[k]];
    }
}

}  // namespace rtr
