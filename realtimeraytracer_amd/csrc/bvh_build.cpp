/* bvh_build.cpp — deterministic binned-SAH BVH2 builder (children-in-parent nodes with fp32 planes), followed by the
 * outward quantisation to the 32-B RtrBvhNode.  See bvh_build.h for what it replaces in the reference. */
#include "bvh_build.h"
#include "../../include/rtr_math.h"

#include <cstdlib>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>

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
};

struct Prim { Box box; float c[3]; uint32_t index; };

constexpr int kBins = 32;
constexpr uint32_t kLeafTarget = 4;        /* SAH may stop at <= this many */
constexpr uint32_t kMedianDepth = 48;      /* beyond this depth fall back to median splits: bounds the stack */
constexpr float kCostTraverse = 1.0f;
constexpr float kCostIntersect = 1.0f;    /* relative to a node step; 0.7 ... 4 move the bench frame's visits and tests by < 2 % and its time not at all */

struct Builder {
    std::vector<Prim> prims;
    const std::vector<WorldTriangle>* src = nullptr;
    BvhResult* out = nullptr;
    std::vector<BvhNodeF> nodesF;
    float pad = 0.f;
    double sah = 0.0;
    float rootArea = 1.f;

    static int32_t leaf_code(uint32_t first, uint32_t count) {
        uint32_t code = (first << 3) | (count - 1u);
        return (int32_t)~code;
    }

    /* emit leaf triangles [lo,hi) and return the child code */
    int32_t emit_leaf(uint32_t lo, uint32_t hi, const Box& box) {
        uint32_t first = (uint32_t)out->tris.size();
        for (uint32_t i = lo; i < hi; ++i) {
            const WorldTriangle& w = (*src)[prims[i].index];
            RtrBvhTri t;
            for (int k = 0; k < 3; ++k) {
                t.v0[k] = w.v[0][k];
                t.e1[k] = w.v[1][k] - w.v[0][k];
                t.e2[k] = w.v[2][k] - w.v[0][k];
            }
            t.customIndex = w.customIndex; t.primitiveId = w.primitiveId; t.flags = w.flags;
            out->tris.push_back(t);
        }
        uint32_t n = hi - lo;
        out->maxLeafSize = std::max(out->maxLeafSize, n);
        sah += (double)(box.half_area() / rootArea) * n * kCostIntersect;
        return leaf_code(first, n);
    }

    void write_box(float* dst, const Box& b) const {
        for (int k = 0; k < 3; ++k) { dst[k] = b.mn[k] - pad; dst[3 + k] = b.mx[k] + pad; }
    }

    /* Find the split of prims[lo,hi); returns mid (lo<mid<hi) or lo when a leaf is cheaper. */
    uint32_t split(uint32_t lo, uint32_t hi, const Box& box, uint32_t depth) {
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
        if (depth >= kMedianDepth) return n <= kLeafTarget ? lo : median_split(longest);

        float bestCost = std::numeric_limits<float>::max();
        int bestAxis = -1, bestBin = -1;
        float parentArea = box.half_area();
        for (int axis = 0; axis < 3; ++axis) {
            if (!(ext[axis] > 0.f)) continue;
            Box bb[kBins]; uint32_t cnt[kBins];
            for (int b = 0; b < kBins; ++b) { bb[b].reset(); cnt[b] = 0; }
            float scale = (float)kBins / ext[axis];
            for (uint32_t i = lo; i < hi; ++i) {
                int b = (int)((prims[i].c[axis] - cb.mn[axis]) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                bb[b].grow(prims[i].box); cnt[b]++;
            }
            float rightArea[kBins]; uint32_t rightCnt[kBins];
            Box acc; acc.reset(); uint32_t c = 0;
            for (int b = kBins - 1; b > 0; --b) { acc.grow(bb[b]); c += cnt[b]; rightArea[b] = acc.half_area(); rightCnt[b] = c; }
            acc.reset(); c = 0;
            for (int b = 0; b < kBins - 1; ++b) {
                acc.grow(bb[b]); c += cnt[b];
                if (c == 0 || rightCnt[b + 1] == 0) continue;
                float cost = kCostTraverse + kCostIntersect * (acc.half_area() * c + rightArea[b + 1] * rightCnt[b + 1]) /
                                                 (parentArea > 0.f ? parentArea : 1.f);
                if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestBin = b; }
            }
        }
        float leafCost = kCostIntersect * n;
        if (bestAxis < 0) {                       /* all centroids coincide */
            return n <= RTR_BVH_MAX_LEAF ? lo : median_split(longest);
        }
        if (n <= kLeafTarget && leafCost <= bestCost) return lo;
        if (n <= RTR_BVH_MAX_LEAF && leafCost <= bestCost) return lo;
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

    /* Builds the subtree over [lo,hi) whose box is `box`; returns the child code (node index or leaf). */
    int32_t build(uint32_t lo, uint32_t hi, const Box& box, uint32_t depth, bool forceInner) {
        uint32_t n = hi - lo;
        uint32_t mid = lo;
        if (n > 1) mid = split(lo, hi, box, depth);
        if (n == 1 || mid == lo) {
            if (!forceInner) return emit_leaf(lo, hi, box);
            /* root must be an inner node: both children point at the same leaf (testing a triangle
             * twice cannot change the (t,id)-minimal hit) */
            uint32_t idx = (uint32_t)nodesF.size();
            nodesF.emplace_back();
            int32_t leaf = emit_leaf(lo, hi, box);
            BvhNodeF& nd = nodesF[idx];
            memset(&nd, 0, sizeof nd);
            write_box(&nd.f[0], box); write_box(&nd.f[6], box);
            nd.child[0] = leaf; nd.child[1] = leaf;
            out->maxDepth = std::max(out->maxDepth, depth + 1);
            return (int32_t)idx;
        }
        uint32_t idx = (uint32_t)nodesF.size();
        nodesF.emplace_back();
        out->maxDepth = std::max(out->maxDepth, depth + 1);
        sah += (double)(box.half_area() / rootArea) * kCostTraverse;
        Box lb = range_box(lo, mid), rb = range_box(mid, hi);
        int32_t lc = build(lo, mid, lb, depth + 1, false);
        int32_t rc = build(mid, hi, rb, depth + 1, false);
        BvhNodeF& nd = nodesF[idx];
        memset(&nd, 0, sizeof nd);
        write_box(&nd.f[0], lb); write_box(&nd.f[6], rb);
        nd.child[0] = lc; nd.child[1] = rc;
        return (int32_t)idx;
    }
};

}  // namespace

bool build_bvh(const std::vector<WorldTriangle>& tris, BvhResult& out, std::string* err) {
    auto t0 = std::chrono::steady_clock::now();
    out = BvhResult();
    Builder b;
    b.src = &tris; b.out = &out;
    static const WorldTriangle kDummy = {{{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, 0xffffffffu, 0xffffffffu, 0};
    std::vector<WorldTriangle> dummy;
    const std::vector<WorldTriangle>* in = &tris;
    if (tris.empty()) {           /* a degenerate triangle never passes Moeller-Trumbore (a == 0) */
        dummy.push_back(kDummy); in = &dummy; b.src = in;
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
    b.pad = std::max(maxAbs, 1e-6f) * 3.814697265625e-06f;
    b.rootArea = std::max(all.half_area(), 1e-30f);
    b.nodesF.reserve(in->size());
    out.tris.reserve(in->size());
    b.build(0, (uint32_t)in->size(), all, 0, true);
    out.nodes.resize(b.nodesF.size());
    quantize_nodes(b.nodesF.data(), b.nodesF.size(), out.grid, out.nodes.data());
    for (int k = 0; k < 3; ++k) { out.boundsMin[k] = all.mn[k]; out.boundsMax[k] = all.mx[k]; }
    out.boxPad = b.pad;
    out.sahCost = (float)b.sah;
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

}  // namespace rtr
