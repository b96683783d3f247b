// Randomised soups through the host tree builder (bvh_build.cpp) in an AddressSanitizer + UBSan build: binned SAH, insertion-based
// optimisation, linearisation, quantisation, cost-driven and greedy 4-wide collapse, host wide view.  Checks the invariants the
// traversal kernels rely on; any memory error or undefined arithmetic aborts.   usage: fuzz_bvh <iterations>
#include "bvh_build.cpp"

#include <cstdio>
#include <random>

using namespace rtr;

static int fail(const char* what, int it) { fprintf(stderr, "fuzz_bvh: iteration %d: %s\n", it, what); return 1; }

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    std::mt19937 rng(12345);
    auto uni = [&](float a, float b) { return std::uniform_real_distribution<float>(a, b)(rng); };
    for (int it = 0; it < iters; ++it) {
        const int kind = it % 8;
        size_t n = (size_t)(rng() % (kind == 7 ? 6000 : 600));
        if (it < 4) n = (size_t)it;                       // 0, 1, 2, 3 triangles
        std::vector<WorldTriangle> soup(n);
        const float scale = kind == 3 ? 1e6f : (kind == 4 ? 1e-4f : 100.f);
        for (size_t i = 0; i < n; ++i) {
            WorldTriangle& w = soup[i];
            float c[3] = {uni(-1, 1) * scale, uni(-1, 1) * scale, uni(-1, 1) * scale};
            if (kind == 1) c[1] = 0.f;                    // everything in one plane
            if (kind == 2) { c[0] = c[1] = c[2] = 0.f; } // every centroid at the origin
            const float ext = kind == 5 ? (i % 7 == 0 ? scale : scale * 1e-3f) : scale * 0.05f;      // kind 5: mixed sizes
            for (int v = 0; v < 3; ++v) for (int k = 0; k < 3; ++k) w.v[v][k] = c[k] + uni(-1, 1) * ext;
            if (kind == 6 && i % 3 == 0) for (int k = 0; k < 3; ++k) w.v[1][k] = w.v[2][k] = w.v[0][k];   // degenerate
            w.customIndex = (uint32_t)(i % 5); w.primitiveId = (uint32_t)i; w.flags = 0;
        }
        BuildOptions opt;
        opt.bins = 4 + rng() % 61; opt.leafTarget = 1 + rng() % 8; opt.maxLeaf = opt.leafTarget + rng() % (9 - opt.leafTarget);
        opt.reinsertPasses = rng() % 4; opt.reinsertFraction = uni(0.05f, 1.0f); opt.wideGreedy = rng() % 2;
        BvhResult r; std::string err;
        if (!build_bvh(soup, r, &err, opt)) return fail(err.c_str(), it);
        const size_t nt = n ? n : 1;
        if (r.tris.size() != nt || r.nodes.empty() || r.maxDepth > 64 || r.maxLeafSize > 8) return fail("sizes", it);
        // every triangle in exactly one leaf, child indices valid, children after their parent
        std::vector<int> seen(nt, 0);
        for (size_t i = 0; i < r.nodes.size(); ++i)
            for (int s = 0; s < 2; ++s) {
                const int32_t c = r.nodes[i].child[s];
                if (c >= 0) { if ((size_t)c >= r.nodes.size() || (size_t)c <= i) return fail("child index", it); }
                else {
                    const uint32_t code = (uint32_t)~c, first = code >> 3, cnt = (code & 7u) + 1u;
                    if (first + cnt > nt) return fail("leaf range", it);
                    if (!(i == 0 && s == 1 && r.nodes[0].child[0] == c)) for (uint32_t k = 0; k < cnt; ++k) seen[first + k]++;
                }
                for (int k = 0; k < 3; ++k) if (r.nodes[i].q[RTR_BVH_QSLOT(s, 0, k)] > r.nodes[i].q[RTR_BVH_QSLOT(s, 1, k)]) return fail("inside-out box", it);
            }
        for (size_t i = 0; i < nt; ++i) if (seen[i] != 1) return fail("a triangle is not in exactly one leaf", it);
        // ids are a permutation
        std::vector<int> ids(nt, 0);
        if (n) for (const RtrBvhTri& t : r.tris) { if (t.primitiveId >= n) return fail("primitive id", it); ids[t.primitiveId]++; }
        if (n) for (size_t i = 0; i < n; ++i) if (ids[i] != 1) return fail("ids not a permutation", it);
        // the 4-wide view reaches every triangle once
        std::vector<RtrWideNode> wide;
        make_wide_host(r.nodes.data(), r.nodes.size(), r.wideShape.size() == r.nodes.size() ? r.wideShape.data() : nullptr, r.grid, wide);
        if (wide.empty()) return fail("no wide records", it);
        std::fill(seen.begin(), seen.end(), 0);
        for (size_t i = 0; i < wide.size(); ++i)
            for (int s = 0; s < 4; ++s) {
                const int32_t c = wide[i].child[s];
                if (c == RTR_WIDE_EMPTY) { if (s < 2) return fail("empty slot 0/1", it); continue; }
                if (c >= 0) { if ((size_t)c >= wide.size() || (size_t)c <= i) return fail("wide child index", it); }
                else {
                    const uint32_t code = (uint32_t)~c, first = code >> 3, cnt = (code & 7u) + 1u;
                    if (first + cnt > nt) return fail("wide leaf range", it);
                    if (!(i == 0 && s == 1 && wide[0].child[0] == c)) for (uint32_t k = 0; k < cnt; ++k) seen[first + k]++;
                }
            }
        for (size_t i = 0; i < nt; ++i) if (seen[i] != 1) return fail("a triangle is not in exactly one leaf of the wide view", it);
        if (!opt.wideGreedy && r.wideCost > r.wideCostGreedy * 1.0001f) return fail("cost-driven collapse costs more than the greedy one", it);
    }
    // non-finite input is refused
    std::vector<WorldTriangle> bad(1);
    memset(&bad[0], 0, sizeof bad[0]); bad[0].v[1][2] = std::numeric_limits<float>::infinity();
    BvhResult r; std::string err;
    if (build_bvh(bad, r, &err, BuildOptions())) return fail("non-finite corner accepted", -1);
    printf("fuzz_bvh: %d soups ok\n", iters);
    return 0;
}
