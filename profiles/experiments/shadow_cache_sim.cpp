// Design experiment (CPU only, not product, not oracle): would a shadow cache cut the any-hit kernel's visits?
// The 6 rays a surface point sends to one area light (2 light triangles x 3 samples, raygen.rgen:165-241) start at the same point
// and end on the same quad.  If one lane walked them one after the other it could try the last occluder (its triangle, the
// 4-wide record that held its leaf, or that record's parent subtree) before a full walk.  This prices those probes on the dumps
// of dump_rays.py (rays.bin + groups.bin): visits and triangle tests per ray with and without the cache, fall-back walks included.
//   g++ -O2 -std=c++17 -pthread shadow_cache_sim.cpp -o /tmp/scsim && /tmp/scsim /tmp/wsim
#define main wide_sim_main
#include "wide_sim.cpp"
#undef main
#include <map>

struct HitAt { bool hit; int node; int tri; int leafFirst, leafCount; };

// any-hit walk of the subtree rooted at wide record `root`; counts into visits / tests
static HitAt walk_from(const std::vector<WNode>& w, const Ray& r, int root, double& visits, double& tests) {
    float id[3]; for (int k = 0; k < 3; ++k) { float a = std::fabs(r.d[k]) < 1e-20f ? 1e-20f : std::fabs(r.d[k]); id[k] = (r.d[k] < 0 ? -1.f : 1.f) / a; }
    int stack[256]; int sp = 0; stack[sp++] = root;
    while (sp) {
        const int n = stack[--sp]; visits++;
        const WNode& nd = w[n];
        struct H { float t; int j; } hs[8]; int nh = 0;
        for (int j = 0; j < (int)nd.ch.size(); ++j) {
            const WChild& c = nd.ch[j];
            if (c.kind == 2) continue;
            float lo = 0.001f, hi = r.tmax;
            for (int k = 0; k < 3; ++k) {
                float t0 = (c.b.mn[k] - r.o[k]) * id[k], t1 = (c.b.mx[k] - r.o[k]) * id[k];
                lo = std::max(lo, std::min(t0, t1)); hi = std::min(hi, std::max(t0, t1));
            }
            if (lo <= hi * 1.0000005f) hs[nh++] = H{lo, j};
        }
        std::sort(hs, hs + nh, [](const H& a, const H& b) { return a.t < b.t; });
        int inner[8], ni = 0;
        for (int k = 0; k < nh; ++k) {
            const WChild& c = nd.ch[hs[k].j];
            if (c.kind == 1) { for (int t = 0; t < c.count; ++t) { tests++; if (mt(r, tris[c.idx + t])) return HitAt{true, n, c.idx + t, c.idx, c.count}; } }
            else inner[ni++] = c.idx;
        }
        for (int k = ni - 1; k >= 0; --k) stack[sp++] = inner[k];
    }
    return HitAt{false, -1, -1, 0, 0};
}

int main(int argc, char** argv) {
    std::string dir = argc > 1 ? argv[1] : "/tmp/wsim";
    nodes2 = load<Node2>(dir + "/nodes.bin"); tris = load<Tri>(dir + "/tris.bin"); rays = load<Ray>(dir + "/rays.bin");
    auto grp = load<uint32_t>(dir + "/groups.bin");
    auto g = load<float>(dir + "/grid.bin"); for (int k = 0; k < 3; ++k) { gorg[k] = g[k]; gscl[k] = g[4 + k]; g_center[k] = gorg[k] + 32768.0f * gscl[k]; }
    const Config cf = {"W4 f16 about the scene centre", 4, false, Q_F16CENTER, O_NEAREST, 64, 8};
    auto w = build(cf);
    std::vector<int> parent(w.size(), -1);
    for (size_t n = 0; n < w.size(); ++n) for (auto& c : w[n].ch) if (c.kind == 0) parent[c.idx] = (int)n;
    // rays of one (surface point, light), in emission order
    std::map<uint64_t, std::vector<uint32_t>> groups;
    for (size_t i = 0; i < rays.size(); ++i) groups[(uint64_t)grp[2 * i] << 16 | (grp[2 * i + 1] & 0xFFFF)].push_back((uint32_t)i);
    printf("%zu rays in %zu groups (%.2f per group), %zu wide records\n", rays.size(), groups.size(), (double)rays.size() / groups.size(), w.size());
    const char* names[] = {"no cache (today)", "last occluder: its triangle", "last occluder: its leaf", "last occluder: the record of its leaf, whole subtree", "last occluder: that record's parent subtree",
                           "triangle, then the record's subtree"};
    for (int mode = 0; mode < 6; ++mode) {
        double visits = 0, tests = 0, probes = 0, probeHits = 0, occluded = 0, full = 0;
        for (auto& kv : groups) {
            HitAt last{false, -1, -1, 0, 0};
            for (uint32_t ri : kv.second) {
                const Ray& r = rays[ri];
                bool done = false;
                if (mode && last.hit) {
                    probes++;
                    if (mode == 1 || mode == 5) { tests++; if (mt(r, tris[last.tri])) done = true; }
                    if (!done && mode == 2) { for (int t = 0; t < last.leafCount && !done; ++t) { tests++; if (mt(r, tris[last.leafFirst + t])) { done = true; last.tri = last.leafFirst + t; } } }
                    if (!done && (mode == 3 || mode == 4 || mode == 5)) {
                        int root = last.node; if (mode == 4 && parent[root] >= 0) root = parent[root];
                        HitAt h = walk_from(w, r, root, visits, tests);
                        if (h.hit) { done = true; last = h; }
                    }
                    probeHits += done;
                }
                if (!done) { full++; HitAt h = walk_from(w, r, 0, visits, tests); if (h.hit) { last = h; done = true; } }
                occluded += done;
            }
        }
        const double n = (double)rays.size();
        printf("%-56s visits %6.2f tests %5.2f per ray | probes %4.1f%% of rays, answered by the probe %4.1f%% of probes, full walks %4.1f%%, occluded %4.1f%%\n", names[mode], visits / n, tests / n,
               100 * probes / n, 100 * probeHits / std::max(1.0, probes), 100 * full / n, 100 * occluded / n);
    }
    return 0;
}
