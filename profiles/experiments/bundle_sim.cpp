// Design experiment (CPU only, not product, not oracle) — VERDICT r03 item 2: price a SHARED VISIT for the three shadow rays one
// pixel-sample sends at one light triangle (raygen.rgen:206-241: s = 0..2 share shadowRayOrigin and aim inside one triangle).
// One lane walks the union of the three rays' walks over the 4-wide view: one 64-B record load and one stack per visit, a slab
// test per (ray still interested, child), a ray leaves the bundle when it is answered.  Walk rule = the kernel's, generalised:
// enter the nearest hit child (t = the smallest entry distance among the rays that hit it; ties to the lower slot), stack the other
// hit children in slot order — each stack entry carries the 3-bit mask of the rays that hit it; a leaf's triangles are tested in
// storage order for the rays of its mask that are still unanswered.
// Counted per RAY and compared with one ray per lane (the same rule with bundles of one): record loads (L1 look-ups: 4 per record
// outside the LDS-resident tree top, 3 per triangle record), slab tests, triangle tests, and a vector-instruction model anchored on
// the kernel's measured 69 instructions per visit / 77 per triangle trip / 130 per refill.
//   python profiles/experiments/dump_rays.py sponza_class 480 270 /tmp/wsim
//   g++ -O2 -std=c++17 -pthread profiles/experiments/bundle_sim.cpp -o /tmp/bsim && /tmp/bsim /tmp/wsim
#define main wide_sim_main
#include "wide_sim.cpp"
#undef main
#include <map>

struct Cnt { double bundles = 0, rays = 0, visits = 0, topVisits = 0, slabs = 0, rayVisits = 0, triLoads = 0, triTests = 0, pushes = 0, maxStack = 0, over16 = 0, occluded = 0, lanesAtVisit = 0; };

static inline void inv_dir(const Ray& r, float* id) { for (int k = 0; k < 3; ++k) { float a = std::fabs(r.d[k]) < 1e-20f ? 1e-20f : std::fabs(r.d[k]); id[k] = (r.d[k] < 0 ? -1.f : 1.f) / a; } }

// walks up to 3 rays that share an origin as ONE bundle (n = 1: today's one ray per lane)
static void walk_bundle(const std::vector<WNode>& w, const Ray* const* rr, int n, Cnt& c) {
    float id[3][3];
    for (int i = 0; i < n; ++i) inv_dir(*rr[i], id[i]);
    struct E { int code; int leafCount; unsigned mask; bool leaf; };
    E stack[256]; int sp = 0;
    unsigned alive = (1u << n) - 1u;
    E cur{0, 0, alive, false};
    c.bundles++; c.rays += n;
    int depthMax = 0;
    for (;;) {
        const unsigned m = cur.mask & alive;
        bool popNext = true;
        if (m) {
            if (!cur.leaf) {
                c.visits++; if (cur.code < 40) c.topVisits++;
                const WNode& nd = w[cur.code];
                float tbest[8]; unsigned hm[8]; int nslots = (int)nd.ch.size();
                int lanes = 0; for (int i = 0; i < n; ++i) if (m >> i & 1) ++lanes;
                c.rayVisits += lanes; c.lanesAtVisit += lanes;
                for (int j = 0; j < nslots; ++j) {
                    hm[j] = 0; tbest[j] = 3e38f;
                    const WChild& ch = nd.ch[j];
                    if (ch.kind == 2) continue;
                    for (int i = 0; i < n; ++i) if (m >> i & 1) {
                        c.slabs++;
                        const Ray& r = *rr[i];
                        float lo = 0.001f, hi = r.tmax;
                        for (int k = 0; k < 3; ++k) { float t0 = (ch.b.mn[k] - r.o[k]) * id[i][k], t1 = (ch.b.mx[k] - r.o[k]) * id[i][k]; lo = std::max(lo, std::min(t0, t1)); hi = std::min(hi, std::max(t0, t1)); }
                        if (lo <= hi * 1.0000005f) { hm[j] |= 1u << i; tbest[j] = std::min(tbest[j], lo); }
                    }
                }
                int next = -1; float tn = 3e38f;
                for (int j = 0; j < nslots; ++j) if (hm[j] && tbest[j] < tn) { tn = tbest[j]; next = j; }       // strict <: ties to the lower slot
                if (next >= 0) {
                    for (int j = 0; j < nslots; ++j) if (hm[j] && j != next) { const WChild& ch = nd.ch[j]; stack[sp++] = E{ch.idx, ch.count, hm[j], ch.kind == 1}; c.pushes++; }
                    depthMax = std::max(depthMax, sp);
                    const WChild& ch = nd.ch[next]; cur = E{ch.idx, ch.count, hm[next], ch.kind == 1};
                    popNext = false;
                }
            } else {
                for (int t = 0; t < cur.leafCount; ++t) {
                    const unsigned mm = cur.mask & alive;
                    if (!mm) break;
                    c.triLoads++;
                    for (int i = 0; i < n; ++i) if (mm >> i & 1) { c.triTests++; if (mt(*rr[i], tris[cur.code + t])) { alive &= ~(1u << i); c.occluded++; } }
                }
            }
        }
        if (!alive) break;
        if (popNext) { if (!sp) break; cur = stack[--sp]; }
    }
    c.maxStack = std::max(c.maxStack, (double)depthMax); if (depthMax > 16) c.over16++;
}

int main(int argc, char** argv) {
    std::string dir = argc > 1 ? argv[1] : "/tmp/wsim";
    nodes2 = load<Node2>(dir + "/nodes.bin"); tris = load<Tri>(dir + "/tris.bin"); rays = load<Ray>(dir + "/rays.bin");
    auto bun = load<uint32_t>(dir + "/bundles.bin");
    auto g = load<float>(dir + "/grid.bin"); for (int k = 0; k < 3; ++k) { gorg[k] = g[k]; gscl[k] = g[4 + k]; g_center[k] = gorg[k] + 32768.0f * gscl[k]; }
    const Config cf = {"W4 f16 about the scene centre", 4, false, Q_F16CENTER, O_NEAREST, 64, 8};
    auto w = build(cf);
    std::map<uint64_t, std::vector<uint32_t>> groups;
    for (size_t i = 0; i < rays.size(); ++i) groups[(uint64_t)bun[2 * i] << 32 | bun[2 * i + 1]].push_back((uint32_t)i);
    size_t three = 0; for (auto& kv : groups) three += kv.second.size() == 3;
    printf("%zu rays in %zu (surface point, light triangle) groups, %zu of them of three rays; %zu wide records (greedy collapse of the host SAH tree, f16 planes)\n", rays.size(), groups.size(), three, w.size());
    Cnt one, b3;
    for (auto& kv : groups) {
        const Ray* rr[3];
        for (uint32_t ri : kv.second) { rr[0] = &rays[ri]; walk_bundle(w, rr, 1, one); }
        size_t i = 0;
        for (; i + 3 <= kv.second.size(); i += 3) { for (int k = 0; k < 3; ++k) rr[k] = &rays[kv.second[i + k]]; walk_bundle(w, rr, 3, b3); }
        for (; i < kv.second.size(); ++i) { rr[0] = &rays[kv.second[i]]; walk_bundle(w, rr, 1, b3); }      // the directional ray and leftovers walk alone
    }
    auto report = [&](const char* name, const Cnt& c) {
        const double R = c.rays;
        // instruction model per lane-visit: 21 fixed (loads + address, speculative pop, loop control, ordering, pushes) + 48 per ray tested
        // (4 children x (6 fma_mix + 4 min3 / max3 / min / max + widen + compare)); bundles add 6 fixed (mask bookkeeping, entry = code + mask)
        // and 4 per ray tested (its hit bits into the child masks, its t into the per-child minimum)
        const bool bundle = c.bundles < c.rays;
        const double nodeInst = c.visits * (21 + (bundle ? 6 : 0)) + c.rayVisits * (48 + (bundle ? 4 : 0));
        const double triInst = c.triLoads * 8 + c.triTests * 69;                  // 3 loads + unpack per record, Moeller-Trumbore per ray
        const double refillLo = bundle ? c.bundles * 60 + R * 70 : R * 130;       // shared: queue record + origin; per ray: 3 IEEE reciprocals, grid constants
        const double lookups = (c.visits - c.topVisits) * 4 + c.triLoads * 3;
        // SIMT: the three ray slots of a bundle are code, not lanes — a wave runs slot i's slab tests whenever ANY of its 64 bundles
        // still has ray i interested in the record it visits, i.e. practically always: every bundle visit costs all three slots
        const double nodeInstSimt = bundle ? c.visits * (21 + 6 + 3 * (48 + 4)) : nodeInst;
        const double triInstSimt = bundle ? c.triLoads * (8 + 3 * 69) : triInst;
        printf("%-28s per RAY: record visits %6.2f (in the LDS top %4.1f%%)  slab tests (ray x record) %6.2f  tri records %5.2f  tri tests %5.2f  pushes %5.2f | L1 look-ups %6.2f | model instructions: node %7.1f + tri %6.1f + refill %5.1f = %7.1f | occluded %.1f%%  rays per visit %.2f  deepest stack %.0f (%.0f bundles over 16)\n",
               name, c.visits / R, 100 * c.topVisits / c.visits, c.rayVisits / R, c.triLoads / R, c.triTests / R, c.pushes / R, lookups / R, nodeInst / R, triInst / R, refillLo / R, (nodeInst + triInst + refillLo) / R,
               100 * c.occluded / R, c.lanesAtVisit / c.visits, c.maxStack, c.over16);
        if (bundle) printf("%-28s          as a wave issues it (all three ray slots per visit / per triangle record): node %7.1f + tri %6.1f + refill %5.1f = %7.1f instructions per ray\n", "", nodeInstSimt / R, triInstSimt / R, refillLo / R,
                           (nodeInstSimt + triInstSimt + refillLo) / R);
    };
    report("one ray per lane (today)", one);
    report("bundles of three", b3);
    return 0;
}
