// Design experiment (CPU only, not product, not oracle): what a shadow ray costs over candidate wide-node layouts.
// Input: the dumps of dump_rays.py (host-built BVH2 with 16-bit planes, triangle records, grid, shadow rays).
// For each configuration it collapses the BVH2 into W-wide nodes (greedy: open the child with the largest box while a slot is
// free — the rule k_wide_nodes uses), re-quantises the child boxes the way the layout would store them, walks every ray
// any-hit and prints per-ray averages: node visits, triangle tests, stack depth (one entry per pushed child vs one entry per
// visited node = "group" entries), and the resulting 16-B loads per ray.
//   g++ -O2 -std=c++17 -pthread wide_sim.cpp -o /tmp/wide_sim && /tmp/wide_sim /tmp/wsim
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

struct Box { float mn[3], mx[3]; };
static float area(const Box& b) { float d[3] = {b.mx[0] - b.mn[0], b.mx[1] - b.mn[1], b.mx[2] - b.mn[2]}; return d[0] * d[1] + d[1] * d[2] + d[2] * d[0]; }
static Box unite(const Box& a, const Box& b) { Box r; for (int k = 0; k < 3; ++k) { r.mn[k] = std::min(a.mn[k], b.mn[k]); r.mx[k] = std::max(a.mx[k], b.mx[k]); } return r; }

struct Node2 { uint16_t q[12]; int32_t child[2]; };
struct Tri { float v0[3]; uint32_t cu; float e1[3]; uint32_t pr; float e2[3]; uint32_t fl; };
struct Ray { float o[3], d[3], tmax; };

static std::vector<Node2> nodes2; static std::vector<Tri> tris; static std::vector<Ray> rays; static float gorg[3], gscl[3];

static Box box2(int n, int side) {          // RTR_BVH_QSLOT
    const Node2& nd = nodes2[n]; Box b;
    for (int ax = 0; ax < 3; ++ax) {
        int lo = ax < 2 ? side * 4 + 0 * 2 + ax : 8 + side * 2 + 0, hi = ax < 2 ? side * 4 + 1 * 2 + ax : 8 + side * 2 + 1;
        b.mn[ax] = gorg[ax] + nd.q[lo] * gscl[ax]; b.mx[ax] = gorg[ax] + nd.q[hi] * gscl[ax];
    }
    return b;
}
static Box tribox(int t) {                  // on the 16-bit scene grid, outward
    const Tri& T = tris[t]; Box b;
    for (int k = 0; k < 3; ++k) {
        float a = T.v0[k], c = T.v0[k] + T.e1[k], e = T.v0[k] + T.e2[k];
        float mn = std::min(a, std::min(c, e)), mx = std::max(a, std::max(c, e));
        b.mn[k] = gorg[k] + std::floor((mn - gorg[k]) / gscl[k]) * gscl[k];
        b.mx[k] = gorg[k] + std::ceil((mx - gorg[k]) / gscl[k]) * gscl[k];
    }
    return b;
}

enum Quant { Q_SCENE16, Q_LOCAL11, Q_LOCAL8, Q_FP6, Q_LOCAL6, Q_LOCAL7, Q_F16CENTER, Q_F16CORNER };
enum Order { O_NEAREST, O_SLOTOCT, O_SLOT, O_SATO, O_SATO_T, O_FAR, O_LEN };

struct WChild { Box b; int kind; int idx; int count; };   // kind 0 inner (idx = wide node), 1 leaf (idx = first tri, count)
struct WNode { std::vector<WChild> ch; };

struct Config { const char* name; int W; bool singleTri; Quant q; Order ord; int nodeBytes; int maxLeaf; };

static const float kFp6[] = {0, .125f, .25f, .375f, .5f, .625f, .75f, .875f, 1, 1.125f, 1.25f, 1.375f, 1.5f, 1.625f, 1.75f, 1.875f,
                             2, 2.25f, 2.5f, 2.75f, 3, 3.25f, 3.5f, 3.75f, 4, 4.5f, 5, 5.5f, 6, 6.5f, 7, 7.5f};
static float fp6_down(float x) {   // largest representable <= x (x in [-7.5, 7.5])
    if (x >= 0) { float r = 0; for (float v : kFp6) if (v <= x) r = v; return r; }
    for (float v : kFp6) if (-v <= x) return -v;
    return -7.5f;
}
static float fp6_up(float x) { return -fp6_down(-x); }

static float f16_down(float x) {      // largest f16-representable value <= x (|x| < 65504), x in scene-grid steps
    if (x == 0) return 0;
    float a = std::fabs(x); int e; std::frexp(a, &e);            // a = m * 2^e, m in [0.5, 1)
    float step = std::ldexp(1.0f, e - 11);                      // 11 significant bits
    if (a < 6.1e-5f) step = 5.96e-8f;
    float d = std::floor(x / step) * step;
    return d;
}
static float f16_up(float x) { return -f16_down(-x); }
static float g_center[3];

static void quantize(WNode& n, Quant q) {
    if (q == Q_SCENE16 || n.ch.empty()) return;
    if (q == Q_F16CENTER || q == Q_F16CORNER) {                 // planes as f16 values on a scene-wide grid whose origin is the scene centre / min corner
        for (int k = 0; k < 3; ++k) {
            const float org = q == Q_F16CENTER ? g_center[k] : gorg[k];
            for (auto& ch : n.ch) {
                ch.b.mn[k] = org + f16_down((ch.b.mn[k] - org) / gscl[k]) * gscl[k];
                ch.b.mx[k] = org + f16_up((ch.b.mx[k] - org) / gscl[k]) * gscl[k];
            }
        }
        return;
    }
    Box nb = n.ch[0].b; for (auto& c : n.ch) nb = unite(nb, c.b);
    for (int k = 0; k < 3; ++k) {
        if (q == Q_FP6) {
            float c = 0.5f * (nb.mn[k] + nb.mx[k]), h = 0.5f * (nb.mx[k] - nb.mn[k]);
            float s = gscl[k]; while (s * 7.5f < h) s *= 2;            // power-of-two multiple of the scene step
            c = gorg[k] + std::round((c - gorg[k]) / gscl[k]) * gscl[k];
            while (s * 7.5f < std::max(nb.mx[k] - c, c - nb.mn[k])) s *= 2;
            for (auto& ch : n.ch) { ch.b.mn[k] = c + fp6_down((ch.b.mn[k] - c) / s) * s; ch.b.mx[k] = c + fp6_up((ch.b.mx[k] - c) / s) * s; }
        } else {
            int bits = q == Q_LOCAL11 ? 11 : (q == Q_LOCAL8 ? 8 : (q == Q_LOCAL7 ? 7 : 6));
            float steps = (float)((1 << bits) - 1), s = gscl[k];
            while (s * steps < nb.mx[k] - nb.mn[k]) s *= 2;
            for (auto& ch : n.ch) {
                ch.b.mn[k] = nb.mn[k] + std::floor((ch.b.mn[k] - nb.mn[k]) / s) * s;
                ch.b.mx[k] = nb.mn[k] + std::ceil((ch.b.mx[k] - nb.mn[k]) / s) * s;
            }
        }
    }
}

static void assign_slots(WNode& n, int W) {     // CWBVH-style: children to slots so that slot ^ octant orders them along the ray
    if (n.ch.empty()) return;
    Box nb = n.ch[0].b; for (auto& c : n.ch) nb = unite(nb, c.b);
    float cc[3] = {0.5f * (nb.mn[0] + nb.mx[0]), 0.5f * (nb.mn[1] + nb.mx[1]), 0.5f * (nb.mn[2] + nb.mx[2])};
    const int nc = (int)n.ch.size();
    std::vector<WChild> out(W, WChild{Box{{1, 1, 1}, {0, 0, 0}}, 2, 0, 0});   // kind 2 = empty
    std::vector<char> usedC(nc, 0), usedS(W, 0);
    for (int it = 0; it < nc; ++it) {
        float best = -1e30f; int bc = -1, bs = -1;
        for (int c = 0; c < nc; ++c) if (!usedC[c])
            for (int s = 0; s < W; ++s) if (!usedS[s]) {
                float v = 0;
                for (int k = 0; k < 3; ++k) { float off = 0.5f * (n.ch[c].b.mn[k] + n.ch[c].b.mx[k]) - cc[k]; v += ((s >> k) & 1) ? -off : off; }
                if (v > best) { best = v; bc = c; bs = s; }
            }
        usedC[bc] = 1; usedS[bs] = 1; out[bs] = n.ch[bc];
    }
    n.ch.swap(out);
}

static std::vector<WNode> build(const Config& cf) {
    std::vector<WNode> w; w.reserve(nodes2.size());
    struct Item { int n2; int wid; };
    std::vector<Item> todo; w.emplace_back(); todo.push_back({0, 0});
    while (!todo.empty()) {
        Item it = todo.back(); todo.pop_back();
        struct E { Box b; int code; };   // code >= 0 BVH2 inner, < 0 leaf code, special: tri (kind) handled below
        std::vector<E> e; std::vector<WChild> single;
        if (it.n2 >= 0) { e.push_back({box2(it.n2, 0), nodes2[it.n2].child[0]}); e.push_back({box2(it.n2, 1), nodes2[it.n2].child[1]}); }
        else {            // a pure triangle node made from a leaf code
            uint32_t code = (uint32_t)~it.n2; uint32_t first = code >> 3, cnt = (code & 7u) + 1u;
            for (uint32_t k = 0; k < cnt; ++k) single.push_back(WChild{tribox(first + k), 1, (int)(first + k), 1});
        }
        while ((int)(e.size() + single.size()) < cf.W) {
            int best = -1; float ba = -1;
            for (int j = 0; j < (int)e.size(); ++j) {
                bool openable = e[j].code >= 0;
                if (!openable && cf.singleTri) { uint32_t cnt = (((uint32_t)~e[j].code) & 7u) + 1u; openable = cnt > 1 && (int)(e.size() + single.size() - 1 + cnt) <= cf.W; }
                if (!openable && cf.maxLeaf < 8 && e[j].code < 0) { uint32_t cnt = (((uint32_t)~e[j].code) & 7u) + 1u; openable = (int)cnt > cf.maxLeaf && (int)(e.size() + single.size() - 1 + cnt) <= cf.W; }
                if (openable) { float a = area(e[j].b); if (a > ba) { ba = a; best = j; } }
            }
            if (best < 0) break;
            E x = e[best]; e.erase(e.begin() + best);
            if (x.code >= 0) { e.push_back({box2(x.code, 0), nodes2[x.code].child[0]}); e.push_back({box2(x.code, 1), nodes2[x.code].child[1]}); }
            else { uint32_t code = (uint32_t)~x.code; uint32_t first = code >> 3, cnt = (code & 7u) + 1u; for (uint32_t k = 0; k < cnt; ++k) single.push_back(WChild{tribox(first + k), 1, (int)(first + k), 1}); }
        }
        WNode nd;
        for (auto& c : single) nd.ch.push_back(c);
        for (auto& x : e) {
            if (x.code >= 0) { int id = (int)w.size(); w.emplace_back(); todo.push_back({x.code, id}); nd.ch.push_back(WChild{x.b, 0, id, 0}); }
            else {
                uint32_t code = (uint32_t)~x.code; uint32_t first = code >> 3, cnt = (code & 7u) + 1u;
                if ((cf.singleTri && cnt > 1) || (int)cnt > cf.maxLeaf) { int id = (int)w.size(); w.emplace_back(); todo.push_back({x.code, id}); nd.ch.push_back(WChild{x.b, 0, id, 0}); }
                else nd.ch.push_back(WChild{x.b, 1, (int)first, (int)cnt});
            }
        }
        quantize(nd, cf.q);
        if (cf.ord == O_SLOTOCT) assign_slots(nd, cf.W);
        w[it.wid] = std::move(nd);
    }
    return w;
}

static bool mt(const Ray& r, const Tri& T) {
    float h[3] = {r.d[1] * T.e2[2] - r.d[2] * T.e2[1], r.d[2] * T.e2[0] - r.d[0] * T.e2[2], r.d[0] * T.e2[1] - r.d[1] * T.e2[0]};
    float a = T.e1[0] * h[0] + T.e1[1] * h[1] + T.e1[2] * h[2];
    if (std::fabs(a) < 1e-5f) return false;
    float f = 1 / a, s[3] = {r.o[0] - T.v0[0], r.o[1] - T.v0[1], r.o[2] - T.v0[2]};
    float u = f * (s[0] * h[0] + s[1] * h[1] + s[2] * h[2]); if (u < 0 || u > 1) return false;
    float q[3] = {s[1] * T.e1[2] - s[2] * T.e1[1], s[2] * T.e1[0] - s[0] * T.e1[2], s[0] * T.e1[1] - s[1] * T.e1[0]};
    float v = f * (r.d[0] * q[0] + r.d[1] * q[1] + r.d[2] * q[2]); if (v < 0 || u + v > 1) return false;
    float t = f * (T.e2[0] * q[0] + T.e2[1] * q[1] + T.e2[2] * q[2]);
    return t > 0.001f && t < r.tmax;
}

struct Stats { double visOcc = 0, visVis = 0, top40 = 0, top85 = 0, top341 = 0; double visits = 0, tris = 0, leafSlots = 0, maxSingle = 0, maxGroup = 0, sumSingle = 0, sumGroup = 0, occluded = 0, over12 = 0, over16s = 0; };

static void walk(const std::vector<WNode>& w, const Config& cf, size_t r0, size_t r1, Stats& st) {
    std::vector<int> stack; std::vector<int> groupDepthAt;
    for (size_t ri = r0; ri < r1; ++ri) {
        const Ray& r = rays[ri];
        float id[3]; for (int k = 0; k < 3; ++k) { float a = std::fabs(r.d[k]) < 1e-20f ? 1e-20f : std::fabs(r.d[k]); id[k] = (r.d[k] < 0 ? -1.f : 1.f) / a; }
        const int oct = (r.d[0] < 0 ? 1 : 0) | (r.d[1] < 0 ? 2 : 0) | (r.d[2] < 0 ? 4 : 0);
        const int nearSlot = (~oct) & (cf.W - 1) & 7;
        stack.clear(); stack.push_back(0);
        // group-depth bookkeeping: an entry per visited node that still has unvisited hit children
        std::vector<std::pair<int, int>> gstack;   // (node, remaining)
        int maxS = 0, maxG = 0; bool hit = false; int myVisits = 0;
        while (!stack.empty() && !hit) {
            maxS = std::max(maxS, (int)stack.size());
            int n = stack.back(); stack.pop_back();
            while (!gstack.empty() && gstack.back().second == 0) gstack.pop_back();
            if (!gstack.empty()) gstack.back().second--;
            st.visits++; ++myVisits; if (n < 40) st.top40++; if (n < 85) st.top85++; if (n < 341) st.top341++;
            const WNode& nd = w[n];
            struct H { float t; int j; float len; };
            H hs[8]; int nh = 0;
            for (int j = 0; j < (int)nd.ch.size(); ++j) {
                const WChild& c = nd.ch[j];
                if (c.kind == 2) continue;
                float lo = 0.001f, hi = r.tmax;
                for (int k = 0; k < 3; ++k) {
                    float t0 = (c.b.mn[k] - r.o[k]) * id[k], t1 = (c.b.mx[k] - r.o[k]) * id[k];
                    lo = std::max(lo, std::min(t0, t1)); hi = std::min(hi, std::max(t0, t1));
                }
                if (lo <= hi * 1.0000005f) hs[nh++] = H{lo, j, hi - lo};
            }
            // leaves first (any-hit: test the triangles of every hit leaf child), then push inner children
            int inner[8], ni = 0;
            if (cf.ord == O_NEAREST) std::sort(hs, hs + nh, [](const H& a, const H& b) { return a.t < b.t; });
            else if (cf.ord == O_SLOT) {}
            else if (cf.ord == O_FAR) std::sort(hs, hs + nh, [](const H& a, const H& b) { return a.t > b.t; });
            else if (cf.ord == O_LEN) std::sort(hs, hs + nh, [](const H& a, const H& b) { return a.len > b.len; });
            else if (cf.ord == O_SATO) std::stable_sort(hs, hs + nh, [&](const H& a, const H& b) { return area(nd.ch[a.j].b) > area(nd.ch[b.j].b); });
            else if (cf.ord == O_SATO_T) std::stable_sort(hs, hs + nh, [&](const H& a, const H& b) { return nd.ch[a.j].count > nd.ch[b.j].count; });
            else std::sort(hs, hs + nh, [&](const H& a, const H& b) { return (a.j ^ nearSlot) < (b.j ^ nearSlot); });
            for (int k = 0; k < nh && !hit; ++k) {
                const WChild& c = nd.ch[hs[k].j];
                if (c.kind == 1) { st.leafSlots++; for (int t = 0; t < c.count && !hit; ++t) { st.tris++; if (mt(r, tris[c.idx + t])) hit = true; } }
                else inner[ni++] = c.idx;
            }
            if (hit) break;
            for (int k = ni - 1; k >= 0; --k) stack.push_back(inner[k]);     // first in order on top
            if (ni > 1) gstack.push_back({n, ni});                           // a group entry holds the children not yet entered
            else if (ni == 1) { if (!gstack.empty()) gstack.back().second++; /* descend directly, no entry */ gstack.push_back({n, 1}); }
            int g = 0; for (auto& e : gstack) if (e.second > 1) ++g;
            maxG = std::max(maxG, g + 1);
        }
        if (hit) st.visOcc += myVisits; else st.visVis += myVisits;
        st.occluded += hit; st.sumSingle += maxS; st.sumGroup += maxG;
        st.maxSingle = std::max(st.maxSingle, (double)maxS); st.maxGroup = std::max(st.maxGroup, (double)maxG);
        if (maxG > 12) st.over12++;
        if (maxS > 16) st.over16s++;
    }
}

template <class T> static std::vector<T> load(const std::string& p) {
    FILE* f = fopen(p.c_str(), "rb"); if (!f) { perror(p.c_str()); exit(1); }
    fseek(f, 0, SEEK_END); size_t n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<T> v(n / sizeof(T)); if (fread(v.data(), sizeof(T), v.size(), f) != v.size()) exit(1); fclose(f); return v;
}

int main(int argc, char** argv) {
    std::string dir = argc > 1 ? argv[1] : "/tmp/wsim";
    nodes2 = load<Node2>(dir + "/nodes.bin"); tris = load<Tri>(dir + "/tris.bin"); rays = load<Ray>(dir + "/rays.bin");
    auto g = load<float>(dir + "/grid.bin"); for (int k = 0; k < 3; ++k) { gorg[k] = g[k]; gscl[k] = g[4 + k]; g_center[k] = gorg[k] + 32768.0f * gscl[k]; }
    if (const char* e = getenv("WSIM_CENTRE")) { float c[3]; if (sscanf(e, "%f,%f,%f", &c[0], &c[1], &c[2]) == 3) for (int k = 0; k < 3; ++k) g_center[k] = gorg[k] + c[k] * gscl[k]; }   /* grid steps */
    const Config cfgs[] = {
        {"W4 scene16 nearest (r01 kernel)    64B", 4, false, Q_SCENE16, O_NEAREST, 64, 8},
        {"W4 f16 about the scene centre      64B", 4, false, Q_F16CENTER, O_NEAREST, 64, 8},
        {"W4 f16 about the min corner        64B", 4, false, Q_F16CORNER, O_NEAREST, 64, 8},
        {"W4 local11 nearest                 64B", 4, false, Q_LOCAL11, O_NEAREST, 64, 8},
    };




    printf("%zu rays, %zu BVH2 nodes, %zu triangles\n", rays.size(), nodes2.size(), tris.size());
    printf("%-42s %8s %8s %8s %8s %8s %9s %9s %7s %7s %8s\n", "config", "nodes", "visits", "leafslt", "tris", "loads16", "stk1 avg", "stkG avg", "max1", "maxG", ">12grp");
    for (const Config& cf : cfgs) {
        auto w = build(cf);
        const int T = 8; std::vector<Stats> st(T); std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back([&, t] { walk(w, cf, rays.size() * t / T, rays.size() * (t + 1) / T, st[t]); });
        for (auto& x : th) x.join();
        Stats s; for (auto& x : st) { s.visits += x.visits; s.tris += x.tris; s.leafSlots += x.leafSlots; s.sumSingle += x.sumSingle; s.sumGroup += x.sumGroup; s.maxSingle = std::max(s.maxSingle, x.maxSingle); s.maxGroup = std::max(s.maxGroup, x.maxGroup); s.occluded += x.occluded; s.over12 += x.over12; s.visOcc += x.visOcc; s.visVis += x.visVis; s.top40 += x.top40; s.top85 += x.top85; s.top341 += x.top341; }
        const double n = (double)rays.size();
        printf("%-42s %8zu %8.2f %8.2f %8.2f %8.1f %9.2f %9.2f %7.0f %7.0f %8.0f   occl %.1f%% visits occ %.1f vis %.1f  top40 %.0f%% top85 %.0f%% top341 %.0f%%\n", cf.name, w.size(), s.visits / n, s.leafSlots / n, s.tris / n,
               s.visits / n * cf.nodeBytes / 16 + s.tris / n * 3, s.sumSingle / n, s.sumGroup / n, s.maxSingle, s.maxGroup, s.over12, 100 * s.occluded / n, s.visOcc / std::max(1.0, s.occluded), s.visVis / std::max(1.0, n - s.occluded), 100 * s.top40 / s.visits, 100 * s.top85 / s.visits, 100 * s.top341 / s.visits);
    }
    return 0;
}
