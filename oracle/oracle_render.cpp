/* oracle_render.cpp — CPU ORACLE (test infrastructure, never shipped, never the thing measured
 * except as bench.py's reported cpu_baseline).
 *
 * Scalar C++ restatement of the reference's per-pixel path.  Each function cites the reference
 * file:line it follows.  Parity status: UNPINNED at traversal/intersection (see oracle.h).
 *
 * Deliberate divergences from the reference text (all recorded in DESIGN.md):
 *   D1 (Q3)  zero vertex normals: reference computes normalize(vec3(0)) = NaN
 *            (closesthit.rchit:74-76); here the geometric normal, flipped to face the ray.
 *   D2       no HDRI supplied -> constant sky colour (same ToLinear); atan/acos are rtr_math.h's own forms.
 *   D3       pow(): own exp2/log2 forms (rtr_math.h); x < FLT_MIN -> 0.
 *   D4       hit accepted iff tmin < t < tmax; closest = min over (t, customIndex, primitiveID).
 *
 * Build: g++ -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).
 */
#include "oracle.h"
#include "../include/rtr_math.h"

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

struct Counters {
    uint64_t rays = 0, primary = 0, shadow = 0, nodes = 0, tris = 0, hits = 0, lightFetch = 0, lightTriFetch = 0;
    uint64_t shadowNodes = 0, shadowTris = 0, texFetch = 0, alphaTests = 0, primaryOverflow = 0, shadowOverflow = 0;
    uint64_t walk[8] = {0, 0, 0, 0, 0, 0, 0, 0};   /* oracle_walk_stats: shadow rays over the wide view, split by their answer */
    void add(const Counters& o) {
        for (int k = 0; k < 8; ++k) walk[k] += o.walk[k];
        shadowNodes += o.shadowNodes; shadowTris += o.shadowTris; texFetch += o.texFetch; alphaTests += o.alphaTests; primaryOverflow += o.primaryOverflow; shadowOverflow += o.shadowOverflow;
        rays += o.rays; primary += o.primary; shadow += o.shadow; nodes += o.nodes; tris += o.tris;
        hits += o.hits; lightFetch += o.lightFetch; lightTriFetch += o.lightTriFetch;
    }
};

struct WorldTri { rtr_v3 v0, e1, e2; uint32_t custom, prim, flags; };

struct Scene {
    const oracle_scene* s;
    std::vector<WorldTri> brute;            /* only when no BVH is supplied */
    std::vector<float> normalMat;           /* 9 floats per instance, indexed by customIndex */
    std::vector<const RtrInstance*> byCustom;
    rtr_v3 skyLinear;
    bool useWide = false;                   /* shadow rays over the wide view (oracle_scene::wide) */
    uint32_t shadowWalk = 0;                /* oracle_scene::shadowWalk */
    bool ownLeafFirst = false;              /* shadow rays that leave INTO their surface start at their own triangle's leaf (the staged pipeline's default; shadowWalk bit 1 turns it off) */
    std::vector<uint32_t> wideParent;       /* only with oracle_scene::walkProfile: record -> parent record * 4 + slot */
    int primaryStackLimit = 0;              /* 16 in the staged pipeline (k_primary_persist / k_primary + k_primary_tail), 0 = unbounded (megakernel) */
    bool primaryPackets = false;            /* camera rays walked tile by tile (trace_packet, k_primary_packet) */
    bool primaryWide = false;               /* camera rays one per lane over the 4-wide view (trace_wide_closest, k_primary4) */
};

struct Hit { bool hit; float t, u, v; uint32_t custom, prim; int32_t leaf; };      /* leaf: code of the leaf the hit triangle was found in (BVH walks; 0 otherwise) */

/* ---- texture(): 8-bit texels, linear filter, repeat addressing, one mip (image_sampler.cppm:26-42) ------------- */
inline void sample_tex(const rtr_texture& tx, float u, float v, float out[4], Counters& c) {
    c.texFetch++;
    if (!(u > -1.0e9f && u < 1.0e9f)) u = 0.0f;        /* NaN / absurd coordinates sample (0,0) */
    if (!(v > -1.0e9f && v < 1.0e9f)) v = 0.0f;
    const int W = (int)tx.width, H = (int)tx.height, ch = (int)tx.channels;
    const float uf = u - __builtin_floorf(u), vf = v - __builtin_floorf(v);      /* repeat */
    const float x = rtr_fma(uf, (float)W, -0.5f), y = rtr_fma(vf, (float)H, -0.5f);
    const float x0f = __builtin_floorf(x), y0f = __builtin_floorf(y);
    const float fx = x - x0f, fy = y - y0f;
    int x0 = (int)x0f, y0 = (int)y0f;
    if (x0 < 0) x0 += W;
    if (x0 >= W) x0 -= W;
    if (y0 < 0) y0 += H;
    if (y0 >= H) y0 -= H;
    int x1 = x0 + 1, y1 = y0 + 1;
    if (x1 >= W) x1 -= W;
    if (y1 >= H) y1 -= H;
    const uint8_t* p = tx.pixels;
    for (int k = 0; k < 4; ++k) {
        if (k >= ch) { out[k] = (k == 3) ? 1.0f : 0.0f; continue; }       /* R8: (r,0,0,1) */
        const float t00 = (float)p[((size_t)y0 * W + x0) * ch + k] / 255.0f, t10 = (float)p[((size_t)y0 * W + x1) * ch + k] / 255.0f;
        const float t01 = (float)p[((size_t)y1 * W + x0) * ch + k] / 255.0f, t11 = (float)p[((size_t)y1 * W + x1) * ch + k] / 255.0f;
        const float a = rtr_fma(t10 - t00, fx, t00), b = rtr_fma(t11 - t01, fx, t01);
        out[k] = rtr_fma(b - a, fy, a);
    }
}

/* opacity.rahit:31-64: invoked for candidates on non-opaque geometry whose object has an opacity map;
 * false -> ignoreIntersectionEXT */
inline bool alpha_pass(const rtr_scene_desc& D, uint32_t custom, uint32_t prim, float bu, float bv, Counters& c) {
    const RtrObjectInfo& oi = D.objects[custom - D.numLights];                           /* :33-34 */
    if (oi.usesOpacityMap == 0) return true;                                             /* :36-38 */
    c.alphaTests++;
    const uint32_t i0 = D.indices[3 * prim + 0 + oi.indexOffset], i1 = D.indices[3 * prim + 1 + oi.indexOffset],
                   i2 = D.indices[3 * prim + 2 + oi.indexOffset];                        /* :44-46 */
    const RtrVertex& v0 = D.vertices[i0 + oi.vertexOffset];
    const RtrVertex& v1 = D.vertices[i1 + oi.vertexOffset];
    const RtrVertex& v2 = D.vertices[i2 + oi.vertexOffset];
    const float b0 = 1.0f - bu - bv;                                                      /* :52 */
    const float uu = rtr_fma(v2.uv[0], bv, rtr_fma(v1.uv[0], bu, v0.uv[0] * b0));        /* :53 */
    const float vv = rtr_fma(v2.uv[1], bv, rtr_fma(v1.uv[1], bu, v0.uv[1] * b0));
    float t[4];
    sample_tex(D.textures[oi.opacityIndex], uu, vv, t, c);                                /* :55 */
    return !(t[0] < 0.9f);                                                                /* :58-60 */
}

/* ---- traversal: the algorithm the HIP kernels restate (DESIGN.md "Traversal") ---------------- */
inline bool id_less(uint32_t c0, uint32_t p0, uint32_t c1, uint32_t p1) {
    return c0 < c1 || (c0 == c1 && p0 < p1);
}

inline void consider(Hit& best, float t, float u, float v, uint32_t custom, uint32_t prim, float tmax) {
    if (!(t < tmax)) return;
    if (!best.hit) {
        best.hit = true; best.t = t; best.u = u; best.v = v; best.custom = custom; best.prim = prim;
        return;
    }
    if (t < best.t || (t == best.t && id_less(custom, prim, best.custom, best.prim))) {
        best.t = t; best.u = u; best.v = v; best.custom = custom; best.prim = prim;
    }
}

Hit trace_brute(const Scene& sc, rtr_v3 o, rtr_v3 d, float tmin, float tmax, bool anyHit, Counters& c) {
    Hit best{}; best.hit = false; best.t = tmax;
    for (const WorldTri& w : sc.brute) {
        float t, u, v;
        c.tris++; if (anyHit) c.shadowTris++;
        if (rtr_mt_intersect(o, d, w.v0, w.e1, w.e2, tmin, &t, &u, &v)) {
            if (!(t < tmax)) continue;
            if ((w.flags & 1u) && !alpha_pass(sc.s->desc, w.custom, w.prim, u, v, c)) continue;
            consider(best, t, u, v, w.custom, w.prim, tmax);
            if (anyHit && best.hit) return best;
        }
    }
    return best;
}

Hit trace_bvh(const Scene& sc, rtr_v3 o, rtr_v3 d, float tmin, float tmax, bool anyHit, Counters& c, int stackLimit = 0) {
    const RtrBvhNode* nodes = sc.s->nodes;
    const RtrBvhTri* tris = sc.s->tris;
    Hit best{}; best.hit = false; best.t = tmax;
    rtr_v3 idir = rtr_mk(rtr_safe_rcp_dir(d.x), rtr_safe_rcp_dir(d.y), rtr_safe_rcp_dir(d.z));
    rtr_v3 ga, gb;                                   /* t(q) = q * ga + gb, include/rtr_math.h */
    rtr_ray_grid(o, idir, sc.s->grid.origin, sc.s->grid.scale, &ga, &gb);
    int32_t stack[128];
    int sp = 0;
    int32_t cur = 0;   /* root is always an inner node */
    for (;;) {
        if (cur >= 0) {
            const RtrBvhNode& n = nodes[cur];
            c.nodes++; if (anyHit) c.shadowNodes++;
            float tl, tr;
            float limit = best.hit ? best.t : tmax;
            const uint16_t* q = n.q;
            int hl = rtr_slab_q(q[RTR_BVH_QSLOT(0, 0, 0)], q[RTR_BVH_QSLOT(0, 0, 1)], q[RTR_BVH_QSLOT(0, 0, 2)],
                                q[RTR_BVH_QSLOT(0, 1, 0)], q[RTR_BVH_QSLOT(0, 1, 1)], q[RTR_BVH_QSLOT(0, 1, 2)], ga, gb, tmin, limit, &tl);
            int hr = rtr_slab_q(q[RTR_BVH_QSLOT(1, 0, 0)], q[RTR_BVH_QSLOT(1, 0, 1)], q[RTR_BVH_QSLOT(1, 0, 2)],
                                q[RTR_BVH_QSLOT(1, 1, 0)], q[RTR_BVH_QSLOT(1, 1, 1)], q[RTR_BVH_QSLOT(1, 1, 2)], ga, gb, tmin, limit, &tr);
            if (hl && hr) {
                int32_t nearC = n.child[0], farC = n.child[1];      /* the child entered first / stacked: closest hit the nearer first, any hit the FARTHER first (ties: child 0) */
                if (anyHit ? tl < tr : tr < tl) { nearC = n.child[1]; farC = n.child[0]; }
                /* the primary kernels of the staged pipeline keep 16 stack entries in LDS; a camera ray that needs a 17th is abandoned
                 * there and re-traced from scratch by k_primary_tail with a full-depth stack (both parts are counted) */
                if (stackLimit > 0 && sp >= stackLimit) { c.primaryOverflow++; return trace_bvh(sc, o, d, tmin, tmax, anyHit, c, 0); }
                stack[sp++] = farC;
                cur = nearC;
                continue;
            } else if (hl) { cur = n.child[0]; continue; }
            else if (hr) { cur = n.child[1]; continue; }
        } else {
            uint32_t code = (uint32_t)~cur;
            uint32_t first = code >> 3, count = (code & 7u) + 1u;
            for (uint32_t i = 0; i < count; ++i) {
                const RtrBvhTri& tr = tris[first + i];
                float t, u, v;
                c.tris++; if (anyHit) c.shadowTris++;
                if (rtr_mt_intersect(o, d, rtr_ld3(tr.v0), rtr_ld3(tr.e1), rtr_ld3(tr.e2), tmin, &t, &u, &v)) {
                    if (!(t < tmax)) continue;
                    if ((tr.flags & 1u) && !alpha_pass(sc.s->desc, tr.customIndex, tr.primitiveId, u, v, c)) continue;
                    consider(best, t, u, v, tr.customIndex, tr.primitiveId, tmax);
                    if (best.hit && best.custom == tr.customIndex && best.prim == tr.primitiveId && best.t == t) best.leaf = cur;
                    if (anyHit && best.hit) return best;
                }
            }
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
    return best;
}

/* The camera rays of one 8x8 tile walking the BVH2 as ONE packet, restating k_primary_packet
 * (realtimeraytracer_amd/csrc/kernels/rtr_kernels.hip): lanes 0..63 = the tile's pixels row by row (`valid`: the lane has a pixel);
 * a node is visited with the mask of the lanes whose rays hit its box at the parent (the root: every valid lane); every lane of the
 * mask tests both child boxes against its own ray and its own closest t so far; the masks of the two children are the lanes that hit
 * them; one child hit: enter it; both: enter the one that is nearer (strictly, t_right < t_left) for more of the lanes that hit both
 * — ties: child 0 — and stack {other child, its mask}; a leaf's triangles are tested in storage order by the lanes of its mask, each
 * keeping its own closest hit (min over (t, customIndex, primitiveID)).  Counters: a node visit / triangle test per lane of the mask. */
void trace_packet(const Scene& sc, rtr_v3 o, const rtr_v3* d, const bool* valid, float tmin, float tmax, Hit* out, Counters& c) {
    const RtrBvhNode* nodes = sc.s->nodes;
    const RtrBvhTri* tris = sc.s->tris;
    rtr_v3 ga[64], gb[64];
    uint64_t active = 0;
    for (int l = 0; l < 64; ++l) {
        out[l].hit = false; out[l].t = tmax; out[l].u = out[l].v = 0.f; out[l].custom = out[l].prim = 0xffffffffu;
        if (!valid[l]) continue;
        active |= 1ull << l;
        c.rays++; c.primary++;
        const rtr_v3 idir = rtr_mk(rtr_safe_rcp_dir(d[l].x), rtr_safe_rcp_dir(d[l].y), rtr_safe_rcp_dir(d[l].z));
        rtr_ray_grid(o, idir, sc.s->grid.origin, sc.s->grid.scale, &ga[l], &gb[l]);
    }
    if (!active) return;
    struct Entry { int32_t code; uint64_t mask; };
    Entry stack[128];
    int sp = 0;
    int32_t cur = 0;
    for (;;) {
        bool pop = true;
        if (cur >= 0) {
            const RtrBvhNode& n = nodes[cur];
            const uint16_t* q = n.q;
            uint64_t mL = 0, mR = 0;
            int rNear = 0, lNear = 0;
            for (int l = 0; l < 64; ++l) {
                if (!((active >> l) & 1ull)) continue;
                c.nodes++;
                float tl, tr;
                const float limit = out[l].t;                 /* tmax until something is hit: the kernel's best.t */
                const int hl = rtr_slab_q(q[RTR_BVH_QSLOT(0, 0, 0)], q[RTR_BVH_QSLOT(0, 0, 1)], q[RTR_BVH_QSLOT(0, 0, 2)],
                                          q[RTR_BVH_QSLOT(0, 1, 0)], q[RTR_BVH_QSLOT(0, 1, 1)], q[RTR_BVH_QSLOT(0, 1, 2)], ga[l], gb[l], tmin, limit, &tl);
                const int hr = rtr_slab_q(q[RTR_BVH_QSLOT(1, 0, 0)], q[RTR_BVH_QSLOT(1, 0, 1)], q[RTR_BVH_QSLOT(1, 0, 2)],
                                          q[RTR_BVH_QSLOT(1, 1, 0)], q[RTR_BVH_QSLOT(1, 1, 1)], q[RTR_BVH_QSLOT(1, 1, 2)], ga[l], gb[l], tmin, limit, &tr);
                if (hl) mL |= 1ull << l;
                if (hr) mR |= 1ull << l;
                if (hl && hr) { if (tr < tl) ++rNear; else ++lNear; }
            }
            if (mL | mR) {
                pop = false;
                if (!mR) { cur = n.child[0]; active = mL; }
                else if (!mL) { cur = n.child[1]; active = mR; }
                else {
                    const bool rFirst = rNear > lNear;
                    stack[sp++] = Entry{rFirst ? n.child[0] : n.child[1], rFirst ? mL : mR};
                    cur = rFirst ? n.child[1] : n.child[0]; active = rFirst ? mR : mL;
                }
            }
        } else {
            const uint32_t code = (uint32_t)~cur;
            const uint32_t first = code >> 3, count = (code & 7u) + 1u;
            for (uint32_t i = 0; i < count; ++i) {
                const RtrBvhTri& tr = tris[first + i];
                for (int l = 0; l < 64; ++l) {
                    if (!((active >> l) & 1ull)) continue;
                    c.tris++;
                    float t, u, v;
                    if (!rtr_mt_intersect(o, d[l], rtr_ld3(tr.v0), rtr_ld3(tr.e1), rtr_ld3(tr.e2), tmin, &t, &u, &v)) continue;
                    if (!(t < tmax)) continue;
                    if ((tr.flags & 1u) && !alpha_pass(sc.s->desc, tr.customIndex, tr.primitiveId, u, v, c)) continue;
                    Hit& b = out[l];
                    if (t < b.t || (t == b.t && id_less(tr.customIndex, tr.primitiveId, b.custom, b.prim))) {
                        b.hit = true; b.t = t; b.u = u; b.v = v; b.custom = tr.customIndex; b.prim = tr.primitiveId; b.leaf = cur;
                    }
                }
            }
        }
        if (pop) {
            if (sp == 0) break;
            --sp; cur = stack[sp].code; active = stack[sp].mask;
        }
    }
}

/* IEEE binary16 -> binary32 (exact), no compiler support needed */
inline float half_bits_to_float(uint32_t h) {
    const uint32_t e = (h >> 10) & 31u, m = h & 1023u, sgn = (h & 0x8000u) << 16;
    uint32_t u;
    if (e == 0) {
        if (m == 0) u = sgn;
        else { int sh = 0; uint32_t mm = m; while (!(mm & 1024u)) { mm <<= 1; ++sh; } u = sgn | ((uint32_t)(113 - sh) << 23) | ((mm & 1023u) << 13); }
    } else if (e == 31) u = sgn | 0x7f800000u | (m << 13);
    else u = sgn | ((e + 112u) << 23) | (m << 13);
    return rtr_u2f(u);
}

/* Any-hit walk over the 4-wide view, restating k_shadow_trace4 / inner_nodes4 (realtimeraytracer_amd/csrc/kernels/rtr_kernels.hip):
 * per visit the four slab tests on half-float planes about the scene's wide centre (fma(plane, ga, gbc) per plane; the kernel's
 * per-octant forms give the same bits as the min / max form here because the fma is monotone in the plane), descend into
 * the BEST child that is hit (the farthest exit by default, the nearest entry in rounds 1-4: `order` below; strict comparison, so ties go to
 * the lower slot), push the other hit children in slot order (skipping a
 * code equal to the one entered, as the kernel's `c != next` does), test a leaf's triangles in storage order until one hits;
 * RTR_WIDE_STACK (16) stack entries, beyond which the ray is redone over the BVH2 as k_shadow_tail does. */
/* firstLeaf != 0 (the leaf code of the triangle the ray starts on, for a ray that leaves its surface point INTO the surface: dot(normal,
 * direction) < 0): that leaf's triangles are tested first, the walk from the root follows if none of them stops the ray — the ray nearly
 * always re-enters the triangle it starts 0.01 above, and an any-hit answer does not depend on the order triangles are met in. */
Hit trace_wide(const Scene& sc, rtr_v3 o, rtr_v3 d, float tmin, float tmax, Counters& c, int32_t firstLeaf = 0) {
    const RtrWideNode* nodes = sc.s->wide;
    const RtrBvhTri* tris = sc.s->tris;
    Hit best{}; best.hit = false; best.t = tmax;
    rtr_v3 idir = rtr_mk(rtr_safe_rcp_dir(d.x), rtr_safe_rcp_dir(d.y), rtr_safe_rcp_dir(d.z));
    rtr_v3 ga, gb;
    rtr_ray_grid_centre(o, idir, sc.s->grid.origin, sc.s->grid.scale, sc.s->grid.wideCentreXY, sc.s->grid.wideCentreZ, &ga, &gb);
    std::vector<int32_t> stack;
    int32_t cur = 0;
    if (firstLeaf < 0) { stack.push_back(0); cur = firstLeaf; }
    /* which hit child is entered first: the product's rule (order 0) is the one whose EXIT distance (clamped to the ray's far limit) is the
     * GREATEST — strict >, ties to the lower slot; the others are stacked in slot order.  A shadow ray's occluder sits, more often than not,
     * towards the light's end of the ray (two thirds of the occluded area-light rays of the bench frame are stopped in the last 40 % of their
     * length), and any-hit does not care which occluder is found.  Orders 1-3 (experiments; 1 = rounds 1-4): nearest entry, farthest entry,
     * nearest exit. */
    const uint32_t order = (sc.shadowWalk >> 2) & 3u;
    const bool slotOrder = (sc.shadowWalk & 1u) != 0u, farFirst = order == 0u || order == 2u, byExit = order == 0u || order == 3u;
    uint64_t visits = 0, tests = 0;            /* of this ray, for the split by its answer (oracle_walk_stats) */
    auto done = [&](bool occluded) {
        c.walk[occluded ? 0 : 3]++; c.walk[occluded ? 1 : 4] += visits; c.walk[occluded ? 2 : 5] += tests; if (occluded && firstLeaf < 0 && visits == 0) c.walk[7]++;
        if (sc.s->walkRays && sc.s->walkRaysCap) {      /* experiments only: one record per shadow ray */
            const uint64_t at = __atomic_fetch_add(sc.s->walkRaysCount, 1ull, __ATOMIC_RELAXED);
            if (at < sc.s->walkRaysCap) { float* r = sc.s->walkRays + 8 * at; r[0] = (float)visits; r[1] = (float)tests; r[2] = occluded ? 1.f : 0.f; r[3] = firstLeaf < 0 ? 1.f : 0.f; r[4] = occluded ? best.t : -1.f; r[5] = tmax; r[6] = d.y; r[7] = o.y; }
        }
    };
    /* walk profile (experiments only, oracle.h): per (record, slot) {entries, work done below it, occluders found below it} */
    uint64_t* const prof = sc.s->walkProfile;
    auto credit = [&](uint32_t rec, uint32_t slot, int what, uint64_t n) {       /* (rec, slot) and every slot above it */
        for (;;) {
            __atomic_fetch_add(&prof[((size_t)rec * 4u + slot) * 3u + (size_t)what], n, __ATOMIC_RELAXED);
            if (rec == 0u) break;
            const uint32_t up = sc.wideParent[rec]; rec = up >> 2; slot = up & 3u;
        }
    };
    std::vector<uint32_t> from;                                                   /* parallel to `stack` */
    uint32_t curFrom = 0;
    if (prof && firstLeaf < 0) from.push_back(0u);
    for (;;) {
        if (cur >= 0) {
            const RtrWideNode& n = nodes[cur];
            c.nodes++; c.shadowNodes++; visits++;
            int hit[4]; float te[4]; float tx[4];
            for (int k = 0; k < 4; ++k) {
                const uint32_t wmin = n.plane[k][0], wmax = n.plane[k][1], wz = n.plane[k][2];
                const float x0 = rtr_fma(half_bits_to_float(wmin & 0xffffu), ga.x, gb.x), x1 = rtr_fma(half_bits_to_float(wmax & 0xffffu), ga.x, gb.x);
                const float y0 = rtr_fma(half_bits_to_float(wmin >> 16), ga.y, gb.y), y1 = rtr_fma(half_bits_to_float(wmax >> 16), ga.y, gb.y);
                const float z0 = rtr_fma(half_bits_to_float(wz & 0xffffu), ga.z, gb.z), z1 = rtr_fma(half_bits_to_float(wz >> 16), ga.z, gb.z);
                const float lo = rtr_hwmax(rtr_hwmax(rtr_hwmin(x0, x1), rtr_hwmin(y0, y1)), rtr_hwmax(rtr_hwmin(z0, z1), tmin));
                const float hi = rtr_hwmin(rtr_hwmin(rtr_hwmax(x0, x1), rtr_hwmax(y0, y1)), rtr_hwmin(rtr_hwmax(z0, z1), tmax));
                te[k] = byExit ? hi : lo; tx[k] = hi;
                hit[k] = lo <= hi * RTR_BOX_WIDEN;
                if (k >= 2 && n.child[k] == RTR_WIDE_EMPTY) hit[k] = 0;
            }
            (void)tx;
            int nextSlot = -1;
            if (slotOrder) {
                /* the any-hit order: the record's slots were put in the order they should be tried when it was made (the builder's
                 * estimate of where an occluder is met soonest); the first hit slot is entered, the others are stacked so that they
                 * pop in slot order — no distances are compared */
                for (int k = 0; k < 4; ++k) if (hit[k]) { nextSlot = k; break; }
                for (int k = 3; k >= 0; --k) if (hit[k] && k != nextSlot) { stack.push_back(n.child[k]); if (prof) from.push_back((uint32_t)cur * 4u + (uint32_t)k); }
            } else if (farFirst) {
                /* the farthest hit child first (strict >, ties to the lower slot): a shadow ray's occluder sits, more often than not, near the
                 * light's end of the ray */
                float tf = -3.0e38f;
                if (hit[0]) { tf = te[0]; nextSlot = 0; }
                for (int k = 1; k < 4; ++k) if (hit[k] && (nextSlot < 0 || te[k] > tf)) { tf = te[k]; nextSlot = k; }
                if (sc.shadowWalk & 16u) {      /* experiment: the others sorted too, so that the farthest of them pops first */
                    int idx[4], m = 0;
                    for (int k = 0; k < 4; ++k) if (hit[k] && k != nextSlot) idx[m++] = k;
                    for (int a = 0; a < m; ++a) for (int b = a + 1; b < m; ++b) if (te[idx[b]] < te[idx[a]]) { int t2 = idx[a]; idx[a] = idx[b]; idx[b] = t2; }      /* ascending: the last pushed = the farthest */
                    for (int a = 0; a < m; ++a) { stack.push_back(n.child[idx[a]]); if (prof) from.push_back((uint32_t)cur * 4u + (uint32_t)idx[a]); }
                } else
                for (int k = 0; k < 4; ++k) if (hit[k] && k != nextSlot) { stack.push_back(n.child[k]); if (prof) from.push_back((uint32_t)cur * 4u + (uint32_t)k); }
            } else {
                float tn = 3.0e38f;
                if (hit[0]) { tn = te[0]; nextSlot = 0; }
                for (int k = 1; k < 4; ++k) if (hit[k] && te[k] < tn) { tn = te[k]; nextSlot = k; }
                for (int k = 0; k < 4; ++k) if (hit[k] && k != nextSlot) { stack.push_back(n.child[k]); if (prof) from.push_back((uint32_t)cur * 4u + (uint32_t)k); }
            }
            if (prof && cur != 0) credit(curFrom >> 2, curFrom & 3u, 1, 1);       /* this visit is work below the slot that led here */
            /* the kernel keeps RTR_WIDE_STACK stack entries in LDS; a ray that would hold more
             * after a visit is abandoned there and re-traced from scratch over the BVH2 by k_shadow_tail (counting form: both parts
             * are counted) */
            if (stack.size() > RTR_WIDE_STACK) {
                c.shadowOverflow++;
                const uint64_t n0 = c.shadowNodes, t0 = c.shadowTris;
                const Hit h = trace_bvh(sc, o, d, tmin, tmax, true, c);
                visits += c.shadowNodes - n0; tests += c.shadowTris - t0;
                done(h.hit);
                return h;
            }
            int32_t next;
            if (nextSlot < 0) {
                if (stack.empty()) { done(false); return best; }
                next = stack.back(); stack.pop_back();
                if (prof) { curFrom = from.back(); from.pop_back(); }
            } else { next = n.child[nextSlot]; curFrom = (uint32_t)cur * 4u + (uint32_t)nextSlot; }
            if (prof) __atomic_fetch_add(&prof[(size_t)curFrom * 3u], 1ull, __ATOMIC_RELAXED);      /* an entry of that slot */
            cur = next;
        } else {
            const uint32_t code = (uint32_t)~cur;
            const uint32_t first = code >> 3, count = (code & 7u) + 1u;
            for (uint32_t i = 0; i < count; ++i) {
                const RtrBvhTri& tr = tris[first + i];
                float t, u, v;
                c.tris++; c.shadowTris++; tests++;
                if (prof) credit(curFrom >> 2, curFrom & 3u, 1, 1);
                if (rtr_mt_intersect(o, d, rtr_ld3(tr.v0), rtr_ld3(tr.e1), rtr_ld3(tr.e2), tmin, &t, &u, &v)) {
                    if (!(t < tmax)) continue;
                    if ((tr.flags & 1u) && !alpha_pass(sc.s->desc, tr.customIndex, tr.primitiveId, u, v, c)) continue;
                    best.hit = true; best.t = t; best.u = u; best.v = v; best.custom = tr.customIndex; best.prim = tr.primitiveId;
                    if (prof) credit(curFrom >> 2, curFrom & 3u, 2, 1);
                    done(true);
                    return best;
                }
            }
            if (stack.empty()) { done(false); return best; }
            cur = stack.back(); stack.pop_back();
            if (prof) { curFrom = from.back(); from.pop_back(); __atomic_fetch_add(&prof[(size_t)curFrom * 3u], 1ull, __ATOMIC_RELAXED); }
        }
    }
}

/* Closest-hit walk of the 4-wide view, restating k_primary4: trace_wide's rule (nearest hit child first — strict <, ties to the lower
 * slot — the others stacked in slot order) with the slab tests against the best t so far and ALL triangles of a leaf tested; 16 stack
 * entries, beyond which the ray is abandoned and walked again from scratch over the BVH2 (k_primary_tail; both parts counted). */
Hit trace_wide_closest(const Scene& sc, rtr_v3 o, rtr_v3 d, float tmin, float tmax, Counters& c) {
    const RtrWideNode* nodes = sc.s->wide;
    const RtrBvhTri* tris = sc.s->tris;
    Hit best{}; best.hit = false; best.t = tmax; best.custom = best.prim = 0xffffffffu;
    rtr_v3 idir = rtr_mk(rtr_safe_rcp_dir(d.x), rtr_safe_rcp_dir(d.y), rtr_safe_rcp_dir(d.z));
    rtr_v3 ga, gb;
    rtr_ray_grid_centre(o, idir, sc.s->grid.origin, sc.s->grid.scale, sc.s->grid.wideCentreXY, sc.s->grid.wideCentreZ, &ga, &gb);
    int32_t stack[RTR_WIDE_STACK];
    int sp = 0;
    int32_t cur = 0;
    for (;;) {
        if (cur >= 0) {
            const RtrWideNode& n = nodes[cur];
            c.nodes++;
            int hit[4]; float te[4];
            const float lim = best.t;
            for (int k = 0; k < 4; ++k) {
                const uint32_t wmin = n.plane[k][0], wmax = n.plane[k][1], wz = n.plane[k][2];
                const float x0 = rtr_fma(half_bits_to_float(wmin & 0xffffu), ga.x, gb.x), x1 = rtr_fma(half_bits_to_float(wmax & 0xffffu), ga.x, gb.x);
                const float y0 = rtr_fma(half_bits_to_float(wmin >> 16), ga.y, gb.y), y1 = rtr_fma(half_bits_to_float(wmax >> 16), ga.y, gb.y);
                const float z0 = rtr_fma(half_bits_to_float(wz & 0xffffu), ga.z, gb.z), z1 = rtr_fma(half_bits_to_float(wz >> 16), ga.z, gb.z);
                const float lo = rtr_hwmax(rtr_hwmax(rtr_hwmin(x0, x1), rtr_hwmin(y0, y1)), rtr_hwmax(rtr_hwmin(z0, z1), tmin));
                const float hi = rtr_hwmin(rtr_hwmin(rtr_hwmax(x0, x1), rtr_hwmax(y0, y1)), rtr_hwmin(rtr_hwmax(z0, z1), lim));
                te[k] = lo;
                hit[k] = lo <= hi * RTR_BOX_WIDEN;
                if (k >= 2 && n.child[k] == RTR_WIDE_EMPTY) hit[k] = 0;
            }
            int nextSlot = -1;
            float tn = 3.0e38f;
            if (hit[0]) { tn = te[0]; nextSlot = 0; }
            for (int k = 1; k < 4; ++k) if (hit[k] && te[k] < tn) { tn = te[k]; nextSlot = k; }
            bool overflow = false;
            for (int k = 0; k < 4; ++k) if (hit[k] && k != nextSlot) { if (sp < RTR_WIDE_STACK) stack[sp++] = n.child[k]; else overflow = true; }
            if (overflow) { c.primaryOverflow++; return trace_bvh(sc, o, d, tmin, tmax, false, c, 0); }
            if (nextSlot >= 0) { cur = n.child[nextSlot]; continue; }
        } else {
            const uint32_t code = (uint32_t)~cur;
            const uint32_t first = code >> 3, count = (code & 7u) + 1u;
            for (uint32_t i = 0; i < count; ++i) {
                const RtrBvhTri& tr = tris[first + i];
                float t, u, v;
                c.tris++;
                if (rtr_mt_intersect(o, d, rtr_ld3(tr.v0), rtr_ld3(tr.e1), rtr_ld3(tr.e2), tmin, &t, &u, &v)) {
                    if (!(t < tmax)) continue;
                    if ((tr.flags & 1u) && !alpha_pass(sc.s->desc, tr.customIndex, tr.primitiveId, u, v, c)) continue;
                    if (t < best.t || (t == best.t && id_less(tr.customIndex, tr.primitiveId, best.custom, best.prim))) {
                        best.hit = true; best.t = t; best.u = u; best.v = v; best.custom = tr.customIndex; best.prim = tr.primitiveId; best.leaf = cur;
                    }
                }
            }
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
    return best;
}

inline Hit trace(const Scene& sc, rtr_v3 o, rtr_v3 d, float tmin, float tmax, bool anyHit, Counters& c, int32_t firstLeaf = 0) {
    c.rays++;
    if (anyHit) c.shadow++; else c.primary++;
    if (!(tmax > tmin)) { Hit h{}; h.hit = false; return h; }
    if (anyHit && sc.useWide) return trace_wide(sc, o, d, tmin, tmax, c, firstLeaf);
    if (!anyHit && sc.primaryWide) return trace_wide_closest(sc, o, d, tmin, tmax, c);
    return sc.s->nodes ? trace_bvh(sc, o, d, tmin, tmax, anyHit, c, anyHit ? 0 : sc.primaryStackLimit) : trace_brute(sc, o, d, tmin, tmax, anyHit, c);
}

/* ---- cook-torrance.glsl ----------------------------------------------------------------------- */
const float PI_F = 3.14159265359f;   /* cook-torrance.glsl:1 */

inline float chiGGX(float v) { return v > 0.0f ? 1.0f : 0.0f; }                       /* :3-6 */
inline float GGX_Distribution(rtr_v3 n, rtr_v3 h, float alpha) {                       /* :11-18 */
    float NoH = rtr_dot(n, h);
    float alpha2 = alpha * alpha;
    float NoH2 = NoH * NoH;
    float den = rtr_max(rtr_fma(NoH2, alpha2, 1.0f - NoH2), 0.001f);
    return (chiGGX(NoH) * alpha2) / (PI_F * den * den);
}
inline float GGX_PartialGeometryTerm(rtr_v3 v, rtr_v3 n, rtr_v3 h, float alpha) {      /* :43-50 */
    float VoH2 = rtr_clamp(rtr_dot(v, h), 0.001f, 1.0f);
    float chi = chiGGX(VoH2 / rtr_clamp(rtr_dot(v, n), 0.001f, 1.0f));
    VoH2 = VoH2 * VoH2;
    float tan2 = (1.0f - VoH2) / VoH2;
    return (chi * 2.0f) / (1.0f + rtr_sqrt(rtr_fma(alpha * alpha, tan2, 1.0f)));
}
inline rtr_v3 Fresnel_Schlick(float cosT, rtr_v3 F0) {                                  /* :58-61 */
    float p = rtr_pow(1.0f - cosT, 5.0f);
    return rtr_mk(rtr_fma(1.0f - F0.x, p, F0.x), rtr_fma(1.0f - F0.y, p, F0.y), rtr_fma(1.0f - F0.z, p, F0.z));
}

/* ---- LTC.glsl --------------------------------------------------------------------------------- */
const float LUT_SIZE = 64.0f;                               /* raygen.rgen:65-67 */
const float LUT_SCALE = (LUT_SIZE - 1.0f) / LUT_SIZE;
const float LUT_BIAS = 0.5f / LUT_SIZE;

/* texture() on a 64x64 RGBA32F image with the reference's sampler: linear filter, repeat
 * addressing, single mip (src/vulkan/memory/image_sampler.cppm:26-42). */
inline void sample_lut(const float* lut, float u, float v, float out[4]) {
    float x = rtr_fma(u, LUT_SIZE, -0.5f), y = rtr_fma(v, LUT_SIZE, -0.5f);
    if (!(x >= -1.0e6f && x <= 1.0e6f)) x = 0.0f;      /* NaN / huge coordinates sample texel (0,0) */
    if (!(y >= -1.0e6f && y <= 1.0e6f)) y = 0.0f;
    float x0f = __builtin_floorf(x), y0f = __builtin_floorf(y);
    float fx = x - x0f, fy = y - y0f;
    int x0 = ((int)x0f) & 63, y0 = ((int)y0f) & 63;
    int x1 = (x0 + 1) & 63, y1 = (y0 + 1) & 63;
    const float* t00 = lut + (y0 * 64 + x0) * 4;
    const float* t10 = lut + (y0 * 64 + x1) * 4;
    const float* t01 = lut + (y1 * 64 + x0) * 4;
    const float* t11 = lut + (y1 * 64 + x1) * 4;
    for (int k = 0; k < 4; ++k) {
        float a = rtr_fma(t10[k] - t00[k], fx, t00[k]);
        float b = rtr_fma(t11[k] - t01[k], fx, t01[k]);
        out[k] = rtr_fma(b - a, fy, a);
    }
}

inline rtr_v3 IntegrateEdgeVec(rtr_v3 v1, rtr_v3 v2) {                                  /* LTC.glsl:2-14 */
    float x = rtr_dot(v1, v2);
    float y = rtr_abs(x);
    float a = rtr_fma(rtr_fma(0.0145206f, y, 0.4965155f), y, 0.8543985f);
    float b = rtr_fma(4.1616724f + y, y, 3.4175940f);
    float v = a / b;
    float theta_sintheta = (x > 0.0f) ? v : 0.5f * (1.0f / rtr_sqrt(rtr_max(rtr_fma(-x, x, 1.0f), 1e-7f))) - v;
    return rtr_scale(rtr_cross(v1, v2), theta_sintheta);
}

/* Minv given as the 4 LUT parameters (t1) of raygen.rgen:153-157, or identity.
 * GLSL mat3(c0,c1,c2) is column-major: Minv = [ t1.x 0 t1.z ; 0 1 0 ; t1.y 0 t1.w ] as rows. */
inline float LTC_Evaluate(rtr_v3 N, rtr_v3 V, rtr_v3 P, bool identity, const float t1[4],
                          const rtr_v3 points[3], rtr_v3 lightNormal, bool twoSided, const float* ltc2) {
    /* LTC.glsl:19-22 */
    rtr_v3 T1 = rtr_normalize(rtr_sub(V, rtr_scale(N, rtr_dot(V, N))));
    rtr_v3 T2 = rtr_cross(N, T1);
    /* :25  Minv * transpose(mat3(T1,T2,N)) applied to a vector w:
     *       first q = (dot(T1,w), dot(T2,w), dot(N,w)), then Minv * q */
    rtr_v3 L[3];
    for (int k = 0; k < 3; ++k) {
        rtr_v3 w = rtr_sub(points[k], P);
        rtr_v3 q = rtr_mk(rtr_dot(T1, w), rtr_dot(T2, w), rtr_dot(N, w));
        if (identity) L[k] = q;
        else L[k] = rtr_mk(rtr_fma(t1[2], q.z, t1[0] * q.x), q.y, rtr_fma(t1[3], q.z, t1[1] * q.x));
        L[k] = rtr_normalize(L[k]);                                                     /* :41-43 */
    }
    rtr_v3 dir = rtr_sub(points[0], P);                                                 /* :36-38 */
    bool behind = rtr_dot(dir, lightNormal) < 0.0f;
    rtr_v3 vsum = IntegrateEdgeVec(L[0], L[1]);                                         /* :46-49 */
    vsum = rtr_add(vsum, IntegrateEdgeVec(L[1], L[2]));
    vsum = rtr_add(vsum, IntegrateEdgeVec(L[2], L[0]));
    float len = rtr_length(vsum);                                                       /* :52 */
    float z = vsum.z / len;
    if (behind) z = -z;
    float uvx = rtr_fma(rtr_fma(z, 0.5f, 0.5f), LUT_SCALE, LUT_BIAS);                   /* :58-59 */
    float uvy = rtr_fma(len, LUT_SCALE, LUT_BIAS);
    float tex[4];
    sample_lut(ltc2, uvx, uvy, tex);                                                    /* :62 */
    float sum = len * tex[3];
    if (!behind && !twoSided) sum = 0.0f;                                               /* :65-66 */
    return sum;
}

/* ---- closesthit.rchit:45-110 ------------------------------------------------------------------ */
struct Surface { rtr_v3 hitPoint, normal, color; float roughness, metallic; };

inline Surface closest_hit_shader(const Scene& sc, const Hit& h, rtr_v3 rayDir, Counters& c) {
    const rtr_scene_desc& D = sc.s->desc;
    Surface sf;
    c.hits++;
    uint32_t objIndex = h.custom - D.numLights;                                          /* :53 */
    const RtrObjectInfo& oi = D.objects[objIndex];
    uint32_t i0 = D.indices[3 * h.prim + 0 + oi.indexOffset];                            /* :59-61 */
    uint32_t i1 = D.indices[3 * h.prim + 1 + oi.indexOffset];
    uint32_t i2 = D.indices[3 * h.prim + 2 + oi.indexOffset];
    const RtrVertex& v0 = D.vertices[i0 + oi.vertexOffset];                              /* :63-65 */
    const RtrVertex& v1 = D.vertices[i1 + oi.vertexOffset];
    const RtrVertex& v2 = D.vertices[i2 + oi.vertexOffset];
    float b0 = 1.0f - h.u - h.v, b1 = h.u, b2 = h.v;                                     /* :71 */
    rtr_v3 p0 = rtr_ld3(v0.position), p1 = rtr_ld3(v1.position), p2 = rtr_ld3(v2.position);
    rtr_v3 localPos = rtr_madd(rtr_madd(rtr_scale(p0, b0), p1, b1), p2, b2);             /* :72 */
    const RtrInstance* inst = sc.byCustom[h.custom];
    sf.hitPoint = rtr_xform_point34(inst->transform, localPos);                          /* :73 */
    rtr_v3 n0 = rtr_ld3(v0.normal), n1 = rtr_ld3(v1.normal), n2 = rtr_ld3(v2.normal);
    rtr_v3 nsum = rtr_madd(rtr_madd(rtr_scale(n0, b0), n1, b1), n2, b2);                 /* :74 */
    const float* nm = &sc.normalMat[9 * (size_t)h.custom];
    if (rtr_dot(nsum, nsum) > 0.0f) {
        sf.normal = rtr_normalize(rtr_mul33(nm, rtr_normalize(nsum)));                   /* :74-76 */
    } else {
        /* D1: geometric normal of the local triangle, through the normal matrix, facing the ray */
        rtr_v3 g = rtr_cross(rtr_sub(p1, p0), rtr_sub(p2, p0));
        rtr_v3 n = rtr_normalize(rtr_mul33(nm, rtr_normalize(g)));
        if (rtr_dot(n, rayDir) > 0.0f) n = rtr_neg(n);
        sf.normal = n;
    }
    /* :77 uv, :79-101 materials from constants or texSamplers[] */
    const float uu = rtr_fma(v2.uv[0], b2, rtr_fma(v1.uv[0], b1, v0.uv[0] * b0));
    const float vv = rtr_fma(v2.uv[1], b2, rtr_fma(v1.uv[1], b1, v0.uv[1] * b0));
    rtr_v3 col = rtr_ld3(oi.color);
    float rough = oi.specular;
    sf.metallic = oi.metallic;
    float tex[4];
    if (oi.usesColorMap != 0) { sample_tex(D.textures[oi.colorIndex], uu, vv, tex, c); col = rtr_mk(tex[0], tex[1], tex[2]); }
    if (oi.usesSpecularMap != 0) { sample_tex(D.textures[oi.specularIndex], uu, vv, tex, c); rough = tex[0]; }
    if (oi.usesMetallicMap != 0) { sample_tex(D.textures[oi.metallicIndex], uu, vv, tex, c); sf.metallic = tex[0]; }
    sf.color = rtr_mk(rtr_to_linear(col.x), rtr_to_linear(col.y), rtr_to_linear(col.z)); /* :104 */
    sf.roughness = 1.0f - rough;                                                         /* :106 */
    return sf;
}

/* ---- raygen.rgen:71-366 for one pixel --------------------------------------------------------- */
struct PixelOut { rtr_v3 analytic, shadowed, unshadowed, avgNormal, avgPosition; };

/* preHits: the pixel's camera-ray hits (one per sample) when they were found tile by tile (trace_packet), else null */
PixelOut shade_pixel(const Scene& sc, const RtrCameraData& cam, const RtrSceneInfo& info,
                     const rtr_render_params& prm, uint32_t px, uint32_t py, bool wantAnalytic, bool wantUnshadowed, Counters& c, const Hit* preHits = nullptr) {
    const rtr_scene_desc& D = sc.s->desc;
    (void)wantUnshadowed;          /* every sum is evaluated for every sample, as the shader does; the flag only says which are stored */
    PixelOut o;
    o.analytic = o.shadowed = o.unshadowed = o.avgNormal = o.avgPosition = rtr_mk(0, 0, 0);
    const rtr_v3 camPos = rtr_ld3(cam.position);
    const rtr_v3 TL = rtr_ld3(cam.topLeftViewportCorner);
    const rtr_v3 dH = rtr_ld3(cam.horizontalViewportDelta);
    const rtr_v3 dV = rtr_ld3(cam.verticalViewportDelta);

    for (uint32_t i = 0; i < prm.spp; ++i) {                                             /* :81 */
        /* :83 int(pixelCoord + i) is the x coordinate only (quirk Q1) */
        float jx = rtr_random(px + i), jy = rtr_random(px + i * 322u);
        float offx = ((float)px + jx) - 0.5f, offy = ((float)py + jy) - 0.5f;            /* :84 */
        rtr_v3 pw = rtr_madd(rtr_madd(TL, dH, offx), dV, offy);                          /* :86-89 */
        rtr_v3 rayDir = rtr_normalize(rtr_sub(pw, camPos));                              /* :91-92 */
        Hit h = preHits ? preHits[i] : trace(sc, camPos, rayDir, 0.001f, 10000.0f, false, c);   /* :99-107 */
        if (!h.hit) {                                                                    /* :110-115, miss.rmiss:15-27 */
            rtr_v3 sky = sc.skyLinear;
            if (D.hdri) {
                const rtr_v3 dir = rtr_normalize(rayDir);                                /* miss.rmiss:19 */
                const float hu = rtr_atan2(dir.z, dir.x) / (2.0f * 3.14159265f) + 0.5f;  /* :20 */
                float hv = rtr_acos(rtr_clamp(dir.y, -1.0f, 1.0f)) / 3.14159265f;        /* :21 */
                hv = 1.0f - hv;                                                          /* :22 */
                float tex[4];
                sample_tex(*D.hdri, hu, hv, tex, c);                                     /* :23 */
                sky = rtr_mk(rtr_to_linear(tex[0]), rtr_to_linear(tex[1]), rtr_to_linear(tex[2]));   /* :24 */
            }
            o.analytic = rtr_add(o.analytic, sky);
            o.unshadowed = rtr_add(o.unshadowed, sky);
            o.shadowed = rtr_add(o.shadowed, sky);
            continue;
        }
        if (h.custom < D.numLights) {                                                    /* :116-121, closesthit.rchit:46-50 */
            rtr_v3 lc = rtr_ld3(D.lights[h.custom].color);
            o.analytic = rtr_add(o.analytic, lc);
            o.unshadowed = rtr_add(o.unshadowed, lc);
            o.shadowed = rtr_add(o.shadowed, lc);
            continue;
        }
        Surface sf = closest_hit_shader(sc, h, rayDir, c);
        rtr_v3 hitPoint = sf.hitPoint, hitNormal = sf.normal, color = sf.color;          /* :125-130 */
        rtr_v3 viewDir = rtr_normalize(rtr_sub(camPos, hitPoint));
        float roughness = sf.roughness, metallic = sf.metallic;
        o.avgNormal = rtr_add(o.avgNormal, hitNormal);                                   /* :132-133 */
        o.avgPosition = rtr_add(o.avgPosition, hitPoint);
        float om = 1.0f - metallic;
        rtr_v3 mDiffuse = rtr_scale(color, om);                                          /* :135 */
        rtr_v3 mSpecular = rtr_mk(rtr_fma(color.x, metallic, 0.04f * om),                /* :136 mix(vec3(.04),color,metallic) */
                                  rtr_fma(color.y, metallic, 0.04f * om),
                                  rtr_fma(color.z, metallic, 0.04f * om));
        float dotNV = rtr_clamp(rtr_dot(hitNormal, viewDir), 0.0f, 1.0f);                /* :140 */
        float t1[4] = {1, 0, 0, 1}, t2[4] = {0, 0, 0, 0};
        if (wantAnalytic) {                                                              /* :144-151 */
            float lu = rtr_fma(roughness, LUT_SCALE, LUT_BIAS);
            float lv = rtr_fma(rtr_sqrt(1.0f - dotNV), LUT_SCALE, LUT_BIAS);
            sample_lut(D.ltc1, lu, lv, t1);
            sample_lut(D.ltc2, lu, lv, t2);
        }
        rtr_v3 shadowOrigin = rtr_madd(hitPoint, hitNormal, 0.01f);                      /* :224, :297 */

        for (uint32_t li = 0; li < info.numAreaLights; ++li) {                           /* :165 */
            const RtrAreaLightInfo& L = D.lights[li];
            c.lightFetch++;
            rtr_v3 lcol = rtr_ld3(L.color);
            for (uint32_t ti = 0; ti < L.numTriangles; ++ti) {                           /* :172 */
                c.lightTriFetch++;
                uint32_t i0 = D.indices[ti * 3 + 0 + L.indexOffset];                     /* :176-182 */
                uint32_t i1 = D.indices[ti * 3 + 1 + L.indexOffset];
                uint32_t i2 = D.indices[ti * 3 + 2 + L.indexOffset];
                rtr_v3 P[3];
                P[0] = rtr_xform_point44cm(L.transform, rtr_ld3(D.vertices[i0 + L.vertexOffset].position)); /* :184-189 */
                P[1] = rtr_xform_point44cm(L.transform, rtr_ld3(D.vertices[i1 + L.vertexOffset].position));
                P[2] = rtr_xform_point44cm(L.transform, rtr_ld3(D.vertices[i2 + L.vertexOffset].position));
                rtr_v3 lightNormal = rtr_cross(rtr_sub(P[2], P[1]), rtr_sub(P[0], P[1])); /* :192 */
                float area = rtr_length(lightNormal) * 0.5f;                              /* :193 */
                float pdf = 1.0f / (area * 0.7f);                                         /* :194 */
                lightNormal = rtr_normalize(lightNormal);                                 /* :195 */
                if (!L.isTwoSided) {                                                      /* :197-201 */
                    if (rtr_dot(lightNormal, rtr_sub(hitPoint, P[0])) < 0.0f) continue;
                }
                rtr_v3 shadowedSample = rtr_mk(0, 0, 0), unshadowedSample = rtr_mk(0, 0, 0);
                for (uint32_t s = 0; s < prm.numShadowRays; ++s) {                        /* :206 */
                    uint32_t seed = s + px * 733u + py * 1933u + info.frame;              /* :209 */
                    float r1 = rtr_random(seed), r2 = rtr_random(seed + 100u);            /* :210-211 */
                    if (r1 + r2 > 1.0f) { r1 = 1.0f - r1; r2 = 1.0f - r2; }               /* :215-218 */
                    rtr_v3 lightSamplePos = rtr_madd(rtr_madd(P[0], rtr_sub(P[1], P[0]), r1), rtr_sub(P[2], P[0]), r2); /* :219 */
                    rtr_v3 lightVec = rtr_sub(lightSamplePos, hitPoint);
                    rtr_v3 sampledLightDir = rtr_normalize(lightVec);                     /* :221 */
                    float lightDistance = rtr_length(lightVec);                           /* :222 */
                    const int32_t ownLeaf = (sc.ownLeafFirst && rtr_dot(hitNormal, lightVec) < 0.0f) ? h.leaf : 0;      /* the un-normalised direction: what the queue build's count pass has */
                    if (ownLeaf) c.walk[6]++;
                    Hit sh = trace(sc, shadowOrigin, sampledLightDir, 0.001f, lightDistance - 0.5f, true, c, ownLeaf); /* :226-241 */
                    float currShadow = sh.hit ? 0.0f : 1.0f;                              /* :244 */
                    /* every sample's BRDF is evaluated and multiplied by currShadow, as the shader does (the product skips the
                     * BRDF of an occluded sample when only the shadowed image is kept: same bits whenever contrib is finite,
                     * divergence D6 otherwise) */
                    rtr_v3 halfVector = rtr_normalize(rtr_add(viewDir, sampledLightDir)); /* :247 */
                    float cosTheta = rtr_clamp(rtr_dot(viewDir, halfVector), 0.0f, 1.0f); /* :250 */
                    float Dg = GGX_Distribution(hitNormal, halfVector, roughness);        /* :252 */
                    float G = GGX_PartialGeometryTerm(viewDir, hitNormal, halfVector, roughness) *
                              GGX_PartialGeometryTerm(sampledLightDir, hitNormal, halfVector, roughness); /* :253 */
                    rtr_v3 F = Fresnel_Schlick(cosTheta, mSpecular);                      /* :254 */
                    float NdotV = rtr_max(rtr_dot(hitNormal, viewDir), 0.1f);             /* :256 */
                    float NdotL = rtr_max(rtr_dot(hitNormal, sampledLightDir), 0.1f);     /* :257 */
                    float den = 4.0f * NdotV * NdotL;                                     /* :259: (D * F * G) / (4.0 * NdotV * NdotL), left to right */
                    rtr_v3 currSpecular = rtr_mk(((Dg * F.x) * G) / den, ((Dg * F.y) * G) / den, ((Dg * F.z) * G) / den);
                    rtr_v3 currDiffuse = rtr_mk((om * color.x) / PI_F, (om * color.y) / PI_F, (om * color.z) / PI_F); /* :260 */
                    float attenuation = 1.0f / (lightDistance * lightDistance);           /* :262-264 */
                    rtr_v3 BRDF = rtr_add(currSpecular, currDiffuse);                     /* :266 */
                    /* :267 currLight.color * currLight.intensity * NdotL * attenuation * 10.0, left to right */
                    rtr_v3 Lr = rtr_mk((((lcol.x * L.intensity) * NdotL) * attenuation) * 10.0f, (((lcol.y * L.intensity) * NdotL) * attenuation) * 10.0f,
                                       (((lcol.z * L.intensity) * NdotL) * attenuation) * 10.0f);
                    /* :269 currShadow * BRDF * L / pdf and :270 BRDF * L / pdf, left to right */
                    shadowedSample = rtr_add(shadowedSample, rtr_mk((((currShadow * BRDF.x) * Lr.x) / pdf), (((currShadow * BRDF.y) * Lr.y) / pdf),
                                                                    (((currShadow * BRDF.z) * Lr.z) / pdf)));
                    unshadowedSample = rtr_add(unshadowedSample, rtr_mk((BRDF.x * Lr.x) / pdf, (BRDF.y * Lr.y) / pdf, (BRDF.z * Lr.z) / pdf));
                }
                float ns = (float)prm.numShadowRays;                                      /* :272-273 */
                shadowedSample = rtr_mk(shadowedSample.x / ns, shadowedSample.y / ns, shadowedSample.z / ns);
                unshadowedSample = rtr_mk(unshadowedSample.x / ns, unshadowedSample.y / ns, unshadowedSample.z / ns);
                if (wantAnalytic) {                                                       /* :277-283 */
                    bool twoSided = L.isTwoSided != 0;
                    float diffuse = LTC_Evaluate(hitNormal, viewDir, hitPoint, true, t1, P, lightNormal, twoSided, D.ltc2);
                    float spec = LTC_Evaluate(hitNormal, viewDir, hitPoint, false, t1, P, lightNormal, twoSided, D.ltc2);
                    rtr_v3 fres = rtr_mk(rtr_fma(1.0f - mSpecular.x, t2[1], mSpecular.x * t2[0]),
                                         rtr_fma(1.0f - mSpecular.y, t2[1], mSpecular.y * t2[0]),
                                         rtr_fma(1.0f - mSpecular.z, t2[1], mSpecular.z * t2[0]));
                    /* :283 currLight.color * currLight.intensity * (specular + mDiffuse * diffuse) * 5.0, left to right */
                    o.analytic = rtr_add(o.analytic, rtr_mk(
                        ((lcol.x * L.intensity) * rtr_fma(mDiffuse.x, diffuse, spec * fres.x)) * 5.0f,
                        ((lcol.y * L.intensity) * rtr_fma(mDiffuse.y, diffuse, spec * fres.y)) * 5.0f,
                        ((lcol.z * L.intensity) * rtr_fma(mDiffuse.z, diffuse, spec * fres.z)) * 5.0f));
                }
                o.shadowed = rtr_add(o.shadowed, shadowedSample);                         /* :284-285 */
                o.unshadowed = rtr_add(o.unshadowed, unshadowedSample);
            }
        }
        /* :289-338 directional light */
        const rtr_v3 directLightDir = rtr_normalize(rtr_mk(-1.0f, 1.0f, -0.5f));
        const rtr_v3 directLightColor = rtr_mk(1.0f, 1.0f, 0.5f);
        const float directLightIntensity = 0.2f;
        if (rtr_dot(hitNormal, directLightDir) <= 0.0f) continue;                         /* :293 */
        Hit sh = trace(sc, shadowOrigin, directLightDir, 0.001f, 10000.0f, true, c);      /* :303-313 */
        float currShadow = sh.hit ? 0.0f : 1.0f;                                          /* :316 */
        rtr_v3 halfVector = rtr_normalize(rtr_add(viewDir, directLightDir));              /* :318 */
        float cosTheta = rtr_clamp(rtr_dot(viewDir, halfVector), 0.0f, 1.0f);             /* :321 */
        float Dg = GGX_Distribution(hitNormal, halfVector, roughness);                    /* :323 */
        float G = GGX_PartialGeometryTerm(viewDir, hitNormal, halfVector, roughness) *
                  GGX_PartialGeometryTerm(directLightDir, hitNormal, halfVector, roughness);
        rtr_v3 F = Fresnel_Schlick(cosTheta, mSpecular);
        float NdotV = rtr_max(rtr_dot(hitNormal, viewDir), 5.0f);                         /* :327 (quirk Q6) */
        float NdotL = rtr_max(rtr_dot(hitNormal, directLightDir), 0.0001f);               /* :328 */
        float den = 4.0f * NdotV * NdotL;                                                 /* :330 (D * F * G) / (4.0 * NdotV * NdotL), left to right */
        rtr_v3 currSpecular = rtr_mk(((Dg * F.x) * G) / den, ((Dg * F.y) * G) / den, ((Dg * F.z) * G) / den);
        rtr_v3 currDiffuse = rtr_mk((om * color.x) / PI_F, (om * color.y) / PI_F, (om * color.z) / PI_F);
        rtr_v3 BRDF = rtr_add(currSpecular, currDiffuse);
        /* :334 directLightColor * directLightIntensity * NdotL * 20.0, left to right */
        rtr_v3 Lr = rtr_mk(((directLightColor.x * directLightIntensity) * NdotL) * 20.0f, ((directLightColor.y * directLightIntensity) * NdotL) * 20.0f,
                           ((directLightColor.z * directLightIntensity) * NdotL) * 20.0f);
        /* :336-338 currShadow * BRDF * L, BRDF * L */
        o.shadowed = rtr_add(o.shadowed, rtr_mk((currShadow * BRDF.x) * Lr.x, (currShadow * BRDF.y) * Lr.y, (currShadow * BRDF.z) * Lr.z));
        o.unshadowed = rtr_add(o.unshadowed, rtr_mul(BRDF, Lr));
        o.analytic = rtr_add(o.analytic, rtr_mul(BRDF, Lr));
    }
    float n = (float)prm.spp;                                                             /* :341-343 */
    o.shadowed = rtr_mk(o.shadowed.x / n, o.shadowed.y / n, o.shadowed.z / n);
    o.unshadowed = rtr_mk(o.unshadowed.x / n, o.unshadowed.y / n, o.unshadowed.z / n);
    o.analytic = rtr_mk(o.analytic.x / n, o.analytic.y / n, o.analytic.z / n);
    return o;
}

inline uint32_t tonemap_pack(rtr_v3 c) {                                                  /* :345-357 */
    return rtr_pack_bgra8(rtr_to_srgb(rtr_aces(c.x)), rtr_to_srgb(rtr_aces(c.y)), rtr_to_srgb(rtr_aces(c.z)));
}

bool prepare(const oracle_scene* s, Scene& sc) {
    const rtr_scene_desc& D = s->desc;
    sc.s = s;
    sc.byCustom.assign(D.numInstances, nullptr);
    sc.normalMat.assign(9 * (size_t)D.numInstances, 0.0f);
    for (uint32_t i = 0; i < D.numInstances; ++i) {
        const RtrInstance& in = D.instances[i];
        if (in.customIndex >= D.numInstances || in.meshIndex >= D.numMeshes) return false;
        sc.byCustom[in.customIndex] = &in;
        rtr_normal_matrix(in.transform, &sc.normalMat[9 * (size_t)in.customIndex]);
    }
    for (auto p : sc.byCustom) if (!p) return false;
    sc.skyLinear = rtr_mk(rtr_to_linear(D.skyColor[0]), rtr_to_linear(D.skyColor[1]), rtr_to_linear(D.skyColor[2]));
    if (!s->nodes) {
        /* world-space triangle soup in (instance, primitive) order; same formulas as the product's
         * packer: world v = M * v_local, e1 = v1w - v0w, e2 = v2w - v0w */
        for (uint32_t i = 0; i < D.numInstances; ++i) {
            const RtrInstance& in = D.instances[i];
            const RtrMesh& m = D.meshes[in.meshIndex];
            for (uint32_t t = 0; t < m.indexCount / 3; ++t) {
                rtr_v3 w[3];
                for (int k = 0; k < 3; ++k) {
                    uint32_t idx = D.indices[m.indexOffset + 3 * t + k] + m.vertexOffset;
                    w[k] = rtr_xform_point34(in.transform, rtr_ld3(D.vertices[idx].position));
                }
                WorldTri wt;
                wt.v0 = w[0]; wt.e1 = rtr_sub(w[1], w[0]); wt.e2 = rtr_sub(w[2], w[0]);
                wt.custom = in.customIndex; wt.prim = t;
                wt.flags = (in.customIndex >= D.numLights && D.objects[in.customIndex - D.numLights].usesOpacityMap != 0 && m.isOpaque == 0) ? 1u : 0u;
                sc.brute.push_back(wt);
            }
        }
    }
    return true;
}

struct BandMap {
    uint32_t bandRows, shardIndex, shardCount, height;
    /* local row -> global y, or -1 when the (padded) local row has no pixel */
    int64_t global_y(uint32_t localRow) const {
        uint32_t lb = localRow / bandRows, r = localRow % bandRows;
        uint64_t y = ((uint64_t)lb * shardCount + shardIndex) * bandRows + r;
        return y < height ? (int64_t)y : -1;
    }
};

template <class F>
void parallel_rows(uint32_t rows, int threads, F&& fn) {
    if (threads <= 1) { for (uint32_t r = 0; r < rows; ++r) fn(r, 0); return; }
    std::atomic<uint32_t> next{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&, t]() { for (;;) { uint32_t r = next.fetch_add(1); if (r >= rows) break; fn(r, t); } });
    for (auto& th : pool) th.join();
}

}  // namespace

extern "C" {

static uint32_t shard_rows(uint32_t height, uint32_t bandRows, uint32_t shardCount) {
    if (shardCount <= 1) return height;
    uint32_t bands = (height + bandRows - 1) / bandRows;
    uint32_t per = (bands + shardCount - 1) / shardCount;
    return per * bandRows;
}

int oracle_render(const oracle_scene* s, const RtrCameraData* cam, const RtrSceneInfo* info,
                  const rtr_render_params* prmIn, oracle_out* out, int threads) {
    if (!s || !cam || !info || !prmIn || !out) return -1;
    rtr_render_params prm = *prmIn;
    if (prm.bandRows == 0) prm.bandRows = 8;
    if (prm.shardCount == 0) prm.shardCount = 1;
    if (prm.spp == 0 || prm.width == 0 || prm.height == 0 || prm.shardIndex >= prm.shardCount) return -1;
    if (info->numAreaLights > s->desc.numLights) return -1;
    bool wantAnalytic = out->analytic != nullptr;
    if (wantAnalytic && (!s->desc.ltc1 || !s->desc.ltc2)) return -4;
    Scene sc;
    if (!prepare(s, sc)) return -1;
    sc.useWide = s->wide != nullptr && s->nodes != nullptr && s->numWide > 0 && prm.pipeline != 1;
    sc.shadowWalk = s->shadowWalk;
    sc.ownLeafFirst = sc.useWide && (s->shadowWalk & 2u) == 0u;
    if (sc.useWide && s->walkProfile) {
        sc.wideParent.assign(s->numWide, 0u);
        for (uint32_t i = 0; i < s->numWide; ++i)
            for (uint32_t k = 0; k < 4; ++k) if (s->wide[i].child[k] >= 0 && (uint32_t)s->wide[i].child[k] < s->numWide) sc.wideParent[(size_t)s->wide[i].child[k]] = i * 4u + k;
    }
    sc.primaryStackLimit = (s->nodes != nullptr && prm.pipeline != 1) ? 16 : 0;
    sc.primaryPackets = s->nodes != nullptr && prm.pipeline != 1 && s->primaryPackets == 1;
    sc.primaryWide = sc.useWide && s->primaryPackets == 2;
    const uint32_t rows = shard_rows(prm.height, prm.bandRows, prm.shardCount);
    const uint32_t W = prm.width;
    BandMap bm{prm.bandRows, prm.shardIndex, prm.shardCount, prm.height};
    int nthreads = threads < 1 ? 1 : threads;
    std::vector<Counters> counters((size_t)nthreads);
    const float invFrames = (float)(prm.accumulatedFrames + 1u);
    /* camera rays tile by tile, as k_primary_packet walks them: a wave is one 8x8 tile of the LOCAL image (rows of the shard), one
     * sample plane at a time; lane = row in tile * 8 + column in tile */
    std::vector<Hit> pre;
    if (sc.primaryPackets) {
        pre.resize((size_t)rows * W * prm.spp);
        const uint32_t tilesPerRow = (W + 7u) / 8u, bands8 = (rows + 7u) / 8u;
        const rtr_v3 camPos = rtr_ld3(cam->position), TL = rtr_ld3(cam->topLeftViewportCorner);
        const rtr_v3 dH = rtr_ld3(cam->horizontalViewportDelta), dV = rtr_ld3(cam->verticalViewportDelta);
        parallel_rows(bands8, nthreads, [&](uint32_t band8, int tid) {
            for (uint32_t i = 0; i < prm.spp; ++i)
                for (uint32_t tx = 0; tx < tilesPerRow; ++tx) {
                    rtr_v3 d[64]; bool valid[64]; Hit h[64];
                    for (uint32_t l = 0; l < 64; ++l) {
                        const uint32_t x = tx * 8u + (l & 7u), lr = band8 * 8u + (l >> 3);
                        const int64_t gy = lr < rows ? bm.global_y(lr) : -1;
                        valid[l] = x < W && gy >= 0;
                        d[l] = rtr_mk(0.f, 0.f, 1.f);
                        if (!valid[l]) continue;
                        const float jx = rtr_random(x + i), jy = rtr_random(x + i * 322u);
                        const float offx = ((float)x + jx) - 0.5f, offy = ((float)(uint32_t)gy + jy) - 0.5f;
                        d[l] = rtr_normalize(rtr_sub(rtr_madd(rtr_madd(TL, dH, offx), dV, offy), camPos));
                    }
                    trace_packet(sc, camPos, d, valid, 0.001f, 10000.0f, h, counters[(size_t)tid]);
                    for (uint32_t l = 0; l < 64; ++l) if (valid[l]) pre[((size_t)(band8 * 8u + (l >> 3)) * W + tx * 8u + (l & 7u)) * prm.spp + i] = h[l];
                }
        });
    }

    parallel_rows(rows, nthreads, [&](uint32_t lr, int tid) {
        int64_t gy = bm.global_y(lr);
        for (uint32_t x = 0; x < W; ++x) {
            size_t p = (size_t)lr * W + x;
            if (gy < 0) {   /* padded row: defined as zero */
                if (out->analytic) out->analytic[p] = 0;
                if (out->shadowed) out->shadowed[p] = 0;
                if (out->unshadowed) out->unshadowed[p] = 0;
                if (out->normal) out->normal[p] = 0;
                if (out->position) out->position[p] = 0;
                if (out->hdr && !prm.accumulate) { out->hdr[4 * p] = out->hdr[4 * p + 1] = out->hdr[4 * p + 2] = out->hdr[4 * p + 3] = 0; }
                continue;
            }
            PixelOut po = shade_pixel(sc, *cam, *info, prm, x, (uint32_t)gy, wantAnalytic, out->unshadowed != nullptr, counters[(size_t)tid],
                                      sc.primaryPackets ? &pre[p * prm.spp] : nullptr);
            rtr_v3 sh = po.shadowed;
            if (out->hdr) {
                float* h = out->hdr + 4 * p;
                if (prm.accumulate) {
                    h[0] += sh.x; h[1] += sh.y; h[2] += sh.z; h[3] += 1.0f;
                    sh = rtr_mk(h[0] / invFrames, h[1] / invFrames, h[2] / invFrames);
                } else {
                    h[0] = sh.x; h[1] = sh.y; h[2] = sh.z; h[3] = 1.0f;
                }
            }
            if (out->shadowed) out->shadowed[p] = tonemap_pack(sh);
            if (out->unshadowed) out->unshadowed[p] = tonemap_pack(po.unshadowed);
            if (out->analytic) out->analytic[p] = tonemap_pack(po.analytic);
            if (out->normal) {                                                            /* raygen.rgen:359,363 */
                float n = (float)prm.spp;
                rtr_v3 an = rtr_normalize(rtr_mk(po.avgNormal.x / n, po.avgNormal.y / n, po.avgNormal.z / n));
                out->normal[p] = rtr_pack_bgra8(an.x, an.y, an.z);
            }
            if (out->position) {                                                          /* raygen.rgen:360-364 */
                float n = (float)prm.spp;
                out->position[p] = rtr_pack_bgra8(po.avgPosition.x / n, po.avgPosition.y / n, po.avgPosition.z / n);
            }
        }
    });

    Counters tot;
    for (auto& c : counters) tot.add(c);
    rtr_frame_stats& st = out->stats;
    memset(&st, 0, sizeof st);
    st.numRays = tot.rays; st.numPrimaryRays = tot.primary; st.numShadowRays = tot.shadow;
    st.numNodeVisits = tot.nodes; st.numTriTests = tot.tris; st.numHits = tot.hits;
    st.numLightFetches = tot.lightFetch; st.numLightTriFetches = tot.lightTriFetch;
    st.numShadowNodeVisits = tot.shadowNodes; st.numShadowTriTests = tot.shadowTris;
    st.numTexFetches = tot.texFetch; st.numAlphaTests = tot.alphaTests;
    const uint64_t shadowNodeBytes = sc.useWide ? RTR_WIDE_NODE_BYTES : RTR_BVH_NODE_BYTES;
    st.shadowTraceBytes = shadowNodeBytes * tot.shadowNodes + 48 * tot.shadowTris + 37 * tot.shadow;   /* per ray: 20-B queue record + 16-B origin of its pixel-sample + visibility byte */
    st.primaryTailRays = tot.primaryOverflow; st.shadowTailRays = tot.shadowOverflow;
    out->walk.occludedRays = tot.walk[0]; out->walk.occludedVisits = tot.walk[1]; out->walk.occludedTests = tot.walk[2];
    out->walk.visibleRays = tot.walk[3]; out->walk.visibleVisits = tot.walk[4]; out->walk.visibleTests = tot.walk[5];
    out->walk.ownLeafRays = tot.walk[6]; out->walk.ownLeafStopped = tot.walk[7];
    st.localRows = rows; st.localPixels = rows * W;
    uint32_t k = 0;
    k += out->analytic ? 1u : 0u; k += out->shadowed ? 1u : 0u; k += out->unshadowed ? 1u : 0u;
    k += out->normal ? 1u : 0u; k += out->position ? 1u : 0u;
    st.algorithmicBytes = RTR_BVH_NODE_BYTES * (tot.nodes - tot.shadowNodes) + shadowNodeBytes * tot.shadowNodes + 48 * tot.tris + 236 * (tot.hits + tot.alphaTests) + 96 * tot.lightFetch + 156 * tot.lightTriFetch +
                          16 * tot.texFetch +
                          4ull * k * st.localPixels + (out->hdr ? (prm.accumulate ? 32ull : 16ull) * st.localPixels : 0ull);
    return 0;
}

int oracle_primary_hits(const oracle_scene* s, const RtrCameraData* cam, const rtr_render_params* prm,
                        float* t, float* u, float* v, uint32_t* customIndex, uint32_t* primitiveId, int threads) {
    if (!s || !cam || !prm) return -1;
    Scene sc;
    if (!prepare(s, sc)) return -1;
    const uint32_t W = prm->width, H = prm->height, S = prm->spp;
    const rtr_v3 camPos = rtr_ld3(cam->position);
    const rtr_v3 TL = rtr_ld3(cam->topLeftViewportCorner);
    const rtr_v3 dH = rtr_ld3(cam->horizontalViewportDelta);
    const rtr_v3 dV = rtr_ld3(cam->verticalViewportDelta);
    int nthreads = threads < 1 ? 1 : threads;
    std::vector<Counters> counters((size_t)nthreads);
    parallel_rows(H, nthreads, [&](uint32_t py, int tid) {
        for (uint32_t px = 0; px < W; ++px)
            for (uint32_t i = 0; i < S; ++i) {
                float jx = rtr_random(px + i), jy = rtr_random(px + i * 322u);
                float offx = ((float)px + jx) - 0.5f, offy = ((float)py + jy) - 0.5f;
                rtr_v3 pw = rtr_madd(rtr_madd(TL, dH, offx), dV, offy);
                rtr_v3 rayDir = rtr_normalize(rtr_sub(pw, camPos));
                Hit h = trace(sc, camPos, rayDir, 0.001f, 10000.0f, false, counters[(size_t)tid]);
                size_t k = ((size_t)py * W + px) * S + i;
                if (t) t[k] = h.hit ? h.t : -1.0f;
                if (u) u[k] = h.hit ? h.u : 0.0f;
                if (v) v[k] = h.hit ? h.v : 0.0f;
                if (customIndex) customIndex[k] = h.hit ? h.custom : 0xffffffffu;
                if (primitiveId) primitiveId[k] = h.hit ? h.prim : 0xffffffffu;
            }
    });
    return 0;
}

uint32_t oracle_pcg_hash(uint32_t seed) { return rtr_pcg_hash(seed); }
float oracle_random(uint32_t seed) { return rtr_random(seed); }
float oracle_pow(float x, float y) { return rtr_pow(x, y); }
float oracle_log2(float x) { return rtr_log2(x); }
float oracle_exp2(float x) { return rtr_exp2(x); }
uint32_t oracle_pack_bgra8(float r, float g, float b) { return rtr_pack_bgra8(r, g, b); }
float oracle_unorm8_to_float_fast(uint32_t b) { return rtr_unorm8_to_float(b); }
/* rtr_div_by (the product's division by a known divisor) against the IEEE quotient for every float a whose bits lie in
 * [loBits, hiBits], both signs: the number of a for which the two differ as numbers */
uint64_t oracle_div_by_mismatches(float b, uint32_t loBits, uint32_t hiBits) {
    const float r = 1.0f / b;
    uint64_t bad = 0;
    for (uint64_t u = loBits; u <= hiBits; ++u)
        for (uint32_t sign = 0; sign < 2; ++sign) {
            const float a = rtr_u2f((uint32_t)u | (sign << 31));
            if (!(rtr_div_by(a, b, r) == a / b)) ++bad;
        }
    return bad;
}
int oracle_mt(const float* o, const float* d, const float* v0, const float* e1, const float* e2, float tmin, float* tuv) {
    return rtr_mt_intersect(rtr_ld3(o), rtr_ld3(d), rtr_ld3(v0), rtr_ld3(e1), rtr_ld3(e2), tmin, &tuv[0], &tuv[1], &tuv[2]);
}

}  // extern "C"
