/* oracle.h — C interface of the CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT).
 *
 * The oracle is a scalar C++ restatement of the reference's per-pixel ray-tracing path
 * (reference src/shaders/raygen.rgen + closesthit.rchit + miss.rmiss + cook-torrance.glsl +
 * LTC.glsl + intersect.rint).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (realtimeraytracer_amd/) never does.
 *
 * PARITY STATUS: "parity unpinned" at the traversal / intersection boundary — the reference
 * runs BVH build, traversal and the ray-triangle test inside the Vulkan driver / RT hardware
 * and ships no tests or golden images.  What IS pinned: the PCG hash, the
 * camera basis and the OBJ ingest counts (SURVEY Appendix B known answers), and the OBJ reader
 * against the real tinyobjloader compiled from the reference tree (oracle/_ref).
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stdint.h>
#include "../include/rtr.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_scene {
    rtr_scene_desc desc;          /* same arrays the product's rtr_scene_create receives */
    /* Optional BVH exported by the product (rtr_scene_export_bvh).  NULL -> brute force over all
     * world-space triangles in (instance, primitive) order: an independent check of the packer,
     * the BVH builder and the traversal. */
    const RtrBvhNode* nodes;  uint32_t numNodes;
    const RtrBvhTri*  tris;   uint32_t numTris;
    RtrBvhGrid grid;              /* the grid the nodes' 16-bit planes live on (rtr_scene_stats.grid) */
    /* Optional 4-wide view exported by the product (rtr_scene_export_wide).  When given (and params.pipeline != 1: the megakernel
     * walks the BVH2 for its shadow rays too), shadow rays are walked over it exactly as k_shadow_trace4 does — nearest hit child
     * first, ties to the lower slot, the others stacked in slot order, a leaf's triangles in storage order — so that the any-hit
     * work counters (numShadowNodeVisits = 64-B records visited, numShadowTriTests) can be held equal to the kernel's.  The
     * visibility of a ray does not depend on which structure is walked (tests compare all of them with the brute-force mode). */
    const RtrWideNode* wide;  uint32_t numWide;
    /* How the staged pipeline (params.pipeline != 1) walks its CAMERA rays over `nodes` — it decides the work counters, never the hits:
     * 0 (what the product does by default): one ray per lane with 16 stack entries, deeper rays re-traced from scratch (k_primary +
     * k_primary_tail);  1 (tunable primary_packet = 1): the 64 rays of an 8x8 tile as one packet (k_primary_packet: a node is visited
     * when any lane's ray hits it, one stack of {child, lane mask} per tile; counters count every lane of a visited mask);
     * 2 (tunable primary_wide = 1; needs `wide`): one ray per lane over the 4-wide view, closest hit, 16 stack entries + redo over the
     * BVH2 (k_primary4). */
    uint32_t primaryPackets;
    /* How a shadow ray is walked over the wide view — it decides the work counters, never the answer (any-hit is a pure function of ray
     * and triangles).  The product's default (0): a ray that leaves its surface point INTO the surface — dot(hitNormal, light sample - hitPoint) < 0,
     * the area-light samples of raygen.rgen:206-241 only — first tests the triangles of the LEAF its own hit triangle sits in (it starts
     * 0.01 above that triangle and nearly always re-enters it), then walks from the root; at a record the hit child whose EXIT distance
     * (clamped to the ray's far limit) is the greatest is entered first (strict >, ties to the lower slot) — a shadow ray's occluder
     * sits, more often than not, towards the light — and the others are stacked in slot order.
     * bit 0 (experiment, profiles/experiments/anyhit_order_lab.py): the static slot order — the first hit slot in SLOT order, the
     * others popping in slot order, no distance compared.  bit 1: no start at the own leaf (the product's tunable trace_own_leaf = 0).
     * bits 2-3: which child first — 0 farthest exit (the product), 1 nearest entry (rounds 1-4; a build with -DRTR_SHADOW_FAR_FIRST=0),
     * 2 farthest entry, 3 nearest exit.  bit 4 (experiment): the stacked children sorted as well. */
    uint32_t shadowWalk;
    /* Experiments only (profiles/experiments/anyhit_order_lab.py), normally NULL: numWide * 4 * 3 counters the shadow walk adds to —
     * per (record, slot) {times the walk went into the slot, record visits + triangle tests it then spent below it, occluders it found
     * below it} — from which an order "by found occluders per unit of work" is made. */
    uint64_t* walkProfile;
    /* Experiments only, normally NULL: one record of 8 floats per shadow ray walked over the wide view {visits, tests, occluded, started at its
     * own leaf, t of the hit or -1, tmax, direction y, origin y}, the first walkRaysCap of them; *walkRaysCount counts all. */
    float* walkRays; uint64_t walkRaysCap; uint64_t* walkRaysCount;
} oracle_scene;

/* shadow rays walked over the wide view, split by their answer (what an any-hit order is judged by) */
typedef struct oracle_walk_stats {
    uint64_t occludedRays, occludedVisits, occludedTests;
    uint64_t visibleRays, visibleVisits, visibleTests;
    uint64_t ownLeafRays;       /* rays that started at the leaf of their own triangle */
    uint64_t ownLeafStopped;    /* ... and were stopped there, without a record visit */
} oracle_walk_stats;

typedef struct oracle_out {
    /* any pointer may be NULL; sizes are localRows*width elements */
    uint32_t* analytic;      /* RGBA8 packed B,G,R,255 */
    uint32_t* shadowed;
    uint32_t* unshadowed;
    uint32_t* normal;
    uint32_t* position;
    float*    hdr;           /* float4 per pixel: pre-tonemap shadowed radiance (accumulated if params.accumulate) */
    rtr_frame_stats stats;   /* counters filled when params.collectStats */
    oracle_walk_stats walk;
} oracle_out;

/* Renders with the same semantics as rtr_render (same params struct, same band sharding).
 * threads <= 1 -> scalar single-thread loop.  Returns 0 on success. */
int oracle_render(const oracle_scene* scene, const RtrCameraData* camera, const RtrSceneInfo* sceneInfo,
                  const rtr_render_params* params, oracle_out* out, int threads);

/* Primary-visibility only (raygen + traversal + MT): per pixel-sample t,u,v and ids, for the
 * BVH-vs-brute-force cross-check.  Arrays of width*height*spp. */
int oracle_primary_hits(const oracle_scene* scene, const RtrCameraData* camera, const rtr_render_params* params,
                        float* t, float* u, float* v, uint32_t* customIndex, uint32_t* primitiveId, int threads);

/* Denoise + combine restatement (reference src/shaders/denoise.comp, combine.comp and the host
 * protocol src/app/application.cppm:391-445).  All images RGBA8, width*height. */
int oracle_denoise_combine(uint32_t width, uint32_t height, const uint32_t* analytic,
                           uint32_t* shadowed, uint32_t* unshadowed, const uint32_t* normal, const uint32_t* position,
                           uint32_t* denoisedShadowed, uint32_t* denoisedUnshadowed, uint32_t* finalImage,
                           int iterations);

/* known-answer helpers exported for tests */
uint32_t oracle_pcg_hash(uint32_t seed);
float    oracle_random(uint32_t seed);
float    oracle_pow(float x, float y);
float    oracle_log2(float x);
float    oracle_exp2(float x);
uint32_t oracle_pack_bgra8(float r, float g, float b);
float    oracle_unorm8_to_float_fast(uint32_t b);   /* rtr_unorm8_to_float: the product's division-free form */
uint64_t oracle_div_by_mismatches(float b, uint32_t loBits, uint32_t hiBits);   /* rtr_div_by vs a / b over a range of a's bit patterns, both signs */
int      oracle_mt(const float* o, const float* d, const float* v0, const float* e1, const float* e2,
                   float tmin, float* tuv);

#ifdef __cplusplus
}
#endif
#endif
