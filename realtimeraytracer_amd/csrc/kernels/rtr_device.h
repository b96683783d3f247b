/* rtr_device.h — device-side building blocks of the gfx950 ray-tracing path.
 *
 *   trace<ANY,STATS,BLOCK>() : BVH2 traversal, per-lane node stack in LDS laid out
 *                        stack[depth * BLOCK + tid] (lane-interleaved: one ds_read/write_b32 per
 *                        push/pop, 32 consecutive lanes -> 32 distinct banks, conflict-free),
 *                        64-B children-in-parent nodes fetched as dwordx4 loads, Moeller-Trumbore on
 *                        48-B {v0,e1,e2} records.  Replaces traceRayEXT (reference raygen.rgen:99,231,303).
 *   shade_sample<Policy>() : the per-sample body of reference raygen.rgen:81-339 with the hit-shader
 *                        fetch of closesthit.rchit:45-110 inlined; the shadow-ray query is a
 *                        policy so the same loop structure serves the megakernel (trace inline),
 *                        the wavefront generator (enqueue with ballot compaction) and the
 *                        wavefront resolve (look the visibility bit up).
 *
 * All float arithmetic goes through include/rtr_math.h (the numerical contract shared with the
 * CPU oracle); compile with -ffp-contract=off.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../../include/rtr_types.h"
#include "../../../include/rtr_math.h"

namespace rtrdev {

struct DeviceTexture { const uint8_t* pixels; uint32_t width, height, channels, _pad; };

constexpr uint32_t kLightTriRecord = 4;     /* float4s per light-triangle record */

struct DeviceScene {
    const uint4* nodes;              /* RtrBvhNode (layout version 3) as 2 x uint4 */
    const RtrBvhGrid* grid;          /* the grid the 16-bit planes live on; device memory so a refit can rewrite it */
    const uint4* nodes4;             /* RtrWideNode (layout W4.0): 4-wide view for the any-hit kernel (4 x uint4 per entry, breadth-first order), or null */
    uint32_t numNodes4;              /* entries in nodes4 (the first wideReached of them are the tree) */
    const float4* tris;              /* RtrBvhTri  as 3 x float4 */
    const RtrVertex* vertices;
    const uint32_t* indices;
    const RtrObjectInfo* objects;
    const RtrAreaLightInfo* lights;
    const float4* lightTris;         /* per light triangle, kLightTriRecord x float4: {P0, area} {P1, pdf} {P2, -} {unit normal, -} in world space (k_light_tris) */
    const uint32_t* lightTriFirst;   /* first record of light l */
    const float* xforms;             /* 12 floats (3x4 row-major object->world) per customIndex */
    const float* nmats;              /* 12 floats (9 used: transpose(inverse(mat3))) per customIndex */
    const float* ltc1;               /* 64x64x4 or null */
    const float* ltc2;
    float skyLinear[3];
    uint32_t numLights;
    uint32_t numLightTris;           /* light triangles of all the lights (records in lightTris) */
    const DeviceTexture* textures;   /* texSamplers[]: indexed by ObjectInfo.*Index (slots 0,1 unused: LTC) */
    DeviceTexture hdri;              /* pixels == null -> constant sky */
};

struct Counters {                    /* device mirror of rtr_frame_stats' counters */
    unsigned long long rays, primary, shadow, nodes, tris, hits, lightFetch, lightTriFetch, shadowNodes, shadowTris, texFetch, alphaTests;
    /* scheduling of the any-hit kernel (counting form): loop trips of its two phases and the lanes that worked in them */
    unsigned long long innerIters, innerLanes, triIters, triLanes, refills, clockCycles, clockRef;
};

struct LocalStats {
    uint32_t rays = 0, primary = 0, shadow = 0, nodes = 0, tris = 0, hits = 0, lightFetch = 0, lightTriFetch = 0;
    uint32_t shadowNodes = 0, shadowTris = 0, texFetch = 0, alphaTests = 0;
    __device__ void flush(Counters* c) const {
        if (texFetch) atomicAdd(&c->texFetch, (unsigned long long)texFetch);
        if (alphaTests) atomicAdd(&c->alphaTests, (unsigned long long)alphaTests);
        if (shadowNodes) atomicAdd(&c->shadowNodes, (unsigned long long)shadowNodes);
        if (shadowTris) atomicAdd(&c->shadowTris, (unsigned long long)shadowTris);
        if (rays) atomicAdd(&c->rays, (unsigned long long)rays);
        if (primary) atomicAdd(&c->primary, (unsigned long long)primary);
        if (shadow) atomicAdd(&c->shadow, (unsigned long long)shadow);
        if (nodes) atomicAdd(&c->nodes, (unsigned long long)nodes);
        if (tris) atomicAdd(&c->tris, (unsigned long long)tris);
        if (hits) atomicAdd(&c->hits, (unsigned long long)hits);
        if (lightFetch) atomicAdd(&c->lightFetch, (unsigned long long)lightFetch);
        if (lightTriFetch) atomicAdd(&c->lightTriFetch, (unsigned long long)lightTriFetch);
    }
};

struct HitRec { float t, u, v; uint32_t custom, prim; int32_t leaf; };   /* custom == 0xffffffff: miss; leaf: code of the leaf the hit triangle sits in (the shadow rays that leave INTO the surface start their walk there) */
#define RTR_MISS 0xffffffffu

__device__ __forceinline__ rtr_v3 f4xyz(const float4& a) { return rtr_mk(a.x, a.y, a.z); }

/* texture(): 8-bit texels, linear filter, repeat addressing, one mip — the reference's sampler
 * (src/vulkan/memory/image_sampler.cppm:26-42) in software; same arithmetic as oracle sample_tex(). */
template <bool STATS>
__device__ __forceinline__ float4 sample_tex(const DeviceTexture& tx, float u, float v, LocalStats& st) {
    if (STATS) st.texFetch++;
    if (!(u > -1.0e9f && u < 1.0e9f)) u = 0.0f;
    if (!(v > -1.0e9f && v < 1.0e9f)) v = 0.0f;
    const int W = (int)tx.width, H = (int)tx.height, ch = (int)tx.channels;
    const float uf = u - __builtin_floorf(u), vf = v - __builtin_floorf(v);
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
    /* the pointer comes out of a table in memory, so the compiler cannot know its address space and would emit
     * flat_load (slower, and counted on both vmcnt and lgkmcnt); texels always live in global memory */
    typedef const __attribute__((address_space(1))) uint8_t* gptr8;
    typedef const __attribute__((address_space(1))) uint32_t* gptr32;
    const gptr8 p = (gptr8)(uintptr_t)tx.pixels;
    float o[4];
    if (ch == 4) {
        const uint32_t q00 = *(gptr32)(p + ((size_t)y0 * W + x0) * 4);
        const uint32_t q10 = *(gptr32)(p + ((size_t)y0 * W + x1) * 4);
        const uint32_t q01 = *(gptr32)(p + ((size_t)y1 * W + x0) * 4);
        const uint32_t q11 = *(gptr32)(p + ((size_t)y1 * W + x1) * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float t00 = rtr_unorm8_to_float((q00 >> (8 * k)) & 0xffu), t10 = rtr_unorm8_to_float((q10 >> (8 * k)) & 0xffu);
            const float t01 = rtr_unorm8_to_float((q01 >> (8 * k)) & 0xffu), t11 = rtr_unorm8_to_float((q11 >> (8 * k)) & 0xffu);
            const float a = rtr_fma(t10 - t00, fx, t00), b = rtr_fma(t11 - t01, fx, t01);
            o[k] = rtr_fma(b - a, fy, a);
        }
    } else {
        const float t00 = rtr_unorm8_to_float(p[(size_t)y0 * W + x0]), t10 = rtr_unorm8_to_float(p[(size_t)y0 * W + x1]);
        const float t01 = rtr_unorm8_to_float(p[(size_t)y1 * W + x0]), t11 = rtr_unorm8_to_float(p[(size_t)y1 * W + x1]);
        const float a = rtr_fma(t10 - t00, fx, t00), b = rtr_fma(t11 - t01, fx, t01);
        o[0] = rtr_fma(b - a, fy, a); o[1] = 0.0f; o[2] = 0.0f; o[3] = 1.0f;
    }
    return make_float4(o[0], o[1], o[2], o[3]);
}

/* opacity.rahit:31-64 — any-hit for candidates on alpha-tested geometry; false = ignoreIntersectionEXT */
template <bool STATS>
__device__ __forceinline__ bool alpha_pass(const DeviceScene& sc, uint32_t custom, uint32_t prim, float bu, float bv, LocalStats& st) {
    const RtrObjectInfo* oi = sc.objects + (custom - sc.numLights);
    if (oi->usesOpacityMap == 0u) return true;
    if (STATS) st.alphaTests++;
    const uint32_t vOff = oi->vertexOffset, iOff = oi->indexOffset;
    const uint32_t i0 = sc.indices[3u * prim + 0u + iOff], i1 = sc.indices[3u * prim + 1u + iOff], i2 = sc.indices[3u * prim + 2u + iOff];
    const float* uv0 = sc.vertices[i0 + vOff].uv;
    const float* uv1 = sc.vertices[i1 + vOff].uv;
    const float* uv2 = sc.vertices[i2 + vOff].uv;
    const float b0 = 1.0f - bu - bv;
    const float uu = rtr_fma(uv2[0], bv, rtr_fma(uv1[0], bu, uv0[0] * b0));
    const float vv = rtr_fma(uv2[1], bv, rtr_fma(uv1[1], bu, uv0[1] * b0));
    const float4 t = sample_tex<STATS>(sc.textures[oi->opacityIndex], uu, vv, st);
    return !(t.x < 0.9f);
}

/* Slab test of one child box of an RtrBvhNode: wmin = (qminx | qminy << 16), wmax = (qmaxx | qmaxy << 16),
 * wz = (qminz | qmaxz << 16).  Same arithmetic as rtr_slab_q (one fma per plane, identical min/max tree); written on
 * 2-vectors so the six fmas become three v_pk_fma_f32 and the conversions v_cvt_f32_u32 with a 16-bit source select. */
typedef float rtr_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bool slab_pair(uint32_t wmin, uint32_t wmax, uint32_t wz, rtr_v3 ga, rtr_v3 gb,
                                          float tmin, float tmax, float& t_entry) {
    const rtr_f2 axy = {ga.x, ga.y}, bxy = {gb.x, gb.y}, az = {ga.z, ga.z}, bz = {gb.z, gb.z};
    const rtr_f2 t0 = __builtin_elementwise_fma(rtr_f2{(float)(wmin & 0xffffu), (float)(wmin >> 16)}, axy, bxy);
    const rtr_f2 t1 = __builtin_elementwise_fma(rtr_f2{(float)(wmax & 0xffffu), (float)(wmax >> 16)}, axy, bxy);
    const rtr_f2 tz = __builtin_elementwise_fma(rtr_f2{(float)(wz & 0xffffu), (float)(wz >> 16)}, az, bz);
    const float lo = rtr_hwmax(rtr_hwmax(rtr_hwmin(t0.x, t1.x), rtr_hwmin(t0.y, t1.y)), rtr_hwmax(rtr_hwmin(tz.x, tz.y), tmin));
    const float hi = rtr_hwmin(rtr_hwmin(rtr_hwmax(t0.x, t1.x), rtr_hwmax(t0.y, t1.y)), rtr_hwmin(rtr_hwmax(tz.x, tz.y), tmax));
    t_entry = lo;
    return lo <= hi * RTR_BOX_WIDEN;
}

/* The same test when the signs of the ray's direction are known at compile time (OCT bit a = ga.a < 0): q -> q * ga + gb is
 * monotone, so the entry plane of an axis is the box's min plane for ga >= 0 and its max plane otherwise, and min(t0, t1) /
 * max(t0, t1) ARE those two values — the six per-axis min/max disappear, the result is bit-identical (boxes have
 * qmin <= qmax on every axis: rtr_quant_lo/hi; tests/test_oracle_bvh.py::_check_bvh asserts it on host- and device-built and re-fitted trees). */
template <int OCT>
__device__ __forceinline__ bool slab_oct(uint32_t wmin, uint32_t wmax, uint32_t wz, rtr_v3 ga, rtr_v3 gb,
                                         float tmin, float tmax, float& t_entry) {
    const rtr_f2 axy = {ga.x, ga.y}, bxy = {gb.x, gb.y}, az = {ga.z, ga.z}, bz = {gb.z, gb.z};
    const rtr_f2 t0 = __builtin_elementwise_fma(rtr_f2{(float)(wmin & 0xffffu), (float)(wmin >> 16)}, axy, bxy);
    const rtr_f2 t1 = __builtin_elementwise_fma(rtr_f2{(float)(wmax & 0xffffu), (float)(wmax >> 16)}, axy, bxy);
    const rtr_f2 tz = __builtin_elementwise_fma(rtr_f2{(float)(wz & 0xffffu), (float)(wz >> 16)}, az, bz);
    const float nx = (OCT & 1) ? t1.x : t0.x, fx = (OCT & 1) ? t0.x : t1.x;
    const float ny = (OCT & 2) ? t1.y : t0.y, fy = (OCT & 2) ? t0.y : t1.y;
    const float nz = (OCT & 4) ? tz.y : tz.x, fz = (OCT & 4) ? tz.x : tz.y;
    const float lo = rtr_hwmax(rtr_hwmax(nx, ny), rtr_hwmax(nz, tmin));
    const float hi = rtr_hwmin(rtr_hwmin(fx, fy), rtr_hwmin(fz, tmax));
    t_entry = lo;
    return lo <= hi * RTR_BOX_WIDEN;
}
template <>
__device__ __forceinline__ bool slab_oct<8>(uint32_t wmin, uint32_t wmax, uint32_t wz, rtr_v3 ga, rtr_v3 gb,
                                            float tmin, float tmax, float& t_entry) {
    return slab_pair(wmin, wmax, wz, ga, gb, tmin, tmax, t_entry);       /* 8 = signs differ between the lanes of the wave */
}

/* Slab test of one slot of an RtrWideNode: the same three words, but the planes are HALF FLOATS — offsets from the centre of the scene
 * grid in grid steps, rounded outward when the record was made — so t = fma(plane, ga, gbc) is one v_fma_mix_f32 per plane with no
 * conversion (gbc: gb taken about the scene's wide centre, rtr_ray_grid_centre).  OCT as in slab_oct. */
/* v_min_f32 as the instruction.  The compiler's fmin first quiets an operand it cannot prove canonical (v_max x, x); for the ray's far
 * limit — a loaded, loop-carried value — it re-materialised that on EVERY visit.  The limit is a finite number the queue build wrote. */
__device__ __forceinline__ float vmin_raw(float a, float b) {
    float r;
    asm("v_min_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
/* EXIT: the distance handed back is the box's EXIT (clamped to tmax) instead of its entry: what the any-hit walk orders the children of a
 * record by (farthest exit first, inner_nodes4) */
template <int OCT, bool EXIT = false>
__device__ __forceinline__ bool slab_wide(uint32_t wmin, uint32_t wmax, uint32_t wz, rtr_v3 ga, rtr_v3 gbc, float tmin, float tmax, float& t_entry) {
    typedef _Float16 rtr_h2 __attribute__((ext_vector_type(2)));
    const rtr_h2 pmin = __builtin_bit_cast(rtr_h2, wmin), pmax = __builtin_bit_cast(rtr_h2, wmax), pz = __builtin_bit_cast(rtr_h2, wz);
    const float x0 = rtr_fma((float)pmin.x, ga.x, gbc.x), x1 = rtr_fma((float)pmax.x, ga.x, gbc.x);
    const float y0 = rtr_fma((float)pmin.y, ga.y, gbc.y), y1 = rtr_fma((float)pmax.y, ga.y, gbc.y);
    const float z0 = rtr_fma((float)pz.x, ga.z, gbc.z), z1 = rtr_fma((float)pz.y, ga.z, gbc.z);
    float nx, fx, ny, fy, nz, fz;
    if (OCT < 8) {
        nx = (OCT & 1) ? x1 : x0; fx = (OCT & 1) ? x0 : x1;
        ny = (OCT & 2) ? y1 : y0; fy = (OCT & 2) ? y0 : y1;
        nz = (OCT & 4) ? z1 : z0; fz = (OCT & 4) ? z0 : z1;
    } else {
        nx = rtr_hwmin(x0, x1); fx = rtr_hwmax(x0, x1);
        ny = rtr_hwmin(y0, y1); fy = rtr_hwmax(y0, y1);
        nz = rtr_hwmin(z0, z1); fz = rtr_hwmax(z0, z1);
    }
    const float lo = rtr_hwmax(rtr_hwmax(nx, ny), rtr_hwmax(nz, tmin));
    const float hi = rtr_hwmin(rtr_hwmin(fx, fy), vmin_raw(fz, tmax));
    t_entry = EXIT ? hi : lo;
    return lo <= hi * RTR_BOX_WIDEN;
}

/* Direction signs of a ray as trace() will see them (bit a = the grid-space slope of axis a is negative). */
__device__ __forceinline__ uint32_t ray_octant(const DeviceScene& sc, rtr_v3 o, rtr_v3 d) {
    const rtr_v3 idir = rtr_mk(rtr_safe_rcp_dir(d.x), rtr_safe_rcp_dir(d.y), rtr_safe_rcp_dir(d.z));
    rtr_v3 ga, gb;
    rtr_ray_grid(o, idir, sc.grid->origin, sc.grid->scale, &ga, &gb);
    return (ga.x < 0.f ? 1u : 0u) | (ga.y < 0.f ? 2u : 0u) | (ga.z < 0.f ? 4u : 0u);
}

/* ------------------------------------------------------------------------------------------
 * BVH traversal.  Restates, operation for operation, the algorithm of oracle/oracle_render.cpp
 * trace_bvh(): ordered descent (near child first, ties -> left), far child pushed, box culled iff
 * entry > current best t (with the conservative widening of rtr_slab), leaf triangles tested in
 * storage order, closest = min over (t, customIndex, primitiveID).
 * `stack` points at this lane's slot 0; consecutive depths are `BLOCK` ints apart.
 * ------------------------------------------------------------------------------------------ */
/* LIMIT > 0: the stack holds only LIMIT entries; a ray that needs more is abandoned with best.custom = RTR_STACK_OVERFLOW (the
 * caller re-traces it with a full-depth stack), so the common case can run with a small LDS footprint. */
#define RTR_STACK_OVERFLOW 0xfffffffeu
/* OCT 0..7: every lane that calls has these direction signs (slab_oct); 8 = any. */
template <bool ANY, bool STATS, int BLOCK, int LIMIT = 0, int OCT = 8>
__device__ __forceinline__ bool trace(const DeviceScene& sc, int32_t* __restrict__ stack,
                                      rtr_v3 o, rtr_v3 d, float tmin, float tmax, HitRec& best, LocalStats& st) {
    if (STATS) { st.rays++; if (ANY) st.shadow++; else st.primary++; }
    best.custom = RTR_MISS; best.prim = RTR_MISS; best.t = tmax; best.u = 0.f; best.v = 0.f; best.leaf = 0;
    if (!(tmax > tmin)) return false;
    const rtr_v3 idir = rtr_mk(rtr_safe_rcp_dir(d.x), rtr_safe_rcp_dir(d.y), rtr_safe_rcp_dir(d.z));
    rtr_v3 ga, gb;                                        /* t(q) = q * ga + gb (rtr_math.h) */
    rtr_ray_grid(o, idir, sc.grid->origin, sc.grid->scale, &ga, &gb);
    /* nodes and triangles through buffer resources: a visit's address is one 32-bit shift (see k_shadow_trace) */
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t nodeBuf = __builtin_amdgcn_make_buffer_rsrc((void*)sc.nodes, 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t triBuf = __builtin_amdgcn_make_buffer_rsrc((void*)sc.tris, 0, 0xffffffff, 0x00020000);
    bool found = false;
    float limit = tmax;
    int sp = 0;
    int32_t cur = 0;
    /* Two loops, not one with a branch per kind of entry: the walk down the inner nodes carries only (cur, sp) around its back edge;
     * written as one loop with `continue`s the compiler copied the whole hit record (best.*, limit, found: ~30 v_mov per visit, a
     * quarter of a visit's cycles) on every edge.  The order of visits and tests is unchanged. */
    constexpr int32_t kWalkEnd = (int32_t)0x80000000;     /* not a leaf code (rtr_kernels.hip: kDone) */
    bool over = false;                                    /* LIMIT entries were not enough: the ray is abandoned (after the loop) */
    for (;;) {
        while (cur >= 0) {
            const int32_t nodeOff = cur << 5;
            const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(nodeBuf, nodeOff, 0, 0);
            const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(nodeBuf, nodeOff + 16, 0, 0);
            const int2 ch = make_int2((int)b.z, (int)b.w);
            if (STATS) { st.nodes++; if (ANY) st.shadowNodes++; }
            float tl, tr;
            const bool hl = slab_oct<OCT>(a.x, a.y, b.x, ga, gb, tmin, limit, tl);
            const bool hr = slab_oct<OCT>(a.z, a.w, b.y, ga, gb, tmin, limit, tr);
            if (hl && hr) {
                /* closest hit: the nearer child first (ties: child 0).  Any hit: the FARTHER one first (ties: child 0) — a shadow ray's occluder
                 * sits towards the light more often than not (k_shadow_trace4's rule, on entry distances here; the oracle's trace_bvh restates it) */
                const bool swap = ANY ? tl < tr : tr < tl;
                const int32_t nearC = swap ? ch.y : ch.x;
                const int32_t farC = swap ? ch.x : ch.y;
                if (LIMIT > 0 && sp >= LIMIT) { over = true; cur = kWalkEnd; }
                else { stack[sp * BLOCK] = farC; ++sp; cur = nearC; }
            } else if (hl) cur = ch.x;
            else if (hr) cur = ch.y;
            else if (sp == 0) cur = kWalkEnd;
            else { --sp; cur = stack[sp * BLOCK]; }
        }
        if (cur == kWalkEnd) break;
        {
            const uint32_t code = (uint32_t)~cur;
            const uint32_t first = code >> 3, count = (code & 7u) + 1u;
            for (uint32_t i = 0; i < count; ++i) {
                const int32_t triOff = (int32_t)((first + i) * 48u);
                const u32x4 r0 = __builtin_amdgcn_raw_buffer_load_b128(triBuf, triOff, 0, 0);
                const u32x4 r1 = __builtin_amdgcn_raw_buffer_load_b128(triBuf, triOff + 16, 0, 0);
                const u32x4 r2 = __builtin_amdgcn_raw_buffer_load_b128(triBuf, triOff + 32, 0, 0);
                const float4 q0 = make_float4(__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z), __uint_as_float(r0.w));
                const float4 q1 = make_float4(__uint_as_float(r1.x), __uint_as_float(r1.y), __uint_as_float(r1.z), __uint_as_float(r1.w));
                const float4 q2 = make_float4(__uint_as_float(r2.x), __uint_as_float(r2.y), __uint_as_float(r2.z), __uint_as_float(r2.w));
                if (STATS) { st.tris++; if (ANY) st.shadowTris++; }
                float t, u, v;
                if (rtr_mt_intersect(o, d, f4xyz(q0), f4xyz(q1), f4xyz(q2), tmin, &t, &u, &v)) {
                    if (t < tmax) {
                        const uint32_t cu = __float_as_uint(q0.w), pr = __float_as_uint(q1.w);
                        if ((__float_as_uint(q2.w) & 1u) && !alpha_pass<STATS>(sc, cu, pr, u, v, st)) continue;
                        bool take;
                        if (!found) take = true;
                        else take = t < best.t || (t == best.t && (cu < best.custom || (cu == best.custom && pr < best.prim)));
                        if (take) {
                            found = true; best.t = t; best.u = u; best.v = v; best.custom = cu; best.prim = pr; best.leaf = cur;
                            limit = t;
                            if (ANY) return true;
                        }
                    }
                }
            }
        }
        if (sp == 0) break;
        --sp; cur = stack[sp * BLOCK];
    }
    if (LIMIT > 0 && over) { best.custom = RTR_STACK_OVERFLOW; best.prim = RTR_MISS; best.t = tmax; best.u = 0.f; best.v = 0.f; best.leaf = 0; return false; }
    return found;
}

/* ---- cook-torrance.glsl (reference src/shaders/cook-torrance.glsl:1-61) ---------------------- */
#define RTR_PI_F 3.14159265359f
__device__ __forceinline__ float chiGGX(float v) { return v > 0.0f ? 1.0f : 0.0f; }
__device__ __forceinline__ float GGX_Distribution(rtr_v3 n, rtr_v3 h, float alpha) {
    const float NoH = rtr_dot(n, h);
    const float alpha2 = alpha * alpha;
    const float NoH2 = NoH * NoH;
    const float den = rtr_max(rtr_fma(NoH2, alpha2, 1.0f - NoH2), 0.001f);
    return (chiGGX(NoH) * alpha2) / (RTR_PI_F * den * den);
}
__device__ __forceinline__ float GGX_PartialGeometryTerm(rtr_v3 v, rtr_v3 n, rtr_v3 h, float alpha) {
    float VoH2 = rtr_clamp(rtr_dot(v, h), 0.001f, 1.0f);
    const float chi = chiGGX(VoH2 / rtr_clamp(rtr_dot(v, n), 0.001f, 1.0f));
    VoH2 = VoH2 * VoH2;
    const float tan2 = (1.0f - VoH2) / VoH2;
    return (chi * 2.0f) / (1.0f + rtr_sqrt(rtr_fma(alpha * alpha, tan2, 1.0f)));
}
__device__ __forceinline__ rtr_v3 Fresnel_Schlick(float cosT, rtr_v3 F0) {
    const float p = rtr_pow(1.0f - cosT, 5.0f);
    return rtr_mk(rtr_fma(1.0f - F0.x, p, F0.x), rtr_fma(1.0f - F0.y, p, F0.y), rtr_fma(1.0f - F0.z, p, F0.z));
}

/* ---- LTC.glsl (reference src/shaders/LTC.glsl:2-69) + the sampler of image_sampler.cppm:26-42 -- */
#define RTR_LUT_SIZE 64.0f
#define RTR_LUT_SCALE ((RTR_LUT_SIZE - 1.0f) / RTR_LUT_SIZE)
#define RTR_LUT_BIAS (0.5f / RTR_LUT_SIZE)

__device__ __forceinline__ float4 sample_lut(const float* __restrict__ lut, float u, float v) {
    float x = rtr_fma(u, RTR_LUT_SIZE, -0.5f), y = rtr_fma(v, RTR_LUT_SIZE, -0.5f);
    if (!(x >= -1.0e6f && x <= 1.0e6f)) x = 0.0f;      /* NaN / huge coordinates sample texel (0,0) */
    if (!(y >= -1.0e6f && y <= 1.0e6f)) y = 0.0f;
    const float x0f = __builtin_floorf(x), y0f = __builtin_floorf(y);
    const float fx = x - x0f, fy = y - y0f;
    const int x0 = ((int)x0f) & 63, y0 = ((int)y0f) & 63;
    const int x1 = (x0 + 1) & 63, y1 = (y0 + 1) & 63;
    const float4 t00 = *reinterpret_cast<const float4*>(lut + (y0 * 64 + x0) * 4);
    const float4 t10 = *reinterpret_cast<const float4*>(lut + (y0 * 64 + x1) * 4);
    const float4 t01 = *reinterpret_cast<const float4*>(lut + (y1 * 64 + x0) * 4);
    const float4 t11 = *reinterpret_cast<const float4*>(lut + (y1 * 64 + x1) * 4);
    float4 r;
    { const float a = rtr_fma(t10.x - t00.x, fx, t00.x), b = rtr_fma(t11.x - t01.x, fx, t01.x); r.x = rtr_fma(b - a, fy, a); }
    { const float a = rtr_fma(t10.y - t00.y, fx, t00.y), b = rtr_fma(t11.y - t01.y, fx, t01.y); r.y = rtr_fma(b - a, fy, a); }
    { const float a = rtr_fma(t10.z - t00.z, fx, t00.z), b = rtr_fma(t11.z - t01.z, fx, t01.z); r.z = rtr_fma(b - a, fy, a); }
    { const float a = rtr_fma(t10.w - t00.w, fx, t00.w), b = rtr_fma(t11.w - t01.w, fx, t01.w); r.w = rtr_fma(b - a, fy, a); }
    return r;
}

__device__ __forceinline__ rtr_v3 IntegrateEdgeVec(rtr_v3 v1, rtr_v3 v2) {
    const float x = rtr_dot(v1, v2);
    const float y = rtr_abs(x);
    const float a = rtr_fma(rtr_fma(0.0145206f, y, 0.4965155f), y, 0.8543985f);
    const float b = rtr_fma(4.1616724f + y, y, 3.4175940f);
    const float v = a / b;
    const float theta_sintheta = (x > 0.0f) ? v : 0.5f * (1.0f / rtr_sqrt(rtr_max(rtr_fma(-x, x, 1.0f), 1e-7f))) - v;
    return rtr_scale(rtr_cross(v1, v2), theta_sintheta);
}

__device__ __forceinline__ float LTC_Evaluate(rtr_v3 N, rtr_v3 V, rtr_v3 P, bool identity, float4 t1,
                                              const rtr_v3* points, rtr_v3 lightNormal, bool twoSided,
                                              const float* __restrict__ ltc2) {
    const rtr_v3 T1 = rtr_normalize(rtr_sub(V, rtr_scale(N, rtr_dot(V, N))));
    const rtr_v3 T2 = rtr_cross(N, T1);
    rtr_v3 L[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const rtr_v3 w = rtr_sub(points[k], P);
        const rtr_v3 q = rtr_mk(rtr_dot(T1, w), rtr_dot(T2, w), rtr_dot(N, w));
        rtr_v3 l;
        if (identity) l = q;
        else l = rtr_mk(rtr_fma(t1.z, q.z, t1.x * q.x), q.y, rtr_fma(t1.w, q.z, t1.y * q.x));
        L[k] = rtr_normalize(l);
    }
    const rtr_v3 dir = rtr_sub(points[0], P);
    const bool behind = rtr_dot(dir, lightNormal) < 0.0f;
    rtr_v3 vsum = IntegrateEdgeVec(L[0], L[1]);
    vsum = rtr_add(vsum, IntegrateEdgeVec(L[1], L[2]));
    vsum = rtr_add(vsum, IntegrateEdgeVec(L[2], L[0]));
    const float len = rtr_length(vsum);
    float z = vsum.z / len;
    if (behind) z = -z;
    const float uvx = rtr_fma(rtr_fma(z, 0.5f, 0.5f), RTR_LUT_SCALE, RTR_LUT_BIAS);
    const float uvy = rtr_fma(len, RTR_LUT_SCALE, RTR_LUT_BIAS);
    const float4 tex = sample_lut(ltc2, uvx, uvy);
    float sum = len * tex.w;
    if (!behind && !twoSided) sum = 0.0f;
    return sum;
}

/* ---- per-frame arguments -------------------------------------------------------------------- */
struct RenderArgs {
    RtrCameraData cam;        /* 64-B UBO of raygen.rgen:19-21 */
    RtrSceneInfo info;        /* 32-B push constant of raygen.rgen:35-43 */
    uint32_t width, height;   /* full frame */
    uint32_t spp, numShadowRays;
    uint32_t bandRows, shardIndex, shardCount;
    uint32_t localRows;       /* rows of the local (shard) image */
    uint32_t tilesPerRow;     /* ceil(width / 8) */
    uint32_t maxRaysPerSample;/* slots per (pixel,sample) in the wavefront visibility array */
    uint32_t images;          /* RTR_IMG_BIT mask */
    uint32_t accumulate, accumulatedFrames;
    uint32_t directRows;      /* 1: the output images are the FULL frame and this shard writes its rows where they belong (row gy), not into a compact
                               * (localRows x width) image: the parts of rtr_render_split_async, which share one set of images and need no gather */
};

/* Canonical pixel order: 8x8 tiles, row-major inside a tile, tiles left->right inside a band of 8
 * local rows, bands top->bottom.  q -> (x, localRow); returns false for padding. */
__device__ __forceinline__ bool pixel_of(const RenderArgs& ra, uint32_t q, uint32_t& x, uint32_t& lrow, uint32_t& gy) {
    const uint32_t lane = q & 63u, tile = q >> 6;
    const uint32_t tx = tile % ra.tilesPerRow, band8 = tile / ra.tilesPerRow;
    x = tx * 8u + (lane & 7u);
    lrow = band8 * 8u + (lane >> 3);
    const uint32_t lb = lrow / ra.bandRows, r = lrow % ra.bandRows;
    gy = (lb * ra.shardCount + ra.shardIndex) * ra.bandRows + r;
    return x < ra.width && lrow < ra.localRows && gy < ra.height;
}

/* index of pixel (px, local row lrow = global row gy) in the images a launch writes */
__device__ __forceinline__ size_t out_index(const RenderArgs& ra, uint32_t px, uint32_t lrow, uint32_t gy) {
    return (size_t)(ra.directRows ? gy : lrow) * ra.width + px;
}

/* raygen.rgen:83-92 */
__device__ __forceinline__ rtr_v3 primary_dir(const RenderArgs& ra, uint32_t px, uint32_t py, uint32_t i) {
    const float jx = rtr_random(px + i), jy = rtr_random(px + i * 322u);
    const float offx = ((float)px + jx) - 0.5f, offy = ((float)py + jy) - 0.5f;
    const rtr_v3 pw = rtr_madd(rtr_madd(rtr_ld3(ra.cam.topLeftViewportCorner), rtr_ld3(ra.cam.horizontalViewportDelta), offx),
                               rtr_ld3(ra.cam.verticalViewportDelta), offy);
    return rtr_normalize(rtr_sub(pw, rtr_ld3(ra.cam.position)));
}

struct Accum { rtr_v3 analytic, shadowed, unshadowed, avgNormal, avgPosition; };

/* What the light loops need to know about a primary hit: closesthit.rchit:53-106 + raygen.rgen:123-163. */
struct Surface {
    rtr_v3 hitPoint, hitNormal, viewDir, color, mDiffuse, mSpecular;
    float roughness, metallic, om;
    float4 t1, t2;
};

/* Miss / light-hit handling (raygen.rgen:110-121, miss.rmiss:15-27) and the surface fetch.  Returns false when the
 * sample is finished (miss or light).  SHADE == false (shadow-ray generator) skips material and colour work but
 * computes hitPoint / hitNormal with exactly the same operations. */
template <bool SHADE, bool STATS>
__device__ __forceinline__ bool fetch_surface(const DeviceScene& sc, const RenderArgs& ra, const HitRec& h, rtr_v3 rayDir,
                                              bool wantAnalytic, Accum& o, Surface& sf, LocalStats& st) {
    if (h.custom == RTR_MISS) {                                                           /* :110-115, miss.rmiss:15-27 */
        if (SHADE) {
            rtr_v3 sky = rtr_ld3(sc.skyLinear);
            if (sc.hdri.pixels) {
                const rtr_v3 dir = rtr_normalize(rayDir);
                const float hu = rtr_atan2(dir.z, dir.x) / (2.0f * 3.14159265f) + 0.5f;
                float hv = rtr_acos(rtr_clamp(dir.y, -1.0f, 1.0f)) / 3.14159265f;
                hv = 1.0f - hv;
                const float4 tx = sample_tex<STATS>(sc.hdri, hu, hv, st);
                sky = rtr_mk(rtr_to_linear(tx.x), rtr_to_linear(tx.y), rtr_to_linear(tx.z));
            }
            o.analytic = rtr_add(o.analytic, sky); o.unshadowed = rtr_add(o.unshadowed, sky); o.shadowed = rtr_add(o.shadowed, sky);
        }
        return false;
    }
    if (h.custom < sc.numLights) {                                                        /* :116-121 */
        if (SHADE) {
            const rtr_v3 lc = rtr_ld3(sc.lights[h.custom].color);
            o.analytic = rtr_add(o.analytic, lc); o.unshadowed = rtr_add(o.unshadowed, lc); o.shadowed = rtr_add(o.shadowed, lc);
        }
        return false;
    }
    const rtr_v3 camPos = rtr_ld3(ra.cam.position);
    /* closesthit.rchit:53-106 */
    if (STATS) st.hits++;
    const RtrObjectInfo* oi = sc.objects + (h.custom - sc.numLights);
    const uint32_t vOff = oi->vertexOffset, iOff = oi->indexOffset;
    const uint32_t i0 = sc.indices[3u * h.prim + 0u + iOff];
    const uint32_t i1 = sc.indices[3u * h.prim + 1u + iOff];
    const uint32_t i2 = sc.indices[3u * h.prim + 2u + iOff];
    const float4* va = reinterpret_cast<const float4*>(sc.vertices + (i0 + vOff));
    const float4* vb = reinterpret_cast<const float4*>(sc.vertices + (i1 + vOff));
    const float4* vc = reinterpret_cast<const float4*>(sc.vertices + (i2 + vOff));
    const rtr_v3 p0 = f4xyz(va[0]), p1 = f4xyz(vb[0]), p2 = f4xyz(vc[0]);
    const rtr_v3 n0 = f4xyz(va[1]), n1 = f4xyz(vb[1]), n2 = f4xyz(vc[1]);
    const float b0 = 1.0f - h.u - h.v, b1 = h.u, b2 = h.v;
    const rtr_v3 localPos = rtr_madd(rtr_madd(rtr_scale(p0, b0), p1, b1), p2, b2);
    const rtr_v3 hitPoint = rtr_xform_point34(sc.xforms + 12u * h.custom, localPos);
    const rtr_v3 nsum = rtr_madd(rtr_madd(rtr_scale(n0, b0), n1, b1), n2, b2);
    const float* nm = sc.nmats + 12u * h.custom;
    rtr_v3 hitNormal;
    if (rtr_dot(nsum, nsum) > 0.0f) {
        hitNormal = rtr_normalize(rtr_mul33(nm, rtr_normalize(nsum)));
    } else {
        const rtr_v3 g = rtr_cross(rtr_sub(p1, p0), rtr_sub(p2, p0));
        rtr_v3 n = rtr_normalize(rtr_mul33(nm, rtr_normalize(g)));
        if (rtr_dot(n, rayDir) > 0.0f) n = rtr_neg(n);
        hitNormal = n;
    }
    float metallic = oi->metallic;
    float rough = oi->specular;
    rtr_v3 color = rtr_mk(0, 0, 0);
    if (SHADE) {
        rtr_v3 col = rtr_ld3(oi->color);
        if (oi->usesColorMap | oi->usesSpecularMap | oi->usesMetallicMap) {
            const float4 ta = va[2], tb = vb[2], tc = vc[2];                             /* uv in floats 8,9 of the 48-B vertex */
            const float uu = rtr_fma(tc.x, b2, rtr_fma(tb.x, b1, ta.x * b0));
            const float vv = rtr_fma(tc.y, b2, rtr_fma(tb.y, b1, ta.y * b0));
            if (oi->usesColorMap != 0u) { const float4 t = sample_tex<STATS>(sc.textures[oi->colorIndex], uu, vv, st); col = rtr_mk(t.x, t.y, t.z); }
            if (oi->usesSpecularMap != 0u) rough = sample_tex<STATS>(sc.textures[oi->specularIndex], uu, vv, st).x;
            if (oi->usesMetallicMap != 0u) metallic = sample_tex<STATS>(sc.textures[oi->metallicIndex], uu, vv, st).x;
        }
        color = rtr_mk(rtr_to_linear(col.x), rtr_to_linear(col.y), rtr_to_linear(col.z));
    }
    const float roughness = 1.0f - rough;

    const rtr_v3 viewDir = rtr_normalize(rtr_sub(camPos, hitPoint));
    const float om = 1.0f - metallic;
    rtr_v3 mDiffuse = rtr_mk(0, 0, 0), mSpecular = rtr_mk(0, 0, 0);
    float4 t1 = make_float4(1, 0, 0, 1), t2 = make_float4(0, 0, 0, 0);
    if (SHADE) {
        o.avgNormal = rtr_add(o.avgNormal, hitNormal);
        o.avgPosition = rtr_add(o.avgPosition, hitPoint);
        mDiffuse = rtr_scale(color, om);
        mSpecular = rtr_mk(rtr_fma(color.x, metallic, 0.04f * om), rtr_fma(color.y, metallic, 0.04f * om),
                           rtr_fma(color.z, metallic, 0.04f * om));
        if (wantAnalytic) {
            const float dotNV = rtr_clamp(rtr_dot(hitNormal, viewDir), 0.0f, 1.0f);
            const float lu = rtr_fma(roughness, RTR_LUT_SCALE, RTR_LUT_BIAS);
            const float lv = rtr_fma(rtr_sqrt(1.0f - dotNV), RTR_LUT_SCALE, RTR_LUT_BIAS);
            t1 = sample_lut(sc.ltc1, lu, lv);
            t2 = sample_lut(sc.ltc2, lu, lv);
        }
    }
    sf.hitPoint = hitPoint; sf.hitNormal = hitNormal; sf.viewDir = viewDir; sf.color = color;
    sf.mDiffuse = mDiffuse; sf.mSpecular = mSpecular; sf.roughness = roughness; sf.metallic = metallic; sf.om = om;
    sf.t1 = t1; sf.t2 = t2;
    return true;
}

/* One area-light sample's BRDF * L / pdf (raygen.rgen:244-267) and the directional light's BRDF * L (raygen.rgen:316-336): the
 * arithmetic of the light loops, as functions of what a sample needs of its surface point — so that the per-pixel loops (light_loops)
 * and the compacted form of k_resolve_compact (one lane per VISIBLE sample, whichever pixel it belongs to) run the same operations in
 * the same order.  currDiffuse = (om * color) / pi per channel, formed once per surface point. */
__device__ __forceinline__ rtr_v3 surface_diffuse(float om, rtr_v3 color) {
    return rtr_mk((om * color.x) / RTR_PI_F, (om * color.y) / RTR_PI_F, (om * color.z) / RTR_PI_F);
}
__device__ __forceinline__ rtr_v3 area_sample_contrib(rtr_v3 hitNormal, rtr_v3 viewDir, float roughness, rtr_v3 mSpecular, rtr_v3 currDiffuse,
                                                      rtr_v3 lcol, float lintensity, float pdf, rtr_v3 sampledLightDir, float lightDistance) {
    const rtr_v3 halfVector = rtr_normalize(rtr_add(viewDir, sampledLightDir));
    const float cosTheta = rtr_clamp(rtr_dot(viewDir, halfVector), 0.0f, 1.0f);
    const float Dg = GGX_Distribution(hitNormal, halfVector, roughness);
    const float G = GGX_PartialGeometryTerm(viewDir, hitNormal, halfVector, roughness) *
                    GGX_PartialGeometryTerm(sampledLightDir, hitNormal, halfVector, roughness);
    const rtr_v3 F = Fresnel_Schlick(cosTheta, mSpecular);
    const float NdotV = rtr_max(rtr_dot(hitNormal, viewDir), 0.1f);
    const float NdotL = rtr_max(rtr_dot(hitNormal, sampledLightDir), 0.1f);
    /* the products in the order GLSL evaluates them, left to right (raygen.rgen:259-267): (D * F * G) / (4 NdotV NdotL),
     * color * intensity * NdotL * attenuation * 10, BRDF * L / pdf */
    const float den = 4.0f * NdotV * NdotL;
    const rtr_v3 currSpecular = rtr_mk(((Dg * F.x) * G) / den, ((Dg * F.y) * G) / den, ((Dg * F.z) * G) / den);
    const float attenuation = 1.0f / (lightDistance * lightDistance);
    const rtr_v3 BRDF = rtr_add(currSpecular, currDiffuse);
    const rtr_v3 Lr = rtr_mk((((lcol.x * lintensity) * NdotL) * attenuation) * 10.0f, (((lcol.y * lintensity) * NdotL) * attenuation) * 10.0f,
                             (((lcol.z * lintensity) * NdotL) * attenuation) * 10.0f);
    return rtr_mk((BRDF.x * Lr.x) / pdf, (BRDF.y * Lr.y) / pdf, (BRDF.z * Lr.z) / pdf);
}
__device__ __forceinline__ rtr_v3 directional_light_dir() { return rtr_normalize(rtr_mk(-1.0f, 1.0f, -0.5f)); }      /* raygen.rgen:289 */
__device__ __forceinline__ rtr_v3 directional_contrib(rtr_v3 hitNormal, rtr_v3 viewDir, float roughness, rtr_v3 mSpecular, rtr_v3 currDiffuse) {
    const rtr_v3 directLightDir = directional_light_dir();
    const rtr_v3 directLightColor = rtr_mk(1.0f, 1.0f, 0.5f);
    const float directLightIntensity = 0.2f;
    const rtr_v3 halfVector = rtr_normalize(rtr_add(viewDir, directLightDir));
    const float cosTheta = rtr_clamp(rtr_dot(viewDir, halfVector), 0.0f, 1.0f);
    const float Dg = GGX_Distribution(hitNormal, halfVector, roughness);
    const float G = GGX_PartialGeometryTerm(viewDir, hitNormal, halfVector, roughness) *
                    GGX_PartialGeometryTerm(directLightDir, hitNormal, halfVector, roughness);
    const rtr_v3 F = Fresnel_Schlick(cosTheta, mSpecular);
    const float NdotV = rtr_max(rtr_dot(hitNormal, viewDir), 5.0f);
    const float NdotL = rtr_max(rtr_dot(hitNormal, directLightDir), 0.0001f);
    const float den = 4.0f * NdotV * NdotL;
    const rtr_v3 currSpecular = rtr_mk(((Dg * F.x) * G) / den, ((Dg * F.y) * G) / den, ((Dg * F.z) * G) / den);
    const rtr_v3 BRDF = rtr_add(currSpecular, currDiffuse);
    /* directLightColor * directLightIntensity * NdotL * 20, left to right (raygen.rgen:334) */
    const rtr_v3 Lr = rtr_mk(((directLightColor.x * directLightIntensity) * NdotL) * 20.0f, ((directLightColor.y * directLightIntensity) * NdotL) * 20.0f,
                             ((directLightColor.z * directLightIntensity) * NdotL) * 20.0f);
    return rtr_mul(BRDF, Lr);
}
/* the point on light triangle P the sample s of pixel (px, py) of frame `frame` aims at (raygen.rgen:206-215) */
__device__ __forceinline__ rtr_v3 light_sample_pos(const rtr_v3* P, uint32_t s, uint32_t px, uint32_t py, uint32_t frame) {
    const uint32_t seed = s + px * 733u + py * 1933u + frame;
    float r1 = rtr_random(seed), r2 = rtr_random(seed + 100u);
    if (r1 + r2 > 1.0f) { r1 = 1.0f - r1; r2 = 1.0f - r2; }
    return rtr_madd(rtr_madd(P[0], rtr_sub(P[1], P[0]), r1), rtr_sub(P[2], P[0]), r2);
}

/* The light loops of raygen.rgen:165-338 for one shaded surface point.  Policy::occluded(origin, dir, tmax, rawDir, intoSurface) (rawDir: dir before normalisation, only its signs are meaningful; intoSurface: dot(hitNormal, rawDir) < 0) answers the
 * shadow query; Policy::kShade == false (counting / emitting the queries) skips the BRDF arithmetic but keeps the exact
 * sequence of queries. */
#ifndef RTR_INTO_EXPR
#define RTR_INTO_EXPR (rtr_dot(hitNormal, lightVec) < 0.0f)
#endif
template <class Policy, bool STATS>
__device__ __forceinline__ void light_loops(const DeviceScene& sc, const RenderArgs& ra, uint32_t px, uint32_t py,
                                            const Surface& sf, uint32_t want, Accum& o, Policy& pol, LocalStats& st) {
    /* want: bit 0 = the analytic (LTC) image is an output, bit 1 = the unshadowed image is.  When the unshadowed image is not
     * asked for, an occluded sample contributes contrib * 0 to the only sum that is kept, so its BRDF is not evaluated here.  The
     * reference (and the oracle) evaluate it and multiply by 0: identical whenever contrib is finite; an occluded sample whose
     * contrib overflows poisons the reference's pixel (0 * inf = NaN) and not this one — divergence D6, DESIGN.md §4. */
    const bool wantAnalytic = (want & 1u) != 0u, wantUnshadowed = (want & 2u) != 0u;
    const rtr_v3 hitPoint = sf.hitPoint, hitNormal = sf.hitNormal, viewDir = sf.viewDir, color = sf.color;
    const rtr_v3 mDiffuse = sf.mDiffuse, mSpecular = sf.mSpecular;
    const float roughness = sf.roughness, om = sf.om;
    const float4 t1 = sf.t1, t2 = sf.t2;
    const rtr_v3 shadowOrigin = rtr_madd(hitPoint, hitNormal, 0.01f);
    const rtr_v3 currDiffuse = surface_diffuse(om, color);

    for (uint32_t li = 0; li < ra.info.numAreaLights; ++li) {                             /* :165 */
        const RtrAreaLightInfo* L = sc.lights + li;
        if (STATS) st.lightFetch++;
        const rtr_v3 lcol = rtr_ld3(L->color);
        const float lintensity = L->intensity;
        const uint32_t lnt = L->numTriangles;
        const bool twoSided = L->isTwoSided != 0u;
        for (uint32_t ti = 0; ti < lnt; ++ti) {                                           /* :172 */
            if (STATS) st.lightTriFetch++;
            /* corners, area, pdf and normal of the light triangle are the same for every pixel: k_light_tris computed them once
             * (raygen.rgen:174-196, same operations) and the addresses are wave-uniform, so these are scalar loads */
            const float4* rec = sc.lightTris + (size_t)(sc.lightTriFirst[li] + ti) * kLightTriRecord;
            const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
            rtr_v3 P[3];
            P[0] = rtr_mk(r0.x, r0.y, r0.z); P[1] = rtr_mk(r1.x, r1.y, r1.z); P[2] = rtr_mk(r2.x, r2.y, r2.z);
            const float pdf = r1.w;
            const rtr_v3 lightNormal = rtr_mk(r3.x, r3.y, r3.z);
            if (!twoSided) {
                if (rtr_dot(lightNormal, rtr_sub(hitPoint, P[0])) < 0.0f) continue;
            }
            rtr_v3 shadowedSample = rtr_mk(0, 0, 0), unshadowedSample = rtr_mk(0, 0, 0);
            for (uint32_t s = 0; s < ra.numShadowRays; ++s) {                             /* :206 */
                const rtr_v3 lightSamplePos = light_sample_pos(P, s, px, py, ra.info.frame);
                const rtr_v3 lightVec = rtr_sub(lightSamplePos, hitPoint);
                const rtr_v3 sampledLightDir = rtr_normalize(lightVec);
                const float lightDistance = rtr_length(lightVec);
                /* a ray that leaves its surface point INTO the surface nearly always re-enters the triangle it starts 0.01 above: the any-hit
                 * kernel tests that triangle's leaf first (k_shadow_trace4's refill).  The directional ray below is only sent with the light in front. */
                const bool occ = pol.occluded(shadowOrigin, sampledLightDir, lightDistance - 0.5f, lightVec, RTR_INTO_EXPR);
                if (Policy::kShade && (wantUnshadowed || !occ)) {
                    const float currShadow = occ ? 0.0f : 1.0f;
                    const rtr_v3 contrib = area_sample_contrib(hitNormal, viewDir, roughness, mSpecular, currDiffuse, lcol, lintensity, pdf, sampledLightDir, lightDistance);
                    shadowedSample = rtr_madd(shadowedSample, contrib, currShadow);
                    unshadowedSample = rtr_add(unshadowedSample, contrib);
                }
            }
            if (Policy::kShade) {
                const float ns = (float)ra.numShadowRays;
                shadowedSample = rtr_mk(shadowedSample.x / ns, shadowedSample.y / ns, shadowedSample.z / ns);
                unshadowedSample = rtr_mk(unshadowedSample.x / ns, unshadowedSample.y / ns, unshadowedSample.z / ns);
                if (wantAnalytic) {
                    const float diffuse = LTC_Evaluate(hitNormal, viewDir, hitPoint, true, t1, P, lightNormal, twoSided, sc.ltc2);
                    const float spec = LTC_Evaluate(hitNormal, viewDir, hitPoint, false, t1, P, lightNormal, twoSided, sc.ltc2);
                    const rtr_v3 fres = rtr_mk(rtr_fma(1.0f - mSpecular.x, t2.y, mSpecular.x * t2.x),
                                               rtr_fma(1.0f - mSpecular.y, t2.y, mSpecular.y * t2.x),
                                               rtr_fma(1.0f - mSpecular.z, t2.y, mSpecular.z * t2.x));
                    /* color * intensity * (specular + mDiffuse * diffuse) * 5, left to right (raygen.rgen:283) */
                    o.analytic = rtr_add(o.analytic, rtr_mk(((lcol.x * lintensity) * rtr_fma(mDiffuse.x, diffuse, spec * fres.x)) * 5.0f,
                                                            ((lcol.y * lintensity) * rtr_fma(mDiffuse.y, diffuse, spec * fres.y)) * 5.0f,
                                                            ((lcol.z * lintensity) * rtr_fma(mDiffuse.z, diffuse, spec * fres.z)) * 5.0f));
                }
                o.shadowed = rtr_add(o.shadowed, shadowedSample);
                o.unshadowed = rtr_add(o.unshadowed, unshadowedSample);
            }
        }
    }
    /* directional light, raygen.rgen:289-338 */
    const rtr_v3 directLightDir = directional_light_dir();
    if (rtr_dot(hitNormal, directLightDir) <= 0.0f) return;
    const bool occ = pol.occluded(shadowOrigin, directLightDir, 10000.0f, directLightDir, false);
    if (Policy::kShade && (wantUnshadowed || wantAnalytic || !occ)) {
        const float currShadow = occ ? 0.0f : 1.0f;
        const rtr_v3 contrib = directional_contrib(hitNormal, viewDir, roughness, mSpecular, currDiffuse);
        o.shadowed = rtr_madd(o.shadowed, contrib, currShadow);
        o.unshadowed = rtr_add(o.unshadowed, contrib);
        o.analytic = rtr_add(o.analytic, contrib);
    }
}

/* The per-light-triangle part of raygen.rgen:174-196, hoisted out of the pixel loops (see light_loops). */
__device__ __forceinline__ void light_tri_record(const RtrAreaLightInfo* L, const RtrVertex* vertices, const uint32_t* indices,
                                                 uint32_t ti, float4* rec) {
    const uint32_t lvOff = L->vertexOffset, liOff = L->indexOffset;
    const uint32_t j0 = indices[ti * 3u + 0u + liOff];
    const uint32_t j1 = indices[ti * 3u + 1u + liOff];
    const uint32_t j2 = indices[ti * 3u + 2u + liOff];
    rtr_v3 P[3];
    P[0] = rtr_xform_point44cm(L->transform, rtr_ld3(vertices[j0 + lvOff].position));
    P[1] = rtr_xform_point44cm(L->transform, rtr_ld3(vertices[j1 + lvOff].position));
    P[2] = rtr_xform_point44cm(L->transform, rtr_ld3(vertices[j2 + lvOff].position));
    rtr_v3 lightNormal = rtr_cross(rtr_sub(P[2], P[1]), rtr_sub(P[0], P[1]));
    const float area = rtr_length(lightNormal) * 0.5f;
    const float pdf = 1.0f / (area * 0.7f);
    lightNormal = rtr_normalize(lightNormal);
    rec[0] = make_float4(P[0].x, P[0].y, P[0].z, area);
    rec[1] = make_float4(P[1].x, P[1].y, P[1].z, pdf);
    rec[2] = make_float4(P[2].x, P[2].y, P[2].z, 0.f);
    rec[3] = make_float4(lightNormal.x, lightNormal.y, lightNormal.z, 0.f);
}

/* One primary sample's contribution: reference raygen.rgen:110-338 (+ closesthit.rchit:45-110, miss.rmiss:15-27). */
template <class Policy, bool STATS>
__device__ __forceinline__ void shade_sample(const DeviceScene& sc, const RenderArgs& ra, uint32_t px, uint32_t py,
                                             const HitRec& h, rtr_v3 rayDir, uint32_t want, Accum& o,
                                             Policy& pol, LocalStats& st) {
    Surface sf;
    if (!fetch_surface<Policy::kShade, STATS>(sc, ra, h, rayDir, (want & 1u) != 0u, o, sf, st)) return;
    light_loops<Policy, STATS>(sc, ra, px, py, sf, want, o, pol, st);
}

__device__ __forceinline__ uint32_t tonemap_pack(rtr_v3 c) {                              /* raygen.rgen:345-357 */
    return rtr_pack_bgra8(rtr_to_srgb(rtr_aces(c.x)), rtr_to_srgb(rtr_aces(c.y)), rtr_to_srgb(rtr_aces(c.z)));
}

struct FrameOut {
    uint32_t* img[8];     /* indexed by rtr_image binding; null when absent */
    float4* hdr;
};

/* Several frames in ONE launch of every kernel of the staged pipeline (rtr_render_batch_async): a frame of the batch is a further
 * run of sample planes — plane = frame * spp + sample, pixel-sample slot k = plane * planeStride + q — so the camera-ray kernel's
 * grid, the queue and the any-hit kernel's work grow with the batch while the launches per frame shrink.  What differs between the
 * frames (camera, seed, accumulation state, output images) is looked up per wave: planeStride is a multiple of 64, so a wave never
 * straddles two frames.  Why: a 1-spp frame (let alone a 1/8 shard of one) is too little work per launch for the latency-bound
 * kernels — the camera-ray kernel takes 0.34 ms for one frame's rays and 0.41 ms for four times as many. */
constexpr uint32_t kMaxBatch = 32;     /* = RTR_MAX_BATCH (include/rtr.h).  Thirty-two frames' arguments are 7 KB passed by value: this stack takes 64 KB and more
                                        * (profiles/microbench/kernarg_size.hip: launches with 2 ... 64 KB of arguments run and read them right), and the size costs
                                        * nothing at launch (profiles/r03/ab_kernarg_batch16_build.log) */
struct FrameBatch {
    RenderArgs ra[kMaxBatch];
    FrameOut   fo[kMaxBatch];
    uint32_t   n;
};
static_assert(sizeof(FrameBatch) + sizeof(DeviceScene) + 256 <= 16384, "a launch passes DeviceScene + FrameBatch + a few pointers by value");
/* lane g of a launch over the whole batch -> frame b (wave-uniform) and pixel slot q inside the frame */
__device__ __forceinline__ uint32_t batch_frame(uint32_t g, uint32_t planeStride, uint32_t& q) {
    const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)(g / planeStride));
    q = g - b * planeStride;
    return b;
}

/* raygen.rgen:341-364 + the HDR accumulation extension */
__device__ __forceinline__ void write_pixel(const RenderArgs& ra, const FrameOut& fo, size_t p, Accum o) {
    const float n = (float)ra.spp;
    rtr_v3 sh = rtr_mk(o.shadowed.x / n, o.shadowed.y / n, o.shadowed.z / n);
    const rtr_v3 un = rtr_mk(o.unshadowed.x / n, o.unshadowed.y / n, o.unshadowed.z / n);
    const rtr_v3 an = rtr_mk(o.analytic.x / n, o.analytic.y / n, o.analytic.z / n);
    if (fo.hdr) {
        if (ra.accumulate) {
            float4 h = fo.hdr[p];
            h.x += sh.x; h.y += sh.y; h.z += sh.z; h.w += 1.0f;
            fo.hdr[p] = h;
            const float inv = (float)(ra.accumulatedFrames + 1u);
            sh = rtr_mk(h.x / inv, h.y / inv, h.z / inv);
        } else {
            fo.hdr[p] = make_float4(sh.x, sh.y, sh.z, 1.0f);
        }
    }
    if (fo.img[1]) fo.img[1][p] = tonemap_pack(sh);
    if (fo.img[2]) fo.img[2][p] = tonemap_pack(un);
    if (fo.img[0]) fo.img[0][p] = tonemap_pack(an);
    if (fo.img[6]) {
        const rtr_v3 a = rtr_normalize(rtr_mk(o.avgNormal.x / n, o.avgNormal.y / n, o.avgNormal.z / n));
        fo.img[6][p] = rtr_pack_bgra8(a.x, a.y, a.z);
    }
    if (fo.img[7]) fo.img[7][p] = rtr_pack_bgra8(o.avgPosition.x / n, o.avgPosition.y / n, o.avgPosition.z / n);
}

/* write_pixel for a launch whose only outputs are the framebuffer (img[1]) and the HDR accumulator: the same operations on the
 * shadowed sum, in the same order */
__device__ __forceinline__ void write_pixel_framebuffer(const RenderArgs& ra, const FrameOut& fo, size_t p, rtr_v3 shadowed) {
    const float n = (float)ra.spp;
    rtr_v3 sh = rtr_mk(shadowed.x / n, shadowed.y / n, shadowed.z / n);
    if (fo.hdr) {
        if (ra.accumulate) {
            float4 h = fo.hdr[p];
            h.x += sh.x; h.y += sh.y; h.z += sh.z; h.w += 1.0f;
            fo.hdr[p] = h;
            const float inv = (float)(ra.accumulatedFrames + 1u);
            sh = rtr_mk(h.x / inv, h.y / inv, h.z / inv);
        } else {
            fo.hdr[p] = make_float4(sh.x, sh.y, sh.z, 1.0f);
        }
    }
    if (fo.img[1]) fo.img[1][p] = tonemap_pack(sh);
}

}  // namespace rtrdev
