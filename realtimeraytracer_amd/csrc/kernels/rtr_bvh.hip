/* rtr_bvh.hip — BVH build and refit ON the device.
 *
 * Replaces what the reference leaves to the driver: vkCmdBuildAccelerationStructuresKHR for BLAS/TLAS
 * (src/vulkan/raytracing/blas.cppm:113-160, tlas.cppm:112-149) and TLAS::updateTransform / refit
 * (tlas.cppm:151-207 — present in the reference, never called by its app).
 *
 *   build  : LBVH (Karras 2012).  k_world_prims flattens instances to world space (same arithmetic as the host
 *            packer) and reduces centroid bounds; 30-bit Morton code | primitive index as a unique 64-bit key;
 *            rocPRIM/hipCUB radix sort (a plain library sort, the one step that is not hand-written);
 *            k_karras builds the radix tree, one lane per internal node; subtrees of <= 4 primitives collapse into
 *            leaves (Morton-sorted primitives of a subtree are contiguous, so a leaf is (first,count) as in the
 *            host builder); k_fit fits the child boxes bottom-up.
 *   refit  : k_world_prims again with new transforms into the existing leaf order, then k_fit on the unchanged
 *            topology (works for host-SAH-built and device-LBVH-built trees alike).
 *
 * k_fit: one lane per inner node fills the slots of its leaf children, then climbs: an agent-scope
 * fence + atomic counter per node lets exactly the second arriver continue with both child boxes visible
 * (cdna guide G16: visibility comes from the release/acquire pair, not from placement).  Every climb ends at the
 * root or at a node whose sibling has not arrived, so all waves exit.
 * Output layout = RTR_BVH_LAYOUT_VERSION 2 (64-B children-in-parent nodes, 48-B {v0,e1,e2} records), boxes padded
 * outwards by 2^-18 of the largest coordinate like the host builder, so traversal results are identical whichever
 * builder made the tree.
 */
#include "rtr_bvh.h"

#include <hipcub/hipcub.hpp>

#include "../../../include/rtr_math.h"

namespace rtrdev {

constexpr int kB = 256;
constexpr uint32_t kLeafMax = 4;

__device__ __forceinline__ uint32_t f2ord(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__host__ __device__ inline float ord2f(uint32_t u) {
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f; memcpy(&f, &u, 4); return f;
}

/* scene reduction words: [0..2] centroid min (ordered uint), [3..5] centroid max, [6] max |coordinate| (float bits, >= 0) */
__global__ __launch_bounds__(kB) void k_world_prims(BvhInputs in, uint32_t n, const uint32_t* __restrict__ slotOfPrim,
                                                    float4* __restrict__ triOut, float4* __restrict__ boxMin, float4* __restrict__ boxMax,
                                                    uint32_t* __restrict__ red) {
    const uint32_t p = blockIdx.x * kB + threadIdx.x;
    float cmin[3] = {3.0e38f, 3.0e38f, 3.0e38f}, cmax[3] = {-3.0e38f, -3.0e38f, -3.0e38f}, mabs = 0.f;
    if (p < n) {
        const PrimRef pr = in.prims[p];
        const InstanceRef ir = in.instances[pr.customIndex];
        rtr_v3 w[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint32_t idx = in.indices[ir.indexOffset + 3u * pr.primitiveId + k] + ir.vertexOffset;
            w[k] = rtr_xform_point34(ir.transform, rtr_ld3(in.vertices[idx].position));
        }
        const rtr_v3 e1 = rtr_sub(w[1], w[0]), e2 = rtr_sub(w[2], w[0]);
        const uint32_t s = slotOfPrim ? slotOfPrim[p] : p;          /* refit writes straight into leaf order */
        triOut[(size_t)s * 3 + 0] = make_float4(w[0].x, w[0].y, w[0].z, __uint_as_float(pr.customIndex));
        triOut[(size_t)s * 3 + 1] = make_float4(e1.x, e1.y, e1.z, __uint_as_float(pr.primitiveId));
        triOut[(size_t)s * 3 + 2] = make_float4(e2.x, e2.y, e2.z, __uint_as_float(pr.flags));
        const float mn[3] = {fminf(fminf(w[0].x, w[1].x), w[2].x), fminf(fminf(w[0].y, w[1].y), w[2].y), fminf(fminf(w[0].z, w[1].z), w[2].z)};
        const float mx[3] = {fmaxf(fmaxf(w[0].x, w[1].x), w[2].x), fmaxf(fmaxf(w[0].y, w[1].y), w[2].y), fmaxf(fmaxf(w[0].z, w[1].z), w[2].z)};
        boxMin[s] = make_float4(mn[0], mn[1], mn[2], 0.f);
        boxMax[s] = make_float4(mx[0], mx[1], mx[2], 0.f);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float c = 0.5f * (mn[k] + mx[k]);
            cmin[k] = c; cmax[k] = c;
            mabs = fmaxf(mabs, fmaxf(fabsf(mn[k]), fabsf(mx[k])));
        }
    }
    /* wave reduction, then one atomic per wave and word */
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { cmin[k] = fminf(cmin[k], __shfl_xor(cmin[k], o)); cmax[k] = fmaxf(cmax[k], __shfl_xor(cmax[k], o)); }
        mabs = fmaxf(mabs, __shfl_xor(mabs, o));
    }
    if ((threadIdx.x & 63u) == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { atomicMin(&red[k], f2ord(cmin[k])); atomicMax(&red[3 + k], f2ord(cmax[k])); }
        atomicMax(&red[6], __float_as_uint(mabs));
    }
}

__device__ __forceinline__ uint32_t expand10(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ __launch_bounds__(kB) void k_morton(uint32_t n, const float4* __restrict__ boxMin, const float4* __restrict__ boxMax,
                                               const uint32_t* __restrict__ red, unsigned long long* __restrict__ keys) {
    const uint32_t p = blockIdx.x * kB + threadIdx.x;
    if (p >= n) return;
    const float4 a = boxMin[p], b = boxMax[p];
    const float c[3] = {0.5f * (a.x + b.x), 0.5f * (a.y + b.y), 0.5f * (a.z + b.z)};
    uint32_t q[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float lo = ord2f(red[k]), hi = ord2f(red[3 + k]);
        const float ext = hi - lo;
        float t = ext > 0.f ? (c[k] - lo) / ext : 0.f;
        t = fminf(fmaxf(t * 1024.0f, 0.0f), 1023.0f);
        q[k] = (uint32_t)t;
    }
    const uint32_t m = (expand10(q[0]) << 2) | (expand10(q[1]) << 1) | expand10(q[2]);
    keys[p] = ((unsigned long long)m << 32) | p;
}

/* gather the canonical-order records into Morton order and remember where every canonical primitive went */
__global__ __launch_bounds__(kB) void k_gather(uint32_t n, const unsigned long long* __restrict__ keys,
                                               const float4* __restrict__ triIn, const float4* __restrict__ minIn, const float4* __restrict__ maxIn,
                                               float4* __restrict__ triOut, float4* __restrict__ minOut, float4* __restrict__ maxOut,
                                               uint32_t* __restrict__ slotOfPrim) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = (uint32_t)(keys[i] & 0xffffffffull);
    triOut[(size_t)i * 3 + 0] = triIn[(size_t)p * 3 + 0];
    triOut[(size_t)i * 3 + 1] = triIn[(size_t)p * 3 + 1];
    triOut[(size_t)i * 3 + 2] = triIn[(size_t)p * 3 + 2];
    minOut[i] = minIn[p]; maxOut[i] = maxIn[p];
    slotOfPrim[p] = i;
}

/* Karras 2012: longest common prefix of the (unique) keys i and j, -1 outside [0,n) */
__device__ __forceinline__ int lcp(const unsigned long long* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));
}

__device__ __forceinline__ int32_t leaf_code(uint32_t first, uint32_t count) { return (int32_t)~((first << 3) | (count - 1u)); }

/* one lane per internal node: range, split, children; child codes with <=4-primitive subtrees collapsed to leaves */
__global__ __launch_bounds__(kB) void k_karras(int n, const unsigned long long* __restrict__ keys, int2* __restrict__ range,
                                               int2* __restrict__ rawChild /* index, with bit 31 = primitive leaf */) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (lcp(keys, n, i, i + 1) - lcp(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = lcp(keys, n, i, i - d);
    int lmax = 2;
    while (lcp(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (lcp(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = lcp(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
        if (lcp(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    range[i] = make_int2(lo, hi);
    rawChild[i] = make_int2(lo == gamma ? (gamma | (int)0x80000000) : gamma, hi == gamma + 1 ? ((gamma + 1) | (int)0x80000000) : gamma + 1);
}

__global__ __launch_bounds__(kB) void k_mark_unused(uint32_t numNodes, int32_t* __restrict__ parent) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i < numNodes) parent[i] = -2;
}

/* child codes in the final node array + parent links for the climb */
__global__ __launch_bounds__(kB) void k_emit(int n, const int2* __restrict__ range, const int2* __restrict__ rawChild,
                                             float4* __restrict__ nodes, int32_t* __restrict__ parent) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n - 1) return;
    const int2 r = range[i];
    if ((uint32_t)(r.y - r.x + 1) <= kLeafMax && i != 0) return;        /* inside / root of a collapsed subtree: never referenced */
    const int2 rc = rawChild[i];
    int32_t code[2];
    const int raw[2] = {rc.x, rc.y};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (raw[s] < 0) code[s] = leaf_code((uint32_t)(raw[s] & 0x7fffffff), 1u);
        else {
            const int2 cr = range[raw[s]];
            const uint32_t cnt = (uint32_t)(cr.y - cr.x + 1);
            if (cnt <= kLeafMax) code[s] = leaf_code((uint32_t)cr.x, cnt);
            else { code[s] = raw[s]; parent[raw[s]] = (i << 1) | s; }
        }
    }
    int4 w = make_int4(code[0], code[1], 0, 0);
    nodes[(size_t)i * 4 + 3] = *reinterpret_cast<float4*>(&w);
    if (i == 0) parent[0] = -1;
}

/* bottom-up fit (build and refit).  counters must be zero on entry; depth[] gets the inner-node height. */
__global__ __launch_bounds__(kB) void k_fit(uint32_t numNodes, float4* nodes, const float4* __restrict__ boxMin, const float4* __restrict__ boxMax,
                                            const int32_t* __restrict__ parent, uint32_t* counters, uint32_t* depth, const uint32_t* __restrict__ red,
                                            uint32_t* maxDepthOut) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= numNodes) return;
    const int32_t par0 = parent[i];
    if (par0 == -2) return;                                   /* slot of the node array that is not part of the tree */
    const float pad = fmaxf(__uint_as_float(red[6]), 1e-6f) * 3.814697265625e-06f;
    float* nf = reinterpret_cast<float*>(nodes + (size_t)i * 4);
    const int2 ch = *reinterpret_cast<const int2*>(nodes + (size_t)i * 4 + 3);
    uint32_t filled = 0;
    const int32_t code[2] = {ch.x, ch.y};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (code[s] >= 0) continue;
        const uint32_t c = (uint32_t)~code[s];
        const uint32_t first = c >> 3, cnt = (c & 7u) + 1u;
        float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        for (uint32_t k = 0; k < cnt; ++k) {
            const float4 a = boxMin[first + k], b = boxMax[first + k];
            mn[0] = fminf(mn[0], a.x); mn[1] = fminf(mn[1], a.y); mn[2] = fminf(mn[2], a.z);
            mx[0] = fmaxf(mx[0], b.x); mx[1] = fmaxf(mx[1], b.y); mx[2] = fmaxf(mx[2], b.z);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { nf[6 * s + k] = mn[k] - pad; nf[6 * s + 3 + k] = mx[k] + pad; }
        ++filled;
    }
    if (filled == 0) return;                                  /* two inner children: their climbs complete this node */
    uint32_t cur = i;
    uint32_t add = filled;
    for (;;) {
        __threadfence();                                      /* release my slot writes (agent scope) */
        const uint32_t before = atomicAdd(&counters[cur], add);
        if (before + add < 2u) return;                        /* sibling subtree not finished: its lane will continue */
        __threadfence();                                      /* acquire the sibling's writes */
        float* cf = reinterpret_cast<float*>(nodes + (size_t)cur * 4);
        const int2 cc = *reinterpret_cast<const int2*>(nodes + (size_t)cur * 4 + 3);
        const uint32_t dl = cc.x >= 0 ? __hip_atomic_load(&depth[cc.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        const uint32_t dr = cc.y >= 0 ? __hip_atomic_load(&depth[cc.y], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        const uint32_t dcur = 1u + (dl > dr ? dl : dr);
        __hip_atomic_store(&depth[cur], dcur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int32_t p = parent[cur];
        if (p < 0) { *maxDepthOut = dcur; return; }           /* root done */
        const uint32_t pi = (uint32_t)p >> 1, ps = (uint32_t)p & 1u;
        float* pf = reinterpret_cast<float*>(nodes + (size_t)pi * 4);
        /* my box = union of my two child boxes (volatile-style loads through the agent-scope path) */
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float a = __hip_atomic_load(&cf[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float b = __hip_atomic_load(&cf[6 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float c = __hip_atomic_load(&cf[3 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float d = __hip_atomic_load(&cf[9 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&pf[6 * ps + k], fminf(a, b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&pf[6 * ps + 3 + k], fmaxf(c, d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        cur = pi; add = 1u;
    }
}

/* scene grid from the root's two child boxes (one lane) */
__global__ void k_grid(const float4* __restrict__ nodesF, RtrBvhGrid* grid) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const float* f = reinterpret_cast<const float*>(nodesF);
    float mn[3], mx[3];
    for (int k = 0; k < 3; ++k) { mn[k] = fminf(f[k], f[6 + k]); mx[k] = fmaxf(f[3 + k], f[9 + k]); }
    RtrBvhGrid g = {};
    rtr_grid_from_bounds(mn, mx, g.origin, g.scale);
    *grid = g;
}

/* fp32 planes -> 16-bit grid coordinates, rounded outward (same arithmetic as rtr::quantize_nodes on the host) */
__global__ __launch_bounds__(kB) void k_quantize(uint32_t numNodes, const float4* __restrict__ nodesF, const int32_t* __restrict__ parent,
                                                 const RtrBvhGrid* __restrict__ grid, uint4* __restrict__ nodes) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= numNodes) return;
    uint4 w0 = make_uint4(0, 0, 0, 0), w1 = make_uint4(0, 0, 0, 0);
    if (parent[i] != -2) {
        const float4 a = nodesF[(size_t)i * 4], b = nodesF[(size_t)i * 4 + 1], c = nodesF[(size_t)i * 4 + 2];
        const int4 d = *reinterpret_cast<const int4*>(nodesF + (size_t)i * 4 + 3);
        const float ox = grid->origin[0], oy = grid->origin[1], oz = grid->origin[2];
        const float sx = grid->scale[0], sy = grid->scale[1], sz = grid->scale[2];
        /* a = (lminx,lminy,lminz,lmaxx)  b = (lmaxy,lmaxz,rminx,rminy)  c = (rminz,rmaxx,rmaxy,rmaxz) */
        w0.x = rtr_quant_lo(a.x, ox, sx) | (rtr_quant_lo(a.y, oy, sy) << 16);
        w0.y = rtr_quant_hi(a.w, ox, sx) | (rtr_quant_hi(b.x, oy, sy) << 16);
        w0.z = rtr_quant_lo(b.z, ox, sx) | (rtr_quant_lo(b.w, oy, sy) << 16);
        w0.w = rtr_quant_hi(c.y, ox, sx) | (rtr_quant_hi(c.z, oy, sy) << 16);
        w1.x = rtr_quant_lo(a.z, oz, sz) | (rtr_quant_hi(b.y, oz, sz) << 16);
        w1.y = rtr_quant_lo(c.x, oz, sz) | (rtr_quant_hi(c.w, oz, sz) << 16);
        w1.z = (uint32_t)d.x; w1.w = (uint32_t)d.y;
    }
    nodes[(size_t)i * 2] = w0; nodes[(size_t)i * 2 + 1] = w1;
}

/* f16 bits of a signed plane offset v (scene-grid steps from the scene's wide centre, |v| <= 65535), rounded towards -inf / +inf to
 * the 11 significant bits of a half float; a magnitude past the largest half float (65504) rounds away from zero to infinity and
 * towards zero to 65504.  Integer arithmetic, so a host restatement agrees bit for bit. */
__host__ __device__ inline uint32_t rtr_f16_bits_of_int(uint32_t a) {      /* a has at most 11 significant bits; a > 65504 (only 65536 can arrive) -> inf */
    if (a == 0) return 0u;
    if (a > 65504u) return 0x7c00u;
    const int e = 31 - __builtin_clz(a);                                   /* a = 1.m * 2^e, e <= 15 */
    const uint32_t m = (e >= 10) ? (a >> (e - 10)) : (a << (10 - e));      /* 11 bits, leading one at bit 10 */
    return ((uint32_t)(e + 15) << 10) | (m & 0x3ffu);
}
__host__ __device__ inline uint32_t rtr_f16_mag_down(uint32_t a) {          /* largest representable <= a */
    if (a <= 2048u) return a;
    if (a > 65504u) return 65504u;
    const int sh = (31 - __builtin_clz(a)) - 10;
    return (a >> sh) << sh;
}
__host__ __device__ inline uint32_t rtr_f16_mag_up(uint32_t a) {            /* smallest representable >= a */
    if (a <= 2048u) return a;
    const int sh = (31 - __builtin_clz(a)) - 10;
    return ((a + (1u << sh) - 1u) >> sh) << sh;
}
__host__ __device__ inline uint32_t rtr_f16_floor_bits(int32_t v) {         /* towards -inf */
    return v >= 0 ? rtr_f16_bits_of_int(rtr_f16_mag_down((uint32_t)v)) : (0x8000u | rtr_f16_bits_of_int(rtr_f16_mag_up((uint32_t)(-v))));
}
__host__ __device__ inline uint32_t rtr_f16_ceil_bits(int32_t v) {          /* towards +inf */
    return v >= 0 ? rtr_f16_bits_of_int(rtr_f16_mag_up((uint32_t)v)) : (0x8000u | rtr_f16_bits_of_int(rtr_f16_mag_down((uint32_t)(-v))));
}

/* The wide centre of a tree: the grid coordinate c (per axis) the 4-wide records' half-float planes are offsets from.  Half floats
 * are exact within 2048 steps of c and lose a bit per doubling beyond, so c decides where the boxes stay tight.  Two candidates per
 * axis, both from the tree's leaf boxes (integer sums in 64-bit atomics: the same bytes on every run):
 *   mean : the mean midpoint of the leaf boxes — tight where most leaves are;
 *   flat : the 4096-step window holding the most face area of leaves that are FLAT on this axis (extent <= 2 steps: floors, walls).
 *          A flat leaf whose planes move outward by more than the 0.01 a shadow ray is lifted off its surface swallows the origin
 *          of every ray leaving that surface; on the bunny-class scene (a small object on a large ground plane, planes about the
 *          grid's own centre) that cost +47 % node visits and +125 % triangle tests per shadow ray.
 * The flat candidate is taken on an axis where it cuts the area-weighted relative inflation of the leaf boxes (capped at the
 * box's own extent) to a quarter or less of the mean candidate's — measured choices and counts in profiles/r02/wide_centre.log. */
constexpr uint32_t kCentreBins = 512;                 /* of 128 grid steps */
constexpr uint32_t kCentreWindow = 32;                /* bins: the +-2048 steps a half float holds exactly */
constexpr uint32_t kCentreWords = 4 + 3 * kCentreBins + 6 + 6;    /* sums[4], flat-area histograms, candidates [axis][2], costs [axis][2] */

struct LeafBox { uint32_t lo[3], hi[3]; };
__device__ __forceinline__ int leaf_boxes(uint32_t i, uint32_t numNodes, const uint4* __restrict__ nodes, const int32_t* __restrict__ parent, LeafBox out[2]) {
    int n = 0;
    if (i < numNodes && (!parent || parent[i] != -2)) {
        const uint4 a = nodes[(size_t)i * 2], b = nodes[(size_t)i * 2 + 1];
        if ((int32_t)b.z < 0) { LeafBox& l = out[n++]; l.lo[0] = a.x & 0xffffu; l.lo[1] = a.x >> 16; l.lo[2] = b.x & 0xffffu; l.hi[0] = a.y & 0xffffu; l.hi[1] = a.y >> 16; l.hi[2] = b.x >> 16; }
        if ((int32_t)b.w < 0) { LeafBox& l = out[n++]; l.lo[0] = a.z & 0xffffu; l.lo[1] = a.z >> 16; l.lo[2] = b.y & 0xffffu; l.hi[0] = a.w & 0xffffu; l.hi[1] = a.w >> 16; l.hi[2] = b.y >> 16; }
    }
    return n;
}
__device__ __forceinline__ unsigned long long wave_total(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

__global__ __launch_bounds__(kB) void k_wide_centre_sum(uint32_t numNodes, const uint4* __restrict__ nodes, const int32_t* __restrict__ parent,
                                                        unsigned long long* __restrict__ w) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    LeafBox lb[2];
    const int nl = leaf_boxes(i, numNodes, nodes, parent, lb);
    unsigned long long s[4] = {0, 0, 0, 0};
    for (int j = 0; j < nl; ++j) {
        for (int k = 0; k < 3; ++k) {
            s[k] += lb[j].lo[k] + lb[j].hi[k];
            if (lb[j].hi[k] - lb[j].lo[k] <= 2u) {          /* flat on axis k: its face area goes to the bins of its two planes */
                const int a = (k + 1) % 3, b = (k + 2) % 3;
                const unsigned long long area = (((unsigned long long)(lb[j].hi[a] - lb[j].lo[a]) * (lb[j].hi[b] - lb[j].lo[b])) >> 8) + 1ull;
                atomicAdd(w + 4 + k * kCentreBins + (lb[j].lo[k] >> 7), area);
                atomicAdd(w + 4 + k * kCentreBins + (lb[j].hi[k] >> 7), area);
            }
        }
        s[3] += 2;
    }
    for (int k = 0; k < 4; ++k) { const unsigned long long t = wave_total(s[k]); if ((threadIdx.x & 63u) == 0 && t) atomicAdd(w + k, t); }
}
__global__ void k_wide_centre_candidates(unsigned long long* __restrict__ w) {
    const unsigned long long n = w[3];
    unsigned long long* cand = w + 4 + 3 * kCentreBins;
    for (int k = 0; k < 3; ++k) {
        unsigned long long mean = 32768ull;
        if (n) { mean = (w[k] + n / 2) / n; if (mean > 65535ull) mean = 65535ull; }
        const unsigned long long* h = w + 4 + k * kCentreBins;
        unsigned long long run = 0, best = 0; uint32_t bestStart = 0;
        for (uint32_t b = 0; b < kCentreBins; ++b) {
            run += h[b];
            if (b >= kCentreWindow) run -= h[b - kCentreWindow];
            if (b + 1 >= kCentreWindow && run > best) { best = run; bestStart = b + 1 - kCentreWindow; }      /* ties: the lowest window */
        }
        cand[2 * k] = mean;
        cand[2 * k + 1] = best ? (unsigned long long)(bestStart * 128u + 2048u) : mean;
    }
}
/* outward movement of a plane q stored as a half float about c, in grid steps */
__device__ __forceinline__ uint32_t plane_slack_lo(uint32_t q, uint32_t c) { const int32_t v = (int32_t)q - (int32_t)c; return v >= 0 ? (uint32_t)v - rtr_f16_mag_down((uint32_t)v) : rtr_f16_mag_up((uint32_t)(-v)) - (uint32_t)(-v); }
__device__ __forceinline__ uint32_t plane_slack_hi(uint32_t q, uint32_t c) { const int32_t v = (int32_t)q - (int32_t)c; return v >= 0 ? rtr_f16_mag_up((uint32_t)v) - (uint32_t)v : (uint32_t)(-v) - rtr_f16_mag_down((uint32_t)(-v)); }
__global__ __launch_bounds__(kB) void k_wide_centre_cost(uint32_t numNodes, const uint4* __restrict__ nodes, const int32_t* __restrict__ parent,
                                                         unsigned long long* __restrict__ w) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    const unsigned long long* cand = w + 4 + 3 * kCentreBins;
    LeafBox lb[2];
    const int nl = leaf_boxes(i, numNodes, nodes, parent, lb);
    unsigned long long cost[6] = {0, 0, 0, 0, 0, 0};
    for (int j = 0; j < nl; ++j)
        for (int k = 0; k < 3; ++k) {
            const int a = (k + 1) % 3, b = (k + 2) % 3;
            const unsigned long long area = (((unsigned long long)(lb[j].hi[a] - lb[j].lo[a]) * (lb[j].hi[b] - lb[j].lo[b])) >> 8) + 1ull;
            const uint32_t ext = lb[j].hi[k] - lb[j].lo[k] + 1u;
            for (int c = 0; c < 2; ++c) {
                const uint32_t cc = (uint32_t)cand[2 * k + c];
                const uint32_t slack = plane_slack_lo(lb[j].lo[k], cc) + plane_slack_hi(lb[j].hi[k], cc);
                uint32_t rel = (slack << 8) / ext;                /* relative inflation in 1/256, capped at the box's own extent */
                if (rel > 256u) rel = 256u;
                cost[2 * k + c] += area * rel;
            }
        }
    for (int q = 0; q < 6; ++q) { const unsigned long long t = wave_total(cost[q]); if ((threadIdx.x & 63u) == 0 && t) atomicAdd(w + 4 + 3 * kCentreBins + 6 + q, t); }
}
__global__ void k_wide_centre_set(const unsigned long long* __restrict__ w, RtrBvhGrid* grid) {
    const unsigned long long* cand = w + 4 + 3 * kCentreBins;
    const unsigned long long* cost = cand + 6;
    uint32_t c[3];
    for (int k = 0; k < 3; ++k) c[k] = (uint32_t)((cost[2 * k + 1] * 4ull <= cost[2 * k]) ? cand[2 * k + 1] : cand[2 * k]);
    grid->wideCentreXY = c[0] | (c[1] << 16);
    grid->wideCentreZ = c[2];
}

/* 4-wide view of the tree for the any-hit kernel (rtr_kernels.hip, k_shadow_trace4): entry n starts from the two children of
 * BVH2 node n and, while a slot is free, opens the inner entry with the largest box into its own two children; boxes are
 * copied from the BVH2 nodes that own them and child codes keep BVH2 node ids, so entry 0 roots a complete 4-wide tree.
 * Word layout (16 words = RtrWideNode): per child (xmin|ymin<<16) (xmax|ymax<<16) (zmin|zmax<<16) as half floats about the scene's wide centre (RtrBvhGrid::wideCentreXY / Z),
 * then the four child codes; an empty slot has the code 0x80000000 and an inside-out infinite box. */
__global__ __launch_bounds__(kB) void k_wide_nodes(uint32_t numNodes, const uint4* __restrict__ nodes, const int32_t* __restrict__ parent,
                                                   const RtrBvhGrid* __restrict__ grid, const uint8_t* __restrict__ shape, uint4* __restrict__ wide) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= numNodes) return;
    uint32_t o[16];
    /* an empty slot: the code RTR_WIDE_EMPTY and an inside-out box with infinite planes (min = +inf, max = -inf as half floats),
     * which no ray enters in the kernel's octant forms: those forms skip the test of the code */
    for (int k = 0; k < 4; ++k) { o[k * 3] = 0x7c007c00u; o[k * 3 + 1] = 0xfc00fc00u; o[k * 3 + 2] = 0xfc007c00u; }
    for (int k = 12; k < 16; ++k) o[k] = 0x80000000u;
    if (!parent || parent[i] != -2) {
        const float sx = grid->scale[0], sy = grid->scale[1], sz = grid->scale[2];
        const int32_t cx = (int32_t)(grid->wideCentreXY & 0xffffu), cy = (int32_t)(grid->wideCentreXY >> 16), cz = (int32_t)(grid->wideCentreZ & 0xffffu);
        /* an entry = (owner node, side); its three plane words and its code come from the owner */
        uint32_t own[4]; int side[4]; int k = 2;
        own[0] = own[1] = i; side[0] = 0; side[1] = 1;
        auto words = [&](uint32_t n, int sd, uint32_t& wmin, uint32_t& wmax, uint32_t& wz, int32_t& code) {
            const uint4 a = nodes[(size_t)n * 2], b = nodes[(size_t)n * 2 + 1];
            wmin = sd ? a.z : a.x; wmax = sd ? a.w : a.y; wz = sd ? b.y : b.x; code = (int32_t)(sd ? b.w : b.z);
        };
        auto area = [&](uint32_t wmin, uint32_t wmax, uint32_t wz) {
            const float dx = (float)((wmax & 0xffffu) - (wmin & 0xffffu)) * sx, dy = (float)((wmax >> 16) - (wmin >> 16)) * sy;
            const float dz = (float)((wz >> 16) - (wz & 0xffffu)) * sz;
            return dx * dy + dy * dz + dz * dx;
        };
        if (shape) {
            /* the host builder chose, by cost, which entries this record opens (bvh_build.cpp collapse_wide): open1 | open2 << 2,
             * each the slot to open + 1 (0 = none); an open replaces the entry by its left child and appends its right child */
            const uint32_t sh = shape[i];
            for (int step = 0; step < 2; ++step) {
                const uint32_t o = (sh >> (2 * step)) & 3u;
                if (o == 0u || (int)o > k) break;
                uint32_t wmin, wmax, wz; int32_t code;
                words(own[o - 1u], side[o - 1u], wmin, wmax, wz, code);
                if (code < 0 || (uint32_t)code >= numNodes) break;
                own[o - 1u] = (uint32_t)code; side[o - 1u] = 0;
                own[k] = (uint32_t)code; side[k] = 1; ++k;
            }
        } else
        while (k < 4) {
            int best = -1; float bestA = -1.0f; int32_t bestCode = 0;
            for (int j = 0; j < k; ++j) {
                uint32_t wmin, wmax, wz; int32_t code;
                words(own[j], side[j], wmin, wmax, wz, code);
                if (code >= 0 && (uint32_t)code < numNodes) { const float a = area(wmin, wmax, wz); if (a > bestA) { bestA = a; best = j; bestCode = code; } }
            }
            if (best < 0) break;
            own[best] = (uint32_t)bestCode; side[best] = 0;
            own[k] = (uint32_t)bestCode; side[k] = 1; ++k;
        }
        for (int j = 0; j < k; ++j) {
            uint32_t wmin, wmax, wz; int32_t code;
            words(own[j], side[j], wmin, wmax, wz, code);
            /* planes leave here as HALF FLOATS: the offset from the scene's wide centre (q - c), rounded outward to 11 significant
             * bits, so the kernel's slab test needs no conversion (one v_fma_mix_f32 per plane).  Exact within 2048 steps of the centre,
             * 2^-11 of the distance from it beyond: +0.7 % node visits, +6 % triangle tests on the bench frame (profiles/r02/wide_sim_f16.log)
             * for 12 % fewer vector instructions per visit */
            const int32_t xmin = (int32_t)(wmin & 0xffffu) - cx, ymin = (int32_t)(wmin >> 16) - cy, zmin = (int32_t)(wz & 0xffffu) - cz;
            const int32_t xmax = (int32_t)(wmax & 0xffffu) - cx, ymax = (int32_t)(wmax >> 16) - cy, zmax = (int32_t)(wz >> 16) - cz;
            o[j * 3] = rtr_f16_floor_bits(xmin) | (rtr_f16_floor_bits(ymin) << 16);
            o[j * 3 + 1] = rtr_f16_ceil_bits(xmax) | (rtr_f16_ceil_bits(ymax) << 16);
            o[j * 3 + 2] = rtr_f16_floor_bits(zmin) | (rtr_f16_ceil_bits(zmax) << 16);
            o[12 + j] = (uint32_t)code;
        }
    }
    for (int q = 0; q < 4; ++q) wide[(size_t)i * 4 + q] = make_uint4(o[q * 4], o[q * 4 + 1], o[q * 4 + 2], o[q * 4 + 3]);
}

/* Moves the 4-wide entries into the order the host chose (breadth-first from the root: rtr_api.cpp, make_wide_nodes), so the
 * top of the tree is entries 0..K-1 — the part k_shadow_trace4 keeps in LDS.  Inner child codes are renumbered with it. */
__global__ __launch_bounds__(kB) void k_permute_wide(uint32_t numNodes, const uint4* __restrict__ in, const uint32_t* __restrict__ remap,
                                                     uint4* __restrict__ out) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= numNodes) return;
    const size_t dst = (size_t)remap[i] * 4;
    for (int q = 0; q < 3; ++q) out[dst + q] = in[(size_t)i * 4 + q];
    uint4 c = in[(size_t)i * 4 + 3];
    if ((int32_t)c.x >= 0) c.x = remap[c.x];
    if ((int32_t)c.y >= 0) c.y = remap[c.y];
    if ((int32_t)c.z >= 0) c.z = remap[c.z];
    if ((int32_t)c.w >= 0) c.w = remap[c.w];
    out[dst + 3] = c;
}

/* ---- host-side drivers ------------------------------------------------------------------------------ */
#define BV_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

hipError_t bvh_refit(const BvhInputs& in, uint32_t numPrims, uint32_t numNodes, const BvhDeviceArrays& a, hipStream_t s) {
    BV_TRY(hipMemsetAsync(a.counters, 0, (size_t)numNodes * sizeof(uint32_t), s));
    BV_TRY(hipMemsetAsync(a.depth, 0, (size_t)numNodes * sizeof(uint32_t), s));
    uint32_t init[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
    BV_TRY(hipMemcpyAsync(a.red, init, sizeof init, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_world_prims, dim3((numPrims + kB - 1) / kB), dim3(kB), 0, s, in, numPrims, a.slotOfPrim, a.tris, a.boxMin, a.boxMax, a.red);
    hipLaunchKernelGGL(k_fit, dim3((numNodes + kB - 1) / kB), dim3(kB), 0, s, numNodes, a.nodesF, a.boxMin, a.boxMax, a.parent, a.counters, a.depth, a.red, a.red + 7);
    hipLaunchKernelGGL(k_grid, dim3(1), dim3(64), 0, s, a.nodesF, a.grid);
    hipLaunchKernelGGL(k_quantize, dim3((numNodes + kB - 1) / kB), dim3(kB), 0, s, numNodes, a.nodesF, a.parent, a.grid, a.nodes);
    return hipGetLastError();
}

hipError_t bvh_build_lbvh(const BvhInputs& in, uint32_t numPrims, const BvhDeviceArrays& a, const BvhScratch& t, hipStream_t s) {
    const uint32_t n = numPrims, numNodes = n - 1;
    uint32_t init[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
    BV_TRY(hipMemcpyAsync(a.red, init, sizeof init, hipMemcpyHostToDevice, s));
    BV_TRY(hipMemsetAsync(a.counters, 0, (size_t)numNodes * sizeof(uint32_t), s));
    BV_TRY(hipMemsetAsync(a.depth, 0, (size_t)numNodes * sizeof(uint32_t), s));
    const dim3 gp((n + kB - 1) / kB), gn((numNodes + kB - 1) / kB);
    hipLaunchKernelGGL(k_world_prims, gp, dim3(kB), 0, s, in, n, (const uint32_t*)nullptr, t.trisCanon, t.minCanon, t.maxCanon, a.red);
    hipLaunchKernelGGL(k_morton, gp, dim3(kB), 0, s, n, t.minCanon, t.maxCanon, a.red, t.keysIn);
    size_t tempBytes = t.sortTempBytes;
    BV_TRY(hipcub::DeviceRadixSort::SortKeys(t.sortTemp, tempBytes, t.keysIn, t.keysOut, (int)n, 0, 64, s));
    hipLaunchKernelGGL(k_gather, gp, dim3(kB), 0, s, n, t.keysOut, t.trisCanon, t.minCanon, t.maxCanon, a.tris, a.boxMin, a.boxMax, a.slotOfPrim);
    hipLaunchKernelGGL(k_karras, gn, dim3(kB), 0, s, (int)n, t.keysOut, t.range, t.rawChild);
    hipLaunchKernelGGL(k_mark_unused, gn, dim3(kB), 0, s, numNodes, a.parent);
    hipLaunchKernelGGL(k_emit, gn, dim3(kB), 0, s, (int)n, t.range, t.rawChild, a.nodesF, a.parent);
    hipLaunchKernelGGL(k_fit, gn, dim3(kB), 0, s, numNodes, a.nodesF, a.boxMin, a.boxMax, a.parent, a.counters, a.depth, a.red, a.red + 7);
    hipLaunchKernelGGL(k_grid, dim3(1), dim3(64), 0, s, a.nodesF, a.grid);
    hipLaunchKernelGGL(k_quantize, gn, dim3(kB), 0, s, numNodes, a.nodesF, a.parent, a.grid, a.nodes);
    return hipGetLastError();
}

hipError_t bvh_make_wide(const uint4* nodes, uint32_t numNodes, const int32_t* parentOrNull, RtrBvhGrid* grid, const uint8_t* shapeOrNull, uint4* wide, unsigned long long* sums4, hipStream_t s) {
    BV_TRY(hipMemsetAsync(sums4, 0, kCentreWords * sizeof(unsigned long long), s));
    const dim3 gn((numNodes + kB - 1) / kB);
    hipLaunchKernelGGL(k_wide_centre_sum, gn, dim3(kB), 0, s, numNodes, nodes, parentOrNull, sums4);
    hipLaunchKernelGGL(k_wide_centre_candidates, dim3(1), dim3(1), 0, s, sums4);
    hipLaunchKernelGGL(k_wide_centre_cost, gn, dim3(kB), 0, s, numNodes, nodes, parentOrNull, sums4);
    hipLaunchKernelGGL(k_wide_centre_set, dim3(1), dim3(1), 0, s, sums4, grid);
    hipLaunchKernelGGL(k_wide_nodes, dim3((numNodes + kB - 1) / kB), dim3(kB), 0, s, numNodes, nodes, parentOrNull, grid, shapeOrNull, wide);
    return hipGetLastError();
}

hipError_t bvh_permute_wide(const uint4* in, uint32_t numNodes, const uint32_t* remap, uint4* out, hipStream_t s) {
    hipLaunchKernelGGL(k_permute_wide, dim3((numNodes + kB - 1) / kB), dim3(kB), 0, s, numNodes, in, remap, out);
    return hipGetLastError();
}

size_t bvh_wide_scratch_words() { return kCentreWords; }

size_t bvh_sort_temp_bytes(uint32_t numPrims) {
    size_t bytes = 0;
    unsigned long long* nul = nullptr;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, nul, nul, (int)numPrims, 0, 64, (hipStream_t)0);
    return bytes;
}

}  // namespace rtrdev
