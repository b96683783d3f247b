/* rtr_kernels.hip — hand-written gfx950 kernels of the ray-tracing hot path.
 *
 * Replaces reference src/shaders/raygen.rgen (+ closesthit/miss hit shaders and the driver's
 * traceRayEXT) dispatched by vkCmdTraceRaysKHR (src/vulkan/ray_tracing_pipeline.cppm:212-214).
 *
 * Two pipelines over the same device functions (rtr_device.h):
 *   megakernel : one lane per pixel does everything (simple, divergent) — the first-slice path
 *                and the fallback when the wavefront scratch would be too large.
 *   wavefront  : k_primary (+ tail) -> k_shadow_gen[_oct] -> k_shadow_trace4 (+ k_shadow_tail) -> k_resolve.  The shadow rays
 *                of all pixels are compacted into one dense queue with a wave-level ballot (mbcnt prefix + one atomic per
 *                workgroup), binned by direction octant for long queues, and drained by persistent waves that refill their
 *                idle lanes by ballot — so the any-hit traversal runs with full waves of coherent rays instead of idling
 *                lanes inside the shading loops.  Camera rays walk the BVH2 (RtrBvhNode), shadow rays the 4-wide view of the
 *                same tree (RtrWideNode).
 *
 * Launch shape: 256-thread workgroups (4 waves), one 8x8 pixel tile per wave in the canonical
 * tile order of rtr_device.h; grids are >> 256 workgroups at 1080p (8100 for the per-pixel
 * kernels) so all 8 XCDs fill; consecutive workgroups (round-robin over XCDs) take consecutive
 * tiles, so each XCD's L2 sees an interleaved slice of every band and the top of the BVH is
 * resident in all eight L2s.
 * No MFMA: the path is pointer-chasing + divergence bound (BASELINE.json north_star); what binds the dominant kernel is vector-
 * instruction issue with the L1's tag look-ups right behind.
 */
#include "rtr_kernels.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace rtrdev {

constexpr int kBlock = 256;

template <bool STATS, int STACK>
struct InlinePolicy {
    static constexpr bool kShade = true;
    const DeviceScene& sc; int32_t* stack; LocalStats& st;
    __device__ __forceinline__ bool occluded(rtr_v3 o, rtr_v3 d, float tmax, rtr_v3, bool) {
        HitRec h;
        return trace<true, STATS, kBlock>(sc, stack, o, d, 0.001f, tmax, h, st);
    }
};

/* The ray queue is written once and read once, 0.8 GB per 1080p frame.  READ past the caches (nt) it does not push the tree's lines
 * out of the L1s and L2s on its way through: the any-hit kernel 1.74 -> 1.71 ms alone and the frame 2.305 -> 2.245 ms with four in
 * flight.  WRITING it that way costs the queue-build kernel its write combining (0.238 -> 0.313 ms, 0.78 -> 0.99 GB of HBM writes), so
 * only the loads are streamed (profiles/r03/ab_deferred_leaf_and_nt_queue.log, ab_nt_split.log). */
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
typedef float rtr_f4 __attribute__((ext_vector_type(4)));
/* one ray out of the queue: direction + far limit, visibility index, and the origin its pixel-sample's rays share.
 * Bit 31 of a ray's slot word (kRayIntoSurface; a visibility index stays below 2^31: rtr_render_batch_limit) says that the ray leaves
 * its surface point INTO the surface — dot(hitNormal, direction) < 0 — and the w of the pixel-sample's origin record holds the code of the
 * leaf its hit triangle sits in: k_shadow_trace4 starts such a ray's walk at that leaf (tunable trace_own_leaf).  Every other reader takes the
 * bit off and walks from the root. */
constexpr uint32_t kRayIntoSurface = 0x80000000u;
__device__ __forceinline__ void queue_load(const RayQueue& q, uint32_t ray, rtr_v3& o, rtr_v3& d, float& tmax, uint32_t& slot, uint32_t nt) {
    float4 a;
    if (nt) {
        const rtr_f4 x = __builtin_nontemporal_load(reinterpret_cast<const rtr_f4*>(q.dt + ray));
        a = make_float4(x.x, x.y, x.z, x.w);
        slot = __builtin_nontemporal_load(q.slot + ray);
    } else { a = q.dt[ray]; slot = q.slot[ray]; }
    slot &= ~kRayIntoSurface;
    const float4 og = q.origin[slot & q.slotMask];
    o = rtr_mk(og.x, og.y, og.z); d = rtr_mk(a.x, a.y, a.z); tmax = a.w;
}
/* the record a pixel-sample's rays share: their origin (hit point + 0.01 normal: light_loops' shadowOrigin, the same expression), written
 * by the queue build once per pixel-sample ahead of its emission loops, and in w the code of the leaf its hit triangle sits in, written by
 * the camera-ray kernel that found the hit (the queue build never sees it: one value less to keep in a kernel capped at 64 registers) */
__device__ __forceinline__ void queue_store_origin(const RayQueue& q, size_t k, rtr_v3 hitPoint, rtr_v3 hitNormal) {
    const rtr_v3 o = rtr_madd(hitPoint, hitNormal, 0.01f);
    float* p = reinterpret_cast<float*>(q.origin + k);
    p[0] = o.x; p[1] = o.y; p[2] = o.z;
}
__device__ __forceinline__ void origin_leaf_store(float4* rayOrigin, size_t k, int32_t leaf) { reinterpret_cast<int32_t*>(rayOrigin + k)[3] = leaf; }
__device__ __forceinline__ void queue_store(const RayQueue& q, size_t idx, rtr_v3 d, float tmax, uint32_t slot, uint32_t nt, bool into) {
    const uint32_t word = slot | ((into && q.ownLeaf) ? kRayIntoSurface : 0u);
    if (nt) {
        __builtin_nontemporal_store(rtr_f4{d.x, d.y, d.z, tmax}, reinterpret_cast<rtr_f4*>(q.dt + idx));
        __builtin_nontemporal_store(word, q.slot + idx);
    } else { q.dt[idx] = make_float4(d.x, d.y, d.z, tmax); q.slot[idx] = word; }
}

/* Wave-level active-ray compaction, two phases so the global queue sees ONE atomic per wave
 * (a single contended counter saturates near 88 atomics/us on this chip — with an atomic per
 * emission step k_shadow_gen spent 4.5 ms of a 12 ms frame on it):
 *   phase 1 (CountPolicy)  walks the shading loops and only counts this lane's shadow queries;
 *                          a wave reduction + one atomicAdd reserves the wave's contiguous chunk;
 *   phase 2 (EmitPolicy)   walks the same loops again; lanes that reach a query together (the
 *                          current EXEC mask) take consecutive entries: ballot + mbcnt prefix on top of
 *                          a per-wave running offset kept in LDS (lanes parked by divergence must see
 *                          the offsets their siblings consumed).
 * Entries of one emission step are contiguous, so k_shadow_trace's waves get rays of neighbouring
 * pixels aimed at the same light triangle / sample index. */
struct CountPolicy {
    static constexpr bool kShade = false;
    uint32_t n;
    __device__ __forceinline__ bool occluded(rtr_v3, rtr_v3, float, rtr_v3, bool) { ++n; return false; }
};

struct EmitPolicy {
    static constexpr bool kShade = false;
    typedef volatile __attribute__((address_space(3))) uint32_t* lds_word;     /* keeps the access a ds_read/ds_write, not a flat_load */
    /* slot: where this query's visibility byte lives.  The bytes are laid out in PLANES — query j of pixel-sample k at
     * j * slotStride + k (slotStride = all pixel-sample slots of the frame) — so the 64 results a traversal wave holds (the same
     * query of 64 neighbouring pixels) are 64 consecutive bytes, whole 32-B sectors written by one wave from one XCD.  Pixel-major
     * (k * maxRays + j) had every byte of a sector written by another wave, mostly on another XCD, at another time: 522 MB of HBM
     * writes for 24.8 MB of payload (profiles/r02/pmc_roofline.json). */
    RayQueue queue; lds_word waveOffset; uint32_t base; uint32_t slot;
    __device__ __forceinline__ bool occluded(rtr_v3, rtr_v3 d, float tmax, rtr_v3, bool into) {
        const unsigned long long m = __ballot(1);
        const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        const uint32_t off = *waveOffset;                      /* same LDS word for the whole wave: broadcast read */
        if (prefix == 0) *waveOffset = off + (uint32_t)__popcll(m);
        queue_store(queue, (size_t)(base + off + prefix), d, tmax, slot, 0u, into);
        slot += queue.slotStride;
        return false;
    }
};

/* The visibility bytes of a pixel-sample's first 32 queries as one mask, fetched TOGETHER (bit j = query j is occluded): read one
 * by one where the light loops ask for them, every byte is a dependent load the wave waits out before it can branch — thirteen
 * memory latencies in a row on the bench frame.  Slots a pixel-sample does not use hold the pre-fill; nobody asks for them. */
__device__ __forceinline__ uint32_t vis_mask32(const uint8_t* __restrict__ vis, uint32_t slot, uint32_t slotStride, uint32_t queries) {
    uint32_t m = 0;
    const uint32_t n = queries < 32u ? queries : 32u;
    for (uint32_t j0 = 0; j0 < n; j0 += 8u) {
        uint32_t b[8];
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j) { const uint32_t jj = j0 + j < n ? j0 + j : n - 1u; b[j] = vis[slot + jj * slotStride]; }      /* eight loads in flight */
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j) m |= (b[j] != 0u && j0 + j < n ? 1u : 0u) << (j0 + j);
    }
    return m;
}

struct LookupPolicy {
    static constexpr bool kShade = true;
    const uint8_t* vis; uint32_t slot, slotStride;
    uint32_t mask, j;                        /* vis_mask32 of this pixel-sample; queries answered so far */
    __device__ __forceinline__ bool occluded(rtr_v3, rtr_v3, float, rtr_v3, bool) {
        const bool occ = j < 32u ? ((mask >> j) & 1u) != 0u : vis[slot] != 0;
        slot += slotStride; ++j;
        return occ;
    }
};

/* The same two phases with the queue binned by direction octant (k_shadow_gen_oct): a workgroup's chunk of the queue is laid
 * out octant by octant, and every run of one octant is cut into batches that are appended to that octant's batch list, from
 * which k_shadow_trace4's waves draw — so the rays a wave holds share their direction signs and the inner-node loop runs in its
 * octant form (slab_oct) nearly always instead of a third of the time.  The octant is taken from the un-normalised direction in
 * both phases (a component that underflows in the normalisation must not move a ray between the count and the emission). */
__device__ __forceinline__ uint32_t raw_octant(rtr_v3 r) {
    return (r.x < 0.f ? 1u : 0u) | (r.y < 0.f ? 2u : 0u) | (r.z < 0.f ? 4u : 0u);
}

struct CountOctPolicy {
    static constexpr bool kShade = false;
    unsigned long long lo, hi;                 /* eight 16-bit counters: octants 0-3, 4-7 */
    __device__ __forceinline__ void add(uint32_t oct, uint32_t n) {
        const unsigned long long v = (unsigned long long)n << ((oct & 3u) * 16u);
        if (oct < 4u) lo += v; else hi += v;
    }
    __device__ __forceinline__ bool occluded(rtr_v3, rtr_v3, float, rtr_v3 raw, bool) { add(raw_octant(raw), 1u); return false; }
};

struct EmitOctPolicy {
    static constexpr bool kShade = false;
    typedef volatile __attribute__((address_space(3))) uint32_t* lds_word;
    RayQueue queue; lds_word run; uint32_t slot, nt;         /* run[o]: next queue index of this wave's part of the octant-o run; slot as in EmitPolicy */
    __device__ __forceinline__ bool occluded(rtr_v3, rtr_v3 d, float tmax, rtr_v3 raw, bool into) {
        const uint32_t oct = raw_octant(raw);
        unsigned long long rem = __ballot(1);
        while (rem != 0ull) {                            /* one round per octant present among the lanes of this emission step */
            const uint32_t oo = (uint32_t)__builtin_amdgcn_readlane((int)oct, (int)__ffsll((long long)rem) - 1);
            const unsigned long long mo = __ballot(oct == oo);
            if (oct == oo) {
                const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(mo >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mo, 0u));
                const uint32_t pos = run[oo];
                if (prefix == 0) run[oo] = pos + (uint32_t)__popcll(mo);
                queue_store(queue, (size_t)(pos + prefix), d, tmax, slot, nt, into);
            }
            rem &= ~mo;
        }
        slot += queue.slotStride;
        return false;
    }
};

__device__ __forceinline__ Accum zero_accum() {
    Accum a;
    a.analytic = a.shadowed = a.unshadowed = a.avgNormal = a.avgPosition = rtr_mk(0, 0, 0);
    return a;
}

/* ---- megakernel ----------------------------------------------------------------------------- */
template <int STACK, bool STATS>
__global__ __launch_bounds__(kBlock) void k_megakernel(DeviceScene sc, RenderArgs ra, FrameOut fo, Counters* stats) {
    __shared__ int32_t s_stack[STACK * kBlock];
    int32_t* stack = s_stack + threadIdx.x;
    const uint32_t q = blockIdx.x * kBlock + threadIdx.x;
    uint32_t px, lrow, py;
    if (!pixel_of(ra, q, px, lrow, py)) return;
    LocalStats st;
    Accum acc = zero_accum();
    const uint32_t want = (fo.img[0] != nullptr ? 1u : 0u) | (fo.img[2] != nullptr ? 2u : 0u);   /* analytic / unshadowed outputs */
    const rtr_v3 camPos = rtr_ld3(ra.cam.position);
    InlinePolicy<STATS, STACK> pol{sc, stack, st};
    for (uint32_t i = 0; i < ra.spp; ++i) {
        const rtr_v3 dir = primary_dir(ra, px, py, i);
        HitRec h;
        trace<false, STATS, kBlock>(sc, stack, camPos, dir, 0.001f, 10000.0f, h, st);
        shade_sample<InlinePolicy<STATS, STACK>, STATS>(sc, ra, px, py, h, dir, want, acc, pol, st);
    }
    write_pixel(ra, fo, out_index(ra, px, lrow, py), acc);
    if (STATS) st.flush(stats);
}

/* ---- wavefront stage 1: primary visibility ---------------------------------------------------- */
/* One camera ray per lane (the comparison form, RTR_PRIMARY_PERSIST=0; k_primary_persist is the production kernel).  16-entry LDS
 * stack (16 KiB per workgroup): ordered traversal seldom holds more, and the rare ray that needs more is listed in `redo` and
 * re-traced by k_primary_tail with a full-depth stack in global memory — the same rule in the timed and the counting form. */
#ifndef RTR_PRIMARY_BLOCK
#define RTR_PRIMARY_BLOCK 64
#endif
/* Lanes per workgroup of k_primary (a multiple of 64 dividing kBlock).  One wave: a workgroup lives as long as its slowest ray, and with
 * four waves the three that finish first keep their wave slots and LDS idle until the fourth does — 0.144 -> 0.134 ms per frame in
 * launches, 0.304 -> 0.286 alone (profiles/r04/ab_primary_block.log). */
constexpr int kPrimBlock = RTR_PRIMARY_BLOCK;
static_assert(kPrimBlock % 64 == 0 && kPrimBlock >= 64 && kPrimBlock <= kBlock && kBlock % kPrimBlock == 0,
              "RTR_PRIMARY_BLOCK: a multiple of 64 that divides the 256-lane tile group (the launcher's grid is blocks * (kBlock / kPrimBlock))");
template <int STACK, bool STATS>
__global__ __launch_bounds__(kPrimBlock) void k_primary(DeviceScene sc, FrameBatch fb, float4* hitTuvp, uint32_t* hitCustom, float4* rayOrigin,
                                                    Counters* stats, uint32_t* redoCount, uint32_t* redoList, uint32_t planeBlocks) {
    __shared__ int32_t s_stack[16 * kPrimBlock];
    int32_t* stack = s_stack + threadIdx.x;
    /* the grid is spp planes of planeBlocks workgroups: one lane = one (sample, pixel slot), so at spp > 1 the samples of a pixel are
     * walked side by side by different waves instead of one after the other by one lane (the kernel is bound by the chain of dependent
     * fetches of its deepest rays: 0.88 -> 0.42 ms for the 4 spp of config 3, whose frame alone goes from 3.36 to 2.92 ms) */
    const uint32_t plane = blockIdx.x / planeBlocks;                 /* frame of the batch * spp + sample */
    const RenderArgs& ra = fb.ra[plane / fb.ra[0].spp];
    const uint32_t i = plane % fb.ra[0].spp;
    const uint32_t q = (blockIdx.x - plane * planeBlocks) * kPrimBlock + threadIdx.x;
    uint32_t px, lrow, py;
    if (!pixel_of(ra, q, px, lrow, py)) return;
    LocalStats st;
    const rtr_v3 camPos = rtr_ld3(ra.cam.position);
    const rtr_v3 dir = primary_dir(ra, px, py, i);
    HitRec h;
    if (STATS) trace<false, true, kPrimBlock, 16, 8>(sc, stack, camPos, dir, 0.001f, 10000.0f, h, st);
    else {
        /* camera rays of one 8x8 tile nearly always share their direction signs: run the traversal compiled for that octant */
        const uint32_t oct = ray_octant(sc, camPos, dir);
        const uint32_t woct = (uint32_t)__builtin_amdgcn_readfirstlane((int)oct);
        switch (__ballot(oct != woct) != 0ull ? 8u : woct) {
            case 0: trace<false, false, kPrimBlock, 16, 0>(sc, stack, camPos, dir, 0.001f, 10000.0f, h, st); break;
            case 1: trace<false, false, kPrimBlock, 16, 1>(sc, stack, camPos, dir, 0.001f, 10000.0f, h, st); break;
            case 2: trace<false, false, kPrimBlock, 16, 2>(sc, stack, camPos, dir, 0.001f, 10000.0f, h, st); break;
            case 3: trace<false, false, kPrimBlock, 16, 3>(sc, stack, camPos, dir, 0.001f, 10000.0f, h, st); break;
            case 4: trace<false, false, kPrimBlock, 16, 4>(sc, stack, camPos, dir, 0.001f, 10000.0f, h, st); break;
            case 5: trace<false, false, kPrimBlock, 16, 5>(sc, stack, camPos, dir, 0.001f, 10000.0f, h, st); break;
            case 6: trace<false, false, kPrimBlock, 16, 6>(sc, stack, camPos, dir, 0.001f, 10000.0f, h, st); break;
            case 7: trace<false, false, kPrimBlock, 16, 7>(sc, stack, camPos, dir, 0.001f, 10000.0f, h, st); break;
            default: trace<false, false, kPrimBlock, 16, 8>(sc, stack, camPos, dir, 0.001f, 10000.0f, h, st); break;
        }
    }
    /* sample-major planes keep each store of a wave contiguous */
    const size_t k = (size_t)plane * planeBlocks * kPrimBlock + q;
    if (h.custom == RTR_STACK_OVERFLOW) redoList[atomicAdd(redoCount, 1u)] = (uint32_t)k;
    else {
        hitTuvp[k] = make_float4(h.t, h.u, h.v, __uint_as_float(h.prim));
        hitCustom[k] = h.custom;
        origin_leaf_store(rayOrigin, k, h.leaf);
    }
    if (STATS) st.flush(stats);
}

/* ---- wavefront stage 1, packet form: the camera rays of one 8x8 tile walk the BVH2 TOGETHER ------------------------------------
 * Camera rays of a tile are the most coherent rays there are: they share their origin and differ by a few hundredths in direction,
 * so the 64 lanes of a wave visit nearly the same nodes.  One ray per lane (k_primary) each lane still chases its own chain of dependent
 * vector fetches with its own LDS stack, and the kernel is as slow as its deepest lane (0.32 ms for a 1080p frame rendered alone,
 * any 8-row band by itself 40-270 us).  Here the wave walks ONE path through the tree:
 *   * a node is visited when ANY lane's ray hits its box; the record's address is wave-uniform, so the fetch is a SCALAR load (the
 *     constant cache, no per-lane look-ups) and there is one stack per wave, in LDS, of {child, 64-bit mask of the lanes that hit it};
 *   * each lane tests the two child boxes against ITS ray and ITS closest hit so far (the same slab test, the same limit as one ray
 *     per lane); a ballot turns the results into the two children's lane masks; where both are hit the child that is nearer for
 *     more of the lanes that hit both is entered first (ties: child 0), the other is stacked with its mask;
 *   * a leaf's triangles are tested — records fetched with scalar loads — by the lanes of the leaf's mask, each keeping its own
 *     closest hit by the rule of trace(): min over (t, customIndex, primitiveID).
 * The stack holds at most one entry per level of the tree (<= 64: the builders' depth bound), so no ray is ever abandoned to a tail
 * kernel.  MEASURED: no faster than one ray per lane (0.168 ms per 1080p frame in an eight-frame launch, 0.334 alone, against 0.167 /
 * 0.320 — profiles/r04/ab_primary_packet.log): the wave now walks the union of its lanes' nodes strictly one after the other, ONE fetch
 * in flight, where 64 lanes chase 64 chains side by side; what it saves in look-ups and stack traffic it pays in serial latency.  Kept
 * as a second implementation (tunable primary_packet = 1), held to the oracle like the first.  A lane's hit is the same as one ray per
 * lane finds: culling is conservative with respect to the lane's own best t and the
 * closest hit does not depend on the order triangles are met in (tests: bit-identical images against k_primary, the megakernel and
 * the oracle).  What does depend on the walk is the WORK — a lane now also looks at nodes its neighbours needed — so the counting
 * form counts a node visit / triangle test for every lane of the visited mask, and the oracle restates this walk tile by tile
 * (oracle_render.cpp: trace_packet) for the counters to be held equal. */
template <bool STATS>
__global__ __launch_bounds__(kBlock) void k_primary_packet(DeviceScene sc, FrameBatch fb, float4* hitTuvp, uint32_t* hitCustom, float4* rayOrigin, Counters* stats, uint32_t planeBlocks) {
    __shared__ uint4 s_stack[kBlock / 64][64];           /* per wave: {child code, mask low, mask high, -} */
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t plane = blockIdx.x / planeBlocks;                 /* frame of the batch * spp + sample */
    const RenderArgs& ra = fb.ra[plane / fb.ra[0].spp];
    const uint32_t i = plane % fb.ra[0].spp;
    const uint32_t q = (blockIdx.x - plane * planeBlocks) * kBlock + threadIdx.x;
    uint32_t px = 0, lrow = 0, py = 0;
    const bool valid = pixel_of(ra, q, px, lrow, py);
    unsigned long long active = __ballot(valid);
    if (active == 0ull) return;                                       /* a wave of padding slots (wave-uniform) */
    LocalStats st;
    const rtr_v3 o = rtr_ld3(ra.cam.position);
    const rtr_v3 d = valid ? primary_dir(ra, px, py, i) : rtr_mk(0.f, 0.f, 1.f);
    const float tmin = 0.001f, tmax = 10000.0f;
    if (STATS && valid) { st.rays++; st.primary++; }
    const rtr_v3 idir = rtr_mk(rtr_safe_rcp_dir(d.x), rtr_safe_rcp_dir(d.y), rtr_safe_rcp_dir(d.z));
    rtr_v3 ga, gb;
    rtr_ray_grid(o, idir, sc.grid->origin, sc.grid->scale, &ga, &gb);
    HitRec best; best.t = tmax; best.u = 0.f; best.v = 0.f; best.custom = RTR_MISS; best.prim = RTR_MISS; best.leaf = 0;
    const uint4* __restrict__ nodes = sc.nodes;
    const float4* __restrict__ tris = sc.tris;
    uint32_t sp = 0;
    int32_t cur = 0;                                                  /* wave-uniform */
    for (;;) {
        const bool mine = ((active >> lane) & 1ull) != 0ull;
        bool pop = true;
        if (cur >= 0) {
            const uint4 a = nodes[2 * cur], b = nodes[2 * cur + 1];  /* uniform address: scalar loads */
            if (STATS && mine) st.nodes++;
            float tl = 0.f, tr = 0.f;
            const bool hl = mine && slab_pair(a.x, a.y, b.x, ga, gb, tmin, best.t, tl);
            const bool hr = mine && slab_pair(a.z, a.w, b.y, ga, gb, tmin, best.t, tr);
            const unsigned long long mL = __ballot(hl), mR = __ballot(hr);
            if ((mL | mR) != 0ull) {
                pop = false;
                if (mR == 0ull) { cur = (int32_t)b.z; active = mL; }
                else if (mL == 0ull) { cur = (int32_t)b.w; active = mR; }
                else {
                    /* the vote of the lanes that hit both boxes: which child is nearer for more of them (ties: child 0) */
                    const uint32_t rNear = (uint32_t)__popcll(__ballot(hl && hr && tr < tl)), lNear = (uint32_t)__popcll(__ballot(hl && hr && !(tr < tl)));
                    const bool rFirst = rNear > lNear;
                    if (lane == 0u) {
                        const unsigned long long m2 = rFirst ? mL : mR;
                        s_stack[wave][sp] = make_uint4(rFirst ? b.z : b.w, (uint32_t)m2, (uint32_t)(m2 >> 32), 0u);
                    }
                    ++sp;
                    cur = (int32_t)(rFirst ? b.w : b.z); active = rFirst ? mR : mL;
                }
            }
        } else {
            const uint32_t code = (uint32_t)~cur;
            const uint32_t first = code >> 3, cnt = (code & 7u) + 1u;
            for (uint32_t k = 0; k < cnt; ++k) {
                const float4 q0 = tris[3u * (first + k)], q1 = tris[3u * (first + k) + 1u], q2 = tris[3u * (first + k) + 2u];      /* uniform: scalar loads */
                if (mine) {
                    if (STATS) st.tris++;
                    float t, u, v;
                    if (rtr_mt_intersect(o, d, f4xyz(q0), f4xyz(q1), f4xyz(q2), tmin, &t, &u, &v) && t < tmax) {
                        const uint32_t cu = __float_as_uint(q0.w), pr = __float_as_uint(q1.w);
                        if (!(__float_as_uint(q2.w) & 1u) || alpha_pass<STATS>(sc, cu, pr, u, v, st)) {
                            if (t < best.t || (t == best.t && (cu < best.custom || (cu == best.custom && pr < best.prim)))) {
                                best.t = t; best.u = u; best.v = v; best.custom = cu; best.prim = pr; best.leaf = cur;
                            }
                        }
                    }
                }
            }
        }
        if (pop) {
            if (sp == 0u) break;
            --sp;
            const uint4 e = s_stack[wave][sp];                        /* the same address in every lane: a broadcast read */
            cur = __builtin_amdgcn_readfirstlane((int32_t)e.x);
            active = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)e.y) | ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)e.z) << 32);
        }
    }
    if (valid) {
        const size_t k = (size_t)plane * planeBlocks * kBlock + q;
        hitTuvp[k] = make_float4(best.t, best.u, best.v, __uint_as_float(best.prim));
        hitCustom[k] = best.custom;
        origin_leaf_store(rayOrigin, k, best.leaf);
    }
    if (STATS) st.flush(stats);
}

/* Re-traces the pixel-samples the primary kernels abandoned: BVH2 walk with a full-depth stack in global memory (no LDS, so it
 * can always run).  STATS: the counting form (the ray itself was counted by the kernel that abandoned it). */
template <bool STATS>
__global__ __launch_bounds__(kBlock) void k_primary_tail(DeviceScene sc, FrameBatch fb, float4* hitTuvp, uint32_t* hitCustom, float4* rayOrigin,
                                                         const uint32_t* __restrict__ redoCount, const uint32_t* __restrict__ redoList,
                                                         int32_t* __restrict__ spill, uint32_t planeStride, Counters* stats) {
    const uint32_t n = *redoCount;
    if (n == 0) return;
    int32_t* stack = spill + blockIdx.x * kBlock + threadIdx.x;
    LocalStats st;
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
        const uint32_t k = redoList[j];
        const uint32_t plane = k / planeStride, q = k % planeStride;       /* frame * spp + sample, pixel slot */
        const RenderArgs& ra = fb.ra[plane / fb.ra[0].spp];               /* (per lane: the list mixes the frames of the batch) */
        const uint32_t i = plane % fb.ra[0].spp;
        uint32_t px, lrow, py;
        if (!pixel_of(ra, q, px, lrow, py)) continue;
        HitRec h;
        trace<false, STATS, 64 * kBlock>(sc, stack, rtr_ld3(ra.cam.position), primary_dir(ra, px, py, i), 0.001f, 10000.0f, h, st);
        if (STATS) { st.rays--; st.primary--; }
        hitTuvp[k] = make_float4(h.t, h.u, h.v, __uint_as_float(h.prim));
        hitCustom[k] = h.custom;
        origin_leaf_store(rayOrigin, k, h.leaf);
    }
    if (STATS) st.flush(stats);
}

/* ---- wavefront stage 2: shadow-ray generation into the compacted queue ------------------------- */
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
/* The sum over the 64 lanes as a wave-uniform value, through the DPP cross-lane modes of gfx9 (quad permutes, row mirrors, row
 * broadcasts: vector instructions) instead of six trips through the LDS crossbar (ds_bpermute): the queue build sums eight counters
 * per wave before it can reserve anything.  All 64 lanes must be active. */
__device__ __forceinline__ uint32_t wave_total(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);       /* quad_perm:[1,0,3,2] */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);       /* quad_perm:[2,3,0,1] */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false);      /* row_half_mirror */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false);      /* row_mirror: every lane holds its row's sum */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);      /* row_bcast:15 into rows 1 and 3 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);      /* row_bcast:31 into rows 2 and 3 */
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

/* 1024-thread workgroups (16 waves): the queue is reserved once per workgroup, see below; planeStride = the sample-plane stride
 * of the hit records k_primary wrote (its grid x 256). */
constexpr uint32_t kGenBlock = 1024;
__global__ __launch_bounds__(kGenBlock) void k_shadow_gen(DeviceScene sc, FrameBatch fb, const float4* hitTuvp,
                                                          const uint32_t* hitCustom, RayQueue queue, uint32_t* count, uint32_t planeStride) {
    __shared__ uint32_t s_off[kGenBlock / 64], s_tot[kGenBlock / 64], s_base;
    uint32_t q;
    const uint32_t frame = batch_frame(blockIdx.x * kGenBlock + threadIdx.x, planeStride, q);
    const RenderArgs& ra = fb.ra[frame < fb.n ? frame : 0u];
    const size_t plane0 = (size_t)frame * ra.spp;                 /* first sample plane of this lane's frame */
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t px = 0, lrow = 0, py = 0;
    const bool live = frame < fb.n && pixel_of(ra, q, px, lrow, py);
    LocalStats st;
    Accum acc = zero_accum();
    /* The surface fetch (hit -> object -> indices -> vertices, a chain of dependent loads) runs ONCE per sample; the
     * light loops then run twice over it, first counting the queries, then emitting them.  spp > 1 keeps the fetched
     * surfaces of the samples in registers only for spp == 1 (the bench case); otherwise the fetch is repeated. */
    const bool single = ra.spp == 1u;
    Surface sf0;
    bool surf0 = false;
    uint32_t n = 0;
    if (live) {
        CountPolicy cp{0};
        for (uint32_t i = 0; i < ra.spp; ++i) {
            const size_t k = (plane0 + i) * planeStride + q;
            const float4 r = hitTuvp[k];
            HitRec h; h.t = r.x; h.u = r.y; h.v = r.z; h.prim = __float_as_uint(r.w); h.custom = hitCustom[k];
            Surface sf;
            const bool surf = fetch_surface<false, false>(sc, ra, h, primary_dir(ra, px, py, i), false, acc, sf, st);
            if (surf) light_loops<CountPolicy, false>(sc, ra, px, py, sf, 0u, acc, cp, st);
            if (i == 0) { sf0 = sf; surf0 = surf; }
        }
        n = cp.n;
    }
    /* ONE reservation per workgroup: the sixteen waves add up their totals in LDS and lane 0 does the atomic.  A single
     * address sustains ~88 atomics/us, so one per wave (32 400 at 1080p) cost 0.37 of this kernel's 0.41 ms; one per
     * 1024-thread workgroup is 2 025.  (Every thread of the block reaches both barriers: padding lanes are not retired early.) */
    const uint32_t total = wave_sum(n);
    if ((threadIdx.x & 63u) == 0) { s_tot[wave] = total; s_off[wave] = 0; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (uint32_t w = 0; w < kGenBlock / 64; ++w) t += s_tot[w];
        s_base = t ? atomicAdd(count, t) : 0u;
    }
    __syncthreads();
    uint32_t base = s_base;
    for (uint32_t w = 0; w < wave; ++w) base += s_tot[w];
    if (!live || n == 0) return;
    /* phase 2: emit */
    for (uint32_t i = 0; i < ra.spp; ++i) {
        const size_t k = (plane0 + i) * planeStride + q;
        EmitPolicy pol{queue, (EmitPolicy::lds_word)&s_off[wave], base, (uint32_t)k};
        if (single) {
            if (surf0) { queue_store_origin(queue, k, sf0.hitPoint, sf0.hitNormal); light_loops<EmitPolicy, false>(sc, ra, px, py, sf0, 0u, acc, pol, st); }
        } else {
            const float4 r = hitTuvp[k];
            HitRec h; h.t = r.x; h.u = r.y; h.v = r.z; h.prim = __float_as_uint(r.w); h.custom = hitCustom[k];
            Surface sf;
            if (fetch_surface<false, false>(sc, ra, h, primary_dir(ra, px, py, i), false, acc, sf, st)) {
                queue_store_origin(queue, k, sf.hitPoint, sf.hitNormal);
                light_loops<EmitPolicy, false>(sc, ra, px, py, sf, 0u, acc, pol, st);
            }
        }
    }
}

constexpr size_t kBinnedMinRays = (size_t)10 << 20;  /* by default the binned queue is used from this queue capacity ... */
constexpr uint32_t kBinnedMinNodes = 1u << 16;       /* ... and this tree size: binning pays where traversal dominates (Cornell at 1080p, 14.6 M
                                                      * cheap rays: 18.8 Grays/s plain, 14.7 binned; the 41 k-node sphere scene: -1 %) */
#ifndef RTR_GEN_OCT_BLOCK
#define RTR_GEN_OCT_BLOCK 512
#endif
constexpr uint32_t kGenOctBlock = RTR_GEN_OCT_BLOCK;     /* two workgroups per CU, so one's reservation round trips hide under the other's work */
/* k_shadow_gen with the queue binned by direction octant (CountOctPolicy / EmitOctPolicy above).  ctrl = Workspace::queueCount:
 * [0] queued rays, [16 + 16 r] / [kQueueListLens + r] cursor / length of batch list r = octant * 8 + xcd; lists: listStride uint2 {first, count} per list. */
/* Seven waves per SIMD: the kernel is neither issue- nor HBM-bound (its records leave in whole lines: WRITE_SIZE = 1.01 x the algorithmic
 * bytes) but waits — on the hit -> object -> index -> vertex chain of its surface fetch, on its reservation atomics and three barriers — so it
 * wants waves more than registers: at the 88 VGPRs it would take by itself (5 waves per SIMD) 0.171 ms per 1080p frame, capped at 64 (8 waves,
 * 13 dwords spilled) 0.148 (profiles/r04/ab_tri2_gen_waves.log).  Round 5: with the into-the-surface mark of every ray (a dot product and
 * a bit per sample) the cap of 64 spilled 31 dwords and cost 0.010 ms; at 72 (7 waves) the kernel is back at 0.151 (profiles/r05/ab_gen_into.log). */
#ifndef RTR_GEN_OCT_WAVES
#define RTR_GEN_OCT_WAVES 7
#endif
#if RTR_GEN_OCT_WAVES > 0
#define RTR_GEN_OCT_ATTR __attribute__((amdgpu_waves_per_eu(RTR_GEN_OCT_WAVES, 8)))
#else
#define RTR_GEN_OCT_ATTR
#endif
__global__ __launch_bounds__(kGenOctBlock) RTR_GEN_OCT_ATTR void k_shadow_gen_oct(DeviceScene sc, FrameBatch fb, const float4* hitTuvp, const uint32_t* hitCustom,
                                                              RayQueue queue, uint32_t* ctrl, uint32_t planeStride, uint2* lists,
                                                              uint32_t listStride, uint32_t kBatch, uint32_t nt) {
    constexpr uint32_t kWaves = kGenOctBlock / 64;
    __shared__ uint32_t s_tot[kWaves][8], s_run[kWaves][8];
    uint32_t q;
    const uint32_t frame = batch_frame(blockIdx.x * kGenOctBlock + threadIdx.x, planeStride, q);
    const RenderArgs& ra = fb.ra[frame < fb.n ? frame : 0u];
    const size_t plane0 = (size_t)frame * ra.spp;
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t px = 0, lrow = 0, py = 0;
    const bool live = frame < fb.n && pixel_of(ra, q, px, lrow, py);
    LocalStats st;
    Accum acc = zero_accum();
    const bool single = ra.spp == 1u;                 /* (fetching the surface again in the emit phase instead of keeping it across the reservation: 74 VGPRs instead of 88, and slower — profiles/r04/ab_gen_refetch.log) */
    Surface sf0;
    bool surf0 = false;
    CountOctPolicy cp{0ull, 0ull};
    if (live) {
        for (uint32_t i = 0; i < ra.spp; ++i) {
            const size_t k = (plane0 + i) * planeStride + q;
            const float4 r = hitTuvp[k];
            HitRec h; h.t = r.x; h.u = r.y; h.v = r.z; h.prim = __float_as_uint(r.w); h.custom = hitCustom[k];
            Surface sf;
            bool surf = fetch_surface<false, false>(sc, ra, h, primary_dir(ra, px, py, i), false, acc, sf, st);
#if RTR_EXP_DOUBLE_FETCH_GEN      /* experiment (profiles/r05/ab_gen_fetch_share.log): what one surface fetch costs the queue build — a DEPENDENT second fetch, so that its chain of gathers is waited for again */
            if (surf) {
                HitRec h2 = h; h2.u = h.u + sf.hitPoint.x * (float)(ra.spp - 1u) * 1.0e-30f; h2.v = h.v + sf.hitNormal.y * (float)(ra.spp - 1u) * 1.0e-30f;
                h2.prim = h.prim + (__float_as_uint(sf.hitPoint.z) == 0x7fc12345u ? 1u : 0u);
                Surface sf2;
                if (fetch_surface<false, false>(sc, ra, h2, primary_dir(ra, px, py, i), false, acc, sf2, st)) sf = sf2;
            }
#endif
            if (surf) light_loops<CountOctPolicy, false>(sc, ra, px, py, sf, 0u, acc, cp, st);
            if (i == 0) { sf0 = sf; surf0 = surf; }
        }
    }
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t o = 0; o < 8; ++o) {
        const uint32_t c = (uint32_t)(((o < 4 ? cp.lo : cp.hi) >> ((o & 3u) * 16u)) & 0xffffull);
        mine += c;
        const uint32_t t = wave_total(c);
        if ((threadIdx.x & 63u) == 0) s_tot[wave][o] = t;
    }
    __syncthreads();
    if (threadIdx.x < kQueueLists) {
        /* The first wave, one lane per (octant, list), makes BOTH reservations at once — the workgroup's chunk of the queue (lane 0: ONE
         * atomic per workgroup, see k_shadow_gen) and the places of its batches in the 64 lists (how many batches a run is cut into
         * depends on its length only) — so the workgroup waits for one device-scope round trip, not for two with a barrier between; then
         * it lays the runs inside the chunk: the waves' write cursors, the batch descriptors.  Every lane sums the per-wave counts
         * itself (64 broadcast LDS reads) instead of waiting for one lane to do it. */
        const uint32_t o = threadIdx.x / kQueueRegions, x = threadIdx.x % kQueueRegions;
        uint32_t before = 0, len = 0, total = 0;                 /* rays of the octants before o, of o, of all */
#pragma unroll
        for (uint32_t oo = 0; oo < 8; ++oo) {
            uint32_t t = 0;
#pragma unroll
            for (uint32_t w = 0; w < kWaves; ++w) t += s_tot[w][oo];
            before += oo < o ? t : 0u;
            len = oo == o ? t : len;
            total += t;
        }
        /* the run's batches are dealt round-robin to the eight lists of this octant (one per consumer XCD); every lane makes its
         * own reservation */
        /* Long queues (batches of >= 512 rays): the last eighth of the workgroups — whose batches land at the ends of the lists, which the
         * any-hit kernel's waves drain last — cut their runs into half batches, so the kernel ends on shorter ones (-0.5 % of the any-hit
         * kernel in 8-frame launches; a single frame's 256-ray batches are left alone: finer ones cost it more than the shorter tail
         * saves — profiles/r04/ab_gen_tail_batches*.log) */
        const uint32_t kB = (kBatch >= 512u && blockIdx.x * 8u >= gridDim.x * 7u) ? kBatch / 2u : kBatch;
        const uint32_t nb = (len + kB - 1) / kB;
        const uint32_t b0 = (x + kQueueRegions - blockIdx.x % kQueueRegions) % kQueueRegions;     /* first batch that goes to list x */
        const uint32_t cnt = b0 < nb ? (nb - b0 + kQueueRegions - 1) / kQueueRegions : 0u;
        uint32_t at = 0, pos = 0;
        if (threadIdx.x == 0 && total) at = atomicAdd(ctrl, total);
        if (cnt) pos = atomicAdd(ctrl + kQueueListLens + threadIdx.x, cnt);
        const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)at) + before;      /* of this lane's octant's run */
        if (x == 0) {
            uint32_t c = first;
            for (uint32_t w = 0; w < kWaves; ++w) { s_run[w][o] = c; c += s_tot[w][o]; }
        }
        for (uint32_t j = 0; j < cnt; ++j) {
            const uint32_t f = (b0 + j * kQueueRegions) * kB;
            lists[(size_t)threadIdx.x * listStride + pos + j] = make_uint2(first + f, len - f < kB ? len - f : kB);
        }
    }
    __syncthreads();
    if (!live || mine == 0) return;
    for (uint32_t i = 0; i < ra.spp; ++i) {
        const size_t k = (plane0 + i) * planeStride + q;
        EmitOctPolicy pol{queue, (EmitOctPolicy::lds_word)&s_run[wave][0], (uint32_t)k, nt};
        if (single) {
            if (surf0) { queue_store_origin(queue, k, sf0.hitPoint, sf0.hitNormal); light_loops<EmitOctPolicy, false>(sc, ra, px, py, sf0, 0u, acc, pol, st); }
        } else {
            const float4 r = hitTuvp[k];
            HitRec h; h.t = r.x; h.u = r.y; h.v = r.z; h.prim = __float_as_uint(r.w); h.custom = hitCustom[k];
            Surface sf;
            if (fetch_surface<false, false>(sc, ra, h, primary_dir(ra, px, py, i), false, acc, sf, st)) {
                queue_store_origin(queue, k, sf.hitPoint, sf.hitNormal);
                light_loops<EmitOctPolicy, false>(sc, ra, px, py, sf, 0u, acc, pol, st);
            }
        }
    }
}

/* ---- wavefront stage 3: any-hit traversal of the queue (the dominant kernel) -------------------
 * Persistent waves with wavefront-ballot active-ray compaction.  Profiling the one-ray-per-lane form
 * showed the SIMDs ~100 % busy issuing instructions at ~30 % lane utilisation: rays of one wave need
 * very different numbers of node visits, and inner-node and leaf work alternated under divergent
 * branches.  Here:
 *   * a wave grabs a batch of consecutive queue entries (256 by default; swept in profiles/r01/sweep_refill.log) with ONE atomic and keeps them in wave-uniform
 *     cursors; whenever at least kRefill lanes are idle (ballot + popcount) the idle lanes take the next
 *     entries (mbcnt prefix), so finished rays are replaced instead of waited for;
 *   * "while-while": all lanes first descend inner nodes until every live lane sits on a leaf (or has
 *     finished), then all leaves are intersected together;
 * A lane's own sequence of node visits / triangle tests is exactly that of trace<true>() — and of the
 * oracle's trace_bvh() — so visibility bits AND work counters are unchanged. */
#ifndef RTR_REFILL_LDS
#define RTR_REFILL_LDS 1
#endif
#ifndef RTR_LIST_SHARE
#define RTR_LIST_SHARE 1
#endif
constexpr int kTailBlocks = 64;            /* grid of the two "redo" kernels; their global stacks are strided by 64 * kBlock lanes */
constexpr int32_t kDone = (int32_t)0x80000000;   /* not a valid leaf code (first would be 2^28-1) */
constexpr int32_t kAbandoned = (int32_t)0x80000001;   /* inner_nodes4 -> its caller: this lane's stack was full (nor is this a leaf code: same first) */
/* queue entries a wave reserves per atomic: 256 for a queue of one frame's length, 512 for the queue of a launch of several frames
 * (kBatchLongQueue: 0.6-1.5 % faster there — fewer cursor atomics, fewer short batches — while a single frame rendered alone ends in
 * a longer tail with it, 2.65 instead of 2.53 ms; profiles/r03/sweep_batch_r03_4.log, bench_*_r03_5.log) */
constexpr uint32_t kBatchDefault = 256, kBatchLong = 512;
constexpr size_t kBatchLongQueue = (size_t)100 << 20;

/* Stack policy of the persistent kernels.  Ordered traversal rarely holds more than ~10 entries, so every lane gets 16 LDS
 * entries (16 KiB per workgroup -> 8 workgroups = 32 waves per CU, the hardware maximum).  The rare ray that needs a 17th entry is
 * abandoned — its queue index goes to an overflow list and k_shadow_tail finishes it from scratch with a full-depth stack — so the
 * hot loop is pure LDS with no spill branch (an LDS/global select made the compiler emit flat_load for EVERY pop). */
/* Production form (no counters).  Same scheduling — persistent waves, ballot refill, early-exit
 * while-while, described above — with the hot loop stripped of everything that is not a node visit:
 *   * a ray's result (0 visible, 1 occluded, 2 "needs a deeper stack") stays in a register and is written when the lane
 *     is retired in the refill block, so the inner and leaf loops contain no global store and no overflow branch;
 *   * the slot below stack entry 0 holds the "ray finished" code, so a pop needs no empty-stack test and the speculative
 *     read of the top needs no index clamp;
 *   * node and triangle addresses are 32-bit offsets from a scalar base (global_load ... saddr), not 64-bit lane math.
 * (This is the 2-wide form, RTR_TRACE_BVH4=0: kept as a second implementation the tests hold against the 4-wide kernel.) */
constexpr uint32_t kResNone = 3u;          /* lane holds no unwritten result */
#ifndef RTR_TRACE_BLOCK
#define RTR_TRACE_BLOCK 256
#endif
/* The any-hit kernel's workgroup: 256 lanes, eight per CU = the 32-wave hardware maximum.  Its LDS: 17 stack entries per lane
 * (17 KiB) + the first kTopNodes four-wide records of the tree (2.5 KiB).  Larger workgroups would share one larger copy of the
 * tree's top (1024 lanes: two copies of 160 records per CU instead of eight of 40) and the kernel alone gets faster with them
 * (1.86 -> 1.80 ms), but a big workgroup frees its LDS only when its last wave retires, which is when the other frames' kernels
 * can start: with four frames in flight 11.57 G rays/s became 11.1 (512 lanes) and 10.3 (1024) — profiles/r02/trace_block_size.log. */
constexpr int kTraceBlock = RTR_TRACE_BLOCK;
/* cache policy of the any-hit kernel's triangle / record loads (raw_buffer_load aux: 1 = sc0, 2 = nt, 16 = sc1): the default policy is the fastest (profiles/r05/ab_cache_policy.log) */
#ifndef RTR_TRI_AUX
#define RTR_TRI_AUX 0
#endif
#ifndef RTR_NODE_AUX
#define RTR_NODE_AUX 0
#endif
#ifndef RTR_SHADOW_STACK
#define RTR_SHADOW_STACK RTR_WIDE_STACK      /* experiments only (profiles/r05/ab_stack_top.log): the oracle restates RTR_WIDE_STACK entries */
#endif
#ifndef RTR_TRACE_TOP
#define RTR_TRACE_TOP 40
#endif
constexpr uint32_t kTopNodes = (uint32_t)RTR_TRACE_TOP * (RTR_TRACE_BLOCK / 256);

template <int STACK>
__global__ __launch_bounds__(kBlock) void k_shadow_trace(DeviceScene sc, const RayQueue queue,
                                                              const uint32_t* __restrict__ count, uint32_t* nextBatch,
                                                              uint8_t* __restrict__ vis, uint32_t visFill, uint32_t kBatch, uint32_t kRefill,
                                                              uint32_t kInnerMin, uint32_t* overflow, uint32_t overflowCap) {
    __shared__ int32_t s_stack[(STACK + 1) * kBlock];        /* slot 0, below the stack, holds kDone for good */
    int32_t* lds = s_stack + threadIdx.x;
    lds[0] = kDone;
    const uint32_t n = *count;
    uint32_t batchPos = 0, batchEnd = 0;     /* wave-uniform */
    bool exhausted = false;                  /* wave-uniform */
    /* The queue is cut into kQueueRegions contiguous regions with one batch cursor each (64 B apart).  A workgroup starts on
     * region (blockIdx mod 8) — workgroups are dealt round-robin to the 8 XCDs — and moves on only when that region is empty.
     * One counter sustains ~88 atomics/us, which is what forced 256-ray batches; eight counters give headroom, but 256 stays
     * the best batch even for the short queue of a 1/8-frame shard (profiles/r01/sweep_batch_sharded.log). */
    const uint32_t regionLen = ((n + kQueueRegions - 1) / kQueueRegions + kBatch - 1) / kBatch * kBatch;
    const uint32_t myRegion = blockIdx.x % kQueueRegions;
    uint32_t regionTry = 0;                  /* wave-uniform: regions found empty so far (cursors only grow) */
    int32_t cur = kDone;
    int sp = 0;                              /* entries held; the top is lds[sp * kBlock] */
    rtr_v3 o = rtr_mk(0, 0, 0), d = rtr_mk(0, 0, 0), ga = rtr_mk(0, 0, 0), gb = rtr_mk(0, 0, 0);   /* t(q) = q * ga + gb */
    float tmax = 0.f;
    uint32_t slot = 0, rayIndex = 0, res = kResNone, occ = 0;
    const float tmin = 0.001f;
    /* nodes and triangles through buffer resources: the address of a visit is one 32-bit shift, not 64-bit lane arithmetic
     * (2.28 -> 2.17 ms, and 62 -> 47 VGPRs) */
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t nodeBuf = __builtin_amdgcn_make_buffer_rsrc((void*)sc.nodes, 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t triBuf = __builtin_amdgcn_make_buffer_rsrc((void*)sc.tris, 0, 0xffffffff, 0x00020000);

    for (;;) {
        /* ---- retire finished rays, refill idle lanes from the wave's batch ---- */
        const unsigned long long idle = __ballot(cur == kDone);
        const uint32_t nIdle = (uint32_t)__popcll(idle);
        if (nIdle >= kRefill || nIdle == 64u) {
            if (cur == kDone && res != kResNone) {
                if (res == 2u) { const uint32_t at = atomicAdd(overflow, 1u); if (at < overflowCap) overflow[1u + at] = rayIndex; }      /* finished by k_shadow_tail (a full list: it redoes the whole queue) */
                else { if (res != visFill) vis[slot] = (uint8_t)res; occ += res; }      /* the array was pre-filled with the commoner outcome: only the other one is stored */
                res = kResNone;
            }
            if (!exhausted) {
                if (batchPos == batchEnd) {
                    for (;;) {
                        if (regionTry >= kQueueRegions) { exhausted = true; break; }
                        const uint32_t r = (myRegion + regionTry) % kQueueRegions;
                        const uint32_t lo = r * regionLen;
                        uint32_t hi = lo + regionLen; if (hi > n) hi = n;
                        uint32_t b = hi;
                        if (lo < hi) {
                            uint32_t got = 0;
                            if ((threadIdx.x & 63u) == 0) got = atomicAdd(nextBatch + 16u * r, kBatch);
                            got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
                            b = got < regionLen ? lo + got : hi;
                        }
                        if (b < hi) { batchPos = b; batchEnd = (b + kBatch < hi) ? b + kBatch : hi; break; }
                        ++regionTry;
                    }
                }
                if (!exhausted) {
                    const uint32_t avail = batchEnd - batchPos;
                    const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                    if (cur == kDone && prefix < avail) {
                        rayIndex = batchPos + prefix;
                        queue_load(queue, rayIndex, o, d, tmax, slot, 0u);
                        if (!(tmax > tmin)) {
                            if (visFill != 0u) vis[slot] = 0;                 /* empty interval: nothing can be hit (oracle trace(): same rule) */
                        } else {
                            const rtr_v3 idir = rtr_mk(rtr_safe_rcp_dir(d.x), rtr_safe_rcp_dir(d.y), rtr_safe_rcp_dir(d.z));
                            rtr_ray_grid(o, idir, sc.grid->origin, sc.grid->scale, &ga, &gb);
                            cur = 0; sp = 0; res = 0u;
                        }
                    }
                    batchPos += (nIdle < avail) ? nIdle : avail;
                }
            }
        }
        if (__ballot(cur != kDone) == 0ull) {
            if (exhausted) break;                /* every lane is idle and was retired above (nIdle == 64) */
            continue;
        }
        /* ---- inner nodes ("while-while" with an early exit: keep descending while more than kInnerMin lanes are on inner nodes, or
         * while nobody has a leaf to test; then test the leaves that are waiting.  Waiting for EVERY lane to reach a leaf left the
         * early lanes idle for the stragglers: 3.64 -> 3.0 ms at kInnerMin = 20, profiles/r01/sweep_inner.log.  Every pass of the
         * outer loop visits a node or tests a leaf for at least one lane, or exits, so all waves drain) ---- */
        for (;;) {
            const unsigned long long innerMask = __ballot(cur >= 0);
            if (innerMask == 0ull) break;
            if ((uint32_t)__popcll(innerMask) <= kInnerMin && __ballot(cur < 0 && cur != kDone) != 0ull) break;
            if (cur >= 0) {
                const int32_t nodeOff = cur << 5;                   /* one 32-B RtrBvhNode = the whole visit */
                const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(nodeBuf, nodeOff, 0, 0);
                const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(nodeBuf, nodeOff + 16, 0, 0);
                const int32_t top = lds[sp * kBlock];              /* speculative: hides the pop's LDS latency under the node loads */
                float tl, tr;
                const bool hl = slab_pair(a.x, a.y, b.x, ga, gb, tmin, tmax, tl);
                const bool hr = slab_pair(a.z, a.w, b.y, ga, gb, tmin, tmax, tr);
                const bool swap = tr < tl;
                const int32_t c0 = (int32_t)b.z, c1 = (int32_t)b.w;
                const bool both = hl && hr, none = !(hl || hr);
                const bool second = both ? swap : !hl;               /* descend into child 1? (near child if both are hit) */
                int32_t next = second ? c1 : c0;
                const bool push = both && sp < STACK;
                if (push) { ++sp; lds[sp * kBlock] = swap ? c0 : c1; }
                if (both && !push) { res = 2u; next = kDone; }       /* needs a 17th entry: the tail kernel redoes this ray */
                if (none) { next = top; --sp; }                      /* slot 0 holds kDone: popping an empty stack ends the ray (visible) */
                cur = next;
            }
        }
        /* ---- leaves ---- */
        if (cur < 0 && cur != kDone) {
            const uint32_t code = (uint32_t)~cur;
            const uint32_t first = code >> 3, cnt = (code & 7u) + 1u;
            bool hit = false;
            for (uint32_t i = 0; i < cnt && !hit; ++i) {
                const int32_t triOff = (int32_t)((first + i) * 48u);
                const u32x4 r0 = __builtin_amdgcn_raw_buffer_load_b128(triBuf, triOff, 0, 0);
                const u32x4 r1 = __builtin_amdgcn_raw_buffer_load_b128(triBuf, triOff + 16, 0, 0);
                const u32x4 r2 = __builtin_amdgcn_raw_buffer_load_b128(triBuf, triOff + 32, 0, 0);
                const float4 q0 = make_float4(__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z), __uint_as_float(r0.w));
                const float4 q1 = make_float4(__uint_as_float(r1.x), __uint_as_float(r1.y), __uint_as_float(r1.z), __uint_as_float(r1.w));
                const float4 q2 = make_float4(__uint_as_float(r2.x), __uint_as_float(r2.y), __uint_as_float(r2.z), __uint_as_float(r2.w));
                float t, u, v;
                if (rtr_mt_intersect(o, d, f4xyz(q0), f4xyz(q1), f4xyz(q2), tmin, &t, &u, &v) && t < tmax) {
                    hit = true;
                    if (__float_as_uint(q2.w) & 1u) {      /* opacity.rahit on alpha-tested geometry */
                        LocalStats st;
                        hit = alpha_pass<false>(sc, __float_as_uint(q0.w), __float_as_uint(q1.w), u, v, st);
                    }
                }
            }
            if (hit) { res = 1u; cur = kDone; }
            else { cur = lds[sp * kBlock]; --sp; }                   /* slot 0 holds kDone: an empty stack ends the ray (visible) */
        }
    }
    /* how many of this launch's rays were occluded: the host picks the next frame's pre-fill of the visibility array by it */
    const uint32_t occWave = wave_sum_u32(occ);
    if ((threadIdx.x & 63u) == 0 && occWave) atomicAdd(nextBatch - kBatchCursorWord + kOccludedWord, occWave);      /* nextBatch = the control block + kBatchCursorWord (the launcher's expression) */
}

/* The inner-node loop of k_shadow_trace4, compiled per direction octant (OCT 0..7; 8 = any signs, see slab_oct). */
/* One any-hit triangle test of the persistent 4-wide kernel: 48-B record through the buffer resource, Moeller-Trumbore, t < tmax,
 * and opacity.rahit on alpha-tested geometry. */
template <bool STATS>
__device__ __forceinline__ bool tri_any(const DeviceScene& sc, const __amdgpu_buffer_rsrc_t triBuf, const uint32_t tri, const rtr_v3 o, const rtr_v3 d,
                                        const float tmin, const float tmax, LocalStats& st) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const int32_t triOff = (int32_t)(tri * 48u);
    const u32x4 r0 = __builtin_amdgcn_raw_buffer_load_b128(triBuf, triOff, 0, RTR_TRI_AUX);
    const u32x4 r1 = __builtin_amdgcn_raw_buffer_load_b128(triBuf, triOff + 16, 0, RTR_TRI_AUX);
    const u32x4 r2 = __builtin_amdgcn_raw_buffer_load_b128(triBuf, triOff + 32, 0, RTR_TRI_AUX);
    const float4 q0 = make_float4(__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z), __uint_as_float(r0.w));
    const float4 q1 = make_float4(__uint_as_float(r1.x), __uint_as_float(r1.y), __uint_as_float(r1.z), __uint_as_float(r1.w));
    const float4 q2 = make_float4(__uint_as_float(r2.x), __uint_as_float(r2.y), __uint_as_float(r2.z), __uint_as_float(r2.w));
    float t, u, v;
    if (!(rtr_mt_intersect(o, d, f4xyz(q0), f4xyz(q1), f4xyz(q2), tmin, &t, &u, &v) && t < tmax)) return false;
    if (__float_as_uint(q2.w) & 1u) return alpha_pass<STATS>(sc, __float_as_uint(q0.w), __float_as_uint(q1.w), u, v, st);
    return true;
}

__device__ __forceinline__ rtr_f4 load_f4(const __amdgpu_buffer_rsrc_t buf, int32_t off) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(buf, off, 0, 0);
    return rtr_f4{__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)};
}

struct WaveStats { uint32_t innerIters = 0, innerLanes = 0, triIters = 0, triLanes = 0, refills = 0, over64 = 0, leafLanes = 0; };     /* wave-uniform */
#ifndef RTR_STATS_LEAF_PHASE
#define RTR_STATS_LEAF_PHASE 0
#endif
constexpr bool kFarFirst = RTR_SHADOW_FAR_FIRST != 0;       /* which child of a record the any-hit walk enters first (inner_nodes4); the oracle's trace_wide restates both */

template <int STACK, int OCT, bool STATS>
__device__ __forceinline__ void inner_nodes4(const __amdgpu_buffer_rsrc_t nodeBuf, const uint4* ldsTop, const uint32_t topCount,
                                             int32_t* lds, int32_t& curRef, int32_t*& spRef,
                                             const rtr_v3 ga, const rtr_v3 gb, const float tmin, const float tmax, const uint32_t kInnerMin,
                                             WaveStats& ws, LocalStats& st) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    int32_t cur = curRef;
    int32_t* sp = spRef;
    const float tmaxC = tmax;
    for (;;) {
        const unsigned long long innerMask = __ballot(cur >= 0);
        if (innerMask == 0ull) break;
        if ((uint32_t)__popcll(innerMask) <= kInnerMin && __ballot(cur < 0 && cur != kDone) != 0ull) break;
        if (STATS) { ws.innerIters++; ws.innerLanes += (uint32_t)__popcll(innerMask); }
        if (cur >= 0) {
            if (STATS) { st.nodes++; st.shadowNodes++; }
            /* one 64-B four-wide node = the whole visit; the first topCount entries (the top levels: breadth-first order) are in
             * LDS, which takes those visits — 35-40 % of all — off the L1's tag look-ups, the busiest unit of this kernel */
            u32x4 q0, q1, q2, q3;
            const bool inLds = __builtin_amdgcn_inverse_ballot_w64(__builtin_amdgcn_ballot_w64((uint32_t)cur < topCount));      /* the mask once: branch and select from one compare */
            if (inLds) {
                const uint4* t = ldsTop + cur * 4;
                const uint4 a0 = t[0], a1 = t[1], a2 = t[2], a3 = t[3];
                q0 = u32x4{a0.x, a0.y, a0.z, a0.w}; q1 = u32x4{a1.x, a1.y, a1.z, a1.w};
                q2 = u32x4{a2.x, a2.y, a2.z, a2.w}; q3 = u32x4{a3.x, a3.y, a3.z, a3.w};
            } else {
                const int32_t nodeOff = cur << 6;
                q0 = __builtin_amdgcn_raw_buffer_load_b128(nodeBuf, nodeOff, 0, RTR_NODE_AUX);
                q1 = __builtin_amdgcn_raw_buffer_load_b128(nodeBuf, nodeOff + 16, 0, RTR_NODE_AUX);
                q2 = __builtin_amdgcn_raw_buffer_load_b128(nodeBuf, nodeOff + 32, 0, RTR_NODE_AUX);
                q3 = __builtin_amdgcn_raw_buffer_load_b128(nodeBuf, nodeOff + 48, 0, RTR_NODE_AUX);
            }
            const int32_t top = *sp;                           /* speculative: hides the pop's LDS latency under the node loads */
            const int32_t c0 = (int32_t)q3.x, c1 = (int32_t)q3.y, c2 = (int32_t)q3.z, c3 = (int32_t)q3.w;
            float t0, t1, t2, t3;
            /* an empty slot (only slots 2, 3 can be) holds an inside-out box with infinite planes: in the octant forms its entry
             * is +inf and its exit -inf, so it is never hit and needs no test of its code; the mixed form's per-axis min / max would
             * turn it back into an all-enclosing box, so that form looks at the code */
            const bool h0 = slab_wide<OCT, kFarFirst>(q0.x, q0.y, q0.z, ga, gb, tmin, tmaxC, t0);
            const bool h1 = slab_wide<OCT, kFarFirst>(q0.w, q1.x, q1.y, ga, gb, tmin, tmaxC, t1);
            const bool h2 = slab_wide<OCT, kFarFirst>(q1.z, q1.w, q2.x, ga, gb, tmin, tmaxC, t2) & (OCT < 8 || c2 != kDone);
            const bool h3 = slab_wide<OCT, kFarFirst>(q2.y, q2.z, q2.w, ga, gb, tmin, tmaxC, t3) & (OCT < 8 || c3 != kDone);
            /* descend into the BEST hit child — by default the one that exits last (kFarFirst, below), in rounds 1-4 the nearest —
             * strict comparison: ties go to the lower slot; the others go on the stack in slot order.  e_k: slot k displaced the best
             * so far; the slot entered is the last one that did (slot 0 if none did).
             * (Taking the first hit slot instead of the best saves instructions and costs 2 % more time; ordering the others too —
             * a 5-exchange sort, or just the second best on top — costs more instructions than the better order saves:
             * 2.15 / 2.22 ms against 2.05.)  Which slots are stacked is mask arithmetic on the comparison results, not a
             * comparison of codes — written on wave masks, so that every comparison is issued once and the rest is scalar (from bools
             * the compiler derived !(t2 < tn) with a second vector compare). */
            const unsigned long long H0 = __builtin_amdgcn_ballot_w64(h0), H1 = __builtin_amdgcn_ballot_w64(h1), H2 = __builtin_amdgcn_ballot_w64(h2), H3 = __builtin_amdgcn_ballot_w64(h3);
            /* kFarFirst (round 5; the default): t_k is child k's EXIT distance and the child entered is the one that exits LAST (strict >, ties
             * to the lower slot) — a shadow ray's occluder sits, more often than not, towards the light's end of the ray, and any-hit does not
             * care which occluder it finds: 10.6 -> 8.0 record visits and 3.40 -> 2.60 triangle tests per ray on the bench frame for the same
             * instructions per visit (profiles/r05/far_first_lab.log).  Otherwise the nearest entry first, the closest-hit order of rounds 1-4. */
            float tn = h0 ? t0 : (kFarFirst ? -3.0e38f : 3.0e38f);
            const unsigned long long E1 = H1 & __builtin_amdgcn_ballot_w64(kFarFirst ? t1 > tn : t1 < tn);
            const bool e1 = __builtin_amdgcn_inverse_ballot_w64(E1); tn = e1 ? t1 : tn;
            const unsigned long long E2 = H2 & __builtin_amdgcn_ballot_w64(kFarFirst ? t2 > tn : t2 < tn);
            const bool e2 = __builtin_amdgcn_inverse_ballot_w64(E2); tn = e2 ? t2 : tn;
            const unsigned long long E3 = H3 & __builtin_amdgcn_ballot_w64(kFarFirst ? t3 > tn : t3 < tn);
            const bool e3 = __builtin_amdgcn_inverse_ballot_w64(E3);
            int32_t next = e3 ? c3 : (e2 ? c2 : (e1 ? c1 : c0));
            const bool any = __builtin_amdgcn_inverse_ballot_w64(H0 | H1 | H2 | H3);
            /* the stack pointer is the LDS address of the top entry, so a push is a store and an add.  A visit pushes at most three
             * entries: while every lane of the wave has three free (one comparison and a branch the whole wave takes or skips) the pushes
             * need no bound; otherwise the visit runs its checked form, in which a lane whose stack is full is abandoned to the tail
             * kernel.  All STACK entries are usable, none is a guard. */
            const bool p0 = __builtin_amdgcn_inverse_ballot_w64(H0 & (E1 | E2 | E3)), p1 = __builtin_amdgcn_inverse_ballot_w64(H1 & ~(E1 & ~E2 & ~E3));
            const bool p2 = __builtin_amdgcn_inverse_ballot_w64(H2 & ~(E2 & ~E3)), p3 = __builtin_amdgcn_inverse_ballot_w64(H3 & ~E3);
            int32_t* const full = lds + STACK * kTraceBlock;
            if (__ballot(sp > full - 3 * kTraceBlock) == 0ull) {
                if (p0) { sp[kTraceBlock] = c0; sp += kTraceBlock; }
                if (p1) { sp[kTraceBlock] = c1; sp += kTraceBlock; }
                if (p2) { sp[kTraceBlock] = c2; sp += kTraceBlock; }
                if (p3) { sp[kTraceBlock] = c3; sp += kTraceBlock; }
                if (!any) { next = top; sp -= kTraceBlock; }    /* nothing hit: pop (slot 0 holds kDone) */
            } else {
                bool over = false;
                if (p0) { if (sp < full) { sp[kTraceBlock] = c0; sp += kTraceBlock; } else over = true; }
                if (p1) { if (sp < full) { sp[kTraceBlock] = c1; sp += kTraceBlock; } else over = true; }
                if (p2) { if (sp < full) { sp[kTraceBlock] = c2; sp += kTraceBlock; } else over = true; }
                if (p3) { if (sp < full) { sp[kTraceBlock] = c3; sp += kTraceBlock; } else over = true; }
                if (!any) { next = top; sp -= kTraceBlock; }
                if (over) next = kAbandoned;                    /* needs more than the LDS stack: the caller hands the ray to the tail kernel */
            }
            cur = next;
        }
    }
    curRef = cur; spRef = sp;
}

/* The same kernel over the 4-wide view of the tree (DeviceScene::nodes4, built by k_wide_nodes): a visit is one 64-B record
 * (four loads issued together) holding up to four child boxes, so a ray makes about half as many DEPENDENT visits; the
 * instruction and look-up totals stay about the same.  2.17 -> 2.05 ms on the bench frame; identical visibility bits. */
/* STATS: the same kernel with per-ray work counters (a ray's sequence of visits and triangle tests depends only on the ray and the
 * tree: the hit child that exits last first (kFarFirst), ties to the lower slot, the others stacked in slot order — whatever the scheduling, the octant
 * form or the queue mode, so the counting form's numbers are the timed form's, and the oracle restates them), per-trip lane counts
 * of the two phases, and a shader-clock stamp pair per wave. */
template <int STACK, bool LISTS, bool STATS>
__global__ __launch_bounds__(kTraceBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_shadow_trace4(DeviceScene sc, const RayQueue queue,
                                                              const uint32_t* __restrict__ count, uint32_t* nextBatch,
                                                              uint8_t* __restrict__ vis, uint32_t visFill, uint32_t kBatch, uint32_t kRefill,
                                                              uint32_t kInnerMin, uint32_t* overflow, uint32_t overflowCap, uint32_t octForms, uint32_t topCount,
                                                              const uint2* __restrict__ lists, uint32_t listStride, Counters* stats,
                                                              unsigned long long* __restrict__ clk) {
    __shared__ int32_t s_stack[(STACK + 1) * kTraceBlock];    /* slot 0, below the stack, holds kDone for good */
    __shared__ uint4 s_top[kTopNodes * 4];                    /* the first topCount (<= kTopNodes) four-wide entries */
#if RTR_REFILL_LDS
    /* What only the refill block needs — the queue's three arrays, the visibility array, the grid — waits in LDS, not in scalar registers:
     * the kernel holds more wave-uniform values than a wave has SGPRs for, and the ones the compiler parked in VGPR lanes came back
     * through ~20 v_readlane per refill pass (vector-issue slots, the unit this kernel is bound by); four ds_read_b128 are not. */
    struct RefillConsts { uint64_t dt, slot, origin, vis; float cx, cy, cz, sx; float sy, sz; uint32_t slotMask, visFill; };     /* addresses as integers: a pointer read back from LDS is a generic one (flat_load) */
    static_assert(sizeof(RefillConsts) == 64, "four 16-B LDS reads");
    typedef const __attribute__((address_space(1))) rtr_f4* global_f4;
    typedef const __attribute__((address_space(1))) uint32_t* global_u32;
    typedef __attribute__((address_space(1))) uint8_t* global_u8;
    __shared__ RefillConsts s_rc;
    if (threadIdx.x == 0) {
        const rtr_v3 cw = rtr_wide_centre_world(sc.grid->origin, sc.grid->scale, sc.grid->wideCentreXY, sc.grid->wideCentreZ);
        s_rc.dt = (uint64_t)queue.dt; s_rc.slot = (uint64_t)queue.slot; s_rc.origin = (uint64_t)queue.origin; s_rc.vis = (uint64_t)vis;
        s_rc.cx = cw.x; s_rc.cy = cw.y; s_rc.cz = cw.z; s_rc.sx = sc.grid->scale[0]; s_rc.sy = sc.grid->scale[1]; s_rc.sz = sc.grid->scale[2];
        s_rc.slotMask = queue.slotMask; s_rc.visFill = visFill;
    }
#endif
    int32_t* lds = s_stack + threadIdx.x;
    lds[0] = kDone;
    for (uint32_t i = threadIdx.x; i < topCount * 4u; i += kTraceBlock) s_top[i] = sc.nodes4[i];
#if RTR_LIST_SHARE
    __shared__ unsigned long long s_drained;                  /* bit r: batch list r was seen drained by a wave of this workgroup (cursors only grow) */
    if (threadIdx.x == 0) s_drained = 0ull;
#endif
    __syncthreads();
    /* clock of this launch: shader-clock ticks over 100-MHz ticks, lane 0 of the first workgroup of each XCD (bench.py: roofline.clock_mhz) */
    const bool stamp = clk != nullptr && blockIdx.x < kQueueRegions && threadIdx.x == 0;
    unsigned long long c0 = 0, r0 = 0;
    if (stamp || STATS) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    LocalStats st;
    WaveStats ws;
    uint32_t batchPos = 0, batchEnd = 0;     /* wave-uniform */
    bool exhausted = false;                  /* wave-uniform */
    /* LISTS == false: the plain queue, cut into kQueueRegions contiguous regions with one batch cursor each (64 B apart).  A
     * workgroup starts on region (blockIdx mod 8) — workgroups are dealt round-robin to the 8 XCDs, so a cursor is hammered by
     * one XCD's waves — and moves on only when that region is empty (one counter sustains ~88 atomics/us).
     * LISTS == true: the queue was binned by direction octant (k_shadow_gen_oct): list r = octant * 8 + xcd holds batches
     * {first, count} (length lens[r], cursor nextBatch[16 r]).  A workgroup starts on the octant its share of the launch falls
     * into — octants get workgroups in proportion to their batches — and there on the list of its own XCD, then takes the other
     * XCDs' lists of that octant, then the next octant: the rays in a wave share their direction signs except around such a move.
     * (With one list per octant, shared by all XCDs, the same kernel took 3.8 ms instead of 1.9.) */
    const uint32_t* __restrict__ lens = nextBatch - kBatchCursorWord + kQueueListLens;
    const uint32_t n = LISTS ? 0u : *count;
    const uint32_t regionLen = ((n + kQueueRegions - 1) / kQueueRegions + kBatch - 1) / kBatch * kBatch;
    const uint32_t myXcd = blockIdx.x % kQueueRegions;
    uint32_t myRegion = myXcd;
    if (LISTS) {
        uint32_t total = 0;
        for (uint32_t r = 0; r < kQueueLists; ++r) total += lens[r];
        const uint32_t target = (uint32_t)(((unsigned long long)blockIdx.x * total) / gridDim.x);
        uint32_t upTo = 0;
        myRegion = 0;
        for (uint32_t r = 0; r < kQueueLists; ++r) { upTo += lens[r]; if (target < upTo) { myRegion = r / kQueueRegions; break; } }
    }
    uint32_t regionTry = 0;                  /* wave-uniform: regions / lists found empty so far (cursors only grow) */
    int32_t cur = kDone;
    int32_t* sp = lds;                       /* LDS address of the top entry (lds = slot 0 = empty) */
    rtr_v3 o = rtr_mk(0, 0, 0), d = rtr_mk(0, 0, 0), ga = rtr_mk(0, 0, 0), gb = rtr_mk(0, 0, 0);   /* t(q) = q * ga + gb */
    float tmax = 0.f;
    uint32_t slot = 0, rayIndex = 0, res = kResNone, occ = 0;
    const float tmin = 0.001f;
    /* nodes and triangles through buffer resources: the address of a visit is one 32-bit shift, not 64-bit lane arithmetic
     * (2.28 -> 2.17 ms, and 62 -> 47 VGPRs) */
    const __amdgpu_buffer_rsrc_t nodeBuf = __builtin_amdgcn_make_buffer_rsrc((void*)sc.nodes4, 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t triBuf = __builtin_amdgcn_make_buffer_rsrc((void*)sc.tris, 0, 0xffffffff, 0x00020000);

    for (;;) {
        /* ---- retire finished rays, refill idle lanes from the wave's batch ---- */
        const unsigned long long idle = __ballot(cur == kDone);
        const uint32_t nIdle = (uint32_t)__popcll(idle);
        if (nIdle >= kRefill || nIdle == 64u) {
            if (STATS) ws.refills++;
#if RTR_REFILL_LDS
            __asm__ volatile("" ::: "memory");                 /* the constants are read HERE, every pass: hoisted out of the loop they would cost 16 VGPRs for the whole kernel */
            const RefillConsts rc = s_rc;
            const global_u8 visOut = (global_u8)rc.vis;
            const uint32_t fill = rc.visFill;
#else
            uint8_t* const visOut = vis;
            const uint32_t fill = visFill;
#endif
            if (cur == kDone && res != kResNone) {
                if (res == 2u) { const uint32_t at = atomicAdd(overflow, 1u); if (at < overflowCap) overflow[1u + at] = rayIndex; }      /* finished by k_shadow_tail (a full list: it redoes the whole queue) */
                else { if (res != fill) visOut[slot] = (uint8_t)res; occ += res; }      /* the array was pre-filled with the commoner outcome: only the other one is stored */
                res = kResNone;
            }
            if (!exhausted) {
                if (batchPos == batchEnd) {
                    for (;;) {
                        if (regionTry >= (LISTS ? kQueueLists : kQueueRegions)) { exhausted = true; break; }
                        if (LISTS) {
                            const uint32_t r = (myRegion + regionTry / kQueueRegions) % kQueueRegions * kQueueRegions + (myXcd + regionTry) % kQueueRegions;
                            const uint32_t len = lens[r];
#if RTR_LIST_SHARE
                            /* a list one wave of this workgroup found drained is not asked again by its other three: at the end of a launch every
                             * wave walks all the lists, one atomic round trip each on cursors the whole chip is hammering */
                            const bool known = (*(volatile __attribute__((address_space(3))) unsigned long long*)&s_drained >> r) & 1ull;      /* address space kept: a ds_read, not a flat load */
#else
                            const bool known = false;
#endif
                            if (len != 0u && !known) {
                                uint32_t got = 0;
                                if ((threadIdx.x & 63u) == 0) got = atomicAdd(nextBatch + 16u * r, 1u);
                                got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
                                if (got < len) {
                                    const uint2 dsc = lists[(size_t)r * listStride + got];
                                    batchPos = dsc.x; batchEnd = dsc.x + dsc.y;
                                    break;
                                }
#if RTR_LIST_SHARE
                                if ((threadIdx.x & 63u) == 0) atomicOr(&s_drained, 1ull << r);
#endif
                            }
                        } else {
                            const uint32_t r = (myRegion + regionTry) % kQueueRegions;
                            const uint32_t lo = r * regionLen;
                            uint32_t hi = lo + regionLen; if (hi > n) hi = n;
                            uint32_t b = hi;
                            if (lo < hi) {
                                uint32_t got = 0;
                                if ((threadIdx.x & 63u) == 0) got = atomicAdd(nextBatch + 16u * r, kBatch);
                                got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
                                b = got < regionLen ? lo + got : hi;
                            }
                            if (b < hi) { batchPos = b; batchEnd = (b + kBatch < hi) ? b + kBatch : hi; break; }
                        }
                        ++regionTry;
                    }
                }
                if (!exhausted) {
                    const uint32_t avail = batchEnd - batchPos;
                    const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                    if (cur == kDone && prefix < avail) {
                        rayIndex = batchPos + prefix;
                        bool into = false;                               /* the ray leaves its surface point into the surface: its own triangle's leaf first */
                        int32_t ownLeaf = 0;
#if RTR_REFILL_LDS
                        {   /* queue_load() through the addresses kept in LDS */
                            const global_f4 qdt = (global_f4)rc.dt, qorg = (global_f4)rc.origin;
                            const global_u32 qslot = (global_u32)rc.slot;
                            rtr_f4 a;
                            if (octForms & 2u) { a = __builtin_nontemporal_load(qdt + rayIndex); slot = __builtin_nontemporal_load(qslot + rayIndex); }
                            else { a = qdt[rayIndex]; slot = qslot[rayIndex]; }
                            into = (slot & kRayIntoSurface) != 0u;
                            slot &= ~kRayIntoSurface;
                            const rtr_f4 og = qorg[slot & rc.slotMask];
                            o = rtr_mk(og.x, og.y, og.z); d = rtr_mk(a.x, a.y, a.z); tmax = a.w;
                            ownLeaf = __float_as_int(og.w);
                        }
#else
                        into = (queue.slot[rayIndex] & kRayIntoSurface) != 0u;
                        queue_load(queue, rayIndex, o, d, tmax, slot, octForms & 2u);
                        ownLeaf = __float_as_int(queue.origin[slot & queue.slotMask].w);
#endif
                        if (STATS) { st.rays++; st.shadow++; }
                        if (!(tmax > tmin)) {
                            if (fill != 0u) visOut[slot] = 0;                 /* empty interval: nothing can be hit (oracle trace(): same rule) */
                        } else {
                            const rtr_v3 idir = rtr_mk(rtr_safe_rcp_dir(d.x), rtr_safe_rcp_dir(d.y), rtr_safe_rcp_dir(d.z));
                            /* gb about the scene's wide centre: the records' planes are offsets from it */
#if RTR_REFILL_LDS
                            rtr_ray_grid_about(o, idir, rtr_mk(rc.sx, rc.sy, rc.sz), rtr_mk(rc.cx, rc.cy, rc.cz), &ga, &gb);
#else
                            rtr_ray_grid_centre(o, idir, sc.grid->origin, sc.grid->scale, sc.grid->wideCentreXY, sc.grid->wideCentreZ, &ga, &gb);
#endif
                            cur = 0; sp = lds; res = 0u;
                            /* A ray that leaves its surface point INTO the surface (marked by the queue build: dot(hitNormal, direction) < 0) starts
                             * 0.01 above the triangle it comes from and nearly always re-enters it a hair's breadth on: 44 % of the bench frame's
                             * rays.  Its walk starts at that triangle's LEAF, the root waiting on the stack — an any-hit answer does not depend on
                             * the order triangles are met in — so the leaf phase answers it without a single record visit (14.3 -> 10.6 visits
                             * per ray over the frame; the oracle restates the rule: trace_wide's firstLeaf). */
                            if (into) { sp = lds + kTraceBlock; *sp = 0; cur = ownLeaf; }
                        }
                    }
                    batchPos += (nIdle < avail) ? nIdle : avail;
                }
            }
        }
        if (__ballot(cur != kDone) == 0ull) {
            if (exhausted) break;                /* every lane is idle and was retired above (nIdle == 64) */
            continue;
        }
        /* ---- inner nodes ("while-while" with an early exit, see k_shadow_trace) ----
         * When every lane that is at an inner node has the same direction signs (usual: a wave's rays are neighbouring pixels
         * aimed at the same light triangle), the loop runs in the form compiled for that octant, without the per-axis min/max. */
        {
            const unsigned long long innerNow = __ballot(cur >= 0);
            if (innerNow != 0ull) {
                const uint32_t oct = (ga.x < 0.f ? 1u : 0u) | (ga.y < 0.f ? 2u : 0u) | (ga.z < 0.f ? 4u : 0u);
                const uint32_t woct = (uint32_t)__builtin_amdgcn_readlane((int)oct, (int)__ffsll((long long)innerNow) - 1);
                const bool mixed = __ballot(cur >= 0 && oct != woct) != 0ull;
                switch ((mixed || (octForms & 1u) == 0u) ? 8u : woct) {
                    case 0: inner_nodes4<STACK, 0, STATS>(nodeBuf, s_top, topCount, lds, cur, sp, ga, gb, tmin, tmax, kInnerMin, ws, st); break;
                    case 1: inner_nodes4<STACK, 1, STATS>(nodeBuf, s_top, topCount, lds, cur, sp, ga, gb, tmin, tmax, kInnerMin, ws, st); break;
                    case 2: inner_nodes4<STACK, 2, STATS>(nodeBuf, s_top, topCount, lds, cur, sp, ga, gb, tmin, tmax, kInnerMin, ws, st); break;
                    case 3: inner_nodes4<STACK, 3, STATS>(nodeBuf, s_top, topCount, lds, cur, sp, ga, gb, tmin, tmax, kInnerMin, ws, st); break;
                    case 4: inner_nodes4<STACK, 4, STATS>(nodeBuf, s_top, topCount, lds, cur, sp, ga, gb, tmin, tmax, kInnerMin, ws, st); break;
                    case 5: inner_nodes4<STACK, 5, STATS>(nodeBuf, s_top, topCount, lds, cur, sp, ga, gb, tmin, tmax, kInnerMin, ws, st); break;
                    case 6: inner_nodes4<STACK, 6, STATS>(nodeBuf, s_top, topCount, lds, cur, sp, ga, gb, tmin, tmax, kInnerMin, ws, st); break;
                    case 7: inner_nodes4<STACK, 7, STATS>(nodeBuf, s_top, topCount, lds, cur, sp, ga, gb, tmin, tmax, kInnerMin, ws, st); break;
                    default: inner_nodes4<STACK, 8, STATS>(nodeBuf, s_top, topCount, lds, cur, sp, ga, gb, tmin, tmax, kInnerMin, ws, st); break;
                }
                if (cur == kAbandoned) { res = 2u; cur = kDone; }
            }
        }
        /* ---- leaves ---- */
        if (STATS) {          /* the same per-lane tests, as a wave-uniform loop so that its trips and the lanes working in them can be counted */
            const bool atLeaf = cur < 0 && cur != kDone;
            const uint32_t code = (uint32_t)~cur;
            const uint32_t first = code >> 3, cnt = (code & 7u) + 1u;
            bool hit = false;
#if RTR_STATS_LEAF_PHASE      /* experiment (profiles/experiments/leaf_phase_pairs.py): the counters re-used — trips := leaf phases, lanes := (ray, triangle) pairs waiting at their start, refills := phases with more than 64 pairs + (lanes at a leaf << 32) */
            { const uint32_t pairs = wave_sum_u32(atLeaf ? cnt : 0u); ws.triIters++; ws.triLanes += pairs; if (pairs > 64u) ws.over64++; ws.leafLanes += (uint32_t)__popcll(__ballot(atLeaf)); }
#endif
            for (uint32_t i = 0;; ++i) {
                const bool go = atLeaf && i < cnt && !hit;
                const unsigned long long m = __ballot(go);
                if (m == 0ull) break;
#if !RTR_STATS_LEAF_PHASE
                ws.triIters++; ws.triLanes += (uint32_t)__popcll(m);
#endif
                if (go) { st.tris++; st.shadowTris++; hit = tri_any<true>(sc, triBuf, first + i, o, d, tmin, tmax, st); }
            }
            if (atLeaf) {
                if (hit) { res = 1u; cur = kDone; }
                else { cur = *sp; sp -= kTraceBlock; }
            }
        } else if (cur < 0 && cur != kDone) {
            const uint32_t code = (uint32_t)~cur;
            const uint32_t first = code >> 3, cnt = (code & 7u) + 1u;
            bool hit = false;
            for (uint32_t i = 0; i < cnt && !hit; ++i) hit = tri_any<false>(sc, triBuf, first + i, o, d, tmin, tmax, st);
            if (hit) { res = 1u; cur = kDone; }
            else { cur = *sp; sp -= kTraceBlock; }                        /* slot 0 holds kDone: an empty stack ends the ray (visible) */
        }
    }
    {   /* how many of this launch's rays were occluded: the host picks the next frame's pre-fill of the visibility array by it */
        const uint32_t occWave = wave_sum_u32(occ);
        if ((threadIdx.x & 63u) == 0 && occWave) atomicAdd(nextBatch - kBatchCursorWord + kOccludedWord, occWave);      /* nextBatch = the control block + kBatchCursorWord (the launcher's expression) */
    }
    if (stamp || STATS) {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (stamp) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
        if (STATS && (threadIdx.x & 63u) == 0) {
            atomicAdd(&stats->innerIters, (unsigned long long)ws.innerIters); atomicAdd(&stats->innerLanes, (unsigned long long)ws.innerLanes);
            atomicAdd(&stats->triIters, (unsigned long long)ws.triIters); atomicAdd(&stats->triLanes, (unsigned long long)ws.triLanes);
#if RTR_STATS_LEAF_PHASE
            atomicAdd(&stats->refills, (unsigned long long)ws.over64 | ((unsigned long long)ws.leafLanes << 32));
#else
            atomicAdd(&stats->refills, (unsigned long long)ws.refills);
#endif
            atomicAdd(&stats->clockCycles, c1 - c0); atomicAdd(&stats->clockRef, r1 - r0);
        }
    }
    if (STATS) st.flush(stats);
}

/* ---- wavefront stage 1 over the 4-wide view: one camera ray per lane, closest hit (tunable primary_wide) ---------------------------
 * The camera-ray kernel is bound by its deepest rays' chains of DEPENDENT fetches (k_primary: 31.8 BVH2 visits per ray on the bench
 * frame); the 4-wide view the shadow rays walk makes about half as many (round 2 measured 18.3), each twice the vector instructions —
 * which profiles/microbench/visit_mix.hip shows are cheaper than the counters made them look.  Walk rule = k_shadow_trace4's, with a
 * closest hit: slab tests against the ray's best t so far, the nearest hit child entered (strict <: ties to the lower slot), the others
 * stacked in slot order, ALL triangles of a leaf tested (min over (t, customIndex, primitiveID), as trace()), RTR_WIDE_STACK = 16 LDS
 * entries, beyond which the ray goes to the redo list and k_primary_tail walks it over the BVH2 from scratch (both parts counted). */
template <bool STATS>
__global__ __launch_bounds__(kBlock) void k_primary4(DeviceScene sc, FrameBatch fb, float4* hitTuvp, uint32_t* hitCustom, float4* rayOrigin, Counters* stats,
                                                     uint32_t* redoCount, uint32_t* redoList, uint32_t planeBlocks) {
    __shared__ int32_t s_stack[RTR_WIDE_STACK * kBlock];
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t plane = blockIdx.x / planeBlocks;
    const RenderArgs& ra = fb.ra[plane / fb.ra[0].spp];
    const uint32_t i = plane % fb.ra[0].spp;
    const uint32_t q = (blockIdx.x - plane * planeBlocks) * kBlock + threadIdx.x;
    uint32_t px, lrow, py;
    if (!pixel_of(ra, q, px, lrow, py)) return;
    LocalStats st;
    const rtr_v3 o = rtr_ld3(ra.cam.position);
    const rtr_v3 d = primary_dir(ra, px, py, i);
    const float tmin = 0.001f, tmax = 10000.0f;
    if (STATS) { st.rays++; st.primary++; }
    const rtr_v3 idir = rtr_mk(rtr_safe_rcp_dir(d.x), rtr_safe_rcp_dir(d.y), rtr_safe_rcp_dir(d.z));
    rtr_v3 ga, gb;
    rtr_ray_grid_centre(o, idir, sc.grid->origin, sc.grid->scale, sc.grid->wideCentreXY, sc.grid->wideCentreZ, &ga, &gb);
    const __amdgpu_buffer_rsrc_t nodeBuf = __builtin_amdgcn_make_buffer_rsrc((void*)sc.nodes4, 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t triBuf = __builtin_amdgcn_make_buffer_rsrc((void*)sc.tris, 0, 0xffffffff, 0x00020000);
    HitRec best; best.t = tmax; best.u = 0.f; best.v = 0.f; best.custom = RTR_MISS; best.prim = RTR_MISS; best.leaf = 0;
    int32_t* const lds = s_stack + threadIdx.x;
    int sp = 0;                                          /* entries held: lds[0 .. sp-1] at stride kBlock */
    int32_t cur = 0;
    bool over = false;
    for (;;) {
        if (cur >= 0) {
            if (STATS) st.nodes++;
            const int32_t nodeOff = cur << 6;
            const u32x4 q0 = __builtin_amdgcn_raw_buffer_load_b128(nodeBuf, nodeOff, 0, 0), q1 = __builtin_amdgcn_raw_buffer_load_b128(nodeBuf, nodeOff + 16, 0, 0);
            const u32x4 q2 = __builtin_amdgcn_raw_buffer_load_b128(nodeBuf, nodeOff + 32, 0, 0), q3 = __builtin_amdgcn_raw_buffer_load_b128(nodeBuf, nodeOff + 48, 0, 0);
            const int32_t c0 = (int32_t)q3.x, c1 = (int32_t)q3.y, c2 = (int32_t)q3.z, c3 = (int32_t)q3.w;
            const float lim = rtr_hwmin(best.t, best.t);
            float t0, t1, t2, t3;
            const bool h0 = slab_wide<8>(q0.x, q0.y, q0.z, ga, gb, tmin, lim, t0);
            const bool h1 = slab_wide<8>(q0.w, q1.x, q1.y, ga, gb, tmin, lim, t1);
            const bool h2 = slab_wide<8>(q1.z, q1.w, q2.x, ga, gb, tmin, lim, t2) & (c2 != kDone);
            const bool h3 = slab_wide<8>(q2.y, q2.z, q2.w, ga, gb, tmin, lim, t3) & (c3 != kDone);
            float tn = h0 ? t0 : 3.0e38f;
            const bool e1 = h1 & (t1 < tn); tn = e1 ? t1 : tn;
            const bool e2 = h2 & (t2 < tn); tn = e2 ? t2 : tn;
            const bool e3 = h3 & (t3 < tn);
            int32_t next = e3 ? c3 : (e2 ? c2 : (e1 ? c1 : c0));
            const bool any = h0 | h1 | h2 | h3;
            const bool p0 = h0 & (e1 | e2 | e3), p1 = h1 & !(e1 & !e2 & !e3), p2 = h2 & !(e2 & !e3), p3 = h3 & !e3;
            if (p0) { if (sp < RTR_WIDE_STACK) lds[sp++ * kBlock] = c0; else over = true; }
            if (p1) { if (sp < RTR_WIDE_STACK) lds[sp++ * kBlock] = c1; else over = true; }
            if (p2) { if (sp < RTR_WIDE_STACK) lds[sp++ * kBlock] = c2; else over = true; }
            if (p3) { if (sp < RTR_WIDE_STACK) lds[sp++ * kBlock] = c3; else over = true; }
            if (over) break;
            if (any) { cur = next; continue; }
        } else {
            const uint32_t code = (uint32_t)~cur;
            const uint32_t first = code >> 3, cnt = (code & 7u) + 1u;
            for (uint32_t k = 0; k < cnt; ++k) {
                const int32_t off = (int32_t)((first + k) * 48u);
                const rtr_f4 a0 = load_f4(triBuf, off), a1 = load_f4(triBuf, off + 16), a2 = load_f4(triBuf, off + 32);
                if (STATS) st.tris++;
                float t, u, v;
                if (rtr_mt_intersect(o, d, rtr_mk(a0.x, a0.y, a0.z), rtr_mk(a1.x, a1.y, a1.z), rtr_mk(a2.x, a2.y, a2.z), tmin, &t, &u, &v) && t < tmax) {
                    const uint32_t cu = __float_as_uint(a0.w), pr = __float_as_uint(a1.w);
                    if ((__float_as_uint(a2.w) & 1u) && !alpha_pass<STATS>(sc, cu, pr, u, v, st)) continue;
                    if (t < best.t || (t == best.t && (cu < best.custom || (cu == best.custom && pr < best.prim)))) {
                        best.t = t; best.u = u; best.v = v; best.custom = cu; best.prim = pr; best.leaf = cur;
                    }
                }
            }
        }
        if (sp == 0) break;
        cur = lds[--sp * kBlock];
    }
    const size_t k = (size_t)plane * planeBlocks * kBlock + q;
    if (over) redoList[atomicAdd(redoCount, 1u)] = (uint32_t)k;
    else { hitTuvp[k] = make_float4(best.t, best.u, best.v, __uint_as_float(best.prim)); hitCustom[k] = best.custom; origin_leaf_store(rayOrigin, k, best.leaf); }
    if (STATS) st.flush(stats);
}

/* ---- wavefront stage 1, persistent form: closest hit of the camera rays ----------------------------------------------------------
 * The any-hit kernel's machinery around the BVH2 walk of trace(): persistent waves; the work items are (sample, pixel slot) pairs
 * k = sample * planeStride + slot in the canonical 8x8-tile order, cut into kQueueRegions contiguous regions (a band of the image
 * per XCD) with one batch cursor each; a wave refills its idle lanes by ballot, so a lane whose ray ends early takes the next
 * camera ray instead of waiting for the deepest ray of its tile (the one-ray-per-lane k_primary ran at 67 % lane use, 56 % VALU
 * issue and an average of 2.9 resident waves per SIMD: profiles/r02/pmc_summary_bench_sponza1080p_serial.txt).
 * A ray's walk is trace()'s, visit for visit: near child first (ties: child 0), far child stacked, boxes tested against the best t
 * so far, ALL triangles of a leaf tested, closest = min over (t, customIndex, primitiveID); 16 LDS stack entries, beyond which the
 * ray is left to k_primary_tail — so the counters are the oracle's trace_bvh() with its 16-entry rule.
 * (The 4-wide records do not pay here: a closest-hit walk over them makes 18.3 visits per camera ray instead of 31.7, but a
 * 4-wide visit costs ~90 vector instructions against ~45, and this kernel is bound by vector-instruction issue like the any-hit
 * kernel: 349 M wave-instructions and 0.52-0.58 ms against 112 M and 0.35 ms — profiles/r02/primary_wide_kernel.txt.) */
template <int STACK, int OCT, bool STATS>
__device__ __forceinline__ void inner_nodes2(const __amdgpu_buffer_rsrc_t nodeBuf, int32_t* lds, int32_t& cur, int32_t*& sp, uint32_t& res,
                                             const rtr_v3 ga, const rtr_v3 gb, const float tmin, const float tmax, const uint32_t kInnerMin, LocalStats& st) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    for (;;) {
        const unsigned long long innerMask = __ballot(cur >= 0);
        if (innerMask == 0ull) break;
        if ((uint32_t)__popcll(innerMask) <= kInnerMin && __ballot(cur < 0 && cur != kDone) != 0ull) break;
        if (cur >= 0) {
            if (STATS) st.nodes++;
            const int32_t nodeOff = cur << 5;                   /* one 32-B RtrBvhNode = the whole visit */
            const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(nodeBuf, nodeOff, 0, 0);
            const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(nodeBuf, nodeOff + 16, 0, 0);
            const int32_t top = *sp;                             /* speculative: hides the pop's LDS latency under the node loads */
            float tl, tr;
            const bool hl = slab_oct<OCT>(a.x, a.y, b.x, ga, gb, tmin, tmax, tl);
            const bool hr = slab_oct<OCT>(a.z, a.w, b.y, ga, gb, tmin, tmax, tr);
            const bool swap = tr < tl;
            const int32_t c0 = (int32_t)b.z, c1 = (int32_t)b.w;
            const bool both = hl && hr, none = !(hl || hr);
            const bool second = both ? swap : !hl;               /* descend into child 1? (the near child if both are hit) */
            int32_t next = second ? c1 : c0;
            if (both) { sp += kBlock; *sp = swap ? c0 : c1; }    /* one guard entry above the STACK the ray may use */
            if (sp > lds + STACK * kBlock) { res = 2u; next = kDone; }     /* needs a deeper stack: k_primary_tail redoes this ray */
            else if (none) { next = top; sp -= kBlock; }                    /* slot 0 holds kDone */
            cur = next;
        }
    }
}

template <int STACK, bool STATS>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_primary_persist(DeviceScene sc, RenderArgs ra, float4* hitTuvp, uint32_t* hitCustom, float4* rayOrigin,
                                                         Counters* stats, uint32_t* redoCount, uint32_t* redoList, uint32_t* cursors,
                                                         uint32_t planeStride, uint32_t kBatch, uint32_t kRefill, uint32_t kInnerMin) {
    __shared__ int32_t s_stack[(STACK + 1 + 1) * kBlock];    /* slot 0 holds kDone for good; one guard entry above the stack (inner_nodes2) */
    int32_t* lds = s_stack + threadIdx.x;
    lds[0] = kDone;
    LocalStats st;
    uint32_t batchPos = 0, batchEnd = 0;     /* wave-uniform */
    bool exhausted = false;
    const uint32_t n = ra.spp * planeStride;
    const uint32_t regionLen = ((n + kQueueRegions - 1) / kQueueRegions + kBatch - 1) / kBatch * kBatch;
    const uint32_t myRegion = blockIdx.x % kQueueRegions;
    uint32_t regionTry = 0;
    int32_t cur = kDone;
    int32_t* sp = lds;
    const rtr_v3 o = rtr_ld3(ra.cam.position);
    rtr_v3 d = rtr_mk(0, 0, 0), ga = rtr_mk(0, 0, 0), gb = rtr_mk(0, 0, 0);
    HitRec best; best.t = 0.f; best.u = 0.f; best.v = 0.f; best.custom = RTR_MISS; best.prim = RTR_MISS; best.leaf = 0;
    uint32_t item = 0, res = kResNone;
    const float tmin = 0.001f, tmax = 10000.0f;
    const __amdgpu_buffer_rsrc_t nodeBuf = __builtin_amdgcn_make_buffer_rsrc((void*)sc.nodes, 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t triBuf = __builtin_amdgcn_make_buffer_rsrc((void*)sc.tris, 0, 0xffffffff, 0x00020000);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

    for (;;) {
        const unsigned long long idle = __ballot(cur == kDone);
        const uint32_t nIdle = (uint32_t)__popcll(idle);
        if (nIdle >= kRefill || nIdle == 64u) {
            if (cur == kDone && res != kResNone) {
                if (res == 2u) redoList[atomicAdd(redoCount, 1u)] = item;              /* finished by k_primary_tail */
                else { hitTuvp[item] = make_float4(best.t, best.u, best.v, __uint_as_float(best.prim)); hitCustom[item] = best.custom; origin_leaf_store(rayOrigin, item, best.leaf); }
                res = kResNone;
            }
            if (!exhausted) {
                if (batchPos == batchEnd) {
                    for (;;) {
                        if (regionTry >= kQueueRegions) { exhausted = true; break; }
                        const uint32_t r = (myRegion + regionTry) % kQueueRegions;
                        const uint32_t lo = r * regionLen;
                        uint32_t hi = lo + regionLen; if (hi > n) hi = n;
                        uint32_t b = hi;
                        if (lo < hi) {
                            uint32_t got = 0;
                            if ((threadIdx.x & 63u) == 0) got = atomicAdd(cursors + 16u * r, kBatch);
                            got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
                            b = got < regionLen ? lo + got : hi;
                        }
                        if (b < hi) { batchPos = b; batchEnd = (b + kBatch < hi) ? b + kBatch : hi; break; }
                        ++regionTry;
                    }
                }
                if (!exhausted) {
                    const uint32_t avail = batchEnd - batchPos;
                    const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                    if (cur == kDone && prefix < avail) {
                        item = batchPos + prefix;
                        const uint32_t i = item / planeStride, q = item - i * planeStride;      /* sample, pixel slot */
                        uint32_t px, lrow, py;
                        if (pixel_of(ra, q, px, lrow, py)) {                                   /* padding slots issue no ray */
                            d = primary_dir(ra, px, py, i);
                            if (STATS) { st.rays++; st.primary++; }
                            const rtr_v3 idir = rtr_mk(rtr_safe_rcp_dir(d.x), rtr_safe_rcp_dir(d.y), rtr_safe_rcp_dir(d.z));
                            rtr_ray_grid(o, idir, sc.grid->origin, sc.grid->scale, &ga, &gb);
                            best.t = tmax; best.u = 0.f; best.v = 0.f; best.custom = RTR_MISS; best.prim = RTR_MISS; best.leaf = 0;
                            cur = 0; sp = lds; res = 0u;
                        }
                    }
                    batchPos += (nIdle < avail) ? nIdle : avail;
                }
            }
        }
        if (__ballot(cur != kDone) == 0ull) {
            if (exhausted) break;
            continue;
        }
        {
            const unsigned long long innerNow = __ballot(cur >= 0);
            if (innerNow != 0ull) {
                const uint32_t oct = (ga.x < 0.f ? 1u : 0u) | (ga.y < 0.f ? 2u : 0u) | (ga.z < 0.f ? 4u : 0u);
                const uint32_t woct = (uint32_t)__builtin_amdgcn_readlane((int)oct, (int)__ffsll((long long)innerNow) - 1);
                const bool mixed = __ballot(cur >= 0 && oct != woct) != 0ull;
                const float limit = best.t;                  /* best t so far (tmax until something is hit): fixed during the node phase */
                switch (mixed ? 8u : woct) {
                    case 0: inner_nodes2<STACK, 0, STATS>(nodeBuf, lds, cur, sp, res, ga, gb, tmin, limit, kInnerMin, st); break;
                    case 1: inner_nodes2<STACK, 1, STATS>(nodeBuf, lds, cur, sp, res, ga, gb, tmin, limit, kInnerMin, st); break;
                    case 2: inner_nodes2<STACK, 2, STATS>(nodeBuf, lds, cur, sp, res, ga, gb, tmin, limit, kInnerMin, st); break;
                    case 3: inner_nodes2<STACK, 3, STATS>(nodeBuf, lds, cur, sp, res, ga, gb, tmin, limit, kInnerMin, st); break;
                    case 4: inner_nodes2<STACK, 4, STATS>(nodeBuf, lds, cur, sp, res, ga, gb, tmin, limit, kInnerMin, st); break;
                    case 5: inner_nodes2<STACK, 5, STATS>(nodeBuf, lds, cur, sp, res, ga, gb, tmin, limit, kInnerMin, st); break;
                    case 6: inner_nodes2<STACK, 6, STATS>(nodeBuf, lds, cur, sp, res, ga, gb, tmin, limit, kInnerMin, st); break;
                    case 7: inner_nodes2<STACK, 7, STATS>(nodeBuf, lds, cur, sp, res, ga, gb, tmin, limit, kInnerMin, st); break;
                    default: inner_nodes2<STACK, 8, STATS>(nodeBuf, lds, cur, sp, res, ga, gb, tmin, limit, kInnerMin, st); break;
                }
            }
        }
        /* ---- leaves: every triangle, keep the closest (same acceptance and tie rule as trace()) ---- */
        if (cur < 0 && cur != kDone) {
            const int32_t leafCode = cur;
            const uint32_t code = (uint32_t)~cur;
            const uint32_t first = code >> 3, cnt = (code & 7u) + 1u;
            for (uint32_t i = 0; i < cnt; ++i) {
                const int32_t triOff = (int32_t)((first + i) * 48u);
                const u32x4 r0 = __builtin_amdgcn_raw_buffer_load_b128(triBuf, triOff, 0, 0);
                const u32x4 r1 = __builtin_amdgcn_raw_buffer_load_b128(triBuf, triOff + 16, 0, 0);
                const u32x4 r2 = __builtin_amdgcn_raw_buffer_load_b128(triBuf, triOff + 32, 0, 0);
                if (STATS) st.tris++;
                float t, u, v;
                if (rtr_mt_intersect(o, d, rtr_mk(__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z)),
                                     rtr_mk(__uint_as_float(r1.x), __uint_as_float(r1.y), __uint_as_float(r1.z)),
                                     rtr_mk(__uint_as_float(r2.x), __uint_as_float(r2.y), __uint_as_float(r2.z)), tmin, &t, &u, &v) && t < tmax) {
                    const uint32_t cu = r0.w, pr = r1.w;
                    if ((r2.w & 1u) && !alpha_pass<STATS>(sc, cu, pr, u, v, st)) continue;
                    if (t < best.t || (t == best.t && (cu < best.custom || (cu == best.custom && pr < best.prim)))) {
                        best.t = t; best.u = u; best.v = v; best.custom = cu; best.prim = pr; best.leaf = leafCode;
                    }
                }
            }
            cur = *sp; sp -= kBlock;                         /* slot 0 holds kDone: an empty stack ends the ray */
        }
    }
    if (STATS) st.flush(stats);
}

/* Finishes the rays the production kernels abandoned (stack deeper than their 16 LDS entries): one ray per lane, plain
 * trace<true>() with a full-depth 64-entry stack.  Usually zero rays.  The stack lives in GLOBAL memory (the frame's spill
 * area): a 64-KiB LDS stack could not become resident next to another frame's persistent traversal kernel, so with frames
 * in flight this small launch used to wait for that kernel to drain and held up its own frame's resolve behind it. */
template <bool STATS>
__global__ __launch_bounds__(kBlock) void k_shadow_tail(DeviceScene sc, const RayQueue queue, const uint32_t* __restrict__ overflow, uint32_t overflowCap,
                                                        const uint32_t* __restrict__ count, uint8_t* __restrict__ vis, int32_t* __restrict__ spill, Counters* stats) {
    uint32_t n = overflow[0];
    if (n == 0) return;
    /* more abandoned rays than the list holds (its capacity is 1/16 of the queue's, at least a million): every ray of the queue is
     * redone — the same bytes for the ones that had finished; the work counters then include their second walk */
    const bool all = n > overflowCap;
    if (all) n = *count;
    int32_t* stack = spill + blockIdx.x * kBlock + threadIdx.x;          /* depth stride = every lane of the grid */
    LocalStats st;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += kTailBlocks * kBlock) {
        const uint32_t r = all ? i : overflow[1u + i];
        rtr_v3 ro, rd; float rt; uint32_t rs;
        queue_load(queue, r, ro, rd, rt, rs, 0u);
        HitRec h;
        const bool occ = trace<true, STATS, kTailBlocks * kBlock>(sc, stack, ro, rd, 0.001f, rt, h, st);
        if (STATS) { st.rays--; st.shadow--; }              /* the ray itself was counted when the persistent kernel took it from the queue */
        vis[rs] = occ ? 1 : 0;
    }
    if (STATS) st.flush(stats);
}

/* ---- wavefront stage 4: resolve (shade with looked-up visibility, tonemap, store) ------------- */
/* Occupancy of k_resolve: it waits on its fetches (hit record -> object -> indices -> vertices, the visibility bytes) more than it
 * issues, so it wants waves.  All five images: 150 VGPRs by itself = 3 waves per SIMD; capped at 128 (33 dwords spilled) it runs 6 %
 * faster, capped lower the spills cost more than the waves bring (profiles/r04/ab_resolve_waves.log).  Framebuffer only (FULL ==
 * false): 103 VGPRs by itself, RTR_RESOLVE_WAVES_FB waves per SIMD. */
#ifndef RTR_RESOLVE_WAVES
#define RTR_RESOLVE_WAVES 4
#endif
#ifndef RTR_RESOLVE_WAVES_FB
#define RTR_RESOLVE_WAVES_FB 6
#endif
#define RTR_RESOLVE_ATTR __attribute__((amdgpu_waves_per_eu(FULL ? RTR_RESOLVE_WAVES : RTR_RESOLVE_WAVES_FB, 8)))
/* FULL == false: the launch writes only the framebuffer of the north-star path (RTR_IMAGE_SHADOWED, + the HDR accumulator) — what
 * bench.py renders.  Known at compile time, the analytic / unshadowed / normal / position sums and the LTC fetches are dead code and
 * the kernel needs 40 registers fewer (one more wave per SIMD of a kernel that waits on its fetches); the shadowed sum is formed by
 * the same operations in the same order either way (the tests render both forms against the oracle). */
template <bool STATS, bool FULL>
__global__ __launch_bounds__(kBlock) RTR_RESOLVE_ATTR void k_resolve(DeviceScene sc, FrameBatch fb, uint32_t planeStride, const float4* hitTuvp,
                                                    const uint32_t* hitCustom, const uint8_t* vis, uint32_t slotStride, Counters* stats, uint32_t rowWaves) {
    /* which pixel slot this lane resolves.  Upstream a wave is one 8x8 tile (slot q = tile * 64 + row-in-tile * 8 + column-in-tile).
     * rowWaves (set when the tile rows are whole groups of eight tiles): a pair of workgroups takes eight tiles side by side, and a
     * wave is one 64-pixel ROW of them, so every image store of a wave is one contiguous 256-B run of a framebuffer row (and 1 KiB
     * of the float HDR buffer) instead of eight 32-B pieces; its hit-record loads are eight 128-B runs instead of one 1-KiB run.
     * Off by default (RTR_RESOLVE_ROW_WAVES=1 turns it on): the stores are 8 MB of a frame's traffic, while a 64x1 strip shades less
     * coherently than an 8x8 tile — 0.246 -> 0.274 ms, 11.61 -> 11.47 G rays/s (profiles/r02/resolve_row_waves.log). */
    uint32_t q = blockIdx.x * kBlock + threadIdx.x;
    if (rowWaves) {
        const uint32_t row = (threadIdx.x >> 6) + 4u * (blockIdx.x & 1u), col = threadIdx.x & 63u;
        q = (blockIdx.x >> 1) * (2u * kBlock) + (col >> 3) * 64u + row * 8u + (col & 7u);
    }
    const uint32_t frame = batch_frame(q, planeStride, q);           /* q: lane over the whole batch -> pixel slot of its frame */
    if (frame >= fb.n) return;
    const RenderArgs& ra = fb.ra[frame];
    const FrameOut& fo = fb.fo[frame];
    uint32_t px, lrow, py;
    if (!pixel_of(ra, q, px, lrow, py)) return;
    LocalStats st;
    Accum acc = zero_accum();
    const uint32_t want = FULL ? ((fo.img[0] != nullptr ? 1u : 0u) | (fo.img[2] != nullptr ? 2u : 0u)) : 0u;   /* analytic / unshadowed outputs */
    for (uint32_t i = 0; i < ra.spp; ++i) {
        const size_t k = ((size_t)frame * ra.spp + i) * planeStride + q;
        const float4 r = hitTuvp[k];
        HitRec h; h.t = r.x; h.u = r.y; h.v = r.z; h.prim = __float_as_uint(r.w); h.custom = hitCustom[k];
        const rtr_v3 dir = primary_dir(ra, px, py, i);
        LookupPolicy pol{vis, (uint32_t)k, slotStride, vis_mask32(vis, (uint32_t)k, slotStride, ra.maxRaysPerSample), 0u};
        shade_sample<LookupPolicy, STATS>(sc, ra, px, py, h, dir, want, acc, pol, st);
    }
    if (FULL) write_pixel(ra, fo, out_index(ra, px, lrow, py), acc);
    else write_pixel_framebuffer(ra, fo, out_index(ra, px, lrow, py), acc.shadowed);
    if (STATS) st.flush(stats);
}

/* ---- wavefront stage 4, compacted form: the BRDF only for the VISIBLE samples, dealt out densely over the wave's lanes ------------
 * k_resolve walks, lane = pixel, the light loops of raygen.rgen:165-338 and evaluates a sample's BRDF when its shadow ray was not
 * occluded.  Three shadow rays in four ARE occluded, and which ones differs from pixel to pixel (every pixel aims at its own random
 * points of the light): at nearly every step SOME lane of the wave has a visible sample, so the wave issues the ~200-instruction BRDF
 * at nearly every step for a fifth of its lanes (measured lane use 0.53, a quarter of it IEEE divisions).  Here a wave (one 8x8 tile)
 *   1. fetches its 64 surfaces as before and parks what a sample needs of them in LDS (16 floats per pixel);
 *   2. BUILD: walks the light loops without any arithmetic but the back-face test, reads each query's visibility byte and appends
 *      every VISIBLE query as one work item {pixel lane, light, light triangle, sample} to a list in LDS — ballot + mbcnt, the same
 *      wavefront-ballot compaction the ray queue is made with;
 *   3. EVALUATE: the list is dealt out 64 items at a time: a lane takes ANY pixel's sample — the surface from LDS, the light
 *      triangle from its record, the sample point from the PCG seed of that pixel — and leaves BRDF * L / pdf in LDS;
 *   4. CONSUME: each pixel's lane walks the same steps again and adds its samples' contributions in the reference's order
 *      (samples of a light triangle in order, / numShadowRays, then the triangle's sum onto the pixel's: raygen.rgen:268-285).
 * The arithmetic of a sample is the same function of the same operands (area_sample_contrib / directional_contrib,
 * light_sample_pos) and the sums are formed in the same order, so the image is the one k_resolve writes, bit for bit (tests: both
 * forms against the oracle).  A round holds at most CAP items and 32 steps; longer light lists take more rounds.  Only the
 * framebuffer-only launch has this form: with the unshadowed image asked for every sample's BRDF is an output and nothing is sparse. */
#ifndef RTR_EXP_DOUBLE_FETCH_GEN
#define RTR_EXP_DOUBLE_FETCH_GEN 0
#endif
#ifndef RTR_EXP_DOUBLE_FETCH
#define RTR_EXP_DOUBLE_FETCH 0
#endif
#ifndef RTR_RESOLVE_CAP
#define RTR_RESOLVE_CAP 192
#endif
#ifndef RTR_RESOLVE_COMPACT_WAVES
#define RTR_RESOLVE_COMPACT_WAVES 5
#endif
constexpr uint32_t kResolveCap = RTR_RESOLVE_CAP;        /* work items a wave collects before it evaluates them (a round closes once fewer than 64 places are left): a multiple of 32, at least 128 */
static_assert(kResolveCap % 32 == 0 && kResolveCap >= 128 && kResolveCap <= 1024, "RTR_RESOLVE_CAP");
constexpr uint32_t kResolveMaxLights = 255u, kResolveMaxLightTris = 4094u, kResolveMaxSamples = 63u;      /* what a 32-bit work item can name */
constexpr uint32_t kItemDirectional = 0xfffu;

__device__ __forceinline__ void wave_lds_sync() {
    /* LDS traffic of ONE wave: its ds instructions execute in order, so a lane sees what another lane of the wave wrote by an earlier
     * instruction; all that is needed is that the compiler keeps the order */
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(RTR_RESOLVE_COMPACT_WAVES, 8)))
void k_resolve_compact(DeviceScene sc, FrameBatch fb, uint32_t planeStride, const float4* __restrict__ hitTuvp, const uint32_t* __restrict__ hitCustom,
                       const uint8_t* __restrict__ vis, uint32_t slotStride) {
    constexpr uint32_t kWaves = kBlock / 64;
    __shared__ float4 s_surf[kWaves][4][64];             /* per pixel of the tile: {hitPoint, roughness} {hitNormal, cd.x} {viewDir, cd.y} {mSpecular, cd.z} */
    __shared__ uint32_t s_item[kWaves][kResolveCap];     /* lane | sample << 6 | light triangle << 12 | light << 24 */
    __shared__ float s_con[kWaves][3][kResolveCap];      /* an item's BRDF * L / pdf */
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t q = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t frame = batch_frame(q, planeStride, q);
    if (frame >= fb.n) return;                           /* wave-uniform */
    const RenderArgs& ra = fb.ra[frame];
    const FrameOut& fo = fb.fo[frame];
    uint32_t px, lrow, py;
    const bool valid = pixel_of(ra, q, px, lrow, py);
    if (__ballot(valid) == 0ull) return;                 /* a tile of padding */
    /* a wave is one 8x8 tile and bandRows is a multiple of 8: the pixel of lane l is (pxBase + (l & 7), pyBase + (l >> 3)) */
    const uint32_t pxBase = (uint32_t)__builtin_amdgcn_readfirstlane((int)(px - (lane & 7u))), pyBase = (uint32_t)__builtin_amdgcn_readfirstlane((int)(py - (lane >> 3)));
    const uint32_t ns = ra.numShadowRays;
    const float nsf = (float)ns;
    const rtr_v3 directLightDir = directional_light_dir();
    LocalStats st;
    Accum acc = zero_accum();
    for (uint32_t i = 0; i < ra.spp; ++i) {
        const size_t k = ((size_t)frame * ra.spp + i) * planeStride + q;
        bool has = false;
        Surface sf;
        sf.hitPoint = sf.hitNormal = rtr_mk(0, 0, 0);
        if (valid) {
            const float4 r = hitTuvp[k];
            HitRec h; h.t = r.x; h.u = r.y; h.v = r.z; h.prim = __float_as_uint(r.w); h.custom = hitCustom[k];
            has = fetch_surface<true, false>(sc, ra, h, primary_dir(ra, px, py, i), false, acc, sf, st);      /* a miss / a light adds its colour to acc.shadowed, as in k_resolve */
#if RTR_EXP_DOUBLE_FETCH      /* experiment (profiles/r05/ab_resolve_fetch_share.log): the surface fetched a second time — what one fetch costs this kernel is the upper bound of what a surface record written by the queue build could save it */
            if (has) {
                HitRec h2 = h; h2.u = h.u + (float)(ra.spp - 1u) * 1.0e-30f; h2.v = h.v + (float)(ra.spp - 1u) * 1.0e-30f;
                Surface sf2; Accum acc2 = zero_accum();
                const bool has2 = fetch_surface<true, false>(sc, ra, h2, primary_dir(ra, px, py, i), false, acc2, sf2, st);
                if (has2 && sf2.roughness == 12345.0f && sf2.hitPoint.x == 4321.0f && sf2.color.y == 777.0f && sf2.viewDir.z == 55.0f && sf2.hitNormal.y == 99.0f) sf = sf2;
            }
#endif
        }
        wave_lds_sync();                                 /* the last sample's evaluation has read the surfaces */
        if (has) {
            const rtr_v3 cd = surface_diffuse(sf.om, sf.color);
            s_surf[wave][0][lane] = make_float4(sf.hitPoint.x, sf.hitPoint.y, sf.hitPoint.z, sf.roughness);
            s_surf[wave][1][lane] = make_float4(sf.hitNormal.x, sf.hitNormal.y, sf.hitNormal.z, cd.x);
            s_surf[wave][2][lane] = make_float4(sf.viewDir.x, sf.viewDir.y, sf.viewDir.z, cd.y);
            s_surf[wave][3][lane] = make_float4(sf.mSpecular.x, sf.mSpecular.y, sf.mSpecular.z, cd.z);
        }
        const rtr_v3 hitPoint = sf.hitPoint, hitNormal = sf.hitNormal;
        uint32_t qdone = 0;                              /* queries this pixel-sample has issued so far: its next visibility byte is plane qdone (LookupPolicy) */
        rtr_v3 shadowedSample = rtr_mk(0, 0, 0);         /* of the light triangle being summed; lives across rounds */
        uint32_t e0 = 0, total = 0;                      /* steps [0, e0) are done; total: steps of a pixel-sample (set by the first walk) */
        do {
            /* ---- build: steps [e0, e1) ---- */
            uint32_t count = 0, e = 0, e1 = e0;          /* wave-uniform */
            uint32_t vmask = 0, imask = 0;               /* per lane: bit (step - e0) = the query is visible / was issued */
            bool full = false;
            /* a round takes at most 32 steps, so a lane issues at most 32 queries in it: their visibility bytes, fetched together */
            const uint32_t occ = has ? vis_mask32(vis, (uint32_t)k + qdone * slotStride, slotStride, ra.maxRaysPerSample - qdone) : 0xffffffffu;
            uint32_t nq = 0;                             /* queries this lane issued in this round */
            for (uint32_t li = 0; li < ra.info.numAreaLights; ++li) {
                const RtrAreaLightInfo* L = sc.lights + li;
                const uint32_t lnt = L->numTriangles, first = sc.lightTriFirst[li];
                const bool twoSided = L->isTwoSided != 0u;
                for (uint32_t ti = 0; ti < lnt; ++ti) {
                    if (full || e + ns <= e0) { e += ns; continue; }
                    const float4* rec = sc.lightTris + (size_t)(first + ti) * kLightTriRecord;
                    const float4 r0 = rec[0], r3 = rec[3];
                    const bool culled = !twoSided && rtr_dot(rtr_mk(r3.x, r3.y, r3.z), rtr_sub(hitPoint, rtr_mk(r0.x, r0.y, r0.z))) < 0.0f;
                    const bool issued = has && !culled;
                    for (uint32_t s = 0; s < ns; ++s, ++e) {
                        if (e < e0 || full) continue;
                        bool v = false;
                        if (issued) { v = ((occ >> nq) & 1u) == 0u; ++nq; }
                        const unsigned long long m = __ballot(v);
                        if (v) s_item[wave][count + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = lane | (s << 6) | ((first + ti) << 12) | (li << 24);
                        const uint32_t bit = e - e0;
                        vmask |= (v ? 1u : 0u) << bit; imask |= (issued ? 1u : 0u) << bit;
                        count += (uint32_t)__popcll(m);
                        e1 = e + 1u;
                        full = count + 64u > kResolveCap || e1 - e0 == 32u;
                    }
                }
            }
            if (!full && e >= e0) {                      /* the directional light: the last step */
                const bool issued = has && rtr_dot(hitNormal, directLightDir) > 0.0f;
                bool v = false;
                if (issued) { v = ((occ >> nq) & 1u) == 0u; ++nq; }
                const unsigned long long m = __ballot(v);
                if (v) s_item[wave][count + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = lane | (kItemDirectional << 12);
                const uint32_t bit = e - e0;
                vmask |= (v ? 1u : 0u) << bit; imask |= (issued ? 1u : 0u) << bit;
                count += (uint32_t)__popcll(m);
                e1 = e + 1u;
            }
            total = e + 1u;
            qdone += nq;
            wave_lds_sync();
            /* ---- evaluate: any lane, any pixel's sample ---- */
            for (uint32_t c = lane; c < count; c += 64u) {
                const uint32_t item = s_item[wave][c];
                const uint32_t src = item & 63u, s = (item >> 6) & 63u, tri = (item >> 12) & 0xfffu, li = item >> 24;
                const float4 a0 = s_surf[wave][0][src], a1 = s_surf[wave][1][src], a2 = s_surf[wave][2][src], a3 = s_surf[wave][3][src];
                const rtr_v3 hp = rtr_mk(a0.x, a0.y, a0.z), hn = rtr_mk(a1.x, a1.y, a1.z), vd = rtr_mk(a2.x, a2.y, a2.z), ms = rtr_mk(a3.x, a3.y, a3.z), cd = rtr_mk(a1.w, a2.w, a3.w);
                rtr_v3 con;
                if (tri == kItemDirectional) con = directional_contrib(hn, vd, a0.w, ms, cd);
                else {
                    const float4* rec = sc.lightTris + (size_t)tri * kLightTriRecord;
                    const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
                    rtr_v3 P[3];
                    P[0] = rtr_mk(r0.x, r0.y, r0.z); P[1] = rtr_mk(r1.x, r1.y, r1.z); P[2] = rtr_mk(r2.x, r2.y, r2.z);
                    const RtrAreaLightInfo* L = sc.lights + li;
                    const rtr_v3 lightVec = rtr_sub(light_sample_pos(P, s, pxBase + (src & 7u), pyBase + (src >> 3), ra.info.frame), hp);
                    con = area_sample_contrib(hn, vd, a0.w, ms, cd, rtr_ld3(L->color), L->intensity, r1.w, rtr_normalize(lightVec), rtr_length(lightVec));
                }
                s_con[wave][0][c] = con.x; s_con[wave][1][c] = con.y; s_con[wave][2][c] = con.z;
            }
            wave_lds_sync();
            /* ---- consume: each pixel's lane adds its samples in the reference's order ---- */
            uint32_t base = 0;
            e = 0;
            for (uint32_t li = 0; li < ra.info.numAreaLights; ++li) {
                const uint32_t lnt = sc.lights[li].numTriangles;
                for (uint32_t ti = 0; ti < lnt; ++ti) {
                    if (e >= e1 || e + ns <= e0) { e += ns; continue; }
                    for (uint32_t s = 0; s < ns; ++s, ++e) {
                        if (e < e0 || e >= e1) continue;
                        const uint32_t bit = e - e0;
                        const bool v = ((vmask >> bit) & 1u) != 0u;
                        const unsigned long long m = __ballot(v);
                        if (v) {
                            const uint32_t at = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                            shadowedSample = rtr_madd(shadowedSample, rtr_mk(s_con[wave][0][at], s_con[wave][1][at], s_con[wave][2][at]), 1.0f);
                        }
                        base += (uint32_t)__popcll(m);
                        if (s + 1u == ns) {              /* raygen.rgen:268-285: the triangle's sum / numShadowRays onto the pixel's */
                            if ((imask >> bit) & 1u) acc.shadowed = rtr_add(acc.shadowed, rtr_mk(shadowedSample.x / nsf, shadowedSample.y / nsf, shadowedSample.z / nsf));
                            shadowedSample = rtr_mk(0, 0, 0);
                        }
                    }
                }
            }
            if (e >= e0 && e < e1) {
                const uint32_t bit = e - e0;
                const bool v = ((vmask >> bit) & 1u) != 0u;
                const unsigned long long m = __ballot(v);
                if (v) {
                    const uint32_t at = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    acc.shadowed = rtr_madd(acc.shadowed, rtr_mk(s_con[wave][0][at], s_con[wave][1][at], s_con[wave][2][at]), 1.0f);
                }
            }
            e0 = e1;
        } while (e0 < total);
    }
    if (valid) write_pixel_framebuffer(ra, fo, out_index(ra, px, lrow, py), acc.shadowed);
}

/* ---- rank-0 de-interleave after the RCCL gather ------------------------------------------------ */
__global__ __launch_bounds__(kBlock) void k_deinterleave(const uint32_t* __restrict__ gathered, uint32_t* __restrict__ dst,
                                                         uint32_t width, uint32_t height, uint32_t bandRows,
                                                         uint32_t shardCount, uint32_t localRows) {
    const size_t total = (size_t)width * height;
    for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < total; p += (size_t)gridDim.x * kBlock) {
        const uint32_t y = (uint32_t)(p / width), x = (uint32_t)(p % width);
        const uint32_t band = y / bandRows, r = y % bandRows;
        const uint32_t shard = band % shardCount, lb = band / shardCount;
        dst[p] = gathered[((size_t)shard * localRows + (size_t)lb * bandRows + r) * width + x];
    }
}

/* One lane per light: the world-space corners, area, pdf and unit normal of its triangles (read by light_loops in every pixel). */
__global__ __launch_bounds__(64) void k_light_tris(const RtrAreaLightInfo* __restrict__ lights, const RtrVertex* __restrict__ vertices,
                                                   const uint32_t* __restrict__ indices, const uint32_t* __restrict__ first,
                                                   uint32_t numLights, float4* __restrict__ out) {
    const uint32_t l = blockIdx.x * 64u + threadIdx.x;
    if (l >= numLights) return;
    const RtrAreaLightInfo* L = lights + l;
    for (uint32_t ti = 0; ti < L->numTriangles; ++ti) light_tri_record(L, vertices, indices, ti, out + (size_t)(first[l] + ti) * kLightTriRecord);
}

/* Pre-fill of the visibility array: the first planeVec 16-B words of each plane (blockIdx.y), pitchVec words apart — and the launch's
 * control block and overflow count zeroed by the first workgroup (two memset launches less in front of every launch of the pipeline). */
__global__ __launch_bounds__(kBlock) void k_fill_planes(uint4* __restrict__ vis, uint32_t word, uint32_t planeVec, uint32_t pitchVec,
                                                        uint32_t* __restrict__ ctrl, uint32_t ctrlWords, uint32_t* __restrict__ overflowCount) {
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        for (uint32_t i = threadIdx.x; i < ctrlWords; i += kBlock) ctrl[i] = 0u;
        if (threadIdx.x == 0) overflowCount[0] = 0u;
    }
    const uint4 v = make_uint4(word, word, word, word);
    uint4* plane = vis + (size_t)blockIdx.y * pitchVec;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < planeVec; i += gridDim.x * kBlock) plane[i] = v;
}

/* ---- launchers ------------------------------------------------------------------------------- */
namespace {
struct TunableField { const char* name; uint32_t Tunables::* field; uint32_t lo, hi; };
const TunableField kTunables[] = {
    {"primary_packet", &Tunables::primary_packet, 0u, 1u}, {"primary_wide", &Tunables::primary_wide, 0u, 1u}, {"primary_persist", &Tunables::primary_persist, 0u, 2u}, {"primary_persist_min_rays", &Tunables::primary_persist_min_rays, 0u, 0xffffffffu},
    {"primary_batch", &Tunables::primary_batch, 64u, 1u << 16}, {"primary_refill", &Tunables::primary_refill, 1u, 64u},
    {"primary_inner_min", &Tunables::primary_inner_min, 0u, 63u}, {"primary_wgs_per_cu", &Tunables::primary_wgs_per_cu, 1u, 8u},
    {"trace_bvh4", &Tunables::trace_bvh4, 0u, 1u}, {"trace_batch", &Tunables::trace_batch, 0u, 1u << 20}, {"trace_binned", &Tunables::trace_binned, 0u, 2u},
    {"queue_nt", &Tunables::queue_nt, 0u, 3u}, {"trace_wgs_per_cu", &Tunables::trace_wgs_per_cu, 0u, 8u}, {"trace_refill", &Tunables::trace_refill, 1u, 64u},
    {"trace_inner_min", &Tunables::trace_inner_min, 0u, 63u}, {"trace_octant_forms", &Tunables::trace_octant_forms, 0u, 1u}, {"trace_own_leaf", &Tunables::trace_own_leaf, 0u, 1u},
    {"trace_top_nodes", &Tunables::trace_top_nodes, 0u, 0xffffffffu}, {"resolve_row_waves", &Tunables::resolve_row_waves, 0u, 1u}, {"resolve_compact", &Tunables::resolve_compact, 0u, 1u}, {"split_priorities", &Tunables::split_priorities, 0u, 1u},
};
}  // namespace

bool tunable_set(Tunables& t, const char* name, uint32_t value) {
    for (const TunableField& f : kTunables)
        if (name && !strcmp(name, f.name)) { if (value < f.lo || value > f.hi) return false; t.*(f.field) = value; return true; }
    return false;
}
bool tunable_get(const Tunables& t, const char* name, uint32_t* value) {
    for (const TunableField& f : kTunables) if (name && !strcmp(name, f.name)) { if (value) *value = t.*(f.field); return true; }
    return false;
}
/* RTR_TRACE_BINNED=1 and friends: the field's name in capitals behind RTR_; out-of-range values are clamped, as they always were */
Tunables tunables_from_env() {
    Tunables t;
    for (const TunableField& f : kTunables) {
        char env[64] = "RTR_"; size_t n = 4;
        for (const char* c = f.name; *c && n + 1 < sizeof env; ++c) env[n++] = (char)(*c >= 'a' && *c <= 'z' ? *c - 32 : *c);
        env[n] = 0;
        const char* v = getenv(env);
        if (!v || !*v) continue;
        char* end = nullptr;
        const unsigned long x = strtoul(v, &end, 10);
        if (end == v) continue;                               /* not a number: the default stays (RTR_TRACE_BINNED=foo is not "never binned") */
        t.*(f.field) = x < f.lo ? f.lo : (x > f.hi ? f.hi : (uint32_t)x);
    }
    return t;
}

static uint32_t padded_pixels(const RenderArgs& ra) {
    const uint32_t band8 = (ra.localRows + 7u) / 8u;
    return band8 * ra.tilesPerRow * 64u;
}

template <int STACK>
static hipError_t mega_t(const DeviceScene& sc, const RenderArgs& ra, const FrameOut& fo, Counters* stats, hipStream_t s) {
    const uint32_t blocks = (padded_pixels(ra) + kBlock - 1) / kBlock;
    if (stats) hipLaunchKernelGGL((k_megakernel<STACK, true>), dim3(blocks), dim3(kBlock), 0, s, sc, ra, fo, stats);
    else hipLaunchKernelGGL((k_megakernel<STACK, false>), dim3(blocks), dim3(kBlock), 0, s, sc, ra, fo, stats);
    return hipGetLastError();
}

hipError_t launch_megakernel(const DeviceScene& sc, const RenderArgs& ra, const FrameOut& fo, int stackEntries,
                             Counters* stats, hipStream_t stream) {
    switch (stackEntries) {
        case 16: return mega_t<16>(sc, ra, fo, stats, stream);
        case 32: return mega_t<32>(sc, ra, fo, stats, stream);
        case 64: return mega_t<64>(sc, ra, fo, stats, stream);
        default: return hipErrorInvalidValue;
    }
}

template <int STACK>
static hipError_t wave_t(const DeviceScene& sc, const FrameBatch& fb, const Workspace& ws, const Tunables& tun,
                         Counters* stats, hipStream_t s, hipEvent_t* ev, uint32_t numCus) {
    const RenderArgs& ra = fb.ra[0];                   /* extent, spp, sharding: the same for every frame of the batch */
    const uint32_t nb = fb.n;
    const uint32_t blocks = (padded_pixels(ra) + kBlock - 1) / kBlock;      /* per frame and sample plane */
    /* the control block (queue length, batch cursors, list lengths ...) and the count of abandoned rays are zeroed by k_fill_planes */
    /* the visibility array starts as "every ray had the commoner outcome" (the frame object's last launch says which: three shadow
     * rays in four of the bench frame are occluded); the any-hit kernel stores only the other outcome — whole-line fill traffic
     * instead of most of its scattered byte stores (WRITE_SIZE of the launch 136 -> 34 MB + 27 MB of fill) */
    {   /* only the bytes the launch uses: its pixel-sample slots at the start of each plane, not the power-of-two pitch between the planes
         * (a 20-frame 1080p launch: 539 MB instead of 872) */
        const uint32_t planeVec = (uint32_t)(ws.visPlaneBytes / 16u), pitchVec = ws.rayQueue.slotStride / 16u;
        uint32_t fblocks = std::min<uint32_t>((planeVec + kBlock - 1) / kBlock, std::max<uint32_t>(4096u / std::max<uint32_t>(ws.visPlanes, 1u), 64u));
        if (fblocks == 0) fblocks = 1;
        hipLaunchKernelGGL(k_fill_planes, dim3(fblocks, ws.visPlanes ? ws.visPlanes : 1u), dim3(kBlock), 0, s, reinterpret_cast<uint4*>(ws.vis), ws.visFill * 0x01010101u, planeVec, pitchVec,
                           ws.queueCount, kQueueCtrlWords, ws.overflow);
    }
    if (ev) hipEventRecord(ev[0], s);
    /* the storage of ws.overflow is used twice per frame: first as k_primary's redo list (count in queueCount[2], consumed by
     * k_primary_tail), then as the any-hit kernel's overflow list (count in overflow[0]) */
    /* camera rays: the persistent kernel (k_primary_persist), or one ray per lane (k_primary; RTR_PRIMARY_PERSIST=0, for comparison).
     * Both walk the BVH2 and leave the rays that outgrow their 16-entry LDS stack to k_primary_tail; the counting form takes the
     * same path as the timed one. */
    const uint32_t kPersist = tun.primary_persist;          /* 0 never (default), 1 whenever it can, 2 by the size of the launch */
    const uint32_t kPersistMinRays = tun.primary_persist_min_rays;
    const uint32_t kPBatch = tun.primary_batch, kPRefill = tun.primary_refill, kPInnerMin = tun.primary_inner_min, kPWgsPerCu = tun.primary_wgs_per_cu;
    const uint32_t planeStride = blocks * kBlock;
    /* Round 2 measured the persistent kernel faster alone from 8 M camera rays (0.58 against 0.89 ms at 1080p x 4 spp); since k_primary
     * takes one lane per (sample, pixel) it is the other way round at every size (0.77 against 0.98 ms; whole frames 3-9 % slower with
     * RTR_PRIMARY_PERSIST=2, profiles/r03/ab_primary_persist_auto.log), so nothing selects it by default. */
    const unsigned long long camRays = (unsigned long long)planeStride * ra.spp;
    bool packet = false;
    if ((kPersist == 1u || (kPersist == 2u && camRays >= kPersistMinRays)) && nb == 1u && camRays < 0xffffffffull) {
        uint32_t pblocks = numCus * kPWgsPerCu;
        const uint32_t pneeded = (uint32_t)(((unsigned long long)planeStride * ra.spp + kBlock - 1) / kBlock);
        if (pblocks > pneeded) pblocks = pneeded;
        if (pblocks == 0) pblocks = 1;
        if (stats) hipLaunchKernelGGL((k_primary_persist<16, true>), dim3(pblocks), dim3(kBlock), 0, s, sc, fb.ra[0], ws.hitTuvp, ws.hitCustom, ws.rayQueue.origin, stats, ws.queueCount + kPrimaryRedoWord, ws.overflow + 1, ws.queueCount + kPrimaryCursors, planeStride, kPBatch, kPRefill, kPInnerMin);
        else hipLaunchKernelGGL((k_primary_persist<16, false>), dim3(pblocks), dim3(kBlock), 0, s, sc, fb.ra[0], ws.hitTuvp, ws.hitCustom, ws.rayQueue.origin, stats, ws.queueCount + kPrimaryRedoWord, ws.overflow + 1, ws.queueCount + kPrimaryCursors, planeStride, kPBatch, kPRefill, kPInnerMin);
    } else if (tun.primary_wide && sc.nodes4) {
        /* one camera ray per lane over the 4-wide view (not the default); rays that outgrow its 16 entries go to k_primary_tail like k_primary's */
        if (stats) hipLaunchKernelGGL((k_primary4<true>), dim3(blocks * ra.spp * nb), dim3(kBlock), 0, s, sc, fb, ws.hitTuvp, ws.hitCustom, ws.rayQueue.origin, stats, ws.queueCount + kPrimaryRedoWord, ws.overflow + 1, blocks);
        else hipLaunchKernelGGL((k_primary4<false>), dim3(blocks * ra.spp * nb), dim3(kBlock), 0, s, sc, fb, ws.hitTuvp, ws.hitCustom, ws.rayQueue.origin, stats, ws.queueCount + kPrimaryRedoWord, ws.overflow + 1, blocks);
    } else if (tun.primary_packet) {
        /* a tile's camera rays walk the tree as one packet (no ray is ever left to the tail kernel: it is not launched); not the default */
        packet = true;
        if (stats) hipLaunchKernelGGL((k_primary_packet<true>), dim3(blocks * ra.spp * nb), dim3(kBlock), 0, s, sc, fb, ws.hitTuvp, ws.hitCustom, ws.rayQueue.origin, stats, blocks);
        else hipLaunchKernelGGL((k_primary_packet<false>), dim3(blocks * ra.spp * nb), dim3(kBlock), 0, s, sc, fb, ws.hitTuvp, ws.hitCustom, ws.rayQueue.origin, stats, blocks);
    } else if (stats) hipLaunchKernelGGL((k_primary<STACK, true>), dim3(blocks * (kBlock / kPrimBlock) * ra.spp * nb), dim3(kPrimBlock), 0, s, sc, fb, ws.hitTuvp, ws.hitCustom, ws.rayQueue.origin, stats, ws.queueCount + kPrimaryRedoWord, ws.overflow + 1, blocks * (kBlock / kPrimBlock));
    else hipLaunchKernelGGL((k_primary<STACK, false>), dim3(blocks * (kBlock / kPrimBlock) * ra.spp * nb), dim3(kPrimBlock), 0, s, sc, fb, ws.hitTuvp, ws.hitCustom, ws.rayQueue.origin, stats, ws.queueCount + kPrimaryRedoWord, ws.overflow + 1, blocks * (kBlock / kPrimBlock));
    if (!packet) {
        if (stats) hipLaunchKernelGGL(k_primary_tail<true>, dim3(kTailBlocks), dim3(kBlock), 0, s, sc, fb, ws.hitTuvp, ws.hitCustom, ws.rayQueue.origin, ws.queueCount + kPrimaryRedoWord, ws.overflow + 1, ws.spill, planeStride, stats);
        else hipLaunchKernelGGL(k_primary_tail<false>, dim3(kTailBlocks), dim3(kBlock), 0, s, sc, fb, ws.hitTuvp, ws.hitCustom, ws.rayQueue.origin, ws.queueCount + kPrimaryRedoWord, ws.overflow + 1, ws.spill, planeStride, stats);
    }
    if (ev) hipEventRecord(ev[1], s);
    /* run-time tunables of the traversal kernels (profiles/sweep_*.sh); the defaults are the swept optima */
    const uint32_t kBatchEnv = tun.trace_batch;        /* 0 (default): by the length of the queue */
    const uint32_t kBatch = kBatchEnv >= 64u ? kBatchEnv : ((size_t)nb * blocks * kBlock * ra.spp * ra.maxRaysPerSample >= kBatchLongQueue ? kBatchLong : kBatchDefault);
    const uint32_t kWide = tun.trace_bvh4;   /* trace_bvh4 = 0: the 2-wide any-hit kernel on the plain queue (same results, for comparison; it has no counting form: rtr_render refuses collectStats with it) */
    const uint32_t genBlocks = (nb * blocks * kBlock + kGenBlock - 1) / kGenBlock, genOctBlocks = (nb * blocks * kBlock + kGenOctBlock - 1) / kGenOctBlock;
    /* Queue binned by direction octant + per-(octant, XCD) batch lists (k_shadow_gen_oct -> k_shadow_trace4), or the plain queue
     * (k_shadow_gen -> k_shadow_trace4 over eight regions of it).  Binning pays on long queues (+2 % frame rate at 12-25 M rays:
     * the octant forms of the slab test then run 97 % of the time instead of 36 %) and costs on short ones (its 64 short lists
     * drain unevenly: -2 % at 3-6 M rays), hence the thresholds (kBinnedMinRays, kBinnedMinNodes); RTR_TRACE_BINNED=0/1 forces it off/on (the tests run both).
     * The 16-bit per-lane counters of the binned count hold any realistic ray count per pixel.  The counting form (stats) takes
     * the same path as the timed one: same queue, same kernel template. */
    const uint32_t binMode = tun.trace_binned;
    const size_t maxRaysQ = (size_t)nb * blocks * kBlock * ra.spp * ra.maxRaysPerSample;
    const bool wide = kWide && sc.nodes4;
    const bool binned = wide && ws.batchLists && (size_t)ra.spp * ra.maxRaysPerSample <= 65535u &&
                        (size_t)ws.capRays / kBatch / kQueueRegions + genOctBlocks <= ws.listStride &&
                        (binMode == 1u || (binMode == 2u && maxRaysQ >= kBinnedMinRays && sc.numNodes4 >= kBinnedMinNodes));
    const uint32_t kNtQueue = tun.queue_nt;          /* bit 0: any-hit kernel reads the queue past the caches (default); bit 1: the queue-build kernel writes it so (slower, see queue_load) */
    RayQueue rq = ws.rayQueue;
    rq.ownLeaf = (wide && tun.trace_own_leaf) ? 1u : 0u;          /* only the 4-wide any-hit kernel starts a walk at a leaf (the 2-wide comparison kernel takes the mark off) */
    if (binned) hipLaunchKernelGGL(k_shadow_gen_oct, dim3(genOctBlocks), dim3(kGenOctBlock), 0, s, sc, fb, ws.hitTuvp, ws.hitCustom, rq, ws.queueCount, blocks * kBlock, ws.batchLists, ws.listStride, kBatch, kNtQueue >> 1);
    else hipLaunchKernelGGL(k_shadow_gen, dim3(genBlocks), dim3(kGenBlock), 0, s, sc, fb, ws.hitTuvp, ws.hitCustom, rq, ws.queueCount, blocks * kBlock);
    if (ev) hipEventRecord(ev[2], s);
    /* persistent waves: as many workgroups as stay resident (17 KiB of LDS stack + 2.5 KiB of tree top per workgroup -> 8 per CU,
     * the 32-wave hardware maximum), each pulling batches until the queue is empty */
    const size_t maxRays = (size_t)nb * blocks * kBlock * ra.spp * ra.maxRaysPerSample;
    /* 8 workgroups per CU fill every wave slot, which is what a long queue wants; the short queue of a 1/4 or 1/8 shard is
     * drained in a fraction of a millisecond, and then the other frames' small kernels (which can only start where a persistent
     * wave has retired) matter more: with 6 per CU one rank of 8 renders a frame in 0.394 instead of 0.418 ms and one rank of 4 in
     * 0.693 instead of 0.708 (4 frames in flight, profiles/r01/sweep_wgs_per_cu.log); at N = 1, 2 it makes no difference.
     * numCus comes from hipDeviceProp_t::multiProcessorCount (256 on an MI355X in SPX mode). */
    const uint32_t kWgsPerCu = tun.trace_wgs_per_cu;
    uint32_t tblocks = numCus * (kWgsPerCu ? kWgsPerCu : (maxRays >= kBinnedMinRays ? 8u : 6u));
    const uint32_t needed = (uint32_t)((maxRays + kBlock - 1) / kBlock);
    if (tblocks > needed) tblocks = needed;
    if (tblocks == 0) tblocks = 1;
    const uint32_t kRefill = tun.trace_refill, kInnerMin = tun.trace_inner_min;
    (void)sizeof(STACK);   /* the BVH-depth bound only sizes the spill area; the LDS part is always 16 entries */
    const uint32_t kOct = tun.trace_octant_forms | ((kNtQueue & 1u) << 1);        /* bit 0: octant forms of the node loop; bit 1: the queue is read past the caches */
    const uint32_t kTop = tun.trace_top_nodes < kTopNodes ? tun.trace_top_nodes : kTopNodes;
    const uint32_t top = kTop < sc.numNodes4 ? kTop : sc.numNodes4;
    /* the 4-wide kernel's workgroups are kTraceBlock lanes: the same number of waves in fewer workgroups */
    uint32_t tblocks4 = tblocks * (uint32_t)kBlock / (uint32_t)kTraceBlock;
    if (tblocks4 == 0) tblocks4 = 1;
    if (wide) {
        if (stats) {
            if (binned) hipLaunchKernelGGL((k_shadow_trace4<RTR_SHADOW_STACK, true, true>), dim3(tblocks4), dim3(kTraceBlock), 0, s, sc, ws.rayQueue, ws.queueCount, ws.queueCount + kBatchCursorWord, ws.vis, ws.visFill, kBatch, kRefill, kInnerMin, ws.overflow, ws.overflowCap, kOct, top, ws.batchLists, ws.listStride, stats, ws.clk);
            else hipLaunchKernelGGL((k_shadow_trace4<RTR_SHADOW_STACK, false, true>), dim3(tblocks4), dim3(kTraceBlock), 0, s, sc, ws.rayQueue, ws.queueCount, ws.queueCount + kBatchCursorWord, ws.vis, ws.visFill, kBatch, kRefill, kInnerMin, ws.overflow, ws.overflowCap, kOct, top, ws.batchLists, ws.listStride, stats, ws.clk);
        } else {
            if (binned) hipLaunchKernelGGL((k_shadow_trace4<RTR_SHADOW_STACK, true, false>), dim3(tblocks4), dim3(kTraceBlock), 0, s, sc, ws.rayQueue, ws.queueCount, ws.queueCount + kBatchCursorWord, ws.vis, ws.visFill, kBatch, kRefill, kInnerMin, ws.overflow, ws.overflowCap, kOct, top, ws.batchLists, ws.listStride, stats, ws.clk);
            else hipLaunchKernelGGL((k_shadow_trace4<RTR_SHADOW_STACK, false, false>), dim3(tblocks4), dim3(kTraceBlock), 0, s, sc, ws.rayQueue, ws.queueCount, ws.queueCount + kBatchCursorWord, ws.vis, ws.visFill, kBatch, kRefill, kInnerMin, ws.overflow, ws.overflowCap, kOct, top, ws.batchLists, ws.listStride, stats, ws.clk);
        }
    } else hipLaunchKernelGGL((k_shadow_trace<16>), dim3(tblocks), dim3(kBlock), 0, s, sc, ws.rayQueue, ws.queueCount, ws.queueCount + kBatchCursorWord, ws.vis, ws.visFill, kBatch, kRefill, kInnerMin, ws.overflow, ws.overflowCap);
    if (ev) hipEventRecord(ev[5], s);
    if (stats) hipLaunchKernelGGL((k_shadow_tail<true>), dim3(kTailBlocks), dim3(kBlock), 0, s, sc, ws.rayQueue, ws.overflow, ws.overflowCap, ws.queueCount, ws.vis, ws.spill, stats);
    else hipLaunchKernelGGL((k_shadow_tail<false>), dim3(kTailBlocks), dim3(kBlock), 0, s, sc, ws.rayQueue, ws.overflow, ws.overflowCap, ws.queueCount, ws.vis, ws.spill, stats);
    if (ev) hipEventRecord(ev[3], s);
    const uint32_t kRowWaves = tun.resolve_row_waves;
    const uint32_t rowWaves = (kRowWaves && ra.tilesPerRow % 8u == 0u && blocks % 2u == 0u) ? 1u : 0u;
    /* every frame of a launch has the same image set (params.images): framebuffer-only launches take the trimmed form */
    const bool full = (ra.images & ~((1u << 1) | (1u << 16))) != 0u;          /* anything beside RTR_IMAGE_SHADOWED (1) and RTR_IMAGE_HDR (16) */
    /* framebuffer only, timed form: the BRDF of the visible samples compacted over the wave (k_resolve_compact); the counting form stays
     * the per-pixel walk (its counters are the light loops') */
    const bool compact = !stats && !full && tun.resolve_compact && !rowWaves && ra.numShadowRays >= 1u && ra.numShadowRays <= kResolveMaxSamples &&
                         sc.numLights <= kResolveMaxLights && sc.numLightTris <= kResolveMaxLightTris;
    if (compact) hipLaunchKernelGGL(k_resolve_compact, dim3(blocks * nb), dim3(kBlock), 0, s, sc, fb, blocks * kBlock, ws.hitTuvp, ws.hitCustom, ws.vis, ws.rayQueue.slotStride);
    else if (stats) {
        if (full) hipLaunchKernelGGL((k_resolve<true, true>), dim3(blocks * nb), dim3(kBlock), 0, s, sc, fb, blocks * kBlock, ws.hitTuvp, ws.hitCustom, ws.vis, ws.rayQueue.slotStride, stats, rowWaves);
        else hipLaunchKernelGGL((k_resolve<true, false>), dim3(blocks * nb), dim3(kBlock), 0, s, sc, fb, blocks * kBlock, ws.hitTuvp, ws.hitCustom, ws.vis, ws.rayQueue.slotStride, stats, rowWaves);
    } else {
        if (full) hipLaunchKernelGGL((k_resolve<false, true>), dim3(blocks * nb), dim3(kBlock), 0, s, sc, fb, blocks * kBlock, ws.hitTuvp, ws.hitCustom, ws.vis, ws.rayQueue.slotStride, stats, rowWaves);
        else hipLaunchKernelGGL((k_resolve<false, false>), dim3(blocks * nb), dim3(kBlock), 0, s, sc, fb, blocks * kBlock, ws.hitTuvp, ws.hitCustom, ws.vis, ws.rayQueue.slotStride, stats, rowWaves);
    }
    if (ev) hipEventRecord(ev[4], s);
    return hipGetLastError();
}

hipError_t launch_wavefront(const DeviceScene& sc, const FrameBatch& fb, const Workspace& ws, const Tunables& tun,
                            int stackEntries, Counters* stats, hipStream_t stream, hipEvent_t* ev, uint32_t numCus) {
    if (numCus == 0) numCus = 256;
    if (fb.n < 1 || fb.n > kMaxBatch) return hipErrorInvalidValue;
    switch (stackEntries) {
        case 16: return wave_t<16>(sc, fb, ws, tun, stats, stream, ev, numCus);
        case 32: return wave_t<32>(sc, fb, ws, tun, stats, stream, ev, numCus);
        case 64: return wave_t<64>(sc, fb, ws, tun, stats, stream, ev, numCus);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_light_tris(const RtrAreaLightInfo* lights, const RtrVertex* vertices, const uint32_t* indices, const uint32_t* first,
                             uint32_t numLights, float4* out, hipStream_t stream) {
    if (numLights == 0) return hipSuccess;
    hipLaunchKernelGGL(k_light_tris, dim3((numLights + 63u) / 64u), dim3(64), 0, stream, lights, vertices, indices, first, numLights, out);
    return hipGetLastError();
}

hipError_t launch_deinterleave(const uint32_t* gathered, uint32_t* dst, uint32_t width, uint32_t height,
                               uint32_t bandRows, uint32_t shardCount, uint32_t localRows, hipStream_t stream) {
    const size_t total = (size_t)width * height;
    uint32_t blocks = (uint32_t)((total + kBlock - 1) / kBlock);
    if (blocks > 2048u) blocks = 2048u;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(k_deinterleave, dim3(blocks), dim3(kBlock), 0, stream, gathered, dst, width, height, bandRows, shardCount, localRows);
    return hipGetLastError();
}

}  // namespace rtrdev
