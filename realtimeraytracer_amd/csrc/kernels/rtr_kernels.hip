/* rtr_kernels.hip — hand-written gfx950 kernels of the ray-tracing hot path.
 *
 * Replaces reference src/shaders/raygen.rgen (+ closesthit/miss hit shaders and the driver's
 * traceRayEXT) dispatched by vkCmdTraceRaysKHR (src/vulkan/ray_tracing_pipeline.cppm:212-214).
 *
 * Two pipelines over the same device functions (rtr_device.h):
 *   megakernel : one lane per pixel does everything (simple, divergent) — the first-slice path
 *                and the fallback when the wavefront scratch would be too large.
 *   wavefront  : k_primary -> k_shadow_gen -> k_shadow_trace -> k_resolve.  The shadow rays of all
 *                pixels are compacted into one dense queue with a wave-level ballot
 *                (mbcnt prefix + one atomic per wave), so the any-hit traversal kernel runs with full
 *                waves of coherent rays instead of idling lanes inside the shading loops.
 *
 * Launch shape: 256-thread workgroups (4 waves), one 8x8 pixel tile per wave in the canonical
 * tile order of rtr_device.h; grids are >> 256 workgroups at 1080p (8100 for the per-pixel
 * kernels) so all 8 XCDs fill; consecutive workgroups (round-robin over XCDs) take consecutive
 * tiles, so each XCD's L2 sees an interleaved slice of every band and the top of the BVH is
 * resident in all eight L2s.
 * No MFMA: the path is pointer-chasing + divergence bound (BASELINE.json north_star).
 */
#include "rtr_kernels.h"

namespace rtrdev {

constexpr int kBlock = 256;

template <bool STATS, int STACK>
struct InlinePolicy {
    static constexpr bool kShade = true;
    const DeviceScene& sc; int32_t* stack; LocalStats& st;
    __device__ __forceinline__ bool occluded(rtr_v3 o, rtr_v3 d, float tmax) {
        HitRec h;
        return trace<true, STATS, kBlock>(sc, stack, o, d, 0.001f, tmax, h, st);
    }
};

/* Wave-level active-ray compaction: lanes that reach this call together (the current EXEC mask)
 * take consecutive queue entries; one atomic per wave. */
struct EmitPolicy {
    static constexpr bool kShade = false;
    float4* queue; uint32_t* count; uint32_t slot;
    __device__ __forceinline__ bool occluded(rtr_v3 o, rtr_v3 d, float tmax) {
        const unsigned long long m = __ballot(1);
        const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        uint32_t base = 0;
        if (prefix == 0) base = atomicAdd(count, (uint32_t)__popcll(m));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        const size_t idx = (size_t)(base + prefix) * 2;
        queue[idx] = make_float4(o.x, o.y, o.z, tmax);
        queue[idx + 1] = make_float4(d.x, d.y, d.z, __uint_as_float(slot));
        ++slot;
        return false;
    }
};

struct LookupPolicy {
    static constexpr bool kShade = true;
    const uint8_t* vis; uint32_t slot;
    __device__ __forceinline__ bool occluded(rtr_v3, rtr_v3, float) { return vis[slot++] != 0; }
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
    const bool wantAnalytic = fo.img[0] != nullptr;
    const rtr_v3 camPos = rtr_ld3(ra.cam.position);
    InlinePolicy<STATS, STACK> pol{sc, stack, st};
    for (uint32_t i = 0; i < ra.spp; ++i) {
        const rtr_v3 dir = primary_dir(ra, px, py, i);
        HitRec h;
        trace<false, STATS, kBlock>(sc, stack, camPos, dir, 0.001f, 10000.0f, h, st);
        shade_sample<InlinePolicy<STATS, STACK>, STATS>(sc, ra, px, py, h, dir, wantAnalytic, acc, pol, st);
    }
    write_pixel(ra, fo, (size_t)lrow * ra.width + px, acc);
    if (STATS) st.flush(stats);
}

/* ---- wavefront stage 1: primary visibility ---------------------------------------------------- */
template <int STACK, bool STATS>
__global__ __launch_bounds__(kBlock) void k_primary(DeviceScene sc, RenderArgs ra, float4* hitTuvp, uint32_t* hitCustom,
                                                    Counters* stats) {
    __shared__ int32_t s_stack[STACK * kBlock];
    int32_t* stack = s_stack + threadIdx.x;
    const uint32_t q = blockIdx.x * kBlock + threadIdx.x;
    uint32_t px, lrow, py;
    if (!pixel_of(ra, q, px, lrow, py)) return;
    LocalStats st;
    const rtr_v3 camPos = rtr_ld3(ra.cam.position);
    for (uint32_t i = 0; i < ra.spp; ++i) {
        const rtr_v3 dir = primary_dir(ra, px, py, i);
        HitRec h;
        trace<false, STATS, kBlock>(sc, stack, camPos, dir, 0.001f, 10000.0f, h, st);
        /* sample-major planes keep each store of a wave contiguous */
        const size_t k = (size_t)i * gridDim.x * kBlock + q;
        hitTuvp[k] = make_float4(h.t, h.u, h.v, __uint_as_float(h.prim));
        hitCustom[k] = h.custom;
    }
    if (STATS) st.flush(stats);
}

/* ---- wavefront stage 2: shadow-ray generation into the compacted queue ------------------------- */
__global__ __launch_bounds__(kBlock) void k_shadow_gen(DeviceScene sc, RenderArgs ra, const float4* hitTuvp,
                                                       const uint32_t* hitCustom, float4* queue, uint32_t* count) {
    const uint32_t q = blockIdx.x * kBlock + threadIdx.x;
    uint32_t px, lrow, py;
    if (!pixel_of(ra, q, px, lrow, py)) return;
    LocalStats st;
    Accum acc = zero_accum();
    for (uint32_t i = 0; i < ra.spp; ++i) {
        const size_t k = (size_t)i * gridDim.x * kBlock + q;
        const float4 r = hitTuvp[k];
        HitRec h; h.t = r.x; h.u = r.y; h.v = r.z; h.prim = __float_as_uint(r.w); h.custom = hitCustom[k];
        const rtr_v3 dir = primary_dir(ra, px, py, i);
        EmitPolicy pol{queue, count, (uint32_t)(k * ra.maxRaysPerSample)};
        shade_sample<EmitPolicy, false>(sc, ra, px, py, h, dir, false, acc, pol, st);
    }
}

/* ---- wavefront stage 3: any-hit traversal of the queue (the dominant kernel) ------------------- */
template <int STACK, bool STATS>
__global__ __launch_bounds__(kBlock) void k_shadow_trace(DeviceScene sc, const float4* queue, const uint32_t* count,
                                                         uint8_t* vis, Counters* stats) {
    __shared__ int32_t s_stack[STACK * kBlock];
    int32_t* stack = s_stack + threadIdx.x;
    const uint32_t n = *count;
    LocalStats st;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float4 a = queue[(size_t)i * 2], b = queue[(size_t)i * 2 + 1];
        HitRec h;
        const bool occ = trace<true, STATS, kBlock>(sc, stack, rtr_mk(a.x, a.y, a.z), rtr_mk(b.x, b.y, b.z), 0.001f, a.w, h, st);
        vis[__float_as_uint(b.w)] = occ ? 1 : 0;
    }
    if (STATS) st.flush(stats);
}

/* ---- wavefront stage 4: resolve (shade with looked-up visibility, tonemap, store) ------------- */
template <bool STATS>
__global__ __launch_bounds__(kBlock) void k_resolve(DeviceScene sc, RenderArgs ra, FrameOut fo, const float4* hitTuvp,
                                                    const uint32_t* hitCustom, const uint8_t* vis, Counters* stats) {
    const uint32_t q = blockIdx.x * kBlock + threadIdx.x;
    uint32_t px, lrow, py;
    if (!pixel_of(ra, q, px, lrow, py)) return;
    LocalStats st;
    Accum acc = zero_accum();
    const bool wantAnalytic = fo.img[0] != nullptr;
    for (uint32_t i = 0; i < ra.spp; ++i) {
        const size_t k = (size_t)i * gridDim.x * kBlock + q;
        const float4 r = hitTuvp[k];
        HitRec h; h.t = r.x; h.u = r.y; h.v = r.z; h.prim = __float_as_uint(r.w); h.custom = hitCustom[k];
        const rtr_v3 dir = primary_dir(ra, px, py, i);
        LookupPolicy pol{vis, (uint32_t)(k * ra.maxRaysPerSample)};
        shade_sample<LookupPolicy, STATS>(sc, ra, px, py, h, dir, wantAnalytic, acc, pol, st);
    }
    write_pixel(ra, fo, (size_t)lrow * ra.width + px, acc);
    if (STATS) st.flush(stats);
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

/* ---- launchers ------------------------------------------------------------------------------- */
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
static hipError_t wave_t(const DeviceScene& sc, const RenderArgs& ra, const FrameOut& fo, const Workspace& ws,
                         Counters* stats, hipStream_t s, hipEvent_t* ev) {
    const uint32_t blocks = (padded_pixels(ra) + kBlock - 1) / kBlock;
    hipError_t e;
    if ((e = hipMemsetAsync(ws.queueCount, 0, sizeof(uint32_t), s)) != hipSuccess) return e;
    if (ev) hipEventRecord(ev[0], s);
    if (stats) hipLaunchKernelGGL((k_primary<STACK, true>), dim3(blocks), dim3(kBlock), 0, s, sc, ra, ws.hitTuvp, ws.hitCustom, stats);
    else hipLaunchKernelGGL((k_primary<STACK, false>), dim3(blocks), dim3(kBlock), 0, s, sc, ra, ws.hitTuvp, ws.hitCustom, stats);
    if (ev) hipEventRecord(ev[1], s);
    hipLaunchKernelGGL(k_shadow_gen, dim3(blocks), dim3(kBlock), 0, s, sc, ra, ws.hitTuvp, ws.hitCustom, ws.rayQueue, ws.queueCount);
    if (ev) hipEventRecord(ev[2], s);
    /* persistent grid-stride over the queue: enough workgroups to fill 256 CUs several times over */
    const size_t maxRays = (size_t)blocks * kBlock * ra.spp * ra.maxRaysPerSample;
    uint32_t tblocks = (uint32_t)((maxRays + kBlock - 1) / kBlock);
    const uint32_t cap = 256u * 16u;
    if (tblocks > cap) tblocks = cap;
    if (tblocks == 0) tblocks = 1;
    if (stats) hipLaunchKernelGGL((k_shadow_trace<STACK, true>), dim3(tblocks), dim3(kBlock), 0, s, sc, ws.rayQueue, ws.queueCount, ws.vis, stats);
    else hipLaunchKernelGGL((k_shadow_trace<STACK, false>), dim3(tblocks), dim3(kBlock), 0, s, sc, ws.rayQueue, ws.queueCount, ws.vis, stats);
    if (ev) hipEventRecord(ev[3], s);
    if (stats) hipLaunchKernelGGL((k_resolve<true>), dim3(blocks), dim3(kBlock), 0, s, sc, ra, fo, ws.hitTuvp, ws.hitCustom, ws.vis, stats);
    else hipLaunchKernelGGL((k_resolve<false>), dim3(blocks), dim3(kBlock), 0, s, sc, ra, fo, ws.hitTuvp, ws.hitCustom, ws.vis, stats);
    if (ev) hipEventRecord(ev[4], s);
    return hipGetLastError();
}

hipError_t launch_wavefront(const DeviceScene& sc, const RenderArgs& ra, const FrameOut& fo, const Workspace& ws,
                            int stackEntries, Counters* stats, hipStream_t stream, hipEvent_t* ev) {
    switch (stackEntries) {
        case 16: return wave_t<16>(sc, ra, fo, ws, stats, stream, ev);
        case 32: return wave_t<32>(sc, ra, fo, ws, stats, stream, ev);
        case 64: return wave_t<64>(sc, ra, fo, ws, stats, stream, ev);
        default: return hipErrorInvalidValue;
    }
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
