/* rtr.h — C ABI of librtr_hip.so, the MI355X-native replacement for the reference's
 * Vulkan dispatch + GLSL ray-tracing pipeline.
 *
 * The reference has no plugin / FFI interface: Application::run() talks to its Vulkan
 * wrappers directly (reference src/app/application.cppm:99-484).  The boundary cut here is
 * "everything run() hands to the GPU and everything it reads back" (SURVEY.md §8b).  Each
 * entry point cites the reference code it replaces.  Plain pointers and sizes only; no
 * C++/torch types; no exceptions cross this boundary (every call returns an rtr_status and
 * rtr_last_error() gives the thread-local message — the C++ shim in
 * realtimeraytracer_amd/csrc/host rethrows std::runtime_error to keep the reference's
 * caller-visible convention, src/main.cpp:12-15).
 *
 * Threading: handles are not thread-safe; one rtr_ctx per device; calls are synchronous
 * unless stated (reference is single-threaded with waitIdle between passes,
 * src/app/application.cppm:353,396,437).
 */
#ifndef RTR_H
#define RTR_H

#include <stddef.h>
#include <stdint.h>
#include "rtr_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RTR_ABI_VERSION 3

typedef enum rtr_status {
    RTR_OK = 0,
    RTR_ERR_INVALID_ARGUMENT = -1,
    RTR_ERR_HIP = -2,            /* a HIP runtime call failed (message has hipGetErrorString) */
    RTR_ERR_NO_DEVICE = -3,
    RTR_ERR_UNSUPPORTED = -4,    /* a feature that is not built (e.g. an image format the host loader cannot decode) */
    RTR_ERR_OUT_OF_MEMORY = -5,
    RTR_ERR_BVH_TOO_DEEP = -6,   /* traversal stack bound exceeded; fail loudly, never clamp */
    RTR_ERR_IO = -7
} rtr_status;

typedef struct rtr_ctx   rtr_ctx;    /* one per device: replaces Instance/Device/CommandPool (src/vulkan/context/) */
typedef struct rtr_scene rtr_scene;  /* replaces vertex/index/info buffers + BLAS[] + TLAS (src/app/application.cppm:230-271) */
typedef struct rtr_frame rtr_frame;  /* replaces the storage images of descriptor set 0 (src/app/application.cppm:108-138) */

/* Which image of the reference's descriptor set 0 (src/shaders/raygen.rgen:12-16;
 * binding numbers kept).  All are R8G8B8A8_UNORM with bytes B,G,R,255 as raygen stores them. */
typedef enum rtr_image {
    RTR_IMAGE_ANALYTIC = 0,           /* binding 0: LTC analytic, needs LTC tables */
    RTR_IMAGE_SHADOWED = 1,           /* binding 1: THE framebuffer of the north-star path */
    RTR_IMAGE_UNSHADOWED = 2,         /* binding 2 */
    RTR_IMAGE_DENOISED_SHADOWED = 3,  /* binding 3 (denoise.comp output) */
    RTR_IMAGE_DENOISED_UNSHADOWED = 4,/* binding 4 */
    RTR_IMAGE_FINAL = 5,              /* binding 5 (combine.comp output) */
    RTR_IMAGE_NORMAL = 6,             /* binding 6 */
    RTR_IMAGE_POSITION = 7,           /* binding 7 */
    RTR_IMAGE_HDR = 16                /* float4 pre-tonemap accumulation of SHADOWED (build-side extension) */
} rtr_image;

/* Bit mask of images a frame owns / a render call writes. */
#define RTR_IMG_BIT(which) (1u << (which))
#define RTR_IMAGES_FRAMEBUFFER (RTR_IMG_BIT(RTR_IMAGE_SHADOWED))
#define RTR_IMAGES_RAYGEN5 (RTR_IMG_BIT(RTR_IMAGE_ANALYTIC) | RTR_IMG_BIT(RTR_IMAGE_SHADOWED) | \
                            RTR_IMG_BIT(RTR_IMAGE_UNSHADOWED) | RTR_IMG_BIT(RTR_IMAGE_NORMAL) | \
                            RTR_IMG_BIT(RTR_IMAGE_POSITION))
#define RTR_IMAGES_DENOISE (RTR_IMG_BIT(RTR_IMAGE_DENOISED_SHADOWED) | RTR_IMG_BIT(RTR_IMAGE_DENOISED_UNSHADOWED) | \
                            RTR_IMG_BIT(RTR_IMAGE_FINAL))

/* One entry of the reference's texSamplers[] array (src/app/setup/create_scene.cppm:71-141): 8-bit texels as
 * core::file::createTextureImage produces them (src/core/file.cppm:272-311: stb_image, vertical flip,
 * R8G8B8A8_UNORM or R8_UNORM), sampled with the reference's sampler — linear filter, repeat addressing, one mip
 * (src/vulkan/memory/image_sampler.cppm:26-42).  Row 0 of `pixels` is v = 0. */
typedef struct rtr_texture {
    const uint8_t* pixels;     /* width*height*channels bytes; NULL for an unused slot (indices 0,1 = LTC tables) */
    uint32_t       width, height;
    uint32_t       channels;   /* 4 = RGBA8, 1 = R8 */
    uint32_t       _pad;
} rtr_texture;

/* What Application::run() uploads once (src/app/application.cppm:226-271,
 * src/app/setup/geometry_builder.cppm:50-212, src/vulkan/raytracing/tlas.cppm:44-149).
 * The library copies everything; the caller keeps ownership of its arrays. */
typedef struct rtr_scene_desc {
    const RtrVertex*        vertices;      uint32_t numVertices;   /* Vertex SSBO, application.cppm:242-248 */
    const uint32_t*         indices;       uint32_t numIndices;    /* mesh-local indices, geometry_builder.cppm:162-169 */
    const RtrMesh*          meshes;        uint32_t numMeshes;     /* one per BLAS */
    const RtrInstance*      instances;     uint32_t numInstances;  /* TLAS instances: lights first, then objects */
    const RtrObjectInfo*    objects;       uint32_t numObjects;    /* ObjectInfo SSBO, application.cppm:254-261 */
    const RtrAreaLightInfo* lights;        uint32_t numLights;     /* LightInfo SSBO, application.cppm:264-271 */
    /* texSamplers[0], [1]: 64x64 RGBA32F LTC tables (src/app/setup/create_scene.cppm:65-69,162-214).
     * NULL -> RTR_IMAGE_ANALYTIC cannot be rendered (RTR_ERR_UNSUPPORTED if asked for). */
    const float*            ltc1;          /* 64*64*4 floats or NULL */
    const float*            ltc2;          /* 64*64*4 floats or NULL */
    /* miss.rmiss:21-26 samples an equirect HDRI (binding 7); when `hdri` is NULL the sky is this constant
     * (sRGB-encoded; ToLinear applied as the miss shader does). */
    float                   skyColor[3];
    float                   _pad;
    /* texSamplers[] (set 1 binding 4): ObjectInfo.colorIndex / specularIndex / metallicIndex / opacityIndex index
     * this array; the reference keeps the LTC tables at 0 and 1 so material textures start at 2. */
    const rtr_texture*      textures;      uint32_t numTextures;
    const rtr_texture*      hdri;          /* equirect sky (RGBA8, as stbi_load of the .hdr gives, file.cppm:279-291) or NULL */
    /* Acceleration-structure build preference (reference: vk::BuildAccelerationStructureFlagBitsKHR, blas.cppm:115
     * ePreferFastTrace; tlas.cppm eAllowUpdate): RTR_BUILD_HOST_SAH (default, best trace speed) or
     * RTR_BUILD_DEVICE_LBVH (Morton/radix-tree build on the GPU, fastest build).  Rendered images are identical. */
    uint32_t                buildFlags;
    uint32_t                _pad2;
} rtr_scene_desc;

#define RTR_BUILD_HOST_SAH    0u
#define RTR_BUILD_DEVICE_LBVH 1u

typedef struct rtr_scene_stats {
    uint32_t numTriangles;
    uint32_t numNodes;
    uint32_t maxDepth;        /* deepest leaf, root = 1 */
    uint32_t maxLeafSize;
    uint32_t bvhLayoutVersion;
    uint32_t stackEntries;    /* LDS stack entries per lane the kernels were specialised for */
    float    buildMs;
    float    sahCost;
    float    boundsMin[3];
    float    boundsMax[3];
    float    boxPad;
    float    _pad;
    RtrBvhGrid grid;          /* the 16-bit planes of the exported nodes live on this grid (rewritten by a refit) */
    uint32_t numWideNodes;    /* RtrWideNode records of the wide view (rtr_scene_export_wide, rtr_host_build_bvh_wide); 0 from rtr_host_build_bvh */
    uint32_t wideLayoutVersion;
    uint32_t _pad2[2];
} rtr_scene_stats;

/* Per-dispatch arguments: what the reference passes as the traceRaysKHR extent + the two
 * shader constants (src/vulkan/ray_tracing_pipeline.cppm:212-214, src/shaders/raygen.rgen:8-9),
 * plus the band sharding and accumulation that are new in this build. */
typedef struct rtr_render_params {
    uint32_t width;            /* full frame width  (dispatch extent x) */
    uint32_t height;           /* full frame height (dispatch extent y) */
    uint32_t spp;              /* NUM_PRIMARY_RAYS (reference: 4) */
    uint32_t numShadowRays;    /* NUM_SHADOW_RAYS  (reference: 3) */
    uint32_t images;           /* mask of RTR_IMG_BIT(...) to produce; 0 -> RTR_IMAGES_FRAMEBUFFER */
    uint32_t bandRows;         /* rows per band; 0 -> 8 */
    uint32_t shardIndex;       /* this device renders bands b with b % shardCount == shardIndex */
    uint32_t shardCount;       /* 0 or 1 -> whole frame */
    uint32_t accumulate;       /* 0: HDR = this frame; 1: HDR += this frame (needs RTR_IMAGE_HDR in the frame) */
    uint32_t accumulatedFrames;/* frames already summed in HDR (tonemap divides by accumulatedFrames+1) */
    uint32_t collectStats;     /* 1: run the counting variants of the kernels and fill rtr_frame_stats counters */
    uint32_t pipeline;         /* 0: default; 1: megakernel; 2: wavefront (staged) */
} rtr_render_params;

typedef struct rtr_frame_stats {
    /* exact work counters of the last render with collectStats=1 */
    uint64_t numRays;          /* every traceRay-equivalent issued: primary + shadow */
    uint64_t numPrimaryRays;
    uint64_t numShadowRays;
    uint64_t numNodeVisits;    /* N_node: node record fetches */
    uint64_t numTriTests;      /* N_tri : 48-B triangle record fetches */
    uint64_t numHits;          /* N_hit : closest-hit shading fetches (236 B each) */
    uint64_t numLightFetches;  /* N_lightfetch: LightInfo reads (96 B each) */
    uint64_t numLightTriFetches;/* light triangle vertex fetches (3 idx + 3 x 48 B = 156 B each) */
    uint64_t numTexFetches;    /* bilinear texture / HDRI lookups (4 texels, 16 B each) */
    uint64_t numAlphaTests;    /* opacity.rahit invocations that sampled an opacity map (236 B fetch + 1 lookup each) */
    uint64_t algorithmicBytes; /* B = 32 N_node_primary (RTR_BVH_NODE_BYTES) + 64 N_node_shadow (RTR_WIDE_NODE_BYTES; 32 in the megakernel) + 48 N_tri + 236 (N_hit + N_alpha) + 96 N_lf + 156 N_ltf + 16 N_tex + 4 k P (+16/32 P HDR) */
    /* the any-hit share of the above (work of the k_shadow_trace launch, the dominant kernel) */
    uint64_t numShadowNodeVisits;
    uint64_t numShadowTriTests;
    uint64_t shadowTraceBytes; /* 64 N_node_shadow (RTR_WIDE_NODE_BYTES: 4-wide records visited; 32 in the megakernel, which walks the BVH2) + 48 N_tri_shadow + 37 N_shadow_rays (20-B queue record + the 16-B origin its pixel-sample's rays share + the visibility byte) */
    /* timings of the last render (HIP events on the render stream), milliseconds */
    float    totalMs;
    float    primaryMs;        /* k_primary (wavefront) or the whole megakernel */
    float    shadowGenMs;      /* k_shadow_gen */
    float    shadowTraceMs;    /* the any-hit kernel (k_shadow_trace4) alone: the dominant kernel */
    float    resolveMs;        /* k_resolve */
    uint32_t localRows;        /* rows this shard rendered */
    uint32_t localPixels;
    uint32_t pipelineUsed;     /* 1 megakernel, 2 wavefront */
    float    shadowTraceClockMHz; /* shader clock the any-hit launch ran at: s_memtime ticks / s_memrealtime (100 MHz) ticks of one wave per XCD, MEAN over the XCDs (they clock independently; slowest / fastest below) */
    /* scheduling of the any-hit kernel, from its counting form (collectStats = 1): loop trips of its node and triangle phases,
     * summed over waves, and the lanes that had work in those trips (lane utilisation = lanes / (64 trips)) */
    uint64_t shadowInnerIterations, shadowInnerActiveLanes;
    uint64_t shadowTriIterations, shadowTriActiveLanes;
    uint64_t shadowRefills;
    float    shadowTailMs;     /* k_shadow_tail (the rays that outgrew the LDS stack, redone over the BVH2); part of totalMs */
    uint32_t _padTail;
    uint64_t primaryTailRays;  /* camera rays that outgrew the 16-entry LDS stack of k_primary_persist and were redone by k_primary_tail over the BVH2 */
    uint64_t shadowTailRays;   /* rays that outgrew the 16-entry LDS stack and were finished by k_shadow_tail over the BVH2 (both parts of their work are in the counters) */
    float    shadowTraceClockMinMHz, shadowTraceClockMaxMHz;   /* the slowest and the fastest XCD of that launch */
} rtr_frame_stats;

/* ---- context -------------------------------------------------------------------------- */
/* replaces Instance+Device creation (src/app/application.cppm:65-69): picks HIP device `ordinal`. */
int  rtr_ctx_create(int deviceOrdinal, rtr_ctx** out);
/* Scenes and frames created on a context may be destroyed after it (garbage-collected bindings do that): the context's
 * resources are released when its last child is gone.  Creating new objects on a destroyed context is an error. */
void rtr_ctx_destroy(rtr_ctx* ctx);
/* Use an existing HIP stream (e.g. torch's current stream) for all work of this ctx; NULL -> own stream. */
int  rtr_ctx_set_stream(rtr_ctx* ctx, void* hipStream);
/* The HIP stream (hipStream_t) this context's work is enqueued on, for callers that order their own work against it with events. */
int  rtr_ctx_get_stream(rtr_ctx* ctx, void** hipStream);
int  rtr_ctx_device_name(rtr_ctx* ctx, char* buf, size_t bytes);
/* Run-time tunables of the staged pipeline (scheduling knobs of its kernels: queue binning, batch lengths, persistent workgroups per
 * CU ...; the names are the fields of rtrdev::Tunables, kernels/rtr_kernels.h).  A context reads them from the environment ONCE, when
 * it is created (RTR_<NAME IN CAPITALS>); a render uses those of the context of its (leading) frame.  No setting changes a pixel
 * (tested); the defaults are the measured optima.  RTR_ERR_INVALID_ARGUMENT for an unknown name or a value outside its range.
 * The reference has no counterpart: its traversal is the driver's (src/vulkan/ray_tracing_pipeline.cppm:212-214). */
int  rtr_ctx_set_tunable(rtr_ctx* ctx, const char* name, uint32_t value);
int  rtr_ctx_get_tunable(const rtr_ctx* ctx, const char* name, uint32_t* value);

/* ---- scene ---------------------------------------------------------------------------- */
/* replaces createSceneFromObjectsAndLights' GPU half (src/app/setup/create_scene.cppm:48-160):
 * validates, flattens instances to world space, builds the BVH on the host, uploads. */
int  rtr_scene_create(rtr_ctx* ctx, const rtr_scene_desc* desc, rtr_scene** out);
/* The same scene once more — on another context, usually another device — WITHOUT building its tree again: `built` is a scene
 * made from the same `desc` (same arrays, same buildFlags) whose host-side copy of the tree is uploaded as it is.  What
 * librtr_mgpu.so replicates the scene with: one build per node instead of one per GPU.  The reference builds its acceleration
 * structure once, on its one device (src/vulkan/raytracing/blas.cppm:75-167); this is that build shared by N devices.
 * RTR_ERR_INVALID_ARGUMENT if `built` does not match `desc` (triangle count). */
int  rtr_scene_create_like(rtr_ctx* ctx, const rtr_scene_desc* desc, const rtr_scene* built, rtr_scene** out);
void rtr_scene_destroy(rtr_scene* scene);
int  rtr_scene_get_stats(const rtr_scene* scene, rtr_scene_stats* out);
/* Copy out the device BVH arrays (test / oracle hook; sizes and the plane grid from rtr_scene_get_stats). */
int  rtr_scene_export_bvh(const rtr_scene* scene, RtrBvhNode* nodes, size_t nodeBytes,
                          RtrBvhTri* tris, size_t triBytes);
/* Copy out the 4-wide view the any-hit kernel walks (RtrWideNode, layout RTR_WIDE_LAYOUT_VERSION): numWideNodes records; their
 * leaf codes index the triangle array of rtr_scene_export_bvh.  Test / oracle hook: the oracle restates the kernel's walk over
 * it to check the kernel's work counters. */
int  rtr_scene_export_wide(const rtr_scene* scene, RtrWideNode* nodes, size_t nodeBytes);
/* Host-only BVH build (no device needed): validates `desc`, flattens and builds exactly as
 * rtr_scene_create does and copies the result out.  Call with nodes == tris == NULL to get the counts
 * in `stats`.  Used by the CPU-side tests (BVH invariants, oracle BVH-vs-brute-force). */
int  rtr_host_build_bvh(const rtr_scene_desc* desc, rtr_scene_stats* stats, RtrBvhNode* nodes, size_t nodeBytes,
                        RtrBvhTri* tris, size_t triBytes);
/* The same build, plus the 4-wide view rtr_scene_create would put on the device for it: the host restatement of the kernels that
 * make it (the wide centre, the records the builder's cost-driven collapse chose, breadth-first order) — byte-identical to
 * rtr_scene_export_wide of a scene made from `desc` (checked in the GPU tests).  stats->numWideNodes / wideLayoutVersion and the
 * wide centre in stats->grid are filled; call with all arrays NULL for the counts.  No device needed: CPU-only tests walk the wide
 * view with the oracle, and tree experiments price a builder change by the oracle's visit counters before any GPU time is spent. */
int  rtr_host_build_bvh_wide(const rtr_scene_desc* desc, rtr_scene_stats* stats, RtrBvhNode* nodes, size_t nodeBytes,
                             RtrBvhTri* tris, size_t triBytes, RtrWideNode* wide, size_t wideBytes);
/* Limits of the 32-bit record offsets the traversal kernels use: RTR_OK if a scene of numTriangles triangles and numNodes BVH nodes
 * can be addressed (triangle records and 4-wide records each below 2 GiB: at most 44 739 242 triangles, 33 554 431 nodes; and the
 * 2^28 of the leaf encoding), else RTR_ERR_INVALID_ARGUMENT with the reason in rtr_last_error().  rtr_scene_create applies it
 * to the worst case for its triangle count (numTriangles - 1 nodes) before anything is built or allocated. */
int  rtr_check_scene_limits(uint64_t numTriangles, uint64_t numNodes);
/* Dynamic scenes: new instance transforms (same instances, meshes and customIndex as at creation) and, optionally,
 * new light infos (NULL keeps them).  World-space triangle records are recomputed and every BVH box is re-fitted ON
 * THE DEVICE; the topology is kept.  Replaces TLAS::updateTransform + TLAS::refit
 * (src/vulkan/raytracing/tlas.cppm:151-207; present in the reference, never called by its app). */
/* Both update calls rewrite device arrays that renders read (nodes, 4-wide records, triangle records, light tables).  A scene may be
 * rendered by frames of several contexts / streams at once, so the calls first JOIN THE WHOLE DEVICE (hipDeviceSynchronize: every
 * frame in flight on any stream of this process finishes with the old scene), rewrite, and return when the new state is complete.
 * They are therefore safe to call at any time from the thread that enqueues the renders; a caller that renders the scene from
 * other threads or processes must itself keep those from enqueueing new frames of this scene until the call has returned. */
int  rtr_scene_update_instances(rtr_scene* scene, const RtrInstance* instances, uint32_t numInstances,
                                const RtrAreaLightInfo* lights, uint32_t numLights);
/* replaces the host-visible LightInfo buffer rewrite (src/app/application.cppm:264-271). */
int  rtr_scene_update_lights(rtr_scene* scene, const RtrAreaLightInfo* lights, uint32_t numLights);

/* ---- frame ---------------------------------------------------------------------------- */
/* replaces the 8 storage images (src/app/application.cppm:108-138).  `rows` is the number of
 * LOCAL rows (== height when unsharded; see rtr_shard_rows).  `images` is a RTR_IMG_BIT mask. */
int  rtr_frame_create(rtr_ctx* ctx, uint32_t width, uint32_t rows, uint32_t images, rtr_frame** out);
void rtr_frame_destroy(rtr_frame* frame);
/* Let an RGBA8 image of this frame live in caller-owned device memory (e.g. a torch tensor that
 * RCCL will gather).  bytes must be width*rows*4. */
int  rtr_frame_bind_external(rtr_frame* frame, int which, void* devicePtr, size_t bytes);
int  rtr_frame_device_ptr(const rtr_frame* frame, int which, void** devicePtr, size_t* bytes);
/* replaces the image->swapchain copy / readback (src/app/application.cppm:450-457). */
int  rtr_frame_download(const rtr_frame* frame, int which, void* dst, size_t bytes);
int  rtr_frame_clear(rtr_frame* frame);
int  rtr_frame_get_stats(const rtr_frame* frame, rtr_frame_stats* out);

/* Number of local rows a shard owns: `height` when unsharded, otherwise ceil(bands / shardCount) * bandRows so
 * every shard has the same count (equal-size shards for the gather; padding rows stay zero). */
uint32_t rtr_shard_rows(uint32_t height, uint32_t bandRows, uint32_t shardCount);

/* ---- dispatch ------------------------------------------------------------------------- */
/* replaces bind pipeline + push constants + vkCmdTraceRaysKHR + waitIdle
 * (src/app/application.cppm:362-389, src/vulkan/ray_tracing_pipeline.cppm:212-214). Synchronous. */
int  rtr_render(rtr_scene* scene, const RtrCameraData* camera, const RtrSceneInfo* sceneInfo,
                const rtr_render_params* params, rtr_frame* frame);
/* Asynchronous variant: enqueues on the stream of the FRAME's context and returns; rtr_frame_wait() joins.  The scene
 * may belong to another context of the same device (it is read-only during rendering), so frames created on
 * different contexts render concurrently on their own streams against one scene. */
int  rtr_render_async(rtr_scene* scene, const RtrCameraData* camera, const RtrSceneInfo* sceneInfo,
                      const rtr_render_params* params, rtr_frame* frame);
/* Several frames in ONE launch of every kernel of the pipeline: cameras[b] / sceneInfos[b] -> frames[b], b < n <= RTR_MAX_BATCH, all
 * with the same params (extent, spp, sharding, images; accumulate applies to each frame's own HDR image).  A frame at 1 spp — and
 * a 1/N shard of one even more so — is too little work per launch for the latency-bound kernels (the camera-ray kernel takes
 * 0.34 ms for one 1080p frame's rays and 0.41 ms for four times as many): batching trades latency of the individual frame for
 * throughput, like frames in flight do, and composes with them.  Same pixels as n calls of rtr_render_async (tested).  The
 * launch runs on frames[0]'s context stream and its times / counters (rtr_frame_get_stats) are frames[0]'s, for the whole launch;
 * rtr_frame_wait on any of the frames joins it.  The other frames may live on other contexts (streams) of the device: the launch waits
 * for what their streams hold when it is enqueued and their streams wait for the launch, so work enqueued for a frame before and
 * after a batch is ordered around it without a host join (tested).  Staged pipeline only; the frames must live on one device and be distinct.
 * The reference records one vkCmdTraceRaysKHR per frame (src/app/application.cppm:362-389); this is n of them in one. */
#define RTR_MAX_BATCH 32
int  rtr_render_batch_async(rtr_scene* scene, const RtrCameraData* cameras, const RtrSceneInfo* sceneInfos, const rtr_render_params* params,
                            rtr_frame* const* frames, uint32_t n);
/* The LATENCY form: ONE frame as `parts` (<= RTR_MAX_SPLIT) band-shards — bands of params->bandRows rows, band b to part b mod parts,
 * the sharding of the multi-GPU path — each on a stream of its own inside the library, all writing their rows of `frame`'s images in
 * place (no gather: `frame` is the whole frame, rows == height).  The kernels of a frame are a dependency chain and each ends in a
 * tail; split, part k+1's camera rays and queue build run under part k's traversal.  Same pixels as rtr_render (tested), stream-ordered
 * on the frame's context stream like rtr_render_async (fork and join are events); rtr_frame_wait joins, and rtr_frame_get_stats then
 * gives totalMs = the frame's duration, fork to join, and the counters and per-kernel times SUMMED over the parts (they overlap).
 * params->shardCount must be 0 or 1.  parts == 1 is rtr_render_async.  What the reference's loop needs — one frame per iteration, then
 * waitIdle (src/app/application.cppm:352-389,437) — where rtr_render_batch_async trades that latency for throughput. */
#define RTR_MAX_SPLIT 16
int  rtr_render_split_async(rtr_scene* scene, const RtrCameraData* camera, const RtrSceneInfo* sceneInfo, const rtr_render_params* params,
                            rtr_frame* frame, uint32_t parts);
int  rtr_render_split(rtr_scene* scene, const RtrCameraData* camera, const RtrSceneInfo* sceneInfo, const rtr_render_params* params,
                      rtr_frame* frame, uint32_t parts);
/* How many frames of these params one launch takes: min(RTR_MAX_BATCH, what the staged pipeline's scratch can address — its
 * visibility slots are 31-bit indices: pixel-sample slots of the launch, rounded up to a power of two, x queries per pixel-sample).
 * 1 when only a single frame fits (or the megakernel is asked for); a caller that batches asks this first.  Pure arithmetic. */
int  rtr_render_batch_limit(const rtr_scene* scene, const rtr_render_params* params, uint32_t numAreaLights, uint32_t* maxFrames);
int  rtr_frame_wait(rtr_frame* frame);

/* The passes that follow the ray-gen dispatch in the reference's frame loop (src/app/application.cppm:391-445):
 * `iterations` (reference: NUM_DENOISING_ITERATIONS = 4) rounds of {a-trous pass on the unshadowed image, then on the
 * shadowed image} ping-ponging between the sampled (1,2) and denoised (3,4) images with step (i+1), c_phi 1,
 * n_phi = p_phi = 1e-3 (src/shaders/denoise.comp), then combine.comp: FINAL = ANALYTIC * shadowed / max(unshadowed, .001)
 * reading the pair the ping-pong flag points at.  The frame must own images 0-7 and hold a full (unsharded) frame
 * rendered with RTR_IMAGES_RAYGEN5.  Synchronous. */
int  rtr_denoise_combine(rtr_frame* frame, int iterations);

/* Rank-0 step after the RCCL gather: `gathered` holds shardCount blocks of (localRows x width)
 * RGBA8 pixels in rank order; writes the de-interleaved (height x width) image to `dst`.
 * Both are device pointers; ENQUEUED on the ctx stream (asynchronous; synchronise the stream to read). */
int  rtr_deinterleave_bands(rtr_ctx* ctx, const void* gathered, void* dst, uint32_t width, uint32_t height,
                            uint32_t bandRows, uint32_t shardCount);

/* ---- errors --------------------------------------------------------------------------- */
const char* rtr_last_error(void);
const char* rtr_status_string(int status);
int         rtr_abi_version(void);
/* Revision tag of the any-hit kernel + the tree layout it walks (bumped when either changes); measurement files carry it. */
const char* rtr_kernel_revision(void);

#ifdef __cplusplus
}
#endif
#endif /* RTR_H */
