/* rtr_kernels.h — host-callable launchers of the HIP kernels (internal to librtr_hip.so). */
#pragma once
#include <hip/hip_runtime.h>
#include "rtr_device.h"

namespace rtrdev {

/* Bumped whenever the any-hit kernel (k_shadow_trace4) or the tree it walks changes what it executes: the counter files under
 * profiles/ carry the revision they were collected with, and bench.py refuses to mix revisions ("record the layout
 * version next to every number"). */
/* Which hit child of a 4-wide record a shadow ray enters first: 1 (default) the one that EXITS last, 0 the nearest entry (rounds 1-4) — a
 * compile-time choice (inner_nodes4); a build with 0 says so in its revision, and the oracle walks either way (shadowWalk bits 2-3). */
#ifndef RTR_SHADOW_FAR_FIRST
#define RTR_SHADOW_FAR_FIRST 1
#endif
#if RTR_SHADOW_FAR_FIRST
#define RTR_ANYHIT_KERNEL_REVISION "r05.3"
#else
#define RTR_ANYHIT_KERNEL_REVISION "r05.3-nearest-first"
#endif

/* Batch cursors per queue (and batch lists per octant).  Workgroups are dealt round-robin to the XCDs, and a workgroup starts on
 * cursor blockIdx mod 8: with 8, 4, 2 or 1 XCDs visible (SPX, DPX, QPX, CPX partitions of an MI355X) a cursor is still used by the
 * waves of ONE XCD — cursor r by XCD r mod numXccs — which is what matters (a counter that the waves of several XCDs wait on
 * serialises them: 3.75 instead of 1.9 ms).  Eight is therefore right for every partition mode of the part and stays a compile-time
 * constant: `% kQueueRegions` is an AND in the refill path.  (rtr_ctx reads hipDeviceAttributeNumberOfXccs only to say so in
 * rtr_ctx_device_name; a count that does not divide 8 would merely share cursors, never give a wrong result.) */
constexpr uint32_t kQueueRegions = 8;
/* Words of Workspace::queueCount (the control block of a launch, zeroed before it): */
constexpr uint32_t kQueueLenWord = 0;        /* queued rays */
constexpr uint32_t kPrimaryRedoWord = 2;     /* camera rays k_primary left to k_primary_tail */
constexpr uint32_t kOccludedWord = 3;        /* occluded rays of the any-hit launch (the next launch's pre-fill of the visibility array is chosen by it) */
constexpr uint32_t kBatchCursorWord = 16;    /* first batch cursor; cursor r at kBatchCursorWord + 16 r (64 B apart) */
static_assert(kOccludedWord < kBatchCursorWord && kPrimaryRedoWord < kOccludedWord, "the small counters sit below the first batch cursor");
constexpr uint32_t kQueueLists = kQueueRegions * kQueueRegions;        /* batch lists of the binned queue: (direction octant, consumer XCD) */
constexpr uint32_t kQueueListLens = kBatchCursorWord + 16 * kQueueLists;            /* first word of the list lengths (read-only while the queue drains) */
constexpr uint32_t kPrimaryCursors = kQueueListLens + kQueueLists;   /* kQueueRegions batch cursors of k_primary_persist, 64 B apart */
constexpr uint32_t kQueueCtrlWords = kPrimaryCursors + 16 * kQueueRegions;
static_assert(kQueueLists == 64, "k_shadow_gen_oct reserves the batch lists with one lane per list");

constexpr size_t kSpillInts = (size_t)64 * 64 * 256;     /* full-depth stacks of the two redo kernels: 64 entries x (64 workgroups x 256 lanes) */

/* The shadow-ray queue: 20 B per ray + 16 B per pixel-sample.  A pixel-sample's rays (up to 13 on the bench frame) all start at the
 * same point, so the origin is stored once per pixel-sample and a ray record is its direction, its far limit and the index of its
 * visibility byte; that index also names the pixel-sample (slot & slotMask: the planes of the visibility array are slotStride =
 * a power of two apart).  32-B records with the origin in every ray made the queue 0.79 GB per 1080p frame, written by a kernel that
 * is bound by exactly those writes. */
struct RayQueue {
    float4*   dt = nullptr;          /* per ray: direction xyz, tmax */
    uint32_t* slot = nullptr;        /* per ray: visibility index = query * slotStride + pixel-sample */
    float4*   origin = nullptr;      /* per pixel-sample: origin of its shadow rays (hit point + 0.01 normal; the queue build), w = code of the leaf its hit triangle sits in (the camera-ray kernel) */
    uint32_t  slotStride = 0;        /* power of two >= pixel-sample slots of the frame */
    uint32_t  slotMask = 0;          /* slotStride - 1 */
    uint32_t  ownLeaf = 0;           /* 1: rays that leave INTO their surface are marked (bit 31 of the slot word) and start at their own triangle's leaf (tunable trace_own_leaf) */
};

/* Scratch of the wavefront (staged) pipeline, owned by an rtr_frame. */

struct Workspace {
    float4*   hitTuvp = nullptr;     /* per (pixel,sample): t,u,v,bits(primitiveID) */
    uint32_t* hitCustom = nullptr;   /* per (pixel,sample): customIndex or RTR_MISS */
    RayQueue  rayQueue;              /* the queued shadow rays */
    uint8_t*  vis = nullptr;         /* per slot (query-major planes, rayQueue.slotStride apart): 1 = occluded */
    uint32_t  visFill = 1;           /* what the array is pre-filled with before every launch (the commoner outcome); the any-hit kernel stores the other */
    size_t    visPlaneBytes = 0;     /* what a launch pre-fills and reads of each plane: its pixel-sample slots (a multiple of 256) */
    uint32_t  visPlanes = 0;         /* queries per pixel-sample = planes in use */
    uint32_t* queueCount = nullptr;  /* kQueueCtrlWords words: [0] queued rays, [1] batch cursor of the counting kernel, [2] k_primary's redo count, [3] occluded rays of the any-hit launch, [16 + 16 r] batch cursor of queue region r (2-wide kernel, r < 8) or of batch list r = octant * 8 + xcd (64 B apart: a cursor is hammered by one XCD's waves), [kQueueListLens + r] length of list r, [kPrimaryCursors + 16 r] batch cursor of region r of the camera rays (k_primary_persist) */
    uint2*    batchLists = nullptr;  /* kQueueLists lists of listStride batches {first queue index, rays}: the queue binned by direction octant */
    uint32_t  listStride = 0;
    uint32_t* overflow = nullptr;    /* [0] count, then queue indices of rays k_shadow_trace left to k_shadow_tail: overflowCap entries; a count past the capacity makes k_shadow_tail redo the whole queue */
    uint32_t  overflowCap = 0;
    unsigned long long* clk = nullptr;   /* 2 x kQueueRegions words: {shader-clock ticks, 100-MHz ticks} of the any-hit launch, one wave per XCD */
    int32_t*  spill = nullptr;       /* full-depth traversal stacks of the two redo kernels: kSpillInts = 64 entries x (64 workgroups x 256 lanes) */
    size_t    capPixelSamples = 0;
    size_t    capRays = 0;
};

/* Run-time tunables of the staged pipeline.  They belong to an rtr_ctx: read from the environment ONCE, when the context is created
 * (RTR_<NAME IN CAPITALS>, e.g. RTR_TRACE_BINNED=1), settable afterwards with rtr_ctx_set_tunable; a render uses those of its (leading)
 * frame's context.  The defaults are the swept optima (profiles/sweep_*.sh); every setting renders the same pixels (tested). */
struct Tunables {
    uint32_t primary_packet = 0;                  /* camera rays: 0 = one ray per lane (k_primary + k_primary_tail); 1 = a tile's 64 rays walk the BVH2 as one packet (k_primary_packet:
                                                   * scalar node fetches, one stack per tile, no tail kernel — and no faster: 0.168 / 0.334 ms per frame in a launch / alone against
                                                   * 0.167 / 0.320, profiles/r04/ab_primary_packet.log: a wave walks its nodes strictly one after the other, one fetch in flight) */
    uint32_t primary_wide = 0;                    /* 1: camera rays one per lane over the 4-wide view (k_primary4: half the dependent visits, twice the instructions per visit) */
    uint32_t primary_persist = 0;                 /* camera rays by k_primary_persist: 0 never, 1 whenever it can, 2 by the size of the launch */
    uint32_t primary_persist_min_rays = 6u << 20;
    uint32_t primary_batch = 64, primary_refill = 24, primary_inner_min = 20, primary_wgs_per_cu = 8;
    uint32_t trace_bvh4 = 1;                      /* 0: the 2-wide any-hit kernel on the plain queue (comparison form; it has no counting form) */
    uint32_t trace_batch = 0;                     /* rays a wave reserves per cursor atomic; 0: by the length of the queue (256 / 512) */
    uint32_t trace_binned = 2;                    /* queue binned by direction octant: 0 never, 1 always, 2 by queue and tree size */
    uint32_t queue_nt = 1;                        /* bit 0: the any-hit kernel reads the queue past the caches; bit 1: the queue-build kernel writes it so */
    uint32_t trace_wgs_per_cu = 0;                /* persistent workgroups per CU; 0: 8 on long queues, 6 on short ones */
    uint32_t trace_refill = 20, trace_inner_min = 28;
    uint32_t trace_octant_forms = 1;
    uint32_t trace_own_leaf = 1;                  /* 1: a shadow ray that leaves its surface point into the surface (dot(normal, direction) < 0) tests the leaf of the triangle it starts on
                                                   * first, then walks from the root: it nearly always re-enters that triangle (44 % of the bench frame's rays; 14.3 -> 10.6 visits per ray) */
    uint32_t trace_top_nodes = 0xffffffffu;       /* 4-wide records kept in LDS (at most the kernel's kTopNodes) */
    uint32_t resolve_row_waves = 0;
    uint32_t resolve_compact = 1;                 /* framebuffer-only launches: 1 = k_resolve_compact (the BRDF of the VISIBLE samples of a tile, compacted over the wave's lanes), 0 = k_resolve (one lane per pixel walks its samples) */
    uint32_t split_priorities = 0;                /* rtr_render_split: part k's stream gets the k-th highest stream priority (0, the default: all parts default priority — priorities bought nothing, profiles/r04/sweep_split_priorities.log) */
};
Tunables tunables_from_env();
/* name: a field of Tunables (lower case).  false: no such tunable, or a value outside its range. */
bool tunable_set(Tunables& t, const char* name, uint32_t value);
bool tunable_get(const Tunables& t, const char* name, uint32_t* value);

/* stackEntries must be one of 16, 32, 64. */
hipError_t launch_megakernel(const DeviceScene& sc, const RenderArgs& ra, const FrameOut& fo, int stackEntries,
                             Counters* stats, hipStream_t stream);

/* Staged pipeline: primary trace -> shadow-ray generation (ballot-compacted queue) -> any-hit trace
 * -> resolve.  `ev` (6 events, may be null) are recorded between stages for per-stage timing ([5]: after the any-hit kernel,
 * before k_shadow_tail). */
hipError_t launch_wavefront(const DeviceScene& sc, const FrameBatch& batch, const Workspace& ws, const Tunables& tun,
                            int stackEntries, Counters* stats, hipStream_t stream, hipEvent_t* ev, uint32_t numCus);

/* fills DeviceScene::lightTris (4 x float4 per light triangle, light l from first[l]); after create and after light transforms change */
hipError_t launch_light_tris(const RtrAreaLightInfo* lights, const RtrVertex* vertices, const uint32_t* indices, const uint32_t* first,
                             uint32_t numLights, float4* out, hipStream_t stream);
hipError_t launch_deinterleave(const uint32_t* gathered, uint32_t* dst, uint32_t width, uint32_t height,
                               uint32_t bandRows, uint32_t shardCount, uint32_t localRows, hipStream_t stream);

}  // namespace rtrdev
