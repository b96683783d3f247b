/* rtr_mgpu.h — C ABI of librtr_mgpu.so: the frame tile-sharded over the GPUs of one node, assembled on rank 0 with ONE exchange
 * step over xGMI (RCCL grouped send / recv), SURVEY.md §8b "Threading" row and §8e.
 *
 * The reference renders on one GPU (one vkCmdTraceRaysKHR per frame, src/app/application.cppm:362-389); this is the part of
 * the north star that has no counterpart there.  librtr_mgpu.so is a pure client of include/rtr.h (librtr_hip.so) plus RCCL and
 * the HIP runtime: it owns one rtr_ctx + one replicated rtr_scene per rank, the RCCL communicator, one host thread per local
 * rank (enqueueing a shard costs ~25 us of host time, which eight ranks must not pay one after the other), one render stream per
 * frame slot and one communication stream per rank, ordered against each other with events — so that the kernels of several
 * frames overlap on a GPU (a 1/8-frame shard cannot fill it) and the gather of frame n runs under the rendering of frame n+1.
 *
 * Partitioning (same as rtr_render_params::shardIndex / shardCount): interleaved bands of 8 rows, band b -> rank b mod N; every
 * rank renders rtr_shard_rows() rows into a compact image; rank 0 receives the N-1 other shards next to its own (it renders
 * straight into slot 0 of the gather buffer) and de-interleaves them into the full frame (k_deinterleave).  Pixels are
 * independent and the PCG seeds depend only on (x, y, sample, frame), so the assembled frame is bit-identical to the one-GPU frame.
 *
 * Two ways to make the ranks:
 *   rtr_mgpu_create       one process drives n devices (ncclCommInitAll); what a C++ application like the reference's would use;
 *   rtr_mgpu_create_rank  one process per GPU (ncclCommInitRank with an id made by rtr_mgpu_unique_id on rank 0 and handed to the
 *                         others by the launcher's own means); what bench.py does under torch.distributed.run.
 * Handles are not thread-safe; calls return an rtr_status (include/rtr.h) and rtr_mgpu_last_error() has the message.
 */
#ifndef RTR_MGPU_H
#define RTR_MGPU_H

#include "rtr.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rtr_mgpu rtr_mgpu;

#define RTR_MGPU_ID_BYTES 128          /* sizeof(ncclUniqueId) */
#define RTR_MGPU_MAX_SLOTS 64

/* Fills `id` (RTR_MGPU_ID_BYTES bytes) with a fresh communicator id: call on one rank, distribute to all. */
int  rtr_mgpu_unique_id(void* id);

/* One process, n devices (ordinals in `devices`, rank r = devices[r]).  framesInFlight frame slots (1..RTR_MGPU_MAX_SLOTS). */
int  rtr_mgpu_create(const int* devices, int n, int framesInFlight, rtr_mgpu** out);
/* This process is rank `rank` of `nranks`, on HIP device `device`. */
int  rtr_mgpu_create_rank(int device, int rank, int nranks, const void* id, int framesInFlight, rtr_mgpu** out);
void rtr_mgpu_destroy(rtr_mgpu* m);

/* Replicates the scene on every local rank (rtr_scene_create per device; the caller keeps its arrays). */
int  rtr_mgpu_scene_create(rtr_mgpu* m, const rtr_scene_desc* desc);

/* Renders the frame into slot `slot`: every local rank enqueues its shard (rtr_render_async), then the exchange on its
 * communication stream (rank 0: grouped ncclRecv x (N-1) + k_deinterleave; others: ncclSend).  params->shardIndex / shardCount
 * are set by the library; width, height, spp, numShadowRays, bandRows are the caller's; images is RTR_IMAGES_FRAMEBUFFER, plus
 * RTR_IMG_BIT(RTR_IMAGE_HDR) when frames are summed (accumulate / accumulatedFrames as in rtr_render; each shard keeps its own
 * float accumulator and tonemaps it, the RGBA8 result is what travels).  flags: RTR_MGPU_NO_EXCHANGE renders without the
 * exchange — the frames of a sum before its last.  Returns when everything is enqueued; rtr_mgpu_wait joins.  Calls on one slot
 * are ordered; a slot's previous exchange is waited for (on the GPU, by an event) before its buffers are overwritten. */
#define RTR_MGPU_NO_EXCHANGE 1
/* the launch's transfers as one RCCL group PER SLOT instead of one group for the launch (the fallback; also set for every launch of a
 * handle made while RTR_MGPU_GROUP_PER_SLOT=1 is in the environment) */
#define RTR_MGPU_GROUP_PER_SLOT 2
int  rtr_mgpu_render_async(rtr_mgpu* m, int slot, const RtrCameraData* camera, const RtrSceneInfo* sceneInfo, const rtr_render_params* params, int flags);
/* n frames into n distinct slots with ONE launch of the pipeline per rank (rtr_render_batch_async: a 1/N shard of a 1-spp frame is
 * too little work per launch — one rank of eight renders a frame in 0.39 ms one launch per frame and in 0.35 ms four per launch) and
 * ONE exchange for the launch: every slot's shards in one RCCL group (rtr_mgpu_plan_batch is the list that is carried out), then a
 * de-interleave per slot.  n <= RTR_MAX_BATCH. */
int  rtr_mgpu_render_batch_async(rtr_mgpu* m, const int* slots, int n, const RtrCameraData* cameras, const RtrSceneInfo* sceneInfos,
                                 const rtr_render_params* params, int flags);
int  rtr_mgpu_wait(rtr_mgpu* m, int slot);
/* rtr_mgpu_render_async + rtr_mgpu_wait on slot 0. */
int  rtr_mgpu_render(rtr_mgpu* m, const RtrCameraData* camera, const RtrSceneInfo* sceneInfo, const rtr_render_params* params);

/* The assembled frame (height x width RGBA8, bytes B,G,R,255) of a waited-for slot; only where rank 0 is local
 * (RTR_ERR_INVALID_ARGUMENT elsewhere). */
int  rtr_mgpu_frame_download(rtr_mgpu* m, int slot, void* dst, size_t bytes);
int  rtr_mgpu_frame_device_ptr(rtr_mgpu* m, int slot, void** devicePtr, size_t* bytes);
/* Per-kernel times of a local rank's shard render of a waited-for slot (rtr_frame_get_stats of its frame). */
int  rtr_mgpu_frame_stats(rtr_mgpu* m, int slot, int localRank, rtr_frame_stats* out);
/* This rank's own shard (rtr_shard_rows x width) of a waited-for slot: localRank indexes the ranks of this process. */
int  rtr_mgpu_shard_download(rtr_mgpu* m, int slot, int localRank, void* dst, size_t bytes);

typedef struct rtr_mgpu_info {
    int nranks;          /* size of the communicator */
    int nlocal;          /* ranks driven by this process */
    int firstRank;       /* rank of local rank 0 */
    int framesInFlight;
    int selfExchange;    /* 1: a one-rank communicator still sends its shard to itself through RCCL (RTR_MGPU_SELF_EXCHANGE=1; test hook).
                          * (A second test hook, RTR_MGPU_TEST_SHARED_DEVICE=1, lets rtr_mgpu_create accept the same device for several ranks:
                          * RCCL refuses that, tests/fake_rccl/ — preloaded by the GPU tests, never part of the product — does not.) */
    int aborted;         /* 1: a rank failed after the exchange was posted or the watchdog fired; the communicators were aborted */
    int rcclVersion;     /* ncclGetVersion of the library that is actually loaded */
    int timeoutMs;
    /* host time the local ranks' threads spent carrying out plans (enqueue(): stream waits, the launch of the pipeline, RCCL calls, event
     * records) since the handle was made, for the local rank that spent most, and the frames that rank enqueued: ms / frames must stay
     * below the GPU time of a rank's shard (0.28 ms for one rank of eight on the bench frame) or the host is the limit */
    double enqueueHostMs;
    double enqueueRcclMs;              /* the part of enqueueHostMs spent inside RCCL calls (ncclGroupEnd posts the transfers; it may wait for peers) */
    unsigned long long enqueuedFrames;
} rtr_mgpu_info;
int  rtr_mgpu_get_info(const rtr_mgpu* m, rtr_mgpu_info* out);

/* ---- The exchange, as data ------------------------------------------------------------------------------------------------------
 * What one rank does for one frame slot, in stream order, as a list of operations.  rtr_mgpu_render_async() EXECUTES this list
 * (csrc/mgpu/rtr_mgpu.cpp: enqueue() is a switch over it) — it is not a description kept beside the code — so the byte offsets,
 * lengths, peers and event edges of the N > 1 exchange can be checked for every rank of every communicator size without a GPU
 * (tests/test_mgpu_plan.py), and a multi-process CPU test can carry the same list out over another transport
 * (tests/test_distributed.py runs it over gloo send / recv with the oracle's shards).  Needs no device, no communicator.
 *
 *   stream   which of the rank's two streams the operation is enqueued on
 *   peer     RECV: the rank the bytes come from; SEND: the rank they go to; RENDER: shardIndex (= rank); others -1
 *   offset   RENDER: byte offset of the render target inside buffer `buffer`;  RECV: byte offset in the gather buffer the shard
 *            lands at (= shardBytes * peer);  SEND: byte offset in the local shard (0)
 *   bytes    RENDER / RECV / SEND: shardBytes = rtr_shard_rows(height, bandRows, nranks) * width * 4;
 *            DEINTERLEAVE: width * height * 4 (the assembled frame)
 *   buffer   which buffer offset / bytes refer to
 *   event    WAIT / RECORD: which of the slot's two events */
typedef enum rtr_mgpu_op_kind {
    RTR_MGPU_OP_WAIT = 1,          /* stream waits for `event` (skipped the first time a slot is used: nothing to wait for) */
    RTR_MGPU_OP_RENDER = 2,        /* rtr_render_async of shard `peer` of nranks into `buffer` + `offset` */
    RTR_MGPU_OP_RECORD = 3,        /* record `event` on `stream` */
    RTR_MGPU_OP_GROUP_START = 4,   /* ncclGroupStart */
    RTR_MGPU_OP_RECV = 5,          /* ncclRecv(gather + offset, bytes, from peer) */
    RTR_MGPU_OP_SEND = 6,          /* ncclSend(local + offset, bytes, to peer) */
    RTR_MGPU_OP_GROUP_END = 7,     /* ncclGroupEnd */
    RTR_MGPU_OP_DEINTERLEAVE = 8   /* rank 0: gather buffer (nranks shards) -> assembled frame (k_deinterleave) */
} rtr_mgpu_op_kind;
typedef enum rtr_mgpu_stream { RTR_MGPU_STREAM_RENDER = 0, RTR_MGPU_STREAM_COMM = 1 } rtr_mgpu_stream;
typedef enum rtr_mgpu_buffer { RTR_MGPU_BUF_NONE = 0, RTR_MGPU_BUF_LOCAL = 1, RTR_MGPU_BUF_GATHER = 2, RTR_MGPU_BUF_SELF_SRC = 3, RTR_MGPU_BUF_FULL = 4 } rtr_mgpu_buffer;
typedef enum rtr_mgpu_event { RTR_MGPU_EV_NONE = 0, RTR_MGPU_EV_RENDER_DONE = 1, RTR_MGPU_EV_COMM_DONE = 2 } rtr_mgpu_event;
typedef struct rtr_mgpu_op {
    int32_t  kind;       /* rtr_mgpu_op_kind */
    int32_t  stream;     /* rtr_mgpu_stream */
    int32_t  peer;
    int32_t  buffer;     /* rtr_mgpu_buffer */
    int32_t  event;      /* rtr_mgpu_event */
    int32_t  slot;       /* which frame slot of the launch the operation is about: an index 0 .. nslots-1 into the slots the launch was given (0 in a one-frame plan) */
    uint64_t offset;
    uint64_t bytes;
} rtr_mgpu_op;
#define RTR_MGPU_PLAN_MAX_OPS 32     /* 8 + (nranks - 1) operations on rank 0; nranks <= RTR_MGPU_MAX_RANKS */
#define RTR_MGPU_MAX_RANKS 16
/* Fills ops[0 .. *numOps) for `rank` of `nranks`.  flags: RTR_MGPU_NO_EXCHANGE (render only); selfExchange != 0 with nranks == 1:
 * the one rank sends its shard to itself through the communicator (test hook).  RTR_ERR_INVALID_ARGUMENT on a bad rank / extent or
 * when maxOps is too small. */
int  rtr_mgpu_plan(int rank, int nranks, uint32_t width, uint32_t height, uint32_t bandRows, int flags, int selfExchange,
                   rtr_mgpu_op* ops, int maxOps, int* numOps);
/* The plan of a launch of `nslots` frames (rtr_mgpu_render_batch_async; nslots = 1 is rtr_mgpu_plan): the WAITs of every slot, ONE
 * RENDER (slot 0 leads), one render -> communication edge, ONE group holding every slot's transfers in slot order — on rank 0 the
 * (nranks - 1) receives of slot 0, then of slot 1 ...; on the others one send per slot — then a DEINTERLEAVE and a RECORD per slot.
 * At most RTR_MGPU_BATCH_PLAN_MAX_OPS operations. */
#define RTR_MGPU_BATCH_PLAN_MAX_OPS (6 + RTR_MAX_BATCH * (5 + RTR_MGPU_MAX_RANKS))
int  rtr_mgpu_plan_batch(int rank, int nranks, uint32_t width, uint32_t height, uint32_t bandRows, int flags, int selfExchange, int nslots,
                         rtr_mgpu_op* ops, int maxOps, int* numOps);

/* Watchdog of rtr_mgpu_wait: a slot whose exchange has not finished after this many milliseconds is given up — every local
 * communicator is aborted (ncclCommAbort releases peers blocked in a send / recv that will never be matched), the handle refuses
 * further renders and rtr_mgpu_wait returns RTR_ERR_HIP with the stage it was waiting in.  Default 120000, from the environment
 * variable RTR_MGPU_TIMEOUT_MS at creation; 0 = wait for ever. */
int  rtr_mgpu_set_timeout_ms(rtr_mgpu* m, uint32_t ms);

const char* rtr_mgpu_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RTR_MGPU_H */
