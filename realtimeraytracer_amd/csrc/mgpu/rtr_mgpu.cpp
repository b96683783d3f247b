/* rtr_mgpu.cpp — librtr_mgpu.so: the tile-sharded frame over the GPUs of one node (include/rtr_mgpu.h).
 * A pure client of include/rtr.h + RCCL + the HIP runtime: nothing here reaches into librtr_hip.so's internals. */
#include "../../../include/rtr_mgpu.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}

/* One host thread per local rank: jobs run in submission order on the rank's own thread (which keeps its device current). */
class Worker {
public:
    Worker() : th_([this] { loop(); }) {}
    ~Worker() {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; }
        cv_.notify_all();
        th_.join();
    }
    std::future<int> submit(std::function<int(std::string&)> fn) {
        auto task = std::make_shared<std::packaged_task<int()>>([fn, this] { return fn(err_); });
        std::future<int> f = task->get_future();
        { std::lock_guard<std::mutex> l(m_); q_.push_back([task] { (*task)(); }); }
        cv_.notify_one();
        return f;
    }
    const std::string& error() const { return err_; }
private:
    void loop() {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [this] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;
                job = std::move(q_.front()); q_.pop_front();
            }
            job();
        }
    }
    std::mutex m_; std::condition_variable cv_; std::deque<std::function<void()>> q_; bool stop_ = false;
    std::string err_;           /* last error of a job on this worker (read after the job's future is ready) */
    std::thread th_;
};

struct Slot {
    rtr_frame* frame = nullptr;
    uint32_t* gathered = nullptr;     /* rank 0: nranks x rows x width */
    uint32_t* full = nullptr;         /* rank 0: height x width */
    uint32_t* selfSrc = nullptr;      /* one-rank self-exchange test hook: the shard is rendered here and sent to `gathered` */
    void* local = nullptr;            /* this rank's shard (device pointer of the frame's image) */
    hipEvent_t evRender = nullptr, evComm = nullptr;
    uint32_t width = 0, height = 0, rows = 0, bandRows = 0, images = 0;
    bool commPending = false;         /* evComm has been recorded at least once */
    std::shared_future<int> pending;  /* the enqueue job of the render in flight (shared by the slots of one batch) */
    bool inFlight = false;
};

struct Rank {
    int rank = 0, device = 0;
    /* one render context (= one HIP stream) PER FRAME SLOT: the kernels of a frame are a dependency chain with tails, and a 1/N shard
     * cannot fill the GPU, so frames in flight on separate streams are where most of the strong scaling comes from */
    rtr_ctx* ctx[RTR_MGPU_MAX_SLOTS] = {};
    hipStream_t renderStream[RTR_MGPU_MAX_SLOTS] = {};
    rtr_ctx* commCtx = nullptr;       /* one more context whose stream is the communication stream (RCCL ops, k_deinterleave) */
    hipStream_t commStream = nullptr;
    rtr_scene* scene = nullptr;
    ncclComm_t comm = nullptr;        /* written when the rank is made and in destroy_rank (after its worker has joined), read by the worker in between */
    std::atomic<bool> inRccl{false};  /* the worker is inside (or about to enter) an RCCL call on `comm` */
    std::atomic<bool> commAborted{false};     /* ncclCommAbort was called on `comm`: it is gone, destroy_rank must not destroy it again */
    Slot slots[RTR_MGPU_MAX_SLOTS];
    std::unique_ptr<Worker> worker;
    /* what this rank's worker is doing right now (a string literal): the watchdog of rtr_mgpu_wait names it when it gives up, so a
     * hang says where it is */
    std::atomic<const char*> stage{"idle"};
    std::atomic<unsigned long long> enqueueNs{0}, rcclNs{0}, enqueuedFrames{0};     /* host time of enqueue() on this rank's thread, and of the RCCL calls in it (rtr_mgpu_info) */
};

#define W_HIP(expr)  do { hipError_t e_ = (expr); if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return RTR_ERR_HIP; } } while (0)
#define W_NCCL(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { err = std::string(#expr) + ": " + ncclGetErrorString(r_); return RTR_ERR_HIP; } } while (0)
#define W_RTR(expr)  do { int rc_ = (expr); if (rc_ != RTR_OK) { err = std::string(#expr) + ": " + rtr_last_error(); return rc_; } } while (0)

}  // namespace

struct rtr_mgpu {
    int nranks = 0, framesInFlight = 1;
    bool selfExchange = false;
    std::atomic<bool> aborted{false};               /* the communicators were aborted: the handle only waits for / frees things now; workers issue no further RCCL call */
    uint32_t timeoutMs = 120000;
    int planFlags = 0;                /* OR-ed into every launch's flags (RTR_MGPU_GROUP_PER_SLOT=1 in the environment at creation) */
    std::vector<std::unique_ptr<Rank>> ranks;       /* the local ones */
};

namespace {

void release_slot(Rank& r, Slot& s) {
    (void)hipSetDevice(r.device);
    if (s.frame) rtr_frame_destroy(s.frame);
    if (s.gathered) (void)hipFree(s.gathered);
    if (s.full) (void)hipFree(s.full);
    if (s.selfSrc) (void)hipFree(s.selfSrc);
    if (s.evRender) (void)hipEventDestroy(s.evRender);
    if (s.evComm) (void)hipEventDestroy(s.evComm);
    /* device resources only: `pending` / `inFlight` belong to the caller's thread (this may run on the rank's worker) */
    s.frame = nullptr; s.gathered = nullptr; s.full = nullptr; s.selfSrc = nullptr; s.local = nullptr;
    s.evRender = nullptr; s.evComm = nullptr; s.width = s.height = s.rows = s.bandRows = s.images = 0; s.commPending = false;
}

bool slot_matches(const rtr_mgpu* m, const Slot& s, const rtr_render_params& p) {
    const uint32_t band = p.bandRows ? p.bandRows : 8u;
    const uint32_t rows = rtr_shard_rows(p.height, band, (uint32_t)m->nranks);
    const uint32_t images = RTR_IMAGES_FRAMEBUFFER | (p.images & RTR_IMG_BIT(RTR_IMAGE_HDR));
    return s.frame && s.width == p.width && s.height == p.height && s.rows == rows && s.bandRows == band && s.images == images;
}

/* (re)creates the slot's frame and, on rank 0, the gather / full-frame buffers for this extent.  Everything of a frame that can
 * fail for lack of memory happens here, BEFORE any rank posts a send or a receive (rtr_mgpu_render_async joins this phase on all
 * local ranks first): a rank that fails later than that would leave its peers' receives unmatched. */
int prepare_slot(rtr_mgpu* m, Rank& r, Slot& s, const rtr_render_params& p, std::string& err) {
    if (slot_matches(m, s, p)) return RTR_OK;
    const uint32_t band = p.bandRows ? p.bandRows : 8u;
    const uint32_t rows = rtr_shard_rows(p.height, band, (uint32_t)m->nranks);
    const uint32_t images = RTR_IMAGES_FRAMEBUFFER | (p.images & RTR_IMG_BIT(RTR_IMAGE_HDR));
    r.stage = "prepare_slot";
    /* the slot's previous frame may still be running (its join only covered the host-side enqueue) */
    W_HIP(hipSetDevice(r.device));
    const int sl = (int)(&s - r.slots);
    W_HIP(hipStreamSynchronize(r.renderStream[sl]));
    W_HIP(hipStreamSynchronize(r.commStream));
    release_slot(r, s);
    W_RTR(rtr_frame_create(r.ctx[sl], p.width, rows, images, &s.frame));
    W_HIP(hipEventCreateWithFlags(&s.evRender, hipEventDisableTiming));
    W_HIP(hipEventCreateWithFlags(&s.evComm, hipEventDisableTiming));
    const size_t shardBytes = (size_t)rows * p.width * 4;
    if (r.rank == 0) {
        W_HIP(hipMalloc((void**)&s.gathered, shardBytes * (size_t)m->nranks));
        W_HIP(hipMemsetAsync(s.gathered, 0, shardBytes * (size_t)m->nranks, r.commStream));
        W_HIP(hipMalloc((void**)&s.full, (size_t)p.width * p.height * 4));
        if (m->selfExchange && m->nranks == 1) {
            W_HIP(hipMalloc((void**)&s.selfSrc, shardBytes));
            W_HIP(hipMemsetAsync(s.selfSrc, 0, shardBytes, r.commStream));
            W_RTR(rtr_frame_bind_external(s.frame, RTR_IMAGE_SHADOWED, s.selfSrc, shardBytes));
        } else {
            /* rank 0 renders straight into its place in the gather buffer: no copy of its own shard */
            W_RTR(rtr_frame_bind_external(s.frame, RTR_IMAGE_SHADOWED, s.gathered, shardBytes));
        }
        W_HIP(hipStreamSynchronize(r.commStream));
    }
    size_t bytes = 0;
    W_RTR(rtr_frame_device_ptr(s.frame, RTR_IMAGE_SHADOWED, &s.local, &bytes));
    s.width = p.width; s.height = p.height; s.rows = rows; s.bandRows = band; s.images = images;
    r.stage = "idle";
    return RTR_OK;
}

/* The plan (include/rtr_mgpu.h): what `rank` of `nranks` enqueues for one launch of `nslots` frames, in order.  ONE exchange per
 * launch: the shards of all its slots travel in one ncclGroupStart / ncclGroupEnd — on rank 0 the (N - 1) x nslots receives, on the
 * others nslots sends, slot by slot (RCCL matches the transfers between a pair of ranks in posting order) — not one group per slot:
 * a launch of sixteen shards at N = 8 is one RCCL launch on the rank that also renders and de-interleaves, not sixteen. */
#define RTR_MGPU_TEST_WRONG_PLACE 0x40000000      /* internal, test build only (RTR_MGPU_TEST_WRONG_PLACE=1 / 2 at creation) */
#define RTR_MGPU_TEST_WRONG_PLACE_ONE_GROUP 0x20000000
int make_plan(int rank, int nranks, uint32_t width, uint32_t height, uint32_t bandRows, int flags, int selfExchange, int nslots, std::vector<rtr_mgpu_op>& ops) {
    if (nranks < 1 || nranks > RTR_MGPU_MAX_RANKS || rank < 0 || rank >= nranks || width == 0 || height == 0 || nslots < 1 || nslots > RTR_MAX_BATCH) return RTR_ERR_INVALID_ARGUMENT;
    if (bandRows == 0) bandRows = 8;
    const uint64_t shardBytes = (uint64_t)rtr_shard_rows(height, bandRows, (uint32_t)nranks) * width * 4u;
    const bool self = selfExchange && nranks == 1;
    auto op = [&](int kind, int stream, int slot, int peer, int buffer, int event, uint64_t offset, uint64_t bytes) {
        rtr_mgpu_op o; memset(&o, 0, sizeof o);
        o.kind = kind; o.stream = stream; o.slot = slot; o.peer = peer; o.buffer = buffer; o.event = event; o.offset = offset; o.bytes = bytes;
        ops.push_back(o);
    };
    ops.clear();
    /* every slot's previous exchange must be done with the buffers this render overwrites */
    for (int j = 0; j < nslots; ++j) op(RTR_MGPU_OP_WAIT, RTR_MGPU_STREAM_RENDER, j, -1, RTR_MGPU_BUF_NONE, RTR_MGPU_EV_COMM_DONE, 0, 0);
    /* ONE render of all the launch's shards (slot 0 leads: its render stream is "the render stream" of the launch); rank 0 renders
     * every shard straight into its place (shard 0) of that slot's gather buffer */
    op(RTR_MGPU_OP_RENDER, RTR_MGPU_STREAM_RENDER, 0, rank, rank == 0 ? (self ? RTR_MGPU_BUF_SELF_SRC : RTR_MGPU_BUF_GATHER) : RTR_MGPU_BUF_LOCAL, RTR_MGPU_EV_NONE, 0, shardBytes);
    if (flags & RTR_MGPU_NO_EXCHANGE) return RTR_OK;
    op(RTR_MGPU_OP_RECORD, RTR_MGPU_STREAM_RENDER, 0, -1, RTR_MGPU_BUF_NONE, RTR_MGPU_EV_RENDER_DONE, 0, 0);
    op(RTR_MGPU_OP_WAIT, RTR_MGPU_STREAM_COMM, 0, -1, RTR_MGPU_BUF_NONE, RTR_MGPU_EV_RENDER_DONE, 0, 0);
    /* the one exchange step: every other rank's shards -> rank 0 (xGMI: one direct link per peer), grouped so that RCCL posts all
     * of a rank's transfers together — a send and a receive that depend on each other and sit one behind the other on a stream
     * never complete */
    if (nranks > 1 || self) {
        /* RTR_MGPU_GROUP_PER_SLOT: the fallback — one group per slot (nslots RCCL launches instead of one), should a communicator ever
         * mis-order the (N - 1) x nslots receives of one group */
        const bool perSlot = (flags & RTR_MGPU_GROUP_PER_SLOT) != 0;
        if (!perSlot) op(RTR_MGPU_OP_GROUP_START, RTR_MGPU_STREAM_COMM, 0, -1, RTR_MGPU_BUF_NONE, RTR_MGPU_EV_NONE, 0, 0);
        for (int j = 0; j < nslots; ++j) {
            if (perSlot) op(RTR_MGPU_OP_GROUP_START, RTR_MGPU_STREAM_COMM, 0, -1, RTR_MGPU_BUF_NONE, RTR_MGPU_EV_NONE, 0, 0);
            if (self) {
                op(RTR_MGPU_OP_SEND, RTR_MGPU_STREAM_COMM, j, 0, RTR_MGPU_BUF_SELF_SRC, RTR_MGPU_EV_NONE, 0, shardBytes);
                op(RTR_MGPU_OP_RECV, RTR_MGPU_STREAM_COMM, j, 0, RTR_MGPU_BUF_GATHER, RTR_MGPU_EV_NONE, 0, shardBytes);
            } else if (rank == 0) {
                for (int src = 1; src < nranks; ++src) {
                    uint64_t at = shardBytes * (uint64_t)src;
#ifdef RTR_TEST_HOOKS      /* librtr_mgpu_test.so only — the mutation bench.py's verified first exchange must catch: every shard lands in its neighbour's place */
                    if ((flags & RTR_MGPU_TEST_WRONG_PLACE) || ((flags & RTR_MGPU_TEST_WRONG_PLACE_ONE_GROUP) && !(flags & RTR_MGPU_GROUP_PER_SLOT))) at = shardBytes * (uint64_t)((src + 1) % nranks);
#endif
                    op(RTR_MGPU_OP_RECV, RTR_MGPU_STREAM_COMM, j, src, RTR_MGPU_BUF_GATHER, RTR_MGPU_EV_NONE, at, shardBytes);
                }
            } else {
                op(RTR_MGPU_OP_SEND, RTR_MGPU_STREAM_COMM, j, 0, RTR_MGPU_BUF_LOCAL, RTR_MGPU_EV_NONE, 0, shardBytes);
            }
            if (perSlot) op(RTR_MGPU_OP_GROUP_END, RTR_MGPU_STREAM_COMM, 0, -1, RTR_MGPU_BUF_NONE, RTR_MGPU_EV_NONE, 0, 0);
        }
        if (!perSlot) op(RTR_MGPU_OP_GROUP_END, RTR_MGPU_STREAM_COMM, 0, -1, RTR_MGPU_BUF_NONE, RTR_MGPU_EV_NONE, 0, 0);
    }
    if (rank == 0) for (int j = 0; j < nslots; ++j) op(RTR_MGPU_OP_DEINTERLEAVE, RTR_MGPU_STREAM_COMM, j, -1, RTR_MGPU_BUF_FULL, RTR_MGPU_EV_NONE, 0, (uint64_t)width * height * 4u);     /* one rank: a plain copy */
    for (int j = 0; j < nslots; ++j) op(RTR_MGPU_OP_RECORD, RTR_MGPU_STREAM_COMM, j, -1, RTR_MGPU_BUF_NONE, RTR_MGPU_EV_COMM_DONE, 0, 0);
    return RTR_OK;
}

struct BatchJob {                     /* what one call renders: n frames into n distinct slots with ONE launch of the pipeline per rank */
    int n = 0;
    int slots[RTR_MAX_BATCH];
    RtrCameraData cams[RTR_MAX_BATCH];
    RtrSceneInfo infos[RTR_MAX_BATCH];
    rtr_render_params p;
    int flags = 0;
};

/* Carries the launch's plan out on this rank's streams; the slots are prepared (prepare_slot) before this runs.  A plain walk over
 * the list rtr_mgpu_plan_batch returns for (rank, nranks, extent, flags, number of slots): operation by operation, each on the slot
 * its `slot` field names (an index into the launch's slots) — there is no second description of the exchange beside that list. */
int enqueue(rtr_mgpu* m, Rank& r, const BatchJob& job, std::string& err) {
    struct Clock {
        Rank& r; int frames; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        ~Clock() { r.enqueueNs += (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); r.enqueuedFrames += (unsigned long long)frames; }
    } clock{r, job.n};
    W_HIP(hipSetDevice(r.device));
    if (!r.scene) { err = "no scene: call rtr_mgpu_scene_create first"; return RTR_ERR_INVALID_ARGUMENT; }
    rtr_render_params p = job.p;
    for (int j = 0; j < job.n; ++j) if (!slot_matches(m, r.slots[job.slots[j]], p)) { err = "internal: slot not prepared for this extent"; return RTR_ERR_INVALID_ARGUMENT; }
    std::vector<rtr_mgpu_op> ops;
    if (make_plan(r.rank, m->nranks, p.width, p.height, p.bandRows, job.flags, m->selfExchange ? 1 : 0, job.n, ops) != RTR_OK) { err = "internal: no plan for this rank / extent"; return RTR_ERR_INVALID_ARGUMENT; }
    hipStream_t renderStream = r.renderStream[job.slots[0]];
    bool inGroup = false;
    /* an RCCL call is made only while the handle is not being aborted: the watchdog (abort_all, on the caller's thread) first raises
     * `aborted`, then waits for this flag to drop before it gives the communicator up — so a call never STARTS on a communicator
     * that is gone; one that is blocked inside RCCL at that moment is what ncclCommAbort is allowed to interrupt */
    struct InRccl {
        Rank& r; bool ok;
        InRccl(Rank& rr, rtr_mgpu* mm) : r(rr) { r.inRccl.store(true); ok = !mm->aborted.load(); if (!ok) r.inRccl.store(false); }
        ~InRccl() { if (ok) r.inRccl.store(false); }
    };
    for (const rtr_mgpu_op& o : ops) {
        if (o.slot < 0 || o.slot >= job.n) { err = "internal: the plan names a slot outside the launch"; return RTR_ERR_INVALID_ARGUMENT; }
        Slot& s = r.slots[job.slots[o.slot]];
        auto buffer = [&](int which) -> char* {
            switch (which) {
                case RTR_MGPU_BUF_LOCAL: return static_cast<char*>(s.local);
                case RTR_MGPU_BUF_GATHER: return reinterpret_cast<char*>(s.gathered);
                case RTR_MGPU_BUF_SELF_SRC: return reinterpret_cast<char*>(s.selfSrc);
                case RTR_MGPU_BUF_FULL: return reinterpret_cast<char*>(s.full);
                default: return nullptr;
            }
        };
        hipStream_t st = o.stream == RTR_MGPU_STREAM_RENDER ? renderStream : r.commStream;
        hipEvent_t ev = o.event == RTR_MGPU_EV_RENDER_DONE ? s.evRender : s.evComm;
        hipError_t he = hipSuccess; ncclResult_t ne = ncclSuccess;
        int c = RTR_OK;
        switch (o.kind) {
            case RTR_MGPU_OP_WAIT:
                r.stage = "hipStreamWaitEvent";
                if (o.event == RTR_MGPU_EV_COMM_DONE && !s.commPending) break;       /* first use of the slot */
                he = hipStreamWaitEvent(st, ev, 0);
                break;
            case RTR_MGPU_OP_RENDER: {              /* once per launch: every slot's shard, led by slot 0 */
                r.stage = "rtr_render_batch_async";
                rtr_frame* frames[RTR_MAX_BATCH];
                for (int j = 0; j < job.n && c == RTR_OK; ++j) {
                    Slot& sj = r.slots[job.slots[j]];
                    char* target = o.buffer == RTR_MGPU_BUF_LOCAL ? static_cast<char*>(sj.local) : (o.buffer == RTR_MGPU_BUF_GATHER ? reinterpret_cast<char*>(sj.gathered) : reinterpret_cast<char*>(sj.selfSrc));
                    if (target + o.offset != static_cast<char*>(sj.local)) { err = "internal: the frame is not bound to the buffer the plan renders into"; c = RTR_ERR_INVALID_ARGUMENT; }
                    frames[j] = sj.frame;
                }
                if (c != RTR_OK) break;
                p.shardIndex = (uint32_t)o.peer; p.shardCount = (uint32_t)m->nranks;
                p.images = s.images; p.collectStats = 0;
                if (!(s.images & RTR_IMG_BIT(RTR_IMAGE_HDR))) { p.accumulate = 0; p.accumulatedFrames = 0; }
                c = rtr_render_batch_async(r.scene, job.cams, job.infos, &p, frames, (uint32_t)job.n);
                if (c != RTR_OK) err = std::string("rtr_render_batch_async: ") + rtr_last_error();
                break;
            }
            case RTR_MGPU_OP_RECORD:
                r.stage = "hipEventRecord";
                he = hipEventRecord(ev, st);
                if (he == hipSuccess && o.event == RTR_MGPU_EV_COMM_DONE) s.commPending = true;
                break;
            case RTR_MGPU_OP_GROUP_START: case RTR_MGPU_OP_RECV: case RTR_MGPU_OP_SEND: case RTR_MGPU_OP_GROUP_END: {
                const auto t0 = std::chrono::steady_clock::now();
                InRccl guard(r, m);
                if (!guard.ok) { err = "the communicators are being aborted"; c = RTR_ERR_HIP; inGroup = false; break; }     /* (an open group dies with the communicator) */
                if (o.kind == RTR_MGPU_OP_GROUP_START) { r.stage = "ncclGroupStart"; ne = ncclGroupStart(); inGroup = ne == ncclSuccess; }
                else if (o.kind == RTR_MGPU_OP_RECV) { r.stage = "ncclRecv"; ne = ncclRecv(buffer(o.buffer) + o.offset, o.bytes, ncclUint8, o.peer, r.comm, st); }
                else if (o.kind == RTR_MGPU_OP_SEND) { r.stage = "ncclSend"; ne = ncclSend(buffer(o.buffer) + o.offset, o.bytes, ncclUint8, o.peer, r.comm, st); }
                else { r.stage = "ncclGroupEnd"; inGroup = false; ne = ncclGroupEnd(); }
                r.rcclNs += (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
                break;
            }
            case RTR_MGPU_OP_DEINTERLEAVE:
                r.stage = "rtr_deinterleave_bands";
                c = rtr_deinterleave_bands(r.commCtx, s.gathered, s.full, s.width, s.height, s.bandRows, (uint32_t)m->nranks);
                if (c != RTR_OK) err = std::string("rtr_deinterleave_bands: ") + rtr_last_error();
                break;
            default: err = "internal: unknown operation in the plan"; c = RTR_ERR_INVALID_ARGUMENT; break;
        }
        if (he != hipSuccess) { err = std::string(r.stage.load()) + ": " + hipGetErrorString(he); c = RTR_ERR_HIP; }
        if (ne != ncclSuccess) { err = std::string(r.stage.load()) + ": " + ncclGetErrorString(ne); c = RTR_ERR_HIP; }
        if (c != RTR_OK) {
            if (inGroup && !m->aborted.load()) (void)ncclGroupEnd();     /* never leave this thread inside a group */
            return c;                                                      /* the caller aborts the communicators: peers may already have posted their half */
        }
    }
    r.stage = "idle";
    return RTR_OK;
}

void destroy_rank(Rank& r) {
    r.worker.reset();                                   /* joins the thread */
    (void)hipSetDevice(r.device);
    if (r.commStream) (void)hipStreamSynchronize(r.commStream);
    for (int sl = 0; sl < RTR_MGPU_MAX_SLOTS; ++sl) if (r.renderStream[sl]) (void)hipStreamSynchronize(r.renderStream[sl]);
    for (int s = 0; s < RTR_MGPU_MAX_SLOTS; ++s) release_slot(r, r.slots[s]);
    if (r.scene) { rtr_scene_destroy(r.scene); r.scene = nullptr; }
    if (r.comm && !r.commAborted.load()) (void)ncclCommDestroy(r.comm);       /* (an aborted communicator is already gone) */
    r.comm = nullptr;
    if (r.commCtx) { rtr_ctx_destroy(r.commCtx); r.commCtx = nullptr; }
    for (int sl = 0; sl < RTR_MGPU_MAX_SLOTS; ++sl) if (r.ctx[sl]) { rtr_ctx_destroy(r.ctx[sl]); r.ctx[sl] = nullptr; }
}

/* takes ownership of `comm` whatever happens */
int make_rank(rtr_mgpu* m, int rank, int device, ncclComm_t comm) {
    std::unique_ptr<Rank> r(new Rank());
    r->rank = rank; r->device = device; r->comm = comm;
    int rc = RTR_OK;
    for (int sl = 0; sl < m->framesInFlight && rc == RTR_OK; ++sl) {
        rc = rtr_ctx_create(device, &r->ctx[sl]);
        if (rc != RTR_OK) { rc = fail(rc, "rank %d: rtr_ctx_create(%d): %s", rank, device, rtr_last_error()); break; }
        void* st = nullptr;
        if (rtr_ctx_get_stream(r->ctx[sl], &st) != RTR_OK) { rc = fail(RTR_ERR_HIP, "rank %d: rtr_ctx_get_stream: %s", rank, rtr_last_error()); break; }
        r->renderStream[sl] = (hipStream_t)st;
    }
    if (rc == RTR_OK) {
        rc = rtr_ctx_create(device, &r->commCtx);
        if (rc != RTR_OK) rc = fail(rc, "rank %d: rtr_ctx_create(%d): %s", rank, device, rtr_last_error());
    }
    if (rc == RTR_OK) {
        void* s1 = nullptr;
        if (rtr_ctx_get_stream(r->commCtx, &s1) != RTR_OK) rc = fail(RTR_ERR_HIP, "rank %d: rtr_ctx_get_stream: %s", rank, rtr_last_error());
        r->commStream = (hipStream_t)s1;
    }
    if (rc != RTR_OK) { const std::string keep = g_err; destroy_rank(*r); g_err = keep; return rc; }
    r->worker.reset(new Worker());
    m->ranks.push_back(std::move(r));
    return RTR_OK;
}

int check_slots(int framesInFlight) {
    if (framesInFlight < 1 || framesInFlight > RTR_MGPU_MAX_SLOTS) return fail(RTR_ERR_INVALID_ARGUMENT, "framesInFlight %d not in [1, %d]", framesInFlight, RTR_MGPU_MAX_SLOTS);
    return RTR_OK;
}

void read_env(rtr_mgpu* m) {
#ifdef RTR_TEST_HOOKS      /* librtr_mgpu_test.so only: a one-rank communicator still sends its shard to itself through RCCL */
    const char* se = getenv("RTR_MGPU_SELF_EXCHANGE");
    m->selfExchange = se && se[0] == '1';
#endif
    if (const char* t = getenv("RTR_MGPU_TIMEOUT_MS")) m->timeoutMs = (uint32_t)strtoul(t, nullptr, 10);
    if (const char* g = getenv("RTR_MGPU_GROUP_PER_SLOT")) if (g[0] == '1') m->planFlags |= RTR_MGPU_GROUP_PER_SLOT;
#ifdef RTR_TEST_HOOKS
    if (const char* g = getenv("RTR_MGPU_TEST_WRONG_PLACE")) {      /* 1: always; 2: only while the launch's exchange is ONE group (what bench.py's fallback to a group per slot must get around) */
        if (g[0] == '1') m->planFlags |= RTR_MGPU_TEST_WRONG_PLACE;
        if (g[0] == '2') m->planFlags |= RTR_MGPU_TEST_WRONG_PLACE_ONE_GROUP;
    }
#endif
}

/* Gives the communicators up: peers blocked in a send / receive that will never be matched are released.  Cooperative with the
 * ranks' workers: `aborted` is raised first, so a worker makes no further RCCL call (enqueue() checks it before each one); then, per
 * rank, this waits until the worker is outside RCCL — or, after a grace period, concludes it is blocked inside a call, which is the
 * one situation ncclCommAbort may be called in from another thread.  The communicator pointer stays as it is (the worker reads it);
 * destroy_rank, after the worker has joined, knows from commAborted that there is nothing left to destroy. */
void abort_all(rtr_mgpu* m) {
    if (m->aborted.exchange(true)) return;
    for (auto& rp : m->ranks) {
        if (!rp->comm || rp->commAborted.load()) continue;
        const auto until = std::chrono::steady_clock::now() + std::chrono::milliseconds(200);
        while (rp->inRccl.load() && std::chrono::steady_clock::now() < until) std::this_thread::sleep_for(std::chrono::microseconds(200));
        rp->commAborted.store(true);
        (void)ncclCommAbort(rp->comm);
    }
}

std::string stages(const rtr_mgpu* m) {
    std::string out;
    for (auto& rp : m->ranks) { if (!out.empty()) out += ", "; out += "rank " + std::to_string(rp->rank) + ": " + rp->stage.load(); }
    return out;
}

/* Joins a host-side job with the watchdog.  false = it did not come back in time (the communicators are aborted). */
bool join_job(rtr_mgpu* m, std::shared_future<int>& f, int* rc) {
    if (m->timeoutMs == 0) { *rc = f.get(); return true; }
    if (f.wait_for(std::chrono::milliseconds(m->timeoutMs)) != std::future_status::ready) {
        const std::string where = stages(m);
        abort_all(m);                                  /* releases a worker blocked inside RCCL */
        (void)f.wait_for(std::chrono::milliseconds(5000));
        *rc = fail(RTR_ERR_HIP, "watchdog: a rank's enqueue did not return within %u ms (%s); communicators aborted", m->timeoutMs, where.c_str());
        return false;
    }
    *rc = f.get();
    return true;
}

}  // namespace

extern "C" {

const char* rtr_mgpu_last_error(void) { return g_err.c_str(); }

int rtr_mgpu_unique_id(void* id) {
    if (!id) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_unique_id: null");
    static_assert(sizeof(ncclUniqueId) == RTR_MGPU_ID_BYTES, "RTR_MGPU_ID_BYTES must be sizeof(ncclUniqueId)");
    ncclUniqueId u;
    ncclResult_t r = ncclGetUniqueId(&u);
    if (r != ncclSuccess) return fail(RTR_ERR_HIP, "ncclGetUniqueId: %s", ncclGetErrorString(r));
    memcpy(id, &u, sizeof u);
    return RTR_OK;
}

int rtr_mgpu_plan(int rank, int nranks, uint32_t width, uint32_t height, uint32_t bandRows, int flags, int selfExchange,
                  rtr_mgpu_op* ops, int maxOps, int* numOps) {
    if (!ops || !numOps || maxOps < 0) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_plan: null argument");
    return rtr_mgpu_plan_batch(rank, nranks, width, height, bandRows, flags, selfExchange, 1, ops, maxOps, numOps);
}

int rtr_mgpu_plan_batch(int rank, int nranks, uint32_t width, uint32_t height, uint32_t bandRows, int flags, int selfExchange, int nslots,
                        rtr_mgpu_op* ops, int maxOps, int* numOps) {
    if (!ops || !numOps || maxOps < 0) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_plan: null argument");
    std::vector<rtr_mgpu_op> v;
    if (make_plan(rank, nranks, width, height, bandRows, flags, selfExchange, nslots, v) != RTR_OK)
        return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_plan: rank %d of %d (at most %d), frame %ux%u, %d slots (1 to %d)", rank, nranks, RTR_MGPU_MAX_RANKS, width, height, nslots, RTR_MAX_BATCH);
    if ((int)v.size() > maxOps) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_plan: %zu operations, room for %d", v.size(), maxOps);
    memcpy(ops, v.data(), v.size() * sizeof(rtr_mgpu_op));
    *numOps = (int)v.size();
    return RTR_OK;
}

int rtr_mgpu_create(const int* devices, int n, int framesInFlight, rtr_mgpu** out) {
    if (!devices || n < 1 || !out) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_create: bad argument");
    *out = nullptr;
    int rc = check_slots(framesInFlight);
    if (rc != RTR_OK) return rc;
    if (n > RTR_MGPU_MAX_RANKS) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_create: %d devices requested, at most %d ranks", n, RTR_MGPU_MAX_RANKS);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(RTR_ERR_NO_DEVICE, "rtr_mgpu_create: no HIP device; this library has no CPU fallback");
#ifdef RTR_TEST_HOOKS      /* librtr_mgpu_test.so only: ranks may share a device (RCCL itself refuses that; tests/fake_rccl/ stands in for it on a
                            * one-GPU box so that the N > 1 code of this file runs there) */
    const char* shared = getenv("RTR_MGPU_TEST_SHARED_DEVICE");
    const bool sharedOk = shared && shared[0] == '1' && !shared[1];
#else
    const bool sharedOk = false;
#endif
    if (n > count && !sharedOk) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_create: %d devices requested, %d present", n, count);
    for (int i = 0; i < n; ++i) {
        if (devices[i] < 0 || devices[i] >= count) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_create: device %d not in [0,%d)", devices[i], count);
        for (int j = 0; j < i && !sharedOk; ++j) if (devices[j] == devices[i]) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_create: device %d listed twice (one rank per GPU)", devices[i]);
    }
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    ncclResult_t r = ncclCommInitAll(comms.data(), n, devices);
    if (r != ncclSuccess) return fail(RTR_ERR_HIP, "ncclCommInitAll(%d devices): %s", n, ncclGetErrorString(r));
    rtr_mgpu* m = new rtr_mgpu();
    m->nranks = n; m->framesInFlight = framesInFlight;
    read_env(m);
    for (int i = 0; i < n; ++i) {
        rc = make_rank(m, i, devices[i], comms[(size_t)i]);          /* owns comms[i] from here, also when it fails */
        if (rc != RTR_OK) {
            const std::string keep = g_err;
            for (int j = i + 1; j < n; ++j) (void)ncclCommDestroy(comms[(size_t)j]);
            rtr_mgpu_destroy(m);
            g_err = keep;
            return rc;
        }
    }
    *out = m;
    return RTR_OK;
}

int rtr_mgpu_create_rank(int device, int rank, int nranks, const void* id, int framesInFlight, rtr_mgpu** out) {
    if (!id || !out || nranks < 1 || nranks > RTR_MGPU_MAX_RANKS || rank < 0 || rank >= nranks) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_create_rank: bad argument");
    *out = nullptr;
    int rc = check_slots(framesInFlight);
    if (rc != RTR_OK) return rc;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(RTR_ERR_NO_DEVICE, "rtr_mgpu_create_rank: no HIP device; this library has no CPU fallback");
    if (device < 0 || device >= count) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_create_rank: device %d not in [0,%d)", device, count);
    if (hipSetDevice(device) != hipSuccess) return fail(RTR_ERR_HIP, "hipSetDevice(%d) failed", device);
    ncclUniqueId u; memcpy(&u, id, sizeof u);
    ncclComm_t comm = nullptr;
    ncclResult_t r = ncclCommInitRank(&comm, nranks, u, rank);
    if (r != ncclSuccess) return fail(RTR_ERR_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, nranks, ncclGetErrorString(r));
    rtr_mgpu* m = new rtr_mgpu();
    m->nranks = nranks; m->framesInFlight = framesInFlight;
    read_env(m);
    rc = make_rank(m, rank, device, comm);
    if (rc != RTR_OK) { const std::string keep = g_err; rtr_mgpu_destroy(m); g_err = keep; return rc; }
    *out = m;
    return RTR_OK;
}

void rtr_mgpu_destroy(rtr_mgpu* m) {
    if (!m) return;
    for (auto& rp : m->ranks)
        for (int s = 0; s < RTR_MGPU_MAX_SLOTS; ++s)
            if (rp->slots[s].inFlight) { int rc = 0; (void)join_job(m, rp->slots[s].pending, &rc); rp->slots[s].inFlight = false; }
    for (auto& rp : m->ranks) destroy_rank(*rp);
    delete m;
}

int rtr_mgpu_set_timeout_ms(rtr_mgpu* m, uint32_t ms) {
    if (!m) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_set_timeout_ms: null");
    m->timeoutMs = ms;
    return RTR_OK;
}

int rtr_mgpu_get_info(const rtr_mgpu* m, rtr_mgpu_info* out) {
    if (!m || !out) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_get_info: null argument");
    memset(out, 0, sizeof *out);
    out->nranks = m->nranks; out->nlocal = (int)m->ranks.size(); out->firstRank = m->ranks.empty() ? 0 : m->ranks[0]->rank;
    out->framesInFlight = m->framesInFlight; out->selfExchange = m->selfExchange ? 1 : 0;
    out->aborted = m->aborted.load() ? 1 : 0; out->timeoutMs = (int)m->timeoutMs;
    int v = 0; if (ncclGetVersion(&v) == ncclSuccess) out->rcclVersion = v;
    for (auto& rp : m->ranks) {
        const double ms = (double)rp->enqueueNs.load() * 1e-6;
        if (ms >= out->enqueueHostMs) { out->enqueueHostMs = ms; out->enqueueRcclMs = (double)rp->rcclNs.load() * 1e-6; out->enqueuedFrames = rp->enqueuedFrames.load(); }
    }
    return RTR_OK;
}

int rtr_mgpu_scene_create(rtr_mgpu* m, const rtr_scene_desc* desc) {
    if (!m || !desc) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_scene_create: null argument");
    if (m->aborted) return fail(RTR_ERR_HIP, "rtr_mgpu_scene_create: the communicators were aborted; destroy the handle");
    /* the tree is built ONCE, by the first local rank; the others upload that build (rtr_scene_create_like) side by side, each on
     * its own thread and device */
    Rank* first = m->ranks[0].get();
    int rc = RTR_OK;
    {
        std::future<int> f = first->worker->submit([first, desc](std::string& err) -> int {
            first->stage = "rtr_scene_create";
            W_HIP(hipSetDevice(first->device));
            if (first->scene) { rtr_scene_destroy(first->scene); first->scene = nullptr; }
            W_RTR(rtr_scene_create(first->ctx[0], desc, &first->scene));       /* frames of the other slots' contexts render it too (read-only) */
            first->stage = "idle";
            return RTR_OK;
        });
        const int c = f.get();
        if (c != RTR_OK) return fail(c, "rank %d: %s", first->rank, first->worker->error().c_str());
    }
    std::vector<std::future<int>> fs;
    for (size_t i = 1; i < m->ranks.size(); ++i) {
        Rank* r = m->ranks[i].get();
        const rtr_scene* built = first->scene;
        fs.push_back(r->worker->submit([r, desc, built](std::string& err) -> int {
            r->stage = "rtr_scene_create_like";
            W_HIP(hipSetDevice(r->device));
            if (r->scene) { rtr_scene_destroy(r->scene); r->scene = nullptr; }
            W_RTR(rtr_scene_create_like(r->ctx[0], desc, built, &r->scene));
            r->stage = "idle";
            return RTR_OK;
        }));
    }
    size_t i = 1;
    for (auto& f : fs) { const int c = f.get(); if (c != RTR_OK && rc == RTR_OK) rc = fail(c, "rank %d: %s", m->ranks[i]->rank, m->ranks[i]->worker->error().c_str()); ++i; }
    return rc;
}

int rtr_mgpu_render_batch_async(rtr_mgpu* m, const int* slots, int n, const RtrCameraData* cams, const RtrSceneInfo* infos, const rtr_render_params* p, int flags) {
    if (!m || !slots || !cams || !infos || !p) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_render_async: null argument");
    if (m->aborted) return fail(RTR_ERR_HIP, "rtr_mgpu_render_async: the communicators were aborted after an earlier failure; destroy the handle");
    if (n < 1 || n > RTR_MAX_BATCH) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_render_batch_async: %d frames, 1 to %d per launch", n, RTR_MAX_BATCH);
    for (int j = 0; j < n; ++j) {
        if (slots[j] < 0 || slots[j] >= m->framesInFlight) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_render_async: slot %d not in [0,%d)", slots[j], m->framesInFlight);
        for (int k = 0; k < j; ++k) if (slots[k] == slots[j]) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_render_batch_async: slot %d given twice", slots[j]);
    }
    if (p->width == 0 || p->height == 0) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_render_async: empty frame");
    if (p->images & ~(RTR_IMAGES_FRAMEBUFFER | RTR_IMG_BIT(RTR_IMAGE_HDR))) return fail(RTR_ERR_UNSUPPORTED, "rtr_mgpu_render_async: only the RGBA8 framebuffer (RTR_IMAGE_SHADOWED) is gathered; RTR_IMAGE_HDR may be added for accumulation");
    if (p->accumulate && !(p->images & RTR_IMG_BIT(RTR_IMAGE_HDR))) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_render_async: accumulate needs RTR_IMAGE_HDR in params->images");
    for (auto& rp : m->ranks) if (!rp->scene) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_render_async: no scene: call rtr_mgpu_scene_create first");
    /* consecutive calls on a slot are ordered by its render stream; only an enqueue still running on a worker forbids the next call */
    for (int j = 0; j < n; ++j)
        for (auto& rp : m->ranks) if (rp->slots[slots[j]].inFlight) {
            int c = RTR_OK;
            const bool back = join_job(m, rp->slots[slots[j]].pending, &c);
            rp->slots[slots[j]].inFlight = false;
            if (!back) return c;
            if (c != RTR_OK) { const std::string msg = rp->worker->error(); abort_all(m); return fail(c, "rank %d: %s (communicators aborted)", rp->rank, msg.c_str()); }
        }
    /* Phase 1 — everything that can fail for lack of memory, on every local rank, joined BEFORE any rank posts a transfer: a rank
     * that dropped out after its peers had posted theirs would leave them waiting for ever.  In steady state (the slots already have
     * this extent) this is a comparison on the caller's thread. */
    bool prepared = true;
    for (int j = 0; j < n; ++j) for (auto& rp : m->ranks) prepared = prepared && slot_matches(m, rp->slots[slots[j]], *p);
    if (!prepared) {
        std::vector<std::future<int>> fs;
        std::vector<int> sl(slots, slots + n);
        for (auto& rp : m->ranks) {
            Rank* r = rp.get(); const rtr_render_params pp = *p;
            fs.push_back(r->worker->submit([m, r, sl, pp](std::string& err) -> int {
                for (int s1 : sl) { const int c = prepare_slot(m, *r, r->slots[s1], pp, err); if (c != RTR_OK) return c; }
                return RTR_OK;
            }));
        }
        int rc = RTR_OK; size_t i = 0;
        for (auto& f : fs) { const int c = f.get(); if (c != RTR_OK && rc == RTR_OK) rc = fail(c, "rank %d: %s", m->ranks[i]->rank, m->ranks[i]->worker->error().c_str()); ++i; }
        if (rc != RTR_OK) return rc;            /* nothing was posted: the handle stays usable */
    }
    /* Phase 2 — the plan, rank by rank, each on its own thread */
    BatchJob job;
    job.n = n; job.p = *p; job.flags = flags | m->planFlags;
    for (int j = 0; j < n; ++j) { job.slots[j] = slots[j]; job.cams[j] = cams[j]; job.infos[j] = infos[j]; }
    for (auto& rp : m->ranks) {
        Rank* r = rp.get();
        std::shared_future<int> f = r->worker->submit([m, r, job](std::string& err) -> int { return enqueue(m, *r, job, err); }).share();
        for (int j = 0; j < n; ++j) { r->slots[slots[j]].pending = f; r->slots[slots[j]].inFlight = true; }
    }
    return RTR_OK;
}

int rtr_mgpu_render_async(rtr_mgpu* m, int slot, const RtrCameraData* cam, const RtrSceneInfo* info, const rtr_render_params* p, int flags) {
    return rtr_mgpu_render_batch_async(m, &slot, 1, cam, info, p, flags);
}

int rtr_mgpu_wait(rtr_mgpu* m, int slot) {
    if (!m) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_wait: null");
    if (slot < 0 || slot >= m->framesInFlight) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_wait: slot %d not in [0,%d)", slot, m->framesInFlight);
    int rc = RTR_OK;
    /* first every local rank's host-side enqueue ... */
    std::vector<Rank*> waitFor;
    for (auto& rp : m->ranks) {
        Rank& r = *rp; Slot& s = r.slots[slot];
        if (!s.inFlight) continue;
        int c = RTR_OK;
        const bool back = join_job(m, s.pending, &c);
        s.inFlight = false;
        if (!back) return c;
        if (c != RTR_OK) { if (rc == RTR_OK) rc = fail(c, "rank %d: %s", r.rank, r.worker->error().c_str()); continue; }
        waitFor.push_back(&r);
    }
    if (rc != RTR_OK) { const std::string keep = g_err; abort_all(m); g_err = keep + " (communicators aborted)"; return rc; }     /* a rank failed after phase 1: its peers' transfers can never be matched */
    /* ... then the device side, under the watchdog */
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(m->timeoutMs);
    std::vector<Rank*> done;
    for (Rank* rp : waitFor) {
        Rank& r = *rp; Slot& s = r.slots[slot];
        if (hipSetDevice(r.device) != hipSuccess) { if (rc == RTR_OK) rc = fail(RTR_ERR_HIP, "rank %d: hipSetDevice failed", r.rank); continue; }
        if (s.commPending) {
            r.stage = "waiting for the exchange (evComm)";
            hipError_t e = hipErrorNotReady;
            if (m->timeoutMs == 0 || m->aborted) e = hipEventSynchronize(s.evComm);
            else {
                /* the first ~50 us a plain spin (a shard of eight is done in a third of a millisecond), then short sleeps that grow to
                 * 200 us: eight ranks' callers waiting must not hold eight cores against the ranks' enqueue threads */
                unsigned napUs = 10;
                for (unsigned spin = 0;; ++spin) {
                    e = hipEventQuery(s.evComm);
                    if (e != hipErrorNotReady) break;
                    if (spin > 2000u) {
                        if (std::chrono::steady_clock::now() > deadline) break;
                        std::this_thread::sleep_for(std::chrono::microseconds(napUs));
                        if (napUs < 200u) napUs += napUs / 2u;
                    }
                }
            }
            if (e == hipErrorNotReady) {
                const std::string where = stages(m);
                abort_all(m);
                return fail(RTR_ERR_HIP, "watchdog: slot %d's exchange did not finish within %u ms (%s); communicators aborted", slot, m->timeoutMs, where.c_str());
            }
            if (e != hipSuccess) { if (rc == RTR_OK) rc = fail(RTR_ERR_HIP, "rank %d: waiting for the exchange failed: %s", r.rank, hipGetErrorString(e)); continue; }
        }
        r.stage = "idle";
        done.push_back(&r);
    }
    /* the renders are long done (or were not exchanged): collecting their per-kernel times is a stream join and two small copies per
     * rank — side by side on the ranks' own threads, not one after the other on the caller's (eight ranks, 0.4-ms frames) */
    if (done.size() == 1) {          /* one local rank (a process per GPU): no thread hop */
        Rank& r = *done[0];
        if (hipSetDevice(r.device) != hipSuccess || rtr_frame_wait(r.slots[slot].frame) != RTR_OK) { if (rc == RTR_OK) rc = fail(RTR_ERR_HIP, "rank %d: %s", r.rank, rtr_last_error()); }
        return rc;
    }
    std::vector<std::future<int>> fs;
    for (Rank* rp : done) {
        Rank* r = rp; rtr_frame* fr = r->slots[slot].frame;
        fs.push_back(r->worker->submit([r, fr](std::string& err) -> int {
            W_HIP(hipSetDevice(r->device));
            W_RTR(rtr_frame_wait(fr));
            return RTR_OK;
        }));
    }
    for (size_t i = 0; i < fs.size(); ++i) { const int c = fs[i].get(); if (c != RTR_OK && rc == RTR_OK) rc = fail(c, "rank %d: %s", done[i]->rank, done[i]->worker->error().c_str()); }
    return rc;
}

int rtr_mgpu_render(rtr_mgpu* m, const RtrCameraData* cam, const RtrSceneInfo* info, const rtr_render_params* p) {
    int rc = rtr_mgpu_render_async(m, 0, cam, info, p, 0);
    if (rc != RTR_OK) return rc;
    return rtr_mgpu_wait(m, 0);
}

static Rank* rank0_of(rtr_mgpu* m) {
    for (auto& rp : m->ranks) if (rp->rank == 0) return rp.get();
    return nullptr;
}

int rtr_mgpu_frame_device_ptr(rtr_mgpu* m, int slot, void** ptr, size_t* bytes) {
    if (!m || !ptr) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_frame_device_ptr: null argument");
    if (slot < 0 || slot >= m->framesInFlight) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_frame_device_ptr: slot %d not in [0,%d)", slot, m->framesInFlight);
    Rank* r = rank0_of(m);
    if (!r) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_frame_device_ptr: rank 0 is not in this process");
    Slot& s = r->slots[slot];
    /* a worker may be re-creating the slot's buffers right now */
    if (s.inFlight) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_frame_device_ptr: slot %d is in flight; rtr_mgpu_wait it first", slot);
    if (!s.full) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_frame_device_ptr: slot %d has not been rendered", slot);
    *ptr = s.full; if (bytes) *bytes = (size_t)s.width * s.height * 4;
    return RTR_OK;
}

int rtr_mgpu_frame_download(rtr_mgpu* m, int slot, void* dst, size_t bytes) {
    if (!dst) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_frame_download: null destination");
    void* src = nullptr; size_t need = 0;
    int rc = rtr_mgpu_frame_device_ptr(m, slot, &src, &need);
    if (rc != RTR_OK) return rc;
    if (bytes != need) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_frame_download: %zu bytes given, the frame is %zu", bytes, need);
    Rank* r = rank0_of(m);
    if (hipSetDevice(r->device) != hipSuccess || hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail(RTR_ERR_HIP, "rtr_mgpu_frame_download: copy failed");
    return RTR_OK;
}

int rtr_mgpu_frame_stats(rtr_mgpu* m, int slot, int localRank, rtr_frame_stats* out) {
    if (!m || !out) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_frame_stats: null argument");
    if (slot < 0 || slot >= m->framesInFlight || localRank < 0 || localRank >= (int)m->ranks.size()) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_frame_stats: bad slot / rank");
    Slot& s = m->ranks[(size_t)localRank]->slots[slot];
    if (!s.frame || s.inFlight) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_frame_stats: slot %d not rendered or still in flight", slot);
    if (rtr_frame_get_stats(s.frame, out) != RTR_OK) return fail(RTR_ERR_HIP, "rtr_mgpu_frame_stats: %s", rtr_last_error());
    return RTR_OK;
}

int rtr_mgpu_shard_download(rtr_mgpu* m, int slot, int localRank, void* dst, size_t bytes) {
    if (!m || !dst) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_shard_download: null argument");
    if (slot < 0 || slot >= m->framesInFlight || localRank < 0 || localRank >= (int)m->ranks.size()) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_shard_download: bad slot / rank");
    Rank& r = *m->ranks[(size_t)localRank]; Slot& s = r.slots[slot];
    if (!s.frame || s.inFlight) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_shard_download: slot %d not rendered or still in flight", slot);
    const size_t need = (size_t)s.rows * s.width * 4;
    if (bytes != need) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_shard_download: %zu bytes given, the shard is %zu", bytes, need);
    if (hipSetDevice(r.device) != hipSuccess || hipMemcpy(dst, s.local, bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail(RTR_ERR_HIP, "rtr_mgpu_shard_download: copy failed");
    return RTR_OK;
}

}  // extern "C"
