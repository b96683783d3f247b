/* rtr_mgpu.cpp — librtr_mgpu.so: the tile-sharded frame over the GPUs of one node (include/rtr_mgpu.h).
 * A pure client of include/rtr.h + RCCL + the HIP runtime: nothing here reaches into librtr_hip.so's internals. */
#include "../../../include/rtr_mgpu.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

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
    std::future<int> pending;         /* the enqueue job of the render in flight */
    bool inFlight = false;
};

struct Rank {
    int rank = 0, device = 0;
    /* one render context (= one HIP stream) PER FRAME SLOT: the kernels of a frame are a dependency chain with tails, and a 1/N shard
     * cannot fill the GPU, so frames in flight on separate streams are where most of the strong scaling comes from */
    rtr_ctx* ctx[RTR_MGPU_MAX_SLOTS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipStream_t renderStream[RTR_MGPU_MAX_SLOTS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    rtr_ctx* commCtx = nullptr;       /* one more context whose stream is the communication stream (RCCL ops, k_deinterleave) */
    hipStream_t commStream = nullptr;
    rtr_scene* scene = nullptr;
    ncclComm_t comm = nullptr;
    Slot slots[RTR_MGPU_MAX_SLOTS];
    std::unique_ptr<Worker> worker;
};

#define W_HIP(expr)  do { hipError_t e_ = (expr); if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return RTR_ERR_HIP; } } while (0)
#define W_NCCL(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { err = std::string(#expr) + ": " + ncclGetErrorString(r_); return RTR_ERR_HIP; } } while (0)
#define W_RTR(expr)  do { int rc_ = (expr); if (rc_ != RTR_OK) { err = std::string(#expr) + ": " + rtr_last_error(); return rc_; } } while (0)

}  // namespace

struct rtr_mgpu {
    int nranks = 0, framesInFlight = 1;
    bool selfExchange = false;
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

/* (re)creates the slot's frame and, on rank 0, the gather / full-frame buffers for this extent */
int prepare_slot(rtr_mgpu* m, Rank& r, Slot& s, const rtr_render_params& p, std::string& err) {
    const uint32_t band = p.bandRows ? p.bandRows : 8u;
    const uint32_t rows = rtr_shard_rows(p.height, band, (uint32_t)m->nranks);
    const uint32_t images = RTR_IMAGES_FRAMEBUFFER | (p.images & RTR_IMG_BIT(RTR_IMAGE_HDR));
    if (s.frame && s.width == p.width && s.height == p.height && s.rows == rows && s.bandRows == band && s.images == images) return RTR_OK;
    release_slot(r, s);
    W_HIP(hipSetDevice(r.device));
    W_RTR(rtr_frame_create(r.ctx[&s - r.slots], p.width, rows, images, &s.frame));
    W_HIP(hipEventCreateWithFlags(&s.evRender, hipEventDisableTiming));
    W_HIP(hipEventCreateWithFlags(&s.evComm, hipEventDisableTiming));
    const size_t shardBytes = (size_t)rows * p.width * 4;
    if (r.rank == 0) {
        W_HIP(hipMalloc((void**)&s.gathered, shardBytes * (size_t)m->nranks));
        W_HIP(hipMemset(s.gathered, 0, shardBytes * (size_t)m->nranks));
        W_HIP(hipMalloc((void**)&s.full, (size_t)p.width * p.height * 4));
        if (m->selfExchange && m->nranks == 1) {
            W_HIP(hipMalloc((void**)&s.selfSrc, shardBytes));
            W_HIP(hipMemset(s.selfSrc, 0, shardBytes));
            W_RTR(rtr_frame_bind_external(s.frame, RTR_IMAGE_SHADOWED, s.selfSrc, shardBytes));
        } else {
            /* rank 0 renders straight into its place in the gather buffer: no copy of its own shard */
            W_RTR(rtr_frame_bind_external(s.frame, RTR_IMAGE_SHADOWED, s.gathered, shardBytes));
        }
    }
    size_t bytes = 0;
    W_RTR(rtr_frame_device_ptr(s.frame, RTR_IMAGE_SHADOWED, &s.local, &bytes));
    s.width = p.width; s.height = p.height; s.rows = rows; s.bandRows = band; s.images = images;
    return RTR_OK;
}

int enqueue(rtr_mgpu* m, Rank& r, int slot, RtrCameraData cam, RtrSceneInfo info, rtr_render_params p, int flags, std::string& err) {
    W_HIP(hipSetDevice(r.device));
    if (!r.scene) { err = "no scene: call rtr_mgpu_scene_create first"; return RTR_ERR_INVALID_ARGUMENT; }
    Slot& s = r.slots[slot];
    int rc = prepare_slot(m, r, s, p, err);
    if (rc != RTR_OK) return rc;
    /* the previous exchange of this slot must have finished with the buffers the new render overwrites */
    hipStream_t renderStream = r.renderStream[slot];
    if (s.commPending) W_HIP(hipStreamWaitEvent(renderStream, s.evComm, 0));
    p.shardIndex = (uint32_t)r.rank; p.shardCount = (uint32_t)m->nranks;
    p.images = s.images; p.collectStats = 0;
    if (!(s.images & RTR_IMG_BIT(RTR_IMAGE_HDR))) { p.accumulate = 0; p.accumulatedFrames = 0; }
    W_RTR(rtr_render_async(r.scene, &cam, &info, &p, s.frame));
    if (flags & RTR_MGPU_NO_EXCHANGE) return RTR_OK;          /* an accumulation step: the exchange follows the last frame of the sum */
    W_HIP(hipEventRecord(s.evRender, renderStream));
    W_HIP(hipStreamWaitEvent(r.commStream, s.evRender, 0));
    const size_t shardBytes = (size_t)s.rows * s.width * 4;
    /* the one exchange step: every other rank's shard -> rank 0, on the communication stream (xGMI: one direct link per peer) */
    if (m->nranks > 1) {
        W_NCCL(ncclGroupStart());
        if (r.rank == 0) {
            for (int src = 1; src < m->nranks; ++src)
                W_NCCL(ncclRecv(reinterpret_cast<char*>(s.gathered) + shardBytes * (size_t)src, shardBytes, ncclUint8, src, r.comm, r.commStream));
        } else {
            W_NCCL(ncclSend(s.local, shardBytes, ncclUint8, 0, r.comm, r.commStream));
        }
        W_NCCL(ncclGroupEnd());
    } else if (m->selfExchange) {
        W_NCCL(ncclGroupStart());
        W_NCCL(ncclSend(s.selfSrc, shardBytes, ncclUint8, 0, r.comm, r.commStream));
        W_NCCL(ncclRecv(s.gathered, shardBytes, ncclUint8, 0, r.comm, r.commStream));
        W_NCCL(ncclGroupEnd());
    }
    if (r.rank == 0) W_RTR(rtr_deinterleave_bands(r.commCtx, s.gathered, s.full, s.width, s.height, s.bandRows, (uint32_t)m->nranks));   /* one rank: a plain copy */
    W_HIP(hipEventRecord(s.evComm, r.commStream));
    s.commPending = true;
    return RTR_OK;
}

int make_rank(rtr_mgpu* m, int rank, int device, ncclComm_t comm) {
    std::unique_ptr<Rank> r(new Rank());
    r->rank = rank; r->device = device; r->comm = comm;
    int rc = RTR_OK;
    for (int sl = 0; sl < m->framesInFlight; ++sl) {
        rc = rtr_ctx_create(device, &r->ctx[sl]);
        if (rc != RTR_OK) return fail(rc, "rank %d: rtr_ctx_create(%d): %s", rank, device, rtr_last_error());
        void* st = nullptr;
        if (rtr_ctx_get_stream(r->ctx[sl], &st) != RTR_OK) return fail(RTR_ERR_HIP, "rank %d: rtr_ctx_get_stream: %s", rank, rtr_last_error());
        r->renderStream[sl] = (hipStream_t)st;
    }
    rc = rtr_ctx_create(device, &r->commCtx);
    if (rc != RTR_OK) return fail(rc, "rank %d: rtr_ctx_create(%d): %s", rank, device, rtr_last_error());
    void* s1 = nullptr;
    if (rtr_ctx_get_stream(r->commCtx, &s1) != RTR_OK) return fail(RTR_ERR_HIP, "rank %d: rtr_ctx_get_stream: %s", rank, rtr_last_error());
    r->commStream = (hipStream_t)s1;
    r->worker.reset(new Worker());
    m->ranks.push_back(std::move(r));
    return RTR_OK;
}

int check_slots(int framesInFlight) {
    if (framesInFlight < 1 || framesInFlight > RTR_MGPU_MAX_SLOTS) return fail(RTR_ERR_INVALID_ARGUMENT, "framesInFlight %d not in [1, %d]", framesInFlight, RTR_MGPU_MAX_SLOTS);
    return RTR_OK;
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

int rtr_mgpu_create(const int* devices, int n, int framesInFlight, rtr_mgpu** out) {
    if (!devices || n < 1 || !out) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_create: bad argument");
    *out = nullptr;
    int rc = check_slots(framesInFlight);
    if (rc != RTR_OK) return rc;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(RTR_ERR_NO_DEVICE, "rtr_mgpu_create: no HIP device; this library has no CPU fallback");
    for (int i = 0; i < n; ++i) {
        if (devices[i] < 0 || devices[i] >= count) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_create: device %d not in [0,%d)", devices[i], count);
        for (int j = 0; j < i; ++j) if (devices[j] == devices[i]) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_create: device %d listed twice (one rank per GPU)", devices[i]);
    }
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    ncclResult_t r = ncclCommInitAll(comms.data(), n, devices);
    if (r != ncclSuccess) return fail(RTR_ERR_HIP, "ncclCommInitAll(%d devices): %s", n, ncclGetErrorString(r));
    rtr_mgpu* m = new rtr_mgpu();
    m->nranks = n; m->framesInFlight = framesInFlight;
    const char* se = getenv("RTR_MGPU_SELF_EXCHANGE");
    m->selfExchange = se && se[0] == '1';
    for (int i = 0; i < n; ++i) {
        rc = make_rank(m, i, devices[i], comms[(size_t)i]);
        if (rc != RTR_OK) { for (int j = i + 1; j < n; ++j) (void)ncclCommDestroy(comms[(size_t)j]); rtr_mgpu_destroy(m); return rc; }
    }
    *out = m;
    return RTR_OK;
}

int rtr_mgpu_create_rank(int device, int rank, int nranks, const void* id, int framesInFlight, rtr_mgpu** out) {
    if (!id || !out || nranks < 1 || rank < 0 || rank >= nranks) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_create_rank: bad argument");
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
    const char* se = getenv("RTR_MGPU_SELF_EXCHANGE");
    m->selfExchange = se && se[0] == '1';
    rc = make_rank(m, rank, device, comm);
    if (rc != RTR_OK) { rtr_mgpu_destroy(m); return rc; }
    *out = m;
    return RTR_OK;
}

void rtr_mgpu_destroy(rtr_mgpu* m) {
    if (!m) return;
    for (auto& rp : m->ranks) {
        Rank& r = *rp;
        for (int s = 0; s < RTR_MGPU_MAX_SLOTS; ++s) if (r.slots[s].inFlight) { (void)r.slots[s].pending.get(); r.slots[s].inFlight = false; }
        r.worker.reset();                                   /* joins the thread */
        (void)hipSetDevice(r.device);
        if (r.commStream) (void)hipStreamSynchronize(r.commStream);
        for (int sl = 0; sl < RTR_MGPU_MAX_SLOTS; ++sl) if (r.renderStream[sl]) (void)hipStreamSynchronize(r.renderStream[sl]);
        for (int s = 0; s < RTR_MGPU_MAX_SLOTS; ++s) release_slot(r, r.slots[s]);
        if (r.scene) rtr_scene_destroy(r.scene);
        if (r.comm) (void)ncclCommDestroy(r.comm);
        if (r.commCtx) rtr_ctx_destroy(r.commCtx);
        for (int sl = 0; sl < RTR_MGPU_MAX_SLOTS; ++sl) if (r.ctx[sl]) rtr_ctx_destroy(r.ctx[sl]);
    }
    delete m;
}

int rtr_mgpu_get_info(const rtr_mgpu* m, rtr_mgpu_info* out) {
    if (!m || !out) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_get_info: null argument");
    memset(out, 0, sizeof *out);
    out->nranks = m->nranks; out->nlocal = (int)m->ranks.size(); out->firstRank = m->ranks.empty() ? 0 : m->ranks[0]->rank;
    out->framesInFlight = m->framesInFlight; out->selfExchange = m->selfExchange ? 1 : 0;
    return RTR_OK;
}

int rtr_mgpu_scene_create(rtr_mgpu* m, const rtr_scene_desc* desc) {
    if (!m || !desc) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_scene_create: null argument");
    /* one build per device, concurrently, each on its rank's thread (the host BVH build is the long part) */
    std::vector<std::future<int>> fs;
    for (auto& rp : m->ranks) {
        Rank* r = rp.get();
        fs.push_back(r->worker->submit([r, desc](std::string& err) -> int {
            W_HIP(hipSetDevice(r->device));
            if (r->scene) { rtr_scene_destroy(r->scene); r->scene = nullptr; }
            W_RTR(rtr_scene_create(r->ctx[0], desc, &r->scene));       /* frames of the other slots' contexts render it too (read-only) */
            return RTR_OK;
        }));
    }
    int rc = RTR_OK; size_t i = 0;
    for (auto& f : fs) { const int c = f.get(); if (c != RTR_OK && rc == RTR_OK) rc = fail(c, "rank %d: %s", m->ranks[i]->rank, m->ranks[i]->worker->error().c_str()); ++i; }
    return rc;
}

int rtr_mgpu_render_async(rtr_mgpu* m, int slot, const RtrCameraData* cam, const RtrSceneInfo* info, const rtr_render_params* p, int flags) {
    if (!m || !cam || !info || !p) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_render_async: null argument");
    if (slot < 0 || slot >= m->framesInFlight) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_render_async: slot %d not in [0,%d)", slot, m->framesInFlight);
    if (p->width == 0 || p->height == 0) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_render_async: empty frame");
    if (p->images & ~(RTR_IMAGES_FRAMEBUFFER | RTR_IMG_BIT(RTR_IMAGE_HDR))) return fail(RTR_ERR_UNSUPPORTED, "rtr_mgpu_render_async: only the RGBA8 framebuffer (RTR_IMAGE_SHADOWED) is gathered; RTR_IMAGE_HDR may be added for accumulation");
    if (p->accumulate && !(p->images & RTR_IMG_BIT(RTR_IMAGE_HDR))) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_render_async: accumulate needs RTR_IMAGE_HDR in params->images");
    /* consecutive calls on a slot are ordered by its render stream; only an exchange in flight forbids the next call */
    for (auto& rp : m->ranks) if (rp->slots[slot].inFlight) { const int c = rp->slots[slot].pending.get(); rp->slots[slot].inFlight = false; if (c != RTR_OK) return fail(c, "rank %d: %s", rp->rank, rp->worker->error().c_str()); }
    for (auto& rp : m->ranks) {
        Rank* r = rp.get();
        const RtrCameraData c = *cam; const RtrSceneInfo si = *info; const rtr_render_params pp = *p;
        r->slots[slot].pending = r->worker->submit([m, r, slot, c, si, pp, flags](std::string& err) -> int { return enqueue(m, *r, slot, c, si, pp, flags, err); });
        r->slots[slot].inFlight = true;
    }
    return RTR_OK;
}

int rtr_mgpu_wait(rtr_mgpu* m, int slot) {
    if (!m) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_wait: null");
    if (slot < 0 || slot >= m->framesInFlight) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_wait: slot %d not in [0,%d)", slot, m->framesInFlight);
    int rc = RTR_OK;
    for (auto& rp : m->ranks) {
        Rank& r = *rp; Slot& s = r.slots[slot];
        if (!s.inFlight) continue;
        const int c = s.pending.get();
        s.inFlight = false;
        if (c != RTR_OK) { if (rc == RTR_OK) rc = fail(c, "rank %d: %s", r.rank, r.worker->error().c_str()); continue; }
        if (hipSetDevice(r.device) != hipSuccess || hipEventSynchronize(s.evComm) != hipSuccess) { if (rc == RTR_OK) rc = fail(RTR_ERR_HIP, "rank %d: waiting for the exchange failed: %s", r.rank, hipGetErrorString(hipGetLastError())); continue; }
        if (rtr_frame_wait(s.frame) != RTR_OK && rc == RTR_OK) rc = fail(RTR_ERR_HIP, "rank %d: %s", r.rank, rtr_last_error());   /* the render is long done: this collects its per-kernel times */
    }
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
    if (r->slots[slot].inFlight) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_mgpu_frame_download: slot %d is in flight; rtr_mgpu_wait it first", slot);
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
