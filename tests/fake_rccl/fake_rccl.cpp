/* fake_rccl.cpp — TEST INFRASTRUCTURE, never linked into or shipped with the product.
 *
 * A same-device stand-in for the eleven RCCL entry points librtr_mgpu.so uses, so that the library's real N > 1 code — enqueue()
 * executing every rank's plan on real HIP streams, a host thread per rank, the watchdog — runs with 2 ... 8 ranks on a box that has
 * ONE GPU.  Loaded with LD_PRELOAD in front of the real librccl.so by tests/test_gpu_parity.py (child processes only), together with
 * the library's test hook RTR_MGPU_TEST_SHARED_DEVICE=1 (ranks may share a device).  Every rank lives on the same device, so a
 * "transfer" is a device-to-device copy ordered by events:
 *   ncclSend   posts {buffer, bytes, an event recorded on the sender's stream} for (src, dst) — at ncclGroupEnd;
 *   ncclRecv   waits (host side) for the matching post, makes its stream wait for that event, enqueues the copy, records a second
 *              event behind it;
 *   the sender's ncclGroupEnd returns once its posts were matched, after making its stream wait for the copies (its buffer may be
 *   overwritten by the next render).
 * Sends and receives between a pair are matched in posting order, as RCCL matches them.  Only ncclUint8 is supported (all the library
 * sends).  HIP is reached through dlsym at first use: the process must keep ONE HIP runtime (realtimeraytracer_amd/_abi.py), so this
 * file links against none. */
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <deque>
#include <cstdio>
#include <dlfcn.h>
#include <link.h>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

extern "C" {
typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef int ncclResult_t;          /* ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclInvalidArgument = 4, ncclInvalidUsage = 5 */
typedef struct FakeComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
}

namespace {

struct Hip {
    int (*eventCreate)(hipEvent_t*, unsigned) = nullptr;
    int (*eventRecord)(hipEvent_t, hipStream_t) = nullptr;
    int (*eventDestroy)(hipEvent_t) = nullptr;
    int (*streamWaitEvent)(hipStream_t, hipEvent_t, unsigned) = nullptr;
    int (*memcpyAsync)(void*, const void*, size_t, int, hipStream_t) = nullptr;
    bool ok = false;
};
Hip& hip() {
    static Hip h;
    static std::once_flag once;
    std::call_once(once, [] {
        /* the HIP runtime ALREADY in the process (torch's copy or /opt/rocm's, whichever the harness loaded): found among the loaded
         * objects by name and re-opened without loading anything */
        void* lib = nullptr;
        dl_iterate_phdr([](dl_phdr_info* info, size_t, void* out) -> int {
            if (info->dlpi_name && strstr(info->dlpi_name, "libamdhip64")) { *(void**)out = dlopen(info->dlpi_name, RTLD_NOLOAD | RTLD_NOW); return *(void**)out != nullptr; }
            return 0;
        }, &lib);
        if (!lib) lib = RTLD_DEFAULT;
        h.eventCreate = (int (*)(hipEvent_t*, unsigned))dlsym(lib, "hipEventCreateWithFlags");
        h.eventRecord = (int (*)(hipEvent_t, hipStream_t))dlsym(lib, "hipEventRecord");
        h.eventDestroy = (int (*)(hipEvent_t))dlsym(lib, "hipEventDestroy");
        h.streamWaitEvent = (int (*)(hipStream_t, hipEvent_t, unsigned))dlsym(lib, "hipStreamWaitEvent");
        h.memcpyAsync = (int (*)(void*, const void*, size_t, int, hipStream_t))dlsym(lib, "hipMemcpyAsync");
        h.ok = h.eventCreate && h.eventRecord && h.eventDestroy && h.streamWaitEvent && h.memcpyAsync;
        if (!h.ok) fprintf(stderr, "fake rccl: no HIP runtime found in the process\n");
    });
    return h;
}

struct Msg {
    const void* buf; size_t bytes;
    hipEvent_t ready = nullptr;      /* recorded on the sender's stream: the data is there */
    hipEvent_t copied = nullptr;     /* recorded on the receiver's stream behind the copy */
    bool matched = false, failed = false;
};

struct World {
    int n = 0, alive = 0;
    int users = 0;                   /* operations queued or running that still hold this world (add() takes the reference, run_group() gives it back) */
    bool aborted = false;
    std::mutex mu;
    std::condition_variable cv;
    std::map<std::pair<int, int>, std::deque<Msg*>> posted;      /* (src, dst) -> sends not yet received, in posting order */
};

/* an operation keeps what it needs of its communicator BY VALUE (rank, world) and a reference on the world: the communicator may be
 * aborted — and deleted — by the watchdog's thread while the rank's own thread is still inside ncclGroupEnd */
struct Op { bool send; void* buf; size_t bytes; int peer; int rank; World* w; hipStream_t stream; };
/* gives the operations' references on their worlds back; the last one out frees a world every communicator of which is gone */
void release(std::vector<Op>& ops) {
    for (Op& o : ops) {
        World* w = o.w;
        bool last;
        { std::lock_guard<std::mutex> g(w->mu); last = --w->users == 0 && w->alive == 0; }
        if (last) delete w;
    }
    ops.clear();
}
/* the operations queued inside an open ncclGroupStart.  A rank's thread that is told to stop between ncclGroupStart and ncclGroupEnd
 * (librtr_mgpu's cooperative abort makes no further RCCL call) never runs the group: the queue gives its references back when the
 * thread ends, so an aborted world is still freed */
struct Pending { std::vector<Op> ops; ~Pending() { release(ops); } };
thread_local int g_depth = 0;
thread_local Pending g_pending;

}  // namespace

struct FakeComm { int rank; int nranks; World* w; };

namespace {

int run_group(std::vector<Op>& ops) {
    Hip& h = hip();
    if (!h.ok) { release(ops); return 1; }
    std::vector<std::pair<Op*, Msg*>> sent;
    int rc = 0;
    for (Op& o : ops) if (o.send) {                      /* 1: post every send of the group */
        Msg* m = new Msg{o.buf, o.bytes};
        if (h.eventCreate(&m->ready, 2u) || h.eventCreate(&m->copied, 2u) || h.eventRecord(m->ready, o.stream)) { fprintf(stderr, "fake rccl: rank %d: event create / record failed\n", o.rank); rc = 1; delete m; break; }
        World* w = o.w;
        { std::lock_guard<std::mutex> g(w->mu); w->posted[{o.rank, o.peer}].push_back(m); }
        w->cv.notify_all();
        sent.push_back({&o, m});
    }
    for (Op& o : ops) if (!o.send && rc == 0) {          /* 2: every receive: wait for its send, copy behind it */
        World* w = o.w;
        Msg* m = nullptr;
        {
            std::unique_lock<std::mutex> g(w->mu);
            auto& q = w->posted[{o.peer, o.rank}];
            w->cv.wait(g, [&] { return !q.empty() || w->aborted; });
            if (w->aborted) { rc = 1; break; }
            m = q.front(); q.pop_front();
        }
        bool bad = m->bytes != o.bytes;
        if (bad) fprintf(stderr, "fake rccl: rank %d receives %zu bytes from %d, which sent %zu\n", o.rank, o.bytes, o.peer, m->bytes);
        if (!bad) {
            const int e1 = h.streamWaitEvent(o.stream, m->ready, 0u), e2 = e1 ? 0 : h.memcpyAsync(o.buf, m->buf, o.bytes, 3 /* device to device */, o.stream), e3 = (e1 || e2) ? 0 : h.eventRecord(m->copied, o.stream);
            bad = e1 || e2 || e3;
            if (bad) fprintf(stderr, "fake rccl: rank %d: wait %d copy %d record %d\n", o.rank, e1, e2, e3);
        }
        { std::lock_guard<std::mutex> g(w->mu); m->matched = true; m->failed = bad; }
        w->cv.notify_all();
        if (bad) rc = 4;
    }
    for (auto& sm : sent) {                                /* 3: the sender's stream waits for the copies of what it sent */
        World* w = sm.first->w;
        Msg* m = sm.second;
        {
            std::unique_lock<std::mutex> g(w->mu);
            w->cv.wait(g, [&] { return m->matched || w->aborted; });
            if (!m->matched) {                              /* aborted before anyone took it: withdraw the post */
                auto& q = w->posted[{sm.first->rank, sm.first->peer}];
                for (auto it = q.begin(); it != q.end(); ++it) if (*it == m) { q.erase(it); break; }
                rc = rc ? rc : 1;
            }
        }
        if (m->matched && !m->failed) { if (h.streamWaitEvent(sm.first->stream, m->copied, 0u)) rc = rc ? rc : 1; }
        else if (m->failed) rc = rc ? rc : 4;
        /* the events are left to the runtime: destroying an event other streams still wait on is legal in HIP, but this is a test
         * double that sends a few hundred messages — it keeps them */
        delete m;
    }
    release(ops);
    return rc;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetVersion(int* v) { if (!v) return 4; *v = 99999; return 0; }      /* no RCCL has this version: rtr_mgpu_info shows the double was in use */
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { if (!id) return 4; memset(id, 0x5a, sizeof *id); return 0; }
const char* ncclGetErrorString(ncclResult_t r) { return r == 0 ? "no error" : (r == 4 ? "fake rccl: invalid argument (size mismatch between a send and its receive?)" : (r == 5 ? "fake rccl: invalid usage" : "fake rccl: HIP call failed or communicator aborted")); }

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int*) {
    if (!comms || n < 1) return 4;
    World* w = new World; w->n = n; w->alive = n;
    for (int i = 0; i < n; ++i) comms[i] = new FakeComm{i, n, w};
    return 0;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId, int rank) {
    if (!comm || nranks != 1 || rank != 0) return 5;          /* several processes are not what this double is for */
    World* w = new World; w->n = 1; w->alive = 1;
    *comm = new FakeComm{0, 1, w};
    return 0;
}
static void leave(FakeComm* c, bool abort) {
    World* w = c->w;
    bool freeNow;
    {   /* one critical section decides who frees the world: this call, or the last operation still holding it (run_group) */
        std::lock_guard<std::mutex> g(w->mu);
        if (abort) w->aborted = true;
        freeNow = --w->alive == 0 && w->users == 0;
        w->cv.notify_all();
    }
    delete c;
    if (freeNow) delete w;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { if (!c) return 4; leave(c, false); return 0; }
ncclResult_t ncclCommAbort(ncclComm_t c) { if (!c) return 4; leave(c, true); return 0; }

ncclResult_t ncclGroupStart(void) { ++g_depth; return 0; }
ncclResult_t ncclGroupEnd(void) {
    if (g_depth <= 0) return 5;
    if (--g_depth > 0) return 0;
    std::vector<Op> ops; ops.swap(g_pending.ops);
    return run_group(ops);
}
static ncclResult_t add(bool send, void* buf, size_t count, int type, int peer, ncclComm_t c, hipStream_t s) {
    if (!buf || !c || type != 1 /* ncclUint8 */ || peer < 0 || peer >= c->nranks) return 4;
    Op o{send, buf, count, peer, c->rank, c->w, s};
    { std::lock_guard<std::mutex> g(c->w->mu); ++c->w->users; }
    if (g_depth > 0) { g_pending.ops.push_back(o); return 0; }
    std::vector<Op> one{o};
    return run_group(one);
}
ncclResult_t ncclSend(const void* buf, size_t count, int type, int peer, ncclComm_t c, hipStream_t s) { return add(true, const_cast<void*>(buf), count, type, peer, c, s); }
ncclResult_t ncclRecv(void* buf, size_t count, int type, int peer, ncclComm_t c, hipStream_t s) { return add(false, buf, count, type, peer, c, s); }

}  // extern "C"
