// comm.hip -- RCCL called from inside the library (no counterpart in the reference, which has no
// collectives).  The row-sharded build's collectives -- the all-gather of a max-min round's candidates, of the raw rows, of the
// neighbour lists, the all-to-all of the finished rows -- are enqueued on the context's own stream by the C code that
// enqueues the kernels around them: the 32 anchor rounds of a fit (all-gather + pick + sweep each) are ONE C call and never
// return to the host language.  RCCL is loaded at first use with dlopen (librccl.so.1: the one PyTorch-ROCm ships or
// /opt/rocm's): the library has no link-time dependency on it and single-GPU use never touches it.  The communicator is
// created from a 128-byte unique id that rank 0 makes (annchor_comm_unique_id) and the host hands to every rank by any
// means it has (a torch.distributed / MPI broadcast, a file): the C-ABI carries plain bytes.
#include "streamed.h"

#include <dlfcn.h>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

namespace {
// the part of rccl.h this file needs (RCCL 2.x ABI)
typedef struct { char internal[128]; } RcclUniqueId;
typedef void *RcclComm;
typedef int RcclResult;                 // ncclSuccess = 0
enum { kRcclInt8 = 0 };                // ncclInt8 / ncclChar
struct RcclApi {
    void *lib = nullptr;
    RcclResult (*GetUniqueId)(RcclUniqueId *) = nullptr;
    RcclResult (*CommInitRank)(RcclComm *, int, RcclUniqueId, int) = nullptr;
    RcclResult (*CommDestroy)(RcclComm) = nullptr;
    RcclResult (*CommAbort)(RcclComm) = nullptr;            // (optional: without it a timed-out wait is reported but cannot be unblocked)
    RcclResult (*AllGather)(const void *, void *, size_t, int, RcclComm, hipStream_t) = nullptr;
    RcclResult (*Send)(const void *, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    RcclResult (*Recv)(void *, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    RcclResult (*GroupStart)() = nullptr;
    RcclResult (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(RcclResult) = nullptr;
    std::string err;
};
RcclApi g_rccl;
std::once_flag g_rccl_once;

void rccl_load()
{
    const char *names[] = {getenv("ANNCHOR_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
        if (!nm || !*nm) continue;
        g_rccl.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.lib) break;
    }
    if (!g_rccl.lib) { g_rccl.err = "librccl.so.1 not found (set ANNCHOR_RCCL_LIB)"; return; }
#define RCCL_SYM(field, name) \
    *reinterpret_cast<void **>(&g_rccl.field) = dlsym(g_rccl.lib, name); \
    if (!g_rccl.field) { g_rccl.err = std::string("symbol missing in RCCL: ") + name; return; }
    RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
    RCCL_SYM(CommInitRank, "ncclCommInitRank")
    RCCL_SYM(CommDestroy, "ncclCommDestroy")
    RCCL_SYM(AllGather, "ncclAllGather")
    RCCL_SYM(Send, "ncclSend")
    RCCL_SYM(Recv, "ncclRecv")
    RCCL_SYM(GroupStart, "ncclGroupStart")
    RCCL_SYM(GroupEnd, "ncclGroupEnd")
    RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef RCCL_SYM
    *reinterpret_cast<void **>(&g_rccl.CommAbort) = dlsym(g_rccl.lib, "ncclCommAbort");
}
bool rccl_ready()
{
    std::call_once(g_rccl_once, rccl_load);
    return g_rccl.err.empty();
}

// ---- dead-peer guard.  A rank that dies (or never reaches a collective) leaves the others' RCCL kernels spinning and their host
// waits blocked for good.  Every host wait of a context that holds a communicator is therefore armed: a watchdog thread (one
// per context, started with the communicator) sleeps until the deadline and, if the wait is still on, aborts the communicators
// (ncclCommAbort makes the stuck kernels exit), so that the wait returns and the call fails loudly.
struct CommWatch {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    bool armed = false, stop = false, fired = false;
    std::chrono::steady_clock::time_point deadline;
    annchor_ctx *c = nullptr;
    const char *where = "";
};
void watch_main(CommWatch *w)
{
    std::unique_lock<std::mutex> lk(w->m);
    for (;;) {
        w->cv.wait(lk, [&] { return w->stop || w->armed; });
        if (w->stop) return;
        const auto dl = w->deadline;
        if (w->cv.wait_until(lk, dl, [&] { return w->stop || !w->armed || w->deadline != dl; })) continue;   // disarmed / re-armed / stopping
        // the deadline passed with the wait still on
        w->fired = true;
        w->armed = false;
        annchor_ctx *c = w->c;
        fprintf(stderr, "annchor: rank %d of %d: a host wait (%s) on a context with a communicator has lasted %.0f s -- a peer rank is not "
                        "taking part in a collective; aborting the communicator%s\n", c->comm_rank, c->comm_world, w->where, c->comm_timeout_s,
                g_rccl.CommAbort ? "" : " is not possible with this RCCL (no ncclCommAbort): the wait stays blocked");
        if (g_rccl.CommAbort) {
            if (c->comm_side) (void)g_rccl.CommAbort((RcclComm)c->comm_side);
            if (c->comm) (void)g_rccl.CommAbort((RcclComm)c->comm);
        }
    }
}
}   // namespace

// hipStreamSynchronize under the watchdog (ctx.hip's ann_sync and the streamed form's waits call this when c->comm is set)
hipError_t ann_comm_guarded_sync(annchor_ctx *c, hipStream_t stream, const char *where)
{
    CommWatch *w = (CommWatch *)c->comm_watch;
    if (!w || c->comm_timeout_s <= 0) return hipStreamSynchronize(stream);
    if (c->comm_aborted) return hipErrorLaunchFailure;
    {
        std::lock_guard<std::mutex> g(w->m);
        w->armed = true; w->fired = false; w->where = where;
        w->deadline = std::chrono::steady_clock::now() + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(c->comm_timeout_s));
    }
    w->cv.notify_all();
    hipError_t rc = hipStreamSynchronize(stream);
    bool fired;
    {
        std::lock_guard<std::mutex> g(w->m);
        w->armed = false;
        fired = w->fired;
    }
    w->cv.notify_all();
    if (fired) {
        c->comm_aborted = true;
        c->err = std::string("collective timed out after ") + std::to_string((int)c->comm_timeout_s) + " s in " + where +
                 ": a peer rank did not take part (communicator aborted; ANNCHOR_COMM_TIMEOUT_S / annchor_comm_set_timeout change the limit)";
        if (rc == hipSuccess) rc = hipErrorLaunchFailure;
    }
    return rc;
}

#define ANN_CHECK_RCCL(c, call)                                                                                        \
    do {                                                                                                               \
        const RcclResult r_ = (call);                                                                                  \
        if (r_ != 0) {                                                                                                 \
            (c)->err = std::string("RCCL: ") + g_rccl.GetErrorString(r_) + " at " #call;                               \
            return ANNCHOR_EHIP;                                                                                       \
        }                                                                                                              \
    } while (0)

extern "C" int annchor_comm_unique_id(uint8_t *id128)
{
    if (!id128) return ANNCHOR_EINVAL;
    if (!rccl_ready()) return ANNCHOR_ESTATE;
    RcclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != 0) return ANNCHOR_EHIP;
    memcpy(id128, id.internal, 128);
    return ANNCHOR_OK;
}

extern "C" int annchor_comm_init(annchor_ctx *c, const uint8_t *id128, int32_t world, int32_t rank)
{
    if (!c || !id128) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, world >= 1 && rank >= 0 && rank < world, ANNCHOR_EINVAL, "rank %d of %d", rank, world);
    ANN_REQUIRE(c, rccl_ready(), ANNCHOR_ESTATE, "%s", g_rccl.err.c_str());
    ANN_REQUIRE(c, !c->comm, ANNCHOR_ESTATE, "this context already has a communicator");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    RcclUniqueId id;
    memcpy(id.internal, id128, 128);
    RcclComm comm = nullptr;
    ANN_CHECK_RCCL(c, g_rccl.CommInitRank(&comm, world, id, rank));
    c->comm = comm; c->comm_world = world; c->comm_rank = rank;
    c->comm_aborted = false;
    if (const char *t = getenv("ANNCHOR_COMM_TIMEOUT_S")) c->comm_timeout_s = atof(t);
    CommWatch *w = new CommWatch;
    w->c = c;
    w->th = std::thread(watch_main, w);
    c->comm_watch = w;
    return ANNCHOR_OK;
}

int ann_comm_allgather(annchor_ctx *c, const void *send, void *recv, int64_t nbytes);

// Host waits longer than `seconds` on this context abort its communicators and fail (<= 0: no guard).  Default 300 s
// (ANNCHOR_COMM_TIMEOUT_S).
extern "C" int annchor_comm_set_timeout(annchor_ctx *c, double seconds)
{
    if (!c) return ANNCHOR_EINVAL;
    c->comm_timeout_s = seconds;
    return ANNCHOR_OK;
}

// A second communicator (its own 128-byte id, made like the first) and a stream of its own for annchor_comm_allgather_begin.
extern "C" int annchor_comm_init_side(annchor_ctx *c, const uint8_t *id128)
{
    if (!c || !id128) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->comm, ANNCHOR_ESTATE, "annchor_comm_init first");
    ANN_REQUIRE(c, !c->comm_side, ANNCHOR_ESTATE, "this context already has a side communicator");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    RcclUniqueId id;
    memcpy(id.internal, id128, 128);
    RcclComm comm = nullptr;
    ANN_CHECK_RCCL(c, g_rccl.CommInitRank(&comm, c->comm_world, id, c->comm_rank));
    c->comm_side = comm;
    ANN_CHECK_HIP(c, hipStreamCreateWithFlags(&c->comm_side_stream, hipStreamNonBlocking));
    ANN_CHECK_HIP(c, hipEventCreateWithFlags(&c->comm_side_ev, hipEventDisableTiming));
    ANN_CHECK_HIP(c, hipEventCreateWithFlags(&c->comm_main_ev, hipEventDisableTiming));
    return ANNCHOR_OK;
}

// the engine stream waits for the side stream's all-gather (no-op when none is in flight)
int ann_comm_side_join(annchor_ctx *c)
{
    if (!c->comm_side_pending) return ANNCHOR_OK;
    c->comm_side_pending = false;
    ANN_CHECK_HIP(c, hipStreamWaitEvent(c->stream, c->comm_side_ev, 0));
    return ANNCHOR_OK;
}

// An all-gather that runs BESIDE the engine stream's work: ordered after everything queued on the engine stream so far (its input
// may have been written there), on the side communicator and stream; the engine stream waits for it where the library first needs
// the result (annchor_stream_rows_end / annchor_stream_order_end for the raw rows; annchor_comm_side_join for other users).
// Without a side communicator it is the plain all-gather on the engine stream.
extern "C" int annchor_comm_allgather_begin(annchor_ctx *c, const void *send, void *recv, int64_t nbytes)
{
    if (!c || nbytes < 0 || (nbytes > 0 && (!send || !recv))) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->comm, ANNCHOR_ESTATE, "no communicator on this context (annchor_comm_init)");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    if (!c->comm_side) return ann_comm_allgather(c, send, recv, nbytes);
    ANN_TRY(ann_comm_side_join(c));   // (one at a time)
    if (nbytes == 0) return ANNCHOR_OK;
    ANN_CHECK_HIP(c, hipEventRecord(c->comm_main_ev, c->stream));
    ANN_CHECK_HIP(c, hipStreamWaitEvent(c->comm_side_stream, c->comm_main_ev, 0));
    ANN_CHECK_RCCL(c, g_rccl.AllGather(send, recv, (size_t)nbytes, kRcclInt8, (RcclComm)c->comm_side, c->comm_side_stream));
    ANN_CHECK_HIP(c, hipEventRecord(c->comm_side_ev, c->comm_side_stream));
    c->comm_side_pending = true;
    return ANNCHOR_OK;
}
extern "C" int annchor_comm_side_join(annchor_ctx *c)
{
    if (!c) return ANNCHOR_EINVAL;
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    return ann_comm_side_join(c);
}

// Pre-flight of a fresh communicator: every rank contributes 1 KB of (rank, position) bytes, all-gathers them on the engine stream
// (and on the side stream when there is one) and checks what arrived, under a timeout of its own -- a mis-wired job fails here,
// loudly and within `seconds`, not minutes into the first fit.
static __global__ void k_comm_fill(uint8_t *p, int rank, int n) { const int t = blockIdx.x * blockDim.x + threadIdx.x; if (t < n) p[t] = (uint8_t)(rank * 31 + t * 7 + 1); }
extern "C" int annchor_comm_preflight(annchor_ctx *c, double seconds)
{
    if (!c) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->comm, ANNCHOR_ESTATE, "no communicator on this context (annchor_comm_init)");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    const int n = 1024, world = c->comm_world;
    uint8_t *buf = nullptr;
    ANN_CHECK_HIP(c, hipMalloc(&buf, (size_t)n * (world + 1)));
    const double keep = c->comm_timeout_s;
    if (seconds > 0) c->comm_timeout_s = seconds;
    int rc = ANNCHOR_OK;
    std::vector<uint8_t> host((size_t)n * world);
    for (int pass = 0; pass < (c->comm_side ? 2 : 1) && rc == ANNCHOR_OK; ++pass) {
        k_comm_fill<<<(n + 255) / 256, 256, 0, c->stream>>>(buf, c->comm_rank + 64 * pass, n);
        hipError_t he = hipMemsetAsync(buf + n, 0, (size_t)n * world, c->stream);
        if (he == hipSuccess) {
            rc = pass == 0 ? ann_comm_allgather(c, buf, buf + n, n) : annchor_comm_allgather_begin(c, buf, buf + n, n);
            if (rc == ANNCHOR_OK && pass == 1) rc = ann_comm_side_join(c);
        }
        if (rc == ANNCHOR_OK && he == hipSuccess) he = hipMemcpyAsync(host.data(), buf + n, host.size(), hipMemcpyDeviceToHost, c->stream);
        if (rc == ANNCHOR_OK && he == hipSuccess) he = ann_comm_guarded_sync(c, c->stream, "annchor_comm_preflight");
        if (rc == ANNCHOR_OK && he != hipSuccess) {
            if (c->err.empty() || !c->comm_aborted) c->err = std::string("pre-flight all-gather failed: ") + hipGetErrorString(he);
            rc = ANNCHOR_EHIP;
        }
        for (int r = 0; r < world && rc == ANNCHOR_OK; ++r)
            for (int t = 0; t < n; ++t)
                if (host[(size_t)r * n + t] != (uint8_t)((r + 64 * pass) * 31 + t * 7 + 1)) {
                    ann_set_err(c, "pre-flight all-gather (%s communicator): rank %d's block arrived wrong at byte %d on rank %d of %d", pass ? "side" : "main", r, t,
                                c->comm_rank, world);
                    rc = ANNCHOR_EHIP;
                    break;
                }
    }
    c->comm_timeout_s = keep;
    (void)hipFree(buf);
    return rc;
}

extern "C" int annchor_comm_destroy(annchor_ctx *c)
{
    if (!c) return ANNCHOR_EINVAL;
    if (c->comm_watch) {
        CommWatch *w = (CommWatch *)c->comm_watch;
        { std::lock_guard<std::mutex> g(w->m); w->stop = true; }
        w->cv.notify_all();
        if (w->th.joinable()) w->th.join();
        delete w;
        c->comm_watch = nullptr;
    }
    if (c->comm_side) {
        if (!c->comm_aborted) { (void)hipStreamSynchronize(c->comm_side_stream); (void)g_rccl.CommDestroy((RcclComm)c->comm_side); }
        c->comm_side = nullptr; c->comm_side_pending = false;
        if (c->comm_side_stream) (void)hipStreamDestroy(c->comm_side_stream);
        if (c->comm_side_ev) (void)hipEventDestroy(c->comm_side_ev);
        if (c->comm_main_ev) (void)hipEventDestroy(c->comm_main_ev);
        c->comm_side_stream = nullptr; c->comm_side_ev = nullptr; c->comm_main_ev = nullptr;
    }
    if (c->comm) {
        if (!c->comm_aborted) { (void)hipStreamSynchronize(c->stream); (void)g_rccl.CommDestroy((RcclComm)c->comm); }
        c->comm = nullptr; c->comm_world = 1; c->comm_rank = 0;
    }
    return ANNCHOR_OK;
}
void ann_comm_release(annchor_ctx *c) { (void)annchor_comm_destroy(c); }

// every rank's `nbytes` bytes at `send`, concatenated in rank order at `recv` (device pointers); on the context's stream
int ann_comm_allgather(annchor_ctx *c, const void *send, void *recv, int64_t nbytes)
{
    ANN_REQUIRE(c, c->comm, ANNCHOR_ESTATE, "no communicator on this context (annchor_comm_init)");
    ANN_REQUIRE(c, !c->comm_aborted, ANNCHOR_ESTATE, "the communicator was aborted after a timed-out collective");
    if (nbytes == 0) return ANNCHOR_OK;
    ANN_CHECK_RCCL(c, g_rccl.AllGather(send, recv, (size_t)nbytes, kRcclInt8, (RcclComm)c->comm, c->stream));
    return ANNCHOR_OK;
}
extern "C" int annchor_comm_allgather(annchor_ctx *c, const void *send, void *recv, int64_t nbytes)
{
    if (!c || nbytes < 0 || (nbytes > 0 && (!send || !recv))) return ANNCHOR_EINVAL;
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    return ann_comm_allgather(c, send, recv, nbytes);
}

// records of `words` 8-byte words, grouped by destination rank at `send` (send_counts[world] records each), received in
// source-rank order at `recv` (recv_counts[world]): one group of point-to-point transfers
extern "C" int annchor_comm_alltoall_records(annchor_ctx *c, const void *send, const int64_t *send_counts, void *recv,
                                             const int64_t *recv_counts, int32_t words)
{
    if (!c || !send_counts || !recv_counts || words < 1) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->comm, ANNCHOR_ESTATE, "no communicator on this context (annchor_comm_init)");
    ANN_REQUIRE(c, !c->comm_aborted, ANNCHOR_ESTATE, "the communicator was aborted after a timed-out collective");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    const size_t rec = (size_t)words * 8;
    ANN_CHECK_RCCL(c, g_rccl.GroupStart());
    // the first failure is remembered and the group is ALWAYS closed: a return between GroupStart and GroupEnd would leave the
    // communicator's next collective inside a dangling group
    size_t so = 0, ro = 0;
    RcclResult first = 0;
    const char *where = "";
    for (int r = 0; r < c->comm_world && first == 0; ++r) {
        const size_t sb = (size_t)send_counts[r] * rec, rb = (size_t)recv_counts[r] * rec;
        if (sb && first == 0) { first = g_rccl.Send((const char *)send + so, sb, kRcclInt8, r, (RcclComm)c->comm, c->stream); where = "Send"; }
        if (rb && first == 0) { first = g_rccl.Recv((char *)recv + ro, rb, kRcclInt8, r, (RcclComm)c->comm, c->stream); where = "Recv"; }
        so += sb; ro += rb;
    }
    const RcclResult end = g_rccl.GroupEnd();
    if (first == 0 && end != 0) { first = end; where = "GroupEnd"; }
    if (first != 0) {
        c->err = std::string("RCCL: ") + g_rccl.GetErrorString(first) + " at " + where + " (alltoall_records)";
        return ANNCHOR_EHIP;
    }
    return ANNCHOR_OK;
}

// All max-min rounds of the row-sharded build (annchor_stream_anchor_begin has left this rank's first candidate): per round
// the all-gather of the candidates, the winner's pick and the sweep of the local rows -- 3 n_anchors enqueues, no host wait,
// no return to the caller in between.  Without a communicator (one rank) the candidate buffer is its own gather.
extern "C" int annchor_stream_anchor_rounds(annchor_ctx *c, int32_t n_anchors)
{
    if (!c) return ANNCHOR_EINVAL;
    StreamState *s = ann_stream_state(c, false);
    ANN_REQUIRE(c, s && s->cand.p && s->na == n_anchors, ANNCHOR_ESTATE, "annchor_stream_anchor_begin first");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    const int world = c->comm ? c->comm_world : 1;
    const int64_t cand_bytes = (int64_t)sizeof(double) * (2 + s->dim);
    ANN_REQUIRE(c, s->cand_all.cap >= (size_t)cand_bytes * world, ANNCHOR_ESTATE, "annchor_stream_anchor_begin was given a smaller world");
    for (int r = 0; r < n_anchors; ++r) {
        if (c->comm) {
            ANN_TRY(ann_comm_allgather(c, s->cand.p, s->cand_all.p, cand_bytes));
            ANN_TRY(annchor_stream_anchor_step(c, s->cand_all.p, world, r));
        } else {
            ANN_TRY(annchor_stream_anchor_step(c, s->cand.p, 1, r));
        }
    }
    return ANNCHOR_OK;
}
