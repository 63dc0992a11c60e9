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
#include <mutex>

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
}
bool rccl_ready()
{
    std::call_once(g_rccl_once, rccl_load);
    return g_rccl.err.empty();
}
}   // namespace

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
    return ANNCHOR_OK;
}

extern "C" int annchor_comm_destroy(annchor_ctx *c)
{
    if (!c) return ANNCHOR_EINVAL;
    if (c->comm) {
        (void)hipStreamSynchronize(c->stream);
        (void)g_rccl.CommDestroy((RcclComm)c->comm);
        c->comm = nullptr; c->comm_world = 1; c->comm_rank = 0;
    }
    return ANNCHOR_OK;
}
void ann_comm_release(annchor_ctx *c) { (void)annchor_comm_destroy(c); }

// every rank's `nbytes` bytes at `send`, concatenated in rank order at `recv` (device pointers); on the context's stream
int ann_comm_allgather(annchor_ctx *c, const void *send, void *recv, int64_t nbytes)
{
    ANN_REQUIRE(c, c->comm, ANNCHOR_ESTATE, "no communicator on this context (annchor_comm_init)");
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
