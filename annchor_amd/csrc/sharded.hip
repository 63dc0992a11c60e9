// sharded.hip -- the row-sharded (one process per GPU) build of the streamed form with every
// cross-rank buffer resident in device memory.  SURVEY.md section 8(e): the reference has no
// collectives; the protocol is this build's own.  The host (annchor_amd/streamed.py) only moves
// POINTERS: it hands the buffers below to torch.distributed (RCCL over xGMI when the group is
// `nccl`) on the context's own stream, so that a fit runs without host round trips per round.
//
//   anchors   annchor_stream_anchor_begin / _step / _end: per max-min round (pickers.py:44-50) every
//             rank leaves its arg-max candidate -- (value, global row, the row's coordinates), 2 + dim
//             doubles -- in a device buffer; ONE all-gather per round; the next step picks the winner
//             on the device (largest value, first index: np.argmax) and sweeps the local rows with its
//             coordinates, which the all-gather already delivered (no broadcast from the owner).
//   rows      annchor_stream_rows_begin / _end: the raw rows of every rank ([world][most][dim] after one
//             all-gather of padded shards, compacted here, not on the host); the anchor distances of
//             ALL rows are then recomputed in one pass over them (k_sh_all_anchor_dists: the anchors sit
//             in LDS) instead of world-many replicated sweeps per anchor.
//   lists     annchor_stream_lists_all: the all-gather target of the join passes.
//   result    annchor_stream_route_begin / _recv / _end: finished rows are records (global row id,
//             k-1 neighbour ids, k-1 distances) bucketed by owner rank on the device, exchanged with
//             ONE all-to-all, scattered into the shard's own row order and downloaded once.
#include "streamed.h"

#include <algorithm>

#define SH_MAX_WORLD 64

// ------------------------------------------------------------------ anchors
__global__ void k_sh_first_cand(const float *__restrict__ X, int64_t n, int dim, int64_t base, int64_t first, double *__restrict__ cand)
{
    const bool mine = first >= base && first < base + n;
    if (threadIdx.x == 0) { cand[0] = 0.0; cand[1] = mine ? (double)first : -1.0; }
    for (int k = threadIdx.x; k < dim; k += blockDim.x) cand[2 + k] = mine ? (double)X[(size_t)(first - base) * dim + k] : 0.0;
}

// winner of a round among the ranks' candidates: largest value, then smallest global row (np.argmax's
// first-index rule over the concatenated rows); candidates with row < 0 do not take part
__global__ void k_sh_pick(const double *__restrict__ G, int world, int dim, int round, int64_t *__restrict__ A,
                          float *__restrict__ avecs, float *__restrict__ avec)
{
    __shared__ int win;
    if (threadIdx.x == 0) {
        double bv = 0;
        int64_t bi = -1;
        int br = 0;
        for (int r = 0; r < world; ++r) {
            const double v = G[(size_t)r * (2 + dim)];
            const int64_t i = (int64_t)G[(size_t)r * (2 + dim) + 1];
            if (i < 0) continue;
            if (bi < 0 || v > bv || (v == bv && i < bi)) { bv = v; bi = i; br = r; }
        }
        A[round] = bi;
        win = br;
    }
    __syncthreads();
    const double *g = G + (size_t)win * (2 + dim) + 2;
    for (int k = threadIdx.x; k < dim; k += blockDim.x) {
        const float v = (float)g[k];
        avec[k] = v;
        avecs[(size_t)round * dim + k] = v;
    }
}

// final reduction of the sweep's per-workgroup arg-max partials + the candidate record of the next round
__global__ __launch_bounds__(256) void k_sh_cand(const float *__restrict__ redval, const int64_t *__restrict__ redidx, int nb,
                                                 const float *__restrict__ X, int dim, int64_t base, double *__restrict__ cand)
{
    float bv = -INFINITY;
    int64_t bi = 0x7fffffffffffffffll;
    for (int b = threadIdx.x; b < nb; b += blockDim.x) {
        const float v = redval[b];
        const int64_t i = redidx[b];
        if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bv, off);
        const int64_t oi = __shfl_xor(bi, off);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    __shared__ float sv[4];
    __shared__ int64_t si[4];
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    for (int w = 0; w < 4; ++w)
        if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
    if (threadIdx.x == 0) { cand[0] = (double)bv; cand[1] = (double)(base + bi); }
    for (int k = threadIdx.x; k < dim; k += blockDim.x) cand[2 + k] = (double)X[(size_t)bi * dim + k];
}

extern "C" int annchor_stream_hip_stream(annchor_ctx *c, void **stream)
{
    if (!c || !stream) return ANNCHOR_EINVAL;
    *stream = (void *)c->stream;
    return ANNCHOR_OK;
}

extern "C" int annchor_stream_anchor_begin(annchor_ctx *c, int32_t n_anchors, int64_t first_global, int32_t world, void **cand,
                                           void **gathered, int64_t *cand_bytes)
{
    if (!c || !cand || !gathered || !cand_bytes) return ANNCHOR_EINVAL;
    StreamState *s = ann_stream_state(c, false);
    ANN_REQUIRE(c, s && s->n_local > 0, ANNCHOR_EINVAL, "annchor_stream_bind first");
    ANN_REQUIRE(c, n_anchors >= 1 && n_anchors <= 64, ANNCHOR_ELIMIT, "streamed form supports 1 <= n_anchors <= 64");
    ANN_REQUIRE(c, world >= 1 && world <= SH_MAX_WORLD, ANNCHOR_ELIMIT, "1 <= world <= %d", SH_MAX_WORLD);
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    s->na = n_anchors;
    ANN_TRY(ann_stream_reserve(c, s->cand_all, sizeof(double) * (size_t)(2 + s->dim) * (size_t)world));
    ANN_TRY(ann_stream_reserve(c, s->D, sizeof(float) * (size_t)n_anchors * (size_t)s->n_local));
    ANN_TRY(ann_stream_reserve(c, s->avecs, sizeof(float) * (size_t)n_anchors * s->dim));
    ANN_TRY(ann_stream_reserve(c, s->A_dev, sizeof(int64_t) * (size_t)n_anchors));
    ANN_TRY(ann_stream_reserve(c, s->cand, sizeof(double) * (size_t)(2 + s->dim)));
    k_sh_first_cand<<<1, 64, 0, c->stream>>>(s->X.as<float>(), s->n_local, s->dim, s->base, first_global, s->cand.as<double>());
    ANN_CHECK_HIP(c, hipGetLastError());
    *cand = s->cand.p;
    *gathered = s->cand_all.p;
    *cand_bytes = (int64_t)sizeof(double) * (2 + s->dim);
    return ANNCHOR_OK;
}

// `gathered`: double [world][2 + dim], the ranks' candidates of this round in rank order (device; with one rank the
// candidate buffer itself).  Enqueues: winner -> anchor `round`; sweep of the local rows; this rank's candidate of
// the next round into the candidate buffer.  No host wait.
extern "C" int annchor_stream_anchor_step(annchor_ctx *c, const void *gathered, int32_t world, int32_t round)
{
    if (!c || !gathered) return ANNCHOR_EINVAL;
    StreamState *s = ann_stream_state(c, false);
    ANN_REQUIRE(c, s && s->cand.p && s->na > 0, ANNCHOR_ESTATE, "annchor_stream_anchor_begin first");
    ANN_REQUIRE(c, world >= 1 && world <= SH_MAX_WORLD && round >= 0 && round < s->na, ANNCHOR_EINVAL, "bad anchor step");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    k_sh_pick<<<1, 64, 0, c->stream>>>((const double *)gathered, world, s->dim, round, s->A_dev.as<int64_t>(), s->avecs.as<float>(),
                                       s->avec.as<float>());
    int nb = 0;
    ANN_TRY(ann_stream_sweep(c, s, s->avec.as<float>(), round, &nb));
    k_sh_cand<<<1, 256, 0, c->stream>>>(s->red_val.as<float>(), s->red_idx.as<int64_t>(), nb, s->X.as<float>(), s->dim, s->base,
                                        s->cand.as<double>());
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}

// anchors (global row ids, int64 [n_anchors]) and their coordinates (float32 [n_anchors][dim]) to the host: the one wait
// of the anchor stage
extern "C" int annchor_stream_anchor_end(annchor_ctx *c, int64_t *A, float *anchor_vectors)
{
    if (!c || !A || !anchor_vectors) return ANNCHOR_EINVAL;
    StreamState *s = ann_stream_state(c, false);
    ANN_REQUIRE(c, s && s->na > 0 && s->A_dev.p, ANNCHOR_ESTATE, "no anchor rounds on this context");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    return ann_d2h2(c, A, s->A_dev.p, sizeof(int64_t) * (size_t)s->na, anchor_vectors, s->avecs.p, sizeof(float) * (size_t)s->na * s->dim);
}

// ------------------------------------------------------------------ raw rows of every rank
// distances of every row to every anchor in ONE pass over the rows: 16 lanes per row hold the row in registers, the
// anchors sit in LDS.  The arithmetic is k_st_one_to_all's, operation for operation (the library is built with
// -ffp-contract=off), so a rank's D is bit-identical to what per-anchor sweeps of the same rows give.
template <int NV> __global__ __launch_bounds__(256) void k_sh_all_anchor_dists(const float *__restrict__ X, int64_t n, int dim,
                                                                              const float *__restrict__ avecs, int na,
                                                                              float *__restrict__ D)
{
    extern __shared__ float sA[];   // [na][dim]
    for (int t = threadIdx.x; t < na * dim; t += blockDim.x) sA[t] = avecs[t];
    __syncthreads();
    const int sub = threadIdx.x & 15;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4; i < ((n + 15) & ~15ll); i += ((int64_t)gridDim.x * blockDim.x) >> 4) {
        const bool ok = i < n;
        const float *x = X + (size_t)(ok ? i : 0) * dim;
        if constexpr (NV > 0) {
            float4 u[NV];
#pragma unroll
            for (int t = 0; t < NV; ++t) {
                const int k = sub * 4 + 64 * t;
                u[t] = k < dim ? *reinterpret_cast<const float4 *>(x + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            for (int a = 0; a < na; ++a) {
                const float *av = sA + (size_t)a * dim;
                float acc = 0.f;
#pragma unroll
                for (int t = 0; t < NV; ++t) {
                    const int k = sub * 4 + 64 * t;
                    if (k < dim) {
                        const float4 w = *reinterpret_cast<const float4 *>(av + k);
                        float d0 = u[t].x - w.x, d1 = u[t].y - w.y, d2 = u[t].z - w.z, d3 = u[t].w - w.w;
                        acc += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
                    }
                }
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 16);
                if (ok && sub == 0) D[(size_t)a * n + i] = sqrtf(acc);
            }
        } else {
            for (int a = 0; a < na; ++a) {
                const float *av = sA + (size_t)a * dim;
                float acc = 0.f;
                for (int k = sub; k < dim; k += 16) { float d = x[k] - av[k]; acc += d * d; }
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 16);
                if (ok && sub == 0) D[(size_t)a * n + i] = sqrtf(acc);
            }
        }
    }
}

// counts: rows of every rank's shard (host, int64 [world]).  *send: this rank's rows padded with zero rows to the
// largest shard (the shard itself when it is the largest); *recv: room for [world][most][dim]; both device.  The host
// all-gathers *bytes_per_rank bytes from *send into *recv.
extern "C" int annchor_stream_rows_begin(annchor_ctx *c, int32_t world, const int64_t *counts, void **send, void **recv,
                                         int64_t *bytes_per_rank)
{
    if (!c || !counts || !send || !recv || !bytes_per_rank) return ANNCHOR_EINVAL;
    StreamState *s = ann_stream_state(c, false);
    ANN_REQUIRE(c, s && s->n_local > 0, ANNCHOR_EINVAL, "annchor_stream_bind first");
    ANN_REQUIRE(c, world >= 1 && world <= SH_MAX_WORLD, ANNCHOR_ELIMIT, "1 <= world <= %d", SH_MAX_WORLD);
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    int64_t most = 0;
    bool found = false;
    for (int r = 0; r < world; ++r) { most = std::max(most, counts[r]); found = found || counts[r] == s->n_local; }
    ANN_REQUIRE(c, found, ANNCHOR_EINVAL, "this rank's shard (%lld rows) is not among the counts", (long long)s->n_local);
    const size_t row_bytes = sizeof(float) * (size_t)s->dim, bytes = row_bytes * (size_t)most;
    ANN_TRY(ann_stream_reserve(c, s->rows_recv, bytes * (size_t)world));
    if (s->n_local == most) {
        *send = s->X.p;
    } else {
        ANN_TRY(ann_stream_reserve(c, s->rows_send, bytes));
        ANN_CHECK_HIP(c, hipMemcpyAsync(s->rows_send.p, s->X.p, row_bytes * (size_t)s->n_local, hipMemcpyDeviceToDevice, c->stream));
        ANN_CHECK_HIP(c, hipMemsetAsync((char *)s->rows_send.p + row_bytes * (size_t)s->n_local, 0, bytes - row_bytes * (size_t)s->n_local,
                                        c->stream));
        *send = s->rows_send.p;
    }
    *recv = s->rows_recv.p;
    *bytes_per_rank = (int64_t)bytes;
    return ANNCHOR_OK;
}

// The anchor distances of this rank's rows -- the max-min sweeps left them in D [na][n_local] -- on their way to every
// rank: *send float [na][most] (D itself when the shard is the largest, a padded copy otherwise), *recv float
// [world][na][most]; the host all-gathers *bytes_per_rank bytes.  This is the "all-gather of anchor feature vectors":
// annchor_stream_rows_end then assembles D [na][total] from the gathered slices instead of recomputing the distances
// of every rank's rows on every rank.  Call between the anchor rounds and annchor_stream_rows_end.
extern "C" int annchor_stream_anchor_dists_begin(annchor_ctx *c, int32_t world, const int64_t *counts, void **send, void **recv,
                                                 int64_t *bytes_per_rank)
{
    if (!c || !counts || !send || !recv || !bytes_per_rank) return ANNCHOR_EINVAL;
    StreamState *s = ann_stream_state(c, false);
    ANN_REQUIRE(c, s && s->n_local > 0 && s->na > 0 && s->D.p, ANNCHOR_ESTATE, "anchor rounds first");
    ANN_REQUIRE(c, world >= 1 && world <= SH_MAX_WORLD, ANNCHOR_ELIMIT, "1 <= world <= %d", SH_MAX_WORLD);
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    int64_t most = 0;
    for (int r = 0; r < world; ++r) most = std::max(most, counts[r]);
    ANN_REQUIRE(c, s->n_local <= most, ANNCHOR_EINVAL, "this rank's shard (%lld rows) is larger than every count", (long long)s->n_local);
    const size_t bytes = sizeof(float) * (size_t)s->na * (size_t)most;
    ANN_TRY(ann_stream_reserve(c, s->D_recv, bytes * (size_t)world));
    if (s->n_local == most) {
        *send = s->D.p;
    } else {
        ANN_TRY(ann_stream_reserve(c, s->D_send, bytes));
        ANN_CHECK_HIP(c, hipMemsetAsync(s->D_send.p, 0, bytes, c->stream));
        ANN_CHECK_HIP(c, hipMemcpy2DAsync(s->D_send.p, sizeof(float) * (size_t)most, s->D.p, sizeof(float) * (size_t)s->n_local,
                                          sizeof(float) * (size_t)s->n_local, (size_t)s->na, hipMemcpyDeviceToDevice, c->stream));
        *send = s->D_send.p;
    }
    s->D_gathered = true;
    *recv = s->D_recv.p;
    *bytes_per_rank = (int64_t)bytes;
    return ANNCHOR_OK;
}

// After the all-gather: the context takes ALL rows (rank order, padding dropped; global_base 0 -- rows are numbered by
// their position in the concatenation).  Their anchor distances: assembled from the ranks' gathered slices
// (annchor_stream_anchor_dists_begin), or -- a host that did not exchange them -- recomputed from the anchors every rank holds.
extern "C" int annchor_stream_rows_end(annchor_ctx *c, int32_t world, const int64_t *counts)
{
    if (!c || !counts) return ANNCHOR_EINVAL;
    StreamState *s = ann_stream_state(c, false);
    ANN_REQUIRE(c, s && s->rows_recv.p && s->na > 0 && s->avecs.p, ANNCHOR_ESTATE, "annchor_stream_rows_begin / anchor rounds first");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    int64_t most = 0, total = 0;
    for (int r = 0; r < world; ++r) { most = std::max(most, counts[r]); total += counts[r]; }
    ANN_REQUIRE(c, total >= 1 && total < (1ll << 31), ANNCHOR_ELIMIT, "%lld rows in all", (long long)total);
    const size_t row_bytes = sizeof(float) * (size_t)s->dim;
    s->own_base = s->base;
    s->own_n = s->n_local;
    // (the rows' all-gather may still be running on the side stream -- annchor_comm_allgather_begin --: the engine stream waits
    // for it where it first touches the gathered rows: here when they must be compacted, else in annchor_stream_order_end)
    if (total == most * world) {
        std::swap(s->X, s->rows_recv);            // already contiguous
    } else {
        ANN_TRY(ann_comm_side_join(c));
        ANN_TRY(ann_stream_reserve(c, s->rows_all, row_bytes * (size_t)total));
        int64_t at = 0;
        for (int r = 0; r < world; ++r) {
            if (counts[r])
                ANN_CHECK_HIP(c, hipMemcpyAsync((char *)s->rows_all.p + row_bytes * (size_t)at,
                                                (const char *)s->rows_recv.p + row_bytes * (size_t)most * r, row_bytes * (size_t)counts[r],
                                                hipMemcpyDeviceToDevice, c->stream));
            at += counts[r];
        }
        std::swap(s->X, s->rows_all);
    }
    s->n_local = total;
    s->base = 0;
    c->nx = total;
    ANN_TRY(ann_stream_reserve(c, s->runmin, sizeof(float) * (size_t)total));
    static const bool recompute = getenv("ANNCHOR_SH_RECOMPUTE_D") && atoi(getenv("ANNCHOR_SH_RECOMPUTE_D")) != 0;
    const bool gathered = s->D_gathered && !recompute;
    s->D_gathered = false;
    if (gathered) {
        // D_recv [world][na][most] -> D [na][total]: one strided copy per rank (the send buffer may be D itself: the
        // all-gather has read it by now -- stream order -- so D can be re-reserved)
        // (D may have been the all-gather's send buffer and the collective runs asynchronously on this stream under RCCL: it must
        // have read D before D's block is let go -- the parked-block pool would hand it out again without a device-wide wait)
        ANN_CHECK_HIP(c, ann_stream_wait(c, __func__));
        ANN_TRY(ann_stream_reserve(c, s->D, sizeof(float) * (size_t)s->na * (size_t)total));
        ProfScope ps(c, "stream_anchor_dists_assemble", (double)total * s->na * 8.0);
        int64_t at = 0;
        for (int r = 0; r < world; ++r) {
            if (counts[r])
                ANN_CHECK_HIP(c, hipMemcpy2DAsync(s->D.as<float>() + at, sizeof(float) * (size_t)total,
                                                  s->D_recv.as<float>() + (size_t)r * s->na * (size_t)most, sizeof(float) * (size_t)most,
                                                  sizeof(float) * (size_t)counts[r], (size_t)s->na, hipMemcpyDeviceToDevice, c->stream));
            at += counts[r];
        }
        ANN_CHECK_HIP(c, hipGetLastError());
        return ANNCHOR_OK;
    }
    ANN_TRY(ann_comm_side_join(c));   // (the recomputation reads every row)
    ANN_TRY(ann_stream_reserve(c, s->D, sizeof(float) * (size_t)s->na * (size_t)total));
    {
        ProfScope ps(c, "stream_all_anchor_distances", (double)total * (s->dim * 4.0 + s->na * 4.0));
        const size_t lds = sizeof(float) * (size_t)s->na * s->dim;
        ANN_REQUIRE(c, lds <= 150 * 1024, ANNCHOR_ELIMIT, "%d anchors x %d dimensions do not fit the recomputation's LDS: exchange the anchor "
                    "distances instead (annchor_stream_anchor_dists_begin)", s->na, s->dim);
        const int blocks = (int)std::min<int64_t>(ann_blocks(total * 16, 256), 256 * 16);
        const bool vec = (s->dim & 3) == 0;
        const int nv = vec ? (s->dim + 63) / 64 : 0;
#define SH_LAUNCH(NV)                                                                                                                  \
    {                                                                                                                                  \
        ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)k_sh_all_anchor_dists<NV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        k_sh_all_anchor_dists<NV><<<blocks, 256, lds, c->stream>>>(s->X.as<float>(), total, s->dim, s->avecs.as<float>(), s->na, s->D.as<float>()); \
    }
        switch (nv) {
        case 0: SH_LAUNCH(0); break;
        case 1: SH_LAUNCH(1); break;
        case 2: SH_LAUNCH(2); break;
        case 3: SH_LAUNCH(3); break;
        case 4: SH_LAUNCH(4); break;
        default: SH_LAUNCH(0); break;   // (more than 256 dimensions: the scalar form)
        }
#undef SH_LAUNCH
    }
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}

// ------------------------------------------------------------------ join lists
extern "C" int annchor_stream_lists_all(annchor_ctx *c, int32_t world, int64_t bytes_per_rank, void **all)
{
    if (!c || !all || world < 1 || bytes_per_rank < 1) return ANNCHOR_EINVAL;
    StreamState *s = ann_stream_state(c, false);
    ANN_REQUIRE(c, s != nullptr, ANNCHOR_ESTATE, "no streamed build on this context");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    ANN_TRY(ann_stream_reserve(c, s->lists_all, (size_t)bytes_per_rank * (size_t)world));
    *all = s->lists_all.p;
    return ANNCHOR_OK;
}

// ------------------------------------------------------------------ finished rows -> owners
struct RouteTab {
    int64_t starts[SH_MAX_WORLD + 1];   // first position of every rank's shard in the concatenation
    int64_t bases[SH_MAX_WORLD];        // its first global row id
    int world;
};

__device__ __forceinline__ int sh_owner(const RouteTab *t, int64_t pos)
{
    int r = 0;
    while (r + 1 < t->world && pos >= t->starts[r + 1]) ++r;
    return r;
}

// per finished row: owner rank of its global id; per-destination counts
__global__ __launch_bounds__(256) void k_sh_route_count(const int64_t *__restrict__ perm, int64_t row_begin, int64_t rows,
                                                        const RouteTab *__restrict__ tab, int32_t *__restrict__ dest,
                                                        unsigned long long *__restrict__ cnt)
{
    __shared__ unsigned int h[SH_MAX_WORLD];
    if (threadIdx.x < SH_MAX_WORLD) h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) {
        const int64_t pos = perm[row_begin + r];
        const int d = pos < 0 ? -1 : sh_owner(tab, pos);
        dest[r] = d;
        if (d >= 0) atomicAdd(&h[d], 1u);
    }
    __syncthreads();
    if (threadIdx.x < tab->world && h[threadIdx.x]) atomicAdd(&cnt[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// slot of every finished row inside its destination's run of the send buffer: rows are ranked inside their workgroup through
// LDS, a workgroup reserves its rows' room with ONE global atomic per destination (one atomic per row on a handful of addresses
// serialises: 33 ms for 4 x 10^6 rows and two destinations)
__global__ __launch_bounds__(256) void k_sh_route_slot(const int32_t *__restrict__ dest, int64_t rows, unsigned long long *__restrict__ cursor,
                                                      int64_t *__restrict__ slot)
{
    __shared__ unsigned int cnt[SH_MAX_WORLD];
    __shared__ unsigned long long base[SH_MAX_WORLD];
    if (threadIdx.x < SH_MAX_WORLD) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int d = r < rows ? dest[r] : -1;
    const unsigned int mine = d >= 0 ? atomicAdd(&cnt[d], 1u) : 0u;
    __syncthreads();
    if (threadIdx.x < SH_MAX_WORLD && cnt[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], (unsigned long long)cnt[threadIdx.x]);
    __syncthreads();
    if (r < rows) slot[r] = d < 0 ? -1 : (int64_t)(base[d] + mine);
}

// record of a row: [global id, K neighbour ids (global; -1 = none), K distances (float64 bits)]
__global__ void k_sh_route_pack(const int64_t *__restrict__ perm, int64_t row_begin, int64_t rows, int K, const RouteTab *__restrict__ tab,
                                const int64_t *__restrict__ slot, const int64_t *__restrict__ idx, const float *__restrict__ dist,
                                int64_t *__restrict__ send)
{
    const int W = 1 + 2 * K;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * W) return;
    const int64_t r = t / W;
    const int e = (int)(t - r * W);
    const int64_t sl = slot[r];
    if (sl < 0) return;
    int64_t v;
    if (e == 0) {
        const int64_t pos = perm[row_begin + r];
        const int o = sh_owner(tab, pos);
        v = pos - tab->starts[o] + tab->bases[o];
    } else if (e <= K) {
        const int64_t pos = idx[r * K + e - 1];
        if (pos < 0) v = -1;
        else { const int o = sh_owner(tab, pos); v = pos - tab->starts[o] + tab->bases[o]; }
    } else {
        v = __double_as_longlong((double)dist[r * K + e - 1 - K]);
    }
    send[sl * W + e] = v;
}

__global__ void k_sh_route_fill(int64_t n, int64_t *__restrict__ oidx, double *__restrict__ odist)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) { oidx[t] = -1; odist[t] = INFINITY; }
}

__global__ void k_sh_route_scatter(const int64_t *__restrict__ recv, int64_t n_recv, int K, int64_t base, int64_t n_own,
                                   int64_t *__restrict__ oidx, double *__restrict__ odist, unsigned long long *__restrict__ bad)
{
    const int W = 1 + 2 * K, k = K + 1;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_recv * k) return;
    const int64_t r = t / k;
    const int e = (int)(t - r * k);
    const int64_t g = recv[r * W];
    const int64_t loc = g - base;
    if (loc < 0 || loc >= n_own) { if (e == 0) atomicAdd(bad, 1ull); return; }
    oidx[loc * k + e] = e == 0 ? g : recv[r * W + e];
    odist[loc * k + e] = e == 0 ? 0.0 : __longlong_as_double(recv[r * W + K + e]);
}

// Ends the build begun with annchor_stream_knn_begin for a row-sharded run: exact distances and final order of this
// rank's rows (its tile range of the global order), ids mapped from positions in the rank-ordered concatenation back to
// global row ids (starts int64 [world + 1], bases int64 [world]: host), records bucketed by owner rank.
// *send: int64 [rows][*record_words] grouped by destination in rank order (device); send_counts[world] (host): records
// per destination.
extern "C" int annchor_stream_route_begin(annchor_ctx *c, int32_t world, const int64_t *starts, const int64_t *bases, void **send,
                                          int64_t *send_counts, int64_t *record_words, int64_t *tile_evals)
{
    if (!c || !starts || !bases || !send || !send_counts || !record_words) return ANNCHOR_EINVAL;
    StreamState *s = ann_stream_state(c, false);
    ANN_REQUIRE(c, s && s->run, ANNCHOR_ESTATE, "annchor_stream_knn_begin not called");
    ANN_REQUIRE(c, world >= 1 && world <= SH_MAX_WORLD, ANNCHOR_ELIMIT, "1 <= world <= %d", SH_MAX_WORLD);
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    KnnArgs &a = *s->run;
    const int K = a.K, W = 1 + 2 * K;
    const int64_t rows = (int64_t)a.tile_count * ST_T, row_begin = (int64_t)a.tile_begin * ST_T;
    int64_t *d_idx = nullptr;
    float *d_dist = nullptr;
    int rc = ann_stream_knn_finish(c, s, a, s->run_perm, s->run_dimp, &d_idx, &d_dist, tile_evals);
    if (rc != ANNCHOR_OK) { ann_stream_free_run(s); return rc; }
    RouteTab tab;
    memset(&tab, 0, sizeof(tab));
    tab.world = world;
    for (int r = 0; r < world; ++r) { tab.starts[r] = starts[r]; tab.bases[r] = bases[r]; }
    tab.starts[world] = starts[world];
    ANN_TRY(ann_stream_reserve(c, s->route_tab, sizeof(RouteTab)));
    ANN_TRY(ann_stream_reserve(c, s->route_cnt, sizeof(unsigned long long) * (2 * SH_MAX_WORLD + 1)));
    ANN_TRY(ann_stream_reserve(c, s->route_slot, (sizeof(int64_t) + sizeof(int32_t)) * (size_t)rows));
    ANN_TRY(ann_stream_reserve(c, s->route_send, sizeof(int64_t) * (size_t)rows * W));
    ANN_CHECK_HIP(c, hipMemcpyAsync(s->route_tab.p, &tab, sizeof(tab), hipMemcpyHostToDevice, c->stream));
    ANN_CHECK_HIP(c, ann_stream_wait(c, __func__));   // `tab` lives on this stack frame
    unsigned long long *cnt = s->route_cnt.as<unsigned long long>(), *cursor = cnt + SH_MAX_WORLD;
    int64_t *slot = s->route_slot.as<int64_t>();
    int32_t *dest = reinterpret_cast<int32_t *>(slot + rows);
    ANN_CHECK_HIP(c, hipMemsetAsync(cnt, 0, sizeof(unsigned long long) * (2 * SH_MAX_WORLD + 1), c->stream));
    const RouteTab *dtab = s->route_tab.as<RouteTab>();
    k_sh_route_count<<<ann_blocks(rows, 256), 256, 0, c->stream>>>((const int64_t *)s->run_perm, row_begin, rows, dtab, dest, cnt);
    ANN_CHECK_HIP(c, hipGetLastError());
    unsigned long long hc[SH_MAX_WORLD];
    ANN_TRY(ann_d2h(c, hc, cnt, sizeof(unsigned long long) * (size_t)world));
    unsigned long long off[SH_MAX_WORLD];
    unsigned long long at = 0;
    for (int r = 0; r < world; ++r) { off[r] = at; at += hc[r]; send_counts[r] = (int64_t)hc[r]; }
    ANN_TRY(ann_h2d(c, cursor, off, sizeof(unsigned long long) * (size_t)world));
    {
        ProfScope ps(c, "stream_route_rows", (double)rows * (W * 8.0 + K * 12.0 + 24.0));
        k_sh_route_slot<<<ann_blocks(rows, 256), 256, 0, c->stream>>>(dest, rows, cursor, slot);
        k_sh_route_pack<<<ann_blocks(rows * W, 256), 256, 0, c->stream>>>((const int64_t *)s->run_perm, row_begin, rows, K, dtab, slot, d_idx,
                                                                         d_dist, s->route_send.as<int64_t>());
    }
    ANN_CHECK_HIP(c, hipGetLastError());
    s->emit_k = K + 1;
    *send = s->route_send.p;
    *record_words = W;
    ann_stream_free_run(s);
    return ANNCHOR_OK;
}

extern "C" int annchor_stream_route_recv(annchor_ctx *c, int64_t n_recv, void **recv)
{
    if (!c || !recv || n_recv < 0) return ANNCHOR_EINVAL;
    StreamState *s = ann_stream_state(c, false);
    ANN_REQUIRE(c, s && s->emit_k > 0, ANNCHOR_ESTATE, "annchor_stream_route_begin first");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    ANN_TRY(ann_stream_reserve(c, s->route_recv, sizeof(int64_t) * (size_t)std::max<int64_t>(n_recv, 1) * (2 * s->emit_k - 1)));
    *recv = s->route_recv.p;
    return ANNCHOR_OK;
}

// The received records (the receive buffer of annchor_stream_route_recv after the all-to-all) scattered into this rank's
// own row order: graph rows [n_own][k] on the device (kept for annchor_stream_graph_device, padded with (-1, inf) rows up
// to rows_padded) and on the host (ng_idx int64 [n_own][k], ng_dist float64 [n_own][k]; column 0 = self).
extern "C" int annchor_stream_route_end(annchor_ctx *c, int64_t n_recv, int64_t rows_padded, int64_t *ng_idx, double *ng_dist)
{
    if (!c || !ng_idx || !ng_dist) return ANNCHOR_EINVAL;
    StreamState *s = ann_stream_state(c, false);
    ANN_REQUIRE(c, s && s->emit_k > 0 && s->route_recv.p, ANNCHOR_ESTATE, "annchor_stream_route_recv first");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    const int k = s->emit_k, K = k - 1;
    const int64_t n_own = s->own_n;
    ANN_REQUIRE(c, n_recv == n_own, ANNCHOR_ESTATE, "%lld rows arrived for a shard of %lld", (long long)n_recv, (long long)n_own);
    if (rows_padded < n_own) rows_padded = n_own;
    ANN_TRY(ann_stream_reserve(c, s->emit_idx, sizeof(int64_t) * (size_t)rows_padded * k));
    ANN_TRY(ann_stream_reserve(c, s->emit_dist, sizeof(double) * (size_t)rows_padded * k));
    s->emit_rows = rows_padded;
    unsigned long long *bad = s->route_cnt.as<unsigned long long>() + 2 * SH_MAX_WORLD;
    k_sh_route_fill<<<ann_blocks(rows_padded * k, 256), 256, 0, c->stream>>>(rows_padded * k, s->emit_idx.as<int64_t>(), s->emit_dist.as<double>());
    {
        ProfScope ps(c, "stream_route_scatter", (double)n_recv * ((2 * K + 1) * 8.0 + k * 16.0));
        k_sh_route_scatter<<<ann_blocks(n_recv * k, 256), 256, 0, c->stream>>>(s->route_recv.as<int64_t>(), n_recv, K, s->own_base, n_own,
                                                                              s->emit_idx.as<int64_t>(), s->emit_dist.as<double>(), bad);
    }
    ANN_CHECK_HIP(c, hipGetLastError());
    unsigned long long hb = 0;
    ANN_TRY(ann_d2h(c, &hb, bad, sizeof(hb)));
    ANN_REQUIRE(c, hb == 0, ANNCHOR_ESTATE, "%llu received rows do not belong to this shard", hb);
    ANN_TRY(ann_d2h(c, ng_idx, s->emit_idx.p, sizeof(int64_t) * (size_t)n_own * k));
    return ann_d2h(c, ng_dist, s->emit_dist.p, sizeof(double) * (size_t)n_own * k);
}

// the graph rows annchor_stream_route_end left on the device: int64 / float64 [rows_padded][k]
extern "C" int annchor_stream_graph_device(annchor_ctx *c, void **idx, void **dist, int64_t *rows_padded, int32_t *k)
{
    if (!c || !idx || !dist || !rows_padded || !k) return ANNCHOR_EINVAL;
    StreamState *s = ann_stream_state(c, false);
    ANN_REQUIRE(c, s && s->emit_rows > 0 && s->emit_idx.p, ANNCHOR_ESTATE, "no routed graph on this context");
    *idx = s->emit_idx.p; *dist = s->emit_dist.p; *rows_padded = s->emit_rows; *k = s->emit_k;
    return ANNCHOR_OK;
}
