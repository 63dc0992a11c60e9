// picker.hip -- anchor selection and the anchor-distance matrix D.
//
// Replaces MaxMinAnchorPicker.get_anchors (reference annchor/pickers.py:18-52,
// with np_min of annchor/utils.py:47-49), SelectedAnchorPicker / RandomAnchorPicker
// (pickers.py:86-128) and accepts the (A, D, evals) tuple of any other picker.
//
// D is kept anchor-major in HBM, Dt[a][point]: every later kernel walks points with
// consecutive lanes, so all D traffic is coalesced.  The max-min loop is enqueued
// without any host round trip: round r's one-to-all metric sweep reads its anchor
// index from device memory, where round r-1's arg-max reduction left it.
#include "common.h"
#include <memory>

#define RED_THREADS 256

__global__ __launch_bounds__(RED_THREADS) void k_runmin_argmax(const double *__restrict__ row, double *__restrict__ runmin,
                                                              int64_t nx, int reset, double *__restrict__ redval,
                                                              int *__restrict__ redidx)
{
    double bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nx; j += (int64_t)gridDim.x * blockDim.x) {
        double v = row[j];
        if (!reset) v = fmin(runmin[j], v);
        runmin[j] = v;
        argmax_combine(bv, bi, v, (int)j);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double ov = __shfl_xor(bv, off);
        int oi = __shfl_xor(bi, off);
        argmax_combine(bv, bi, ov, oi);
    }
    __shared__ double sv[RED_THREADS / 64];
    __shared__ int si[RED_THREADS / 64];
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < RED_THREADS / 64; ++w) argmax_combine(bv, bi, sv[w], si[w]);
        redval[blockIdx.x] = bv;
        redidx[blockIdx.x] = bi;
    }
}

// small data sets (one workgroup covers them): running minimum + arg-max + next anchor in ONE
// launch -- the rounds are a chain of dependent launches, each one costs a dispatch gap
__global__ __launch_bounds__(1024) void k_runmin_argmax_one(const double *__restrict__ row, double *__restrict__ runmin, int64_t nx,
                                                           int reset, int32_t *__restrict__ next_anchor)
{
    double bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int64_t j = threadIdx.x; j < nx; j += blockDim.x) {
        double v = row[j];
        if (!reset) v = fmin(runmin[j], v);
        runmin[j] = v;
        argmax_combine(bv, bi, v, (int)j);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double ov = __shfl_xor(bv, off);
        int oi = __shfl_xor(bi, off);
        argmax_combine(bv, bi, ov, oi);
    }
    __shared__ double sv[16];
    __shared__ int si[16];
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) argmax_combine(bv, bi, sv[w], si[w]);
        *next_anchor = bi;
    }
}

__global__ void k_argmax_final(const double *__restrict__ redval, const int *__restrict__ redidx, int nblocks,
                               int32_t *__restrict__ next_anchor)
{
    double bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int b = threadIdx.x; b < nblocks; b += 64) argmax_combine(bv, bi, redval[b], redidx[b]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double ov = __shfl_xor(bv, off);
        int oi = __shfl_xor(bi, off);
        argmax_combine(bv, bi, ov, oi);
    }
    if (threadIdx.x == 0) *next_anchor = bi;
}

__global__ void k_anchor_rank(const int32_t *__restrict__ A, int nA, int32_t *__restrict__ rank, int64_t nx)
{
    // sequential on purpose: a later occurrence overrides an earlier one, as the
    // reference's `for a in self.A` loops do (annchor.py:288-289, 369-372)
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int r = 0; r < nA; ++r)
            if (A[r] >= 0 && A[r] < nx) rank[A[r]] = r;
}

__global__ void k_fill_i32(int32_t *p, int32_t v, int64_t n)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) p[t] = v;
}

static int anchor_flags_from_device_A(annchor_ctx *c)
{
    ANN_TRY(ann_reserve(c, c->anchorRank, sizeof(int32_t) * (size_t)c->nx));
    k_fill_i32<<<ann_blocks(c->nx, 256), 256, 0, c->stream>>>(c->anchorRank.as<int32_t>(), -1, c->nx);
    if (c->nA > 0) k_anchor_rank<<<1, 64, 0, c->stream>>>(c->A.as<int32_t>(), c->nA, c->anchorRank.as<int32_t>(), c->nx);
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}

static int begin_anchors(annchor_ctx *c, int32_t na)
{
    ANN_REQUIRE(c, c->nx > 0, ANNCHOR_EINVAL, "no data set bound");
    ANN_REQUIRE(c, na >= 1 && na <= ANN_MAX_ANCHORS, ANNCHOR_ELIMIT, "n_anchors=%d: this build supports 1..%d", na, ANN_MAX_ANCHORS);
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    c->na = na;
    c->n = 0; c->have_bitmap = false;
    c->have_features = c->have_RA = false; c->sel_prepared = false;
    ANN_TRY(ann_reserve(c, c->Dt, sizeof(double) * (size_t)na * (size_t)c->nx));
    ANN_TRY(ann_reserve(c, c->A, sizeof(int32_t) * (size_t)(na + 1)));
    return ANNCHOR_OK;
}

extern "C" int annchor_pick_anchors_maxmin(annchor_ctx *c, int32_t na, int64_t first)
{
    if (!c) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->metric != ANNCHOR_METRIC_NONE, ANNCHOR_EINVAL, "max-min picking needs a device metric");
    ANN_REQUIRE(c, first >= 0 && first < c->nx, ANNCHOR_EINVAL, "first anchor %lld out of range", (long long)first);
    ANN_TRY(begin_anchors(c, na));
    c->nA = na;
    const int64_t nx = c->nx;
    int rblocks = (int)((nx + RED_THREADS * 4 - 1) / (RED_THREADS * 4));
    if (rblocks > 1024) rblocks = 1024;
    ANN_TRY(ann_reserve(c, c->runmin, sizeof(double) * (size_t)nx));
    ANN_TRY(ann_reserve(c, c->redval, sizeof(double) * (size_t)rblocks));
    ANN_TRY(ann_reserve(c, c->redidx, sizeof(int) * (size_t)rblocks));
    int32_t f = (int32_t)first;
    if (c->metric == ANNCHOR_METRIC_LEVENSHTEIN) {
        // every round in one launch where the data set allows it (lev.hip: k_lev_ap); A and anchorRank come out of it
        ANN_CHECK_HIP(c, hipEventRecord(c->call_a, c->stream));
        bool done = false;
        ANN_TRY(ann_lev_anchor_rounds(c, na, f, &done));
        if (done) {
            ANN_CHECK_HIP(c, hipEventRecord(c->call_b, c->stream));
            c->call_timed = true;
            return ANNCHOR_OK;
        }
    }
    ANN_TRY(ann_h2d(c, c->A.p, &f, sizeof f));
    ANN_CHECK_HIP(c, hipEventRecord(c->call_a, c->stream));
    bool fused_prev = false;   // the launch of round r picked its own anchor from row r - 1
    // (the rounds of a fusing metric are nothing but its launches, back to back: one event pair around the run)
    const char *family = c->metric == ANNCHOR_METRIC_LEVENSHTEIN ? "levenshtein_pairs" : nullptr;
    std::unique_ptr<ProfGroup> group;
    for (int r = 0; r < na; ++r) {
        if (r == 1 && fused_prev && family) group.reset(new ProfGroup(c, family));
        PairSource src;
        src.anchor = c->A.as<int32_t>() + r;
        src.n = nx;
        double *row = c->Dt.as<double>() + (size_t)r * nx;
        bool fused = false;
        if (r >= 1 && fused_prev) {   // (round 0 decides whether this metric fuses at all)
            src.pick_row = row - nx;
            src.pick_runmin = c->runmin.as<double>();
            src.pick_out = c->A.as<int32_t>() + r;
            src.pick_reset = r - 1 <= 1 ? 1 : 0;   // pickers.py:47-50: min over all rows for r == 0, over rows 1..r afterwards
        }
        src.pick_fused = &fused;
        ANN_TRY(ann_metric_launch(c, src, row, nullptr, nullptr));
        if (r == 0) fused_prev = fused;             // a capable launch reports it even without a pick request
        else if (fused_prev) ANN_REQUIRE(c, fused, ANNCHOR_ESTATE, "metric launch stopped fusing the anchor pick");
        if (r + 1 < na && !fused_prev) {
            ProfScope ps(c, "maxmin_argmax", (double)nx * 24);
            // pickers.py:47-50: min over all rows for r == 0, over rows 1..r afterwards
            if (nx <= 16384)
                k_runmin_argmax_one<<<1, 1024, 0, c->stream>>>(row, c->runmin.as<double>(), nx, r <= 1 ? 1 : 0,
                                                              c->A.as<int32_t>() + r + 1);
            else {
                k_runmin_argmax<<<rblocks, RED_THREADS, 0, c->stream>>>(row, c->runmin.as<double>(), nx, r <= 1 ? 1 : 0,
                                                                       c->redval.as<double>(), c->redidx.as<int>());
                k_argmax_final<<<1, 64, 0, c->stream>>>(c->redval.as<double>(), c->redidx.as<int>(), rblocks,
                                                       c->A.as<int32_t>() + r + 1);
            }
        }
    }
    group.reset();
    ANN_CHECK_HIP(c, hipEventRecord(c->call_b, c->stream));
    c->call_timed = true;
    ANN_CHECK_HIP(c, hipGetLastError());
    return anchor_flags_from_device_A(c);
}

extern "C" int annchor_pick_anchors_selected(annchor_ctx *c, const int64_t *A, int32_t na)
{
    if (!c || !A) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->metric != ANNCHOR_METRIC_NONE, ANNCHOR_EINVAL, "selected-anchor picking needs a device metric");
    ANN_TRY(begin_anchors(c, na));
    c->nA = na;
    std::vector<int32_t> a32((size_t)na);
    for (int r = 0; r < na; ++r) {
        ANN_REQUIRE(c, A[r] >= 0 && A[r] < c->nx, ANNCHOR_EINVAL, "anchor %lld out of range", (long long)A[r]);
        a32[(size_t)r] = (int32_t)A[r];
    }
    ANN_TRY(ann_h2d(c, c->A.p, a32.data(), sizeof(int32_t) * (size_t)na));
    ANN_CHECK_HIP(c, hipEventRecord(c->call_a, c->stream));
    for (int r = 0; r < na; ++r) {
        PairSource src;
        src.anchor = c->A.as<int32_t>() + r;
        src.n = c->nx;
        ANN_TRY(ann_metric_launch(c, src, c->Dt.as<double>() + (size_t)r * c->nx, nullptr, nullptr));
    }
    ANN_CHECK_HIP(c, hipEventRecord(c->call_b, c->stream));
    c->call_timed = true;
    return anchor_flags_from_device_A(c);
}

__global__ void k_transpose_f64(const double *__restrict__ in, double *__restrict__ out, int64_t rows, int64_t cols)
{
    // in [rows][cols] -> out [cols][rows]; tiny matrices (one side <= 64), keep it simple
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * cols) return;
    int64_t r = t / cols, cc = t - r * cols;
    out[cc * rows + r] = in[t];
}

extern "C" int annchor_set_anchor_distances(annchor_ctx *c, const double *D, int32_t na, const int64_t *A, int32_t nA)
{
    if (!c || !D || (nA > 0 && !A)) return ANNCHOR_EINVAL;
    ANN_TRY(begin_anchors(c, na));
    ANN_REQUIRE(c, nA >= 0 && nA <= na, ANNCHOR_EINVAL, "len(A)=%d exceeds n_anchors=%d", nA, na);
    c->nA = nA;
    const size_t bytes = sizeof(double) * (size_t)na * (size_t)c->nx;
    ANN_TRY(ann_reserve(c, c->stage_in, bytes));
    ANN_TRY(ann_h2d(c, c->stage_in.p, D, bytes));
    k_transpose_f64<<<ann_blocks((int64_t)na * c->nx, 256), 256, 0, c->stream>>>(c->stage_in.as<double>(),
                                                                                c->Dt.as<double>(), c->nx, na);
    std::vector<int32_t> a32((size_t)nA + 1);
    for (int r = 0; r < nA; ++r) {
        ANN_REQUIRE(c, A[r] >= 0 && A[r] < c->nx, ANNCHOR_EINVAL, "anchor %lld out of range", (long long)A[r]);
        a32[(size_t)r] = (int32_t)A[r];
    }
    if (nA > 0) ANN_TRY(ann_h2d(c, c->A.p, a32.data(), sizeof(int32_t) * (size_t)nA));
    ANN_CHECK_HIP(c, hipGetLastError());
    return anchor_flags_from_device_A(c);
}

// used by annchor_download(ANNCHOR_F_D): Dt [na][nx] -> D [nx][na]
int ann_download_D(annchor_ctx *c, double *dst)
{
    const size_t bytes = sizeof(double) * (size_t)c->na * (size_t)c->nx;
    ANN_TRY(ann_reserve(c, c->stage_out, bytes));
    k_transpose_f64<<<ann_blocks((int64_t)c->na * c->nx, 256), 256, 0, c->stream>>>(c->Dt.as<double>(),
                                                                                   c->stage_out.as<double>(), c->na, c->nx);
    ANN_CHECK_HIP(c, hipGetLastError());
    return ann_d2h(c, dst, c->stage_out.p, bytes);
}
