// streamed.hip -- the streamed (tile-granular) form of the path for Euclidean data at
// N >> 10^4, where the pair-list form of the reference cannot exist (its candidate list
// alone would be ~10^11 pairs, SURVEY.md section 7 hard part 3).
//
// Same stages as Annchor.fit() (reference annchor/annchor.py:532-623), lifted from pairs
// to 128-point tiles:
//   anchors   max-min picking, D = distances to the anchors        (pickers.py:18-52)
//   locality  points are ordered by (nearest anchor, distance to it); consecutive 128
//             points form a tile with per-anchor distance intervals [lo, hi]
//   bounds    the triangle inequality of utils.py:274-301 on intervals:
//             lb(I, J) = max_a max(lo_I[a] - hi_J[a], lo_J[a] - hi_I[a], 0)
//   refine    for each row tile the column tiles are visited in ascending lb; the exact
//             metric is evaluated for a whole tile pair as a float32 MFMA GEMM
//             (|x|^2 + |y|^2 - 2 x.y); a tile is skipped as soon as lb^2 >= the worst
//             current k-th squared distance of the row tile; at most p_work * (#tiles)
//             column tiles are evaluated per row tile (the work budget of annchor.py:438-442)
//   top-k     per-row k smallest, kept in LDS while streaming          (utils.py:383-429)
// For float32 inputs evaluating a pair exactly costs 2*d flops on the matrix cores while a
// per-pair triangle bound costs ~5*n_anchors VALU ops at the same peak rate -- so the
// regression / ECDF ranking of single pairs (annchor.py:345-473) has no place here: whole
// tiles are ranked by their bound instead.  With the budget not binding the result is the
// exact k-NN graph.
//
// MFMA mapping (v_mfma_f32_32x32x2_f32, exact f32): a 256-thread workgroup owns a row tile;
// each of its 4 waves keeps its 32 rows x d operand in registers for the whole kernel
// (d/2 VGPRs), column tiles are streamed through LDS in 32-column slabs (row stride d+1
// floats: bank-conflict-free operand reads), accumulators are compared against per-row
// thresholds in LDS and only the survivors are inserted into the per-row lists.

#include "streamed.h"

static std::vector<std::pair<annchor_ctx *, StreamState *>> g_states;

static StreamState *state_of(annchor_ctx *c, bool create)
{
    for (auto &p : g_states)
        if (p.first == c) return p.second;
    if (!create) return nullptr;
    StreamState *s = new StreamState();
    g_states.push_back({c, s});
    return s;
}

void ann_stream_release(annchor_ctx *c)
{
    for (size_t i = 0; i < g_states.size(); ++i)
        if (g_states[i].first == c) {
            StreamState *s = g_states[i].second;
            DevBuf *bufs[] = {&s->X, &s->keys, &s->keys2, &s->vals, &s->vals2, &s->cubtmp, &s->Xs, &s->rs, &s->perm, &s->lo,
                              &s->hi, &s->mid, &s->avec, &s->runmin, &s->red_val, &s->red_idx, &s->D, &s->out_d2, &s->out_col, &s->evals,
                              &s->scr_key, &s->scr_lb, &s->emit_idx, &s->emit_dist, &s->Dt, &s->eval_bits, &s->out_d2b, &s->out_colb,
                              &s->ucand, &s->ucount, &s->rev_cnt, &s->rev_ptr, &s->rev_edges, &s->cand, &s->cand_all, &s->avecs, &s->A_dev,
                              &s->rows_send, &s->rows_recv, &s->rows_all, &s->lists_all, &s->route_tab, &s->route_cnt, &s->route_slot,
                              &s->route_send, &s->route_recv, &s->Xb, &s->rsb, &s->cvec, &s->order_all, &s->rev_all, &s->rev_slice, &s->D_send, &s->D_recv, &s->scr_cl};
            for (DevBuf *b : bufs)
                if (b->p && !b->in_arena) ann_dev_free(c, b->p, b->cap);
            ann_stream_free_run(s);
            delete s;
            g_states.erase(g_states.begin() + (long)i);
            return;
        }
}

static int sreserve(annchor_ctx *c, DevBuf &b, size_t bytes)
{
    // streamed buffers are large: always individual allocations
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return ANNCHOR_OK;
    if (b.p && !b.in_arena) ann_dev_free(c, b.p, b.cap);
    b.p = nullptr; b.cap = 0; b.in_arena = false;
    const size_t want0 = (bytes + 255) & ~(size_t)255;
    size_t got = 0;
    ANN_TRY(ann_dev_alloc(c, &b.p, want0, &got));
    b.cap = got;
    return ANNCHOR_OK;
}

// 32 / 64 / 128 / 256, then multiples of 128 up to 1024 (the k-blocked kernel of knnbk.hip)
static int padded_dim(int dim) { return dim <= 32 ? 32 : dim <= 64 ? 64 : dim <= 128 ? 128 : dim <= 1024 ? (dim + 127) & ~127 : -1; }
StreamState *ann_stream_state(annchor_ctx *c, bool create) { return state_of(c, create); }
// (the column arrays travel through the C-ABI as bare pointers -- a query engine streams another context's columns --
// so the split copy is found from the float32 array it was made from)
bool ann_stream_split_of(const void *Xs, const uint16_t **Xb, const float **rsb, const float **cvec)
{
    for (auto &p : g_states)
        if (p.second->Xs.p == Xs && p.second->Xb.p && p.second->rsb.p && p.second->cvec.p) {
            *Xb = p.second->Xb.as<uint16_t>(); *rsb = p.second->rsb.as<float>(); *cvec = p.second->cvec.as<float>();
            return true;
        }
    *Xb = nullptr; *rsb = nullptr; *cvec = nullptr;
    return false;
}
int ann_stream_reserve(annchor_ctx *c, DevBuf &b, size_t bytes) { return sreserve(c, b, bytes); }
int ann_stream_padded_dim(int dim) { return padded_dim(dim); }

// ------------------------------------------------------------------ bind
extern "C" int annchor_stream_bind(annchor_ctx *c, const float *X, int64_t n_local, int32_t dim, int64_t global_base,
                                   int32_t x_on_device)
{
    if (!c || !X) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, n_local >= 1 && n_local < (1ll << 31), ANNCHOR_ELIMIT, "n_local=%lld out of range", (long long)n_local);
    ANN_REQUIRE(c, padded_dim(dim) > 0, ANNCHOR_ELIMIT, "streamed form supports dim <= 1024 (got %d)", dim);
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    StreamState *s = state_of(c, true);
    s->n_local = n_local; s->dim = dim; s->dimp = padded_dim(dim); s->base = global_base; s->na = 0;
    const size_t bytes = sizeof(float) * (size_t)n_local * dim;
    ANN_TRY(sreserve(c, s->X, bytes));
    ANN_CHECK_HIP(c, hipMemcpyAsync(s->X.p, X, bytes, x_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    ANN_CHECK_HIP(c, ann_stream_wait(c, __func__));
    ANN_TRY(sreserve(c, s->avec, sizeof(float) * 1024));
    ANN_TRY(sreserve(c, s->runmin, sizeof(float) * (size_t)n_local));
    c->metric = ANNCHOR_METRIC_EUCLIDEAN_F32;
    c->nx = n_local;
    return ANNCHOR_OK;
}

// ------------------------------------------------------------ anchor rounds
// one-to-all distances from a vector: 16 lanes per row, 16-byte loads; HBM bound
// (n_local * dim * 4 bytes per round)
__global__ __launch_bounds__(256) void k_st_one_to_all(const float *__restrict__ X, int64_t n, int dim,
                                                      const float *__restrict__ av, float *__restrict__ out)
{
    const int sub = threadIdx.x & 15;
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    float acc = 0.f;
    if (i < n) {
        const float *x = X + (size_t)i * dim;
        if ((dim & 3) == 0) {
            for (int k = sub * 4; k < dim; k += 64) {
                const float4 u = *reinterpret_cast<const float4 *>(x + k), w = *reinterpret_cast<const float4 *>(av + k);
                float d0 = u.x - w.x, d1 = u.y - w.y, d2 = u.z - w.z, d3 = u.w - w.w;
                acc += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
            }
        } else {
            for (int k = sub; k < dim; k += 16) { float d = x[k] - av[k]; acc += d * d; }
        }
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 16);
    if (i < n && sub == 0) out[i] = sqrtf(acc);
}

__global__ __launch_bounds__(256) void k_st_runmin_argmax(const float *__restrict__ row, float *__restrict__ runmin, int64_t n,
                                                         int reset, float *__restrict__ redval, int64_t *__restrict__ redidx)
{
    float bv = -INFINITY;
    int64_t bi = 0x7fffffffffffffffll;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        float v = row[j];
        if (!reset) v = fminf(runmin[j], v);
        runmin[j] = v;
        if (v > bv || (v == bv && j < bi)) { bv = v; bi = j; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        float ov = __shfl_xor(bv, off);
        int64_t oi = __shfl_xor(bi, off);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    __shared__ float sv[4];
    __shared__ int64_t si[4];
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
        redval[blockIdx.x] = bv;
        redidx[blockIdx.x] = bi;
    }
}

// One max-min round on the local shard (pickers.py:44-50): distances of all local rows to
// `anchor_vec`, running-min update (reset for rounds 0 and 1, as the reference's D[1:]
// quirk demands) and the local arg-max (value, local index; first index on ties).
int ann_stream_sweep(annchor_ctx *c, StreamState *s, const float *avec_dev, int round, int *n_partials)
{
    const int64_t n = s->n_local;
    const int rblocks = (int)std::min<int64_t>(1024, (n + 1023) / 1024);
    ANN_TRY(sreserve(c, s->red_val, sizeof(float) * 1024));
    ANN_TRY(sreserve(c, s->red_idx, sizeof(int64_t) * 1024));
    float *row = s->D.as<float>() + (size_t)round * n;
    {
        ProfScope ps(c, "stream_anchor_one_to_all", (double)n * (s->dim * 4.0 + 4.0));
        k_st_one_to_all<<<ann_blocks(n * 16, 256), 256, 0, c->stream>>>(s->X.as<float>(), n, s->dim, avec_dev, row);
    }
    k_st_runmin_argmax<<<rblocks, 256, 0, c->stream>>>(row, s->runmin.as<float>(), n, round <= 1 ? 1 : 0, s->red_val.as<float>(),
                                                      s->red_idx.as<int64_t>());
    ANN_CHECK_HIP(c, hipGetLastError());
    *n_partials = rblocks;
    return ANNCHOR_OK;
}

extern "C" int annchor_stream_anchor_round(annchor_ctx *c, const float *anchor_vec, int32_t round, int32_t n_anchors,
                                           double *local_max, int64_t *local_arg)
{
    if (!c || !anchor_vec || !local_max || !local_arg) return ANNCHOR_EINVAL;
    StreamState *s = state_of(c, false);
    ANN_REQUIRE(c, s && s->n_local > 0, ANNCHOR_EINVAL, "annchor_stream_bind first");
    ANN_REQUIRE(c, n_anchors >= 1 && n_anchors <= 64 && round >= 0 && round < n_anchors, ANNCHOR_ELIMIT, "bad anchor round");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    if (round == 0) {
        s->na = n_anchors;
        ANN_TRY(sreserve(c, s->D, sizeof(float) * (size_t)n_anchors * (size_t)s->n_local));
    }
    ANN_CHECK_HIP(c, hipMemcpyAsync(s->avec.p, anchor_vec, sizeof(float) * (size_t)s->dim, hipMemcpyHostToDevice, c->stream));
    int rblocks = 0;
    ANN_TRY(ann_stream_sweep(c, s, s->avec.as<float>(), round, &rblocks));
    std::vector<float> hv((size_t)rblocks);
    std::vector<int64_t> hi((size_t)rblocks);
    ANN_TRY(ann_d2h(c, hv.data(), s->red_val.p, sizeof(float) * (size_t)rblocks));
    ANN_TRY(ann_d2h(c, hi.data(), s->red_idx.p, sizeof(int64_t) * (size_t)rblocks));
    float bv = -INFINITY;
    int64_t bi = 0;
    for (int b = 0; b < rblocks; ++b)
        if (hv[(size_t)b] > bv || (hv[(size_t)b] == bv && hi[(size_t)b] < bi)) { bv = hv[(size_t)b]; bi = hi[(size_t)b]; }
    *local_max = (double)bv;
    *local_arg = bi;
    return ANNCHOR_OK;
}

extern "C" int annchor_stream_get_row(annchor_ctx *c, int64_t local_idx, float *out)
{
    if (!c || !out) return ANNCHOR_EINVAL;
    StreamState *s = state_of(c, false);
    ANN_REQUIRE(c, s && local_idx >= 0 && local_idx < s->n_local, ANNCHOR_EINVAL, "row out of range");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    ANN_TRY(ann_comm_side_join(c));
    return ann_d2h(c, out, s->X.as<float>() + (size_t)local_idx * s->dim, sizeof(float) * (size_t)s->dim);
}

// ---------------------------------------------------------------- ordering
// ---- locality ordering: balanced k-d splits in anchor-distance space.
// Level l cuts the current order into 2^l equal position ranges ("segments"); every segment
// is sorted along the anchor coordinate on which its points spread most, so that after
// ceil(log2(#tiles)) levels each 128-row tile is a small box in several anchor coordinates:
// tight [lo, hi] intervals, hence strong triangle bounds between tiles.
__host__ __device__ __forceinline__ int st_seg_of(int64_t p, int64_t n, int level) { return (int)((p << level) / n); }

// point-major copy of the anchor distances: a point's whole anchor vector is one contiguous
// (16-byte aligned) run, so the per-level gathers through `order` touch one or two cache lines
// per point instead of one line per (point, anchor)
__global__ void k_st_transpose_D(const float *__restrict__ D, int64_t n, int na, int nap, float *__restrict__ Dt)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * nap) return;
    const int64_t p = t / nap;
    const int a = (int)(t - p * nap);
    Dt[t] = a < na ? D[(size_t)a * n + p] : 0.f;
}

// Split coordinate of every segment of a level: the anchor whose distances vary most inside the
// segment.  One workgroup per segment, no atomics (the sums are reduced in a fixed order, so the
// choice -- and with it the tiling -- is the same on every run): thread = (sample lane, anchor), a
// point's anchor vector is one coalesced 128 / 256-byte read of the point-major copy; segments longer
// than ST_SPLIT_SAMPLE points are judged on that many evenly spaced members.
#ifndef ST_SPLIT_SAMPLE
#define ST_SPLIT_SAMPLE 2048
#endif
__global__ __launch_bounds__(1024) void k_st_split_coord(const float *__restrict__ Dt, int nap, const uint32_t *__restrict__ order,
                                                       int64_t n, int na, int level, int seg0, int32_t *__restrict__ coord)
{
    __shared__ double s1[1024], s2[1024];
    const int sgm = seg0 + blockIdx.x;
    const int64_t b = ((int64_t)sgm * n + (1ll << level) - 1) >> level, e = ((int64_t)(sgm + 1) * n + (1ll << level) - 1) >> level;
    const int64_t len = e - b;
    const int A = nap <= 32 ? 32 : 64;        // anchors per sample lane group (na <= 64)
    const int G = 1024 / A;                    // sample lanes
    const int an = threadIdx.x % A, g = threadIdx.x / A;
    const int64_t m = len < ST_SPLIT_SAMPLE ? len : ST_SPLIT_SAMPLE;
    double a1 = 0.0, a2 = 0.0;
    if (an < na) {
        // a lane's samples are a chain of two dependent gathers each (order[p], then the point's anchor distance): eight samples'
        // reads are issued together -- one at a time a top-level segment (one workgroup, 64 samples per lane) took ~200 us of pure
        // memory latency per level -- and summed in the order they always were (the choice of the split anchor stays the same)
        for (int64_t t0 = g; t0 < m; t0 += 8 * (int64_t)G) {
            uint32_t src[8];
            float v8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t t = t0 + (int64_t)u * G;
                const int64_t p = b + (m == len ? min(t, m - 1) : (min(t, m - 1) * len) / m);
                src[u] = order[p];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) v8[u] = Dt[(size_t)src[u] * nap + an];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (t0 + (int64_t)u * G < m) {
                    const double v = (double)v8[u];
                    a1 += v;
                    a2 += v * v;
                }
        }
    }
    s1[threadIdx.x] = a1;
    s2[threadIdx.x] = a2;
    __syncthreads();
    if (threadIdx.x < A) {   // fixed-order reduction over the sample lanes
        double t1 = 0.0, t2 = 0.0;
        for (int q = 0; q < G; ++q) { t1 += s1[q * A + threadIdx.x]; t2 += s2[q * A + threadIdx.x]; }
        const double cnt = (double)(m > 0 ? m : 1);
        const double mean = t1 / cnt;
        s1[threadIdx.x] = threadIdx.x < na ? t2 / cnt - mean * mean : -1.0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double best = -1.0;
        int ba = 0;
        for (int a = 0; a < na; ++a)
            if (s1[a] > best) { best = s1[a]; ba = a; }   // first maximum
        coord[sgm] = ba;
    }
}

__global__ void k_st_level_keys(const float *__restrict__ D, const uint32_t *__restrict__ order, int64_t n, int level,
                                const int32_t *__restrict__ coord, unsigned long long *__restrict__ keys, int64_t p0, int64_t p1)
{
    const int64_t p = p0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= p1) return;
    const int sgm = st_seg_of(p, n, level);
    const float v = D[(size_t)coord[sgm] * n + order[p]];
    keys[p] = ((unsigned long long)sgm << 32) | (unsigned long long)__float_as_uint(v);
}

// A level of the k-d order whose segments are short (<= ST_SEG_LDS rows): one workgroup per segment sorts it in LDS by (distance to
// the segment's split anchor, position) -- the position makes the bitonic network stable, so the order is the radix sort's, bit for
// bit -- and writes it back in place.  One launch per level instead of 17-20 (key kernel + 5-6 radix passes of three kernels): the
// deep levels of the order were launch-bound, 1.5 ms at N = 10^6 whatever the number of ranks.
#define ST_SEG_LDS 4096
__global__ __launch_bounds__(256) void k_st_level_sort_lds(const float *__restrict__ D, uint32_t *__restrict__ order, int64_t n, int level,
                                                          const int32_t *__restrict__ coord, int seg_lo)
{
    __shared__ unsigned long long key[ST_SEG_LDS];
    __shared__ uint32_t val[ST_SEG_LDS];
    const int sgm = seg_lo + blockIdx.x;
    const int64_t sb = ((int64_t)sgm * n + (1ll << level) - 1) >> level, se = ((int64_t)(sgm + 1) * n + (1ll << level) - 1) >> level;
    const int len = (int)(se - sb);
    int P = 2;
    while (P < len) P <<= 1;
    const float *Dc = D + (size_t)coord[sgm] * n;
    for (int i = threadIdx.x; i < P; i += 256) {
        if (i < len) {
            const uint32_t v = order[sb + i];
            val[i] = v;
            key[i] = ((unsigned long long)__float_as_uint(Dc[v]) << 32) | (unsigned long long)(uint32_t)i;
        } else {
            key[i] = ~0ull;
        }
    }
    __syncthreads();
    for (int k2 = 2; k2 <= P; k2 <<= 1)
        for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
            for (int t = threadIdx.x; t < P / 2; t += 256) {
                const int q = 2 * t - (t & (j2 - 1)), p2 = q + j2;   // the t-th pair of the stage
                const bool up = (q & k2) == 0;
                const unsigned long long a = key[q], b = key[p2];
                if ((a > b) == up) { key[q] = b; key[p2] = a; }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < len; i += 256) order[sb + i] = val[(uint32_t)key[i]];
}

__global__ void k_st_iota(uint32_t *v, int64_t n)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) v[p] = (uint32_t)p;
}

__global__ void k_st_gather(const float *__restrict__ X, const uint32_t *__restrict__ order, int64_t n, int64_t n_pad, int dim,
                            int dimp, int64_t base, float *__restrict__ Xs, float *__restrict__ rs, int64_t *__restrict__ perm)
{
    // 16 lanes per row
    const int sub = threadIdx.x & 15;
    const int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (s >= n_pad) return;
    float acc = 0.f;
    if (s < n) {
        const uint32_t src = order[s];
        for (int k = sub; k < dimp; k += 16) {
            float v = k < dim ? X[(size_t)src * dim + k] : 0.f;
            Xs[(size_t)s * dimp + k] = v;
            acc += v * v;
        }
    } else {
        for (int k = sub; k < dimp; k += 16) Xs[(size_t)s * dimp + k] = 0.f;
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 16);
    if (sub == 0) {
        rs[s] = s < n ? acc : INFINITY;
        perm[s] = s < n ? base + (int64_t)order[s] : -1;
    }
}

// block = tile, thread = row of the tile: the row's whole anchor vector is one contiguous read of the point-major copy Dt (the
// anchor-major D cost one scattered 4-byte read per (row, anchor): 4.6 ms at N = 8 x 10^6), the tile's minimum / maximum / mean
// per anchor are taken by one thread per anchor over an LDS transpose
__global__ __launch_bounds__(ST_T) void k_st_intervals(const float *__restrict__ Dt, int nap, const uint32_t *__restrict__ order, int64_t n,
                                                      int na, int nt, float *__restrict__ lo, float *__restrict__ hi,
                                                      float *__restrict__ mid)
{
    __shared__ float sv[ST_T][65];   // [row][anchor], padded: the per-anchor passes read columns
    const int t = blockIdx.x;
    const int64_t s = (int64_t)t * ST_T + threadIdx.x;
    const bool ok = s < n;
    {
        const float4 *src = reinterpret_cast<const float4 *>(Dt + (size_t)(ok ? order[s] : 0u) * nap);
        for (int q = 0; q < nap / 4; ++q) {
            const float4 v = ok ? src[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            sv[threadIdx.x][4 * q] = v.x; sv[threadIdx.x][4 * q + 1] = v.y; sv[threadIdx.x][4 * q + 2] = v.z; sv[threadIdx.x][4 * q + 3] = v.w;
        }
    }
    __syncthreads();
    const int rows = (int)min<int64_t>(ST_T, max<int64_t>(n - (int64_t)t * ST_T, 0));
    if ((int)threadIdx.x < na) {
        const int a = threadIdx.x;
        float mn = INFINITY, mx = -INFINITY;
        for (int r = 0; r < rows; ++r) { const float v = sv[r][a]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
        lo[(size_t)a * nt + t] = mn;
        hi[(size_t)a * nt + t] = mx;
    }
    __syncthreads();
    // the sums in the association the wave reductions they replace had (rows pair up by xor 32, 16, .. 1 inside each 64-row
    // half, then the two halves; missing rows count 0): the mean keeps its bits, and with it the rank key of every tile pair
    for (int off = 32; off > 0; off >>= 1) {
        for (int idx = threadIdx.x; idx < 2 * off * na; idx += ST_T) {
            const int a = idx % na, rr = idx / na;
            const int r = (rr >= off ? 64 : 0) + (rr % off);
            sv[r][a] += sv[r + off][a];
        }
        __syncthreads();
    }
    if ((int)threadIdx.x < na) {
        const int a = threadIdx.x;
        mid[(size_t)a * nt + t] = rows ? (sv[0][a] + sv[64][a]) / (float)rows : INFINITY;   // mean anchor distance of the tile
    }
}

// ---- stable LSD radix sort of (uint64 key, uint32 value) pairs on bits [0, end_bit): 8-bit digits, per pass a
// per-tile digit count, an exclusive scan of the (digit, tile) counts and a scatter that ranks every key among the
// equal digits of its tile in tile order (wave match by eight ballots, waves ordered through LDS counters).
#define RS_THREADS 256
#define RS_ITEMS 8
#define RS_TILE (RS_THREADS * RS_ITEMS)

__global__ __launch_bounds__(RS_THREADS) void k_rs_count(const unsigned long long *__restrict__ keys, int64_t n, int shift, int nblk,
                                                        uint32_t *__restrict__ cnt)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int64_t t = base + j * RS_THREADS + threadIdx.x;
        if (t < n) atomicAdd(&h[(uint32_t)(keys[t] >> shift) & 255u], 1u);
    }
    __syncthreads();
    cnt[(size_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}

// counts laid out [256 digits][nblk tiles]: one workgroup per digit row turns the row into its exclusive prefix and
// leaves the row total in tot[digit]; the scatter kernel adds the prefix of the totals over the digits itself
__global__ __launch_bounds__(256) void k_rs_scan(uint32_t *__restrict__ cnt, int nblk, uint32_t *__restrict__ tot)
{
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t carry_s;
    uint32_t *row = cnt + (size_t)blockIdx.x * nblk;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int t0 = 0; t0 < nblk; t0 += 256) {
        const int t = t0 + threadIdx.x;
        const uint32_t v = t < nblk ? row[t] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(inc, off); if (lane >= off) inc += o; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t b = carry_s;
        for (int w = 0; w < wave; ++w) b += wsum[w];
        if (t < nblk) row[t] = b + inc - v;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = b + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) tot[blockIdx.x] = carry_s;
}

__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                          int64_t n, int shift, int nblk, const uint32_t *__restrict__ offs,
                                                          const uint32_t *__restrict__ tot,
                                                          unsigned long long *__restrict__ keys_out, uint32_t *__restrict__ vals_out)
{
    // A wave owns a contiguous quarter of the tile (RS_ITEMS rows of 64 keys): tile order = wave, row, lane.  Each
    // wave counts its keys per digit in its own LDS row, one workgroup barrier turns the four rows into every wave's
    // first position per digit, and from there a wave ranks its rows on its own (match by eight ballots, the group's
    // first lane advances the wave's counter): two barriers per tile, not three per row.
    __shared__ uint32_t wc[RS_THREADS / 64][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int w = 0; w < RS_THREADS / 64; ++w) wc[w][threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE + (int64_t)wave * (64 * RS_ITEMS);
    unsigned long long key[RS_ITEMS];
    uint32_t val[RS_ITEMS];
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int64_t t = base + j * 64 + lane;
        key[j] = t < n ? keys[t] : 0ull;
        val[j] = t < n ? vals[t] : 0u;
        if (t < n) atomicAdd(&wc[wave][(uint32_t)(key[j] >> shift) & 255u], 1u);
    }
    __syncthreads();
    {   // thread = digit: counts -> first positions (keys of smaller digits + earlier tiles of this digit + the waves before)
        __shared__ uint32_t dsum[4];
        const uint32_t mine = tot[threadIdx.x];
        uint32_t inc = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(inc, off); if (lane >= off) inc += o; }
        if (lane == 63) dsum[wave] = inc;
        __syncthreads();
        uint32_t dbase = inc - mine;
        for (int w = 0; w < wave; ++w) dbase += dsum[w];
        uint32_t run = dbase + offs[(size_t)threadIdx.x * nblk + blockIdx.x];
        for (int w = 0; w < RS_THREADS / 64; ++w) { const uint32_t v = wc[w][threadIdx.x]; wc[w][threadIdx.x] = run; run += v; }
    }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const bool ok = base + j * 64 + lane < n;
        const uint32_t d = (uint32_t)(key[j] >> shift) & 255u;
        unsigned long long same = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bb = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? bb : ~bb;
        }
        const int before = __popcll(same & below);
        uint32_t pos = 0;
        if (ok) pos = wc[wave][d] + (uint32_t)before;
        __builtin_amdgcn_wave_barrier();   // every lane of the group has read the counter before its first lane advances it
        if (ok && before == 0) wc[wave][d] += (uint32_t)__popcll(same);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (ok) { keys_out[pos] = key[j]; vals_out[pos] = val[j]; }
    }
}

// Sorted result ends up in (keys_a, vals_a) or (keys_b, vals_b): returns 0 / 1 through *where.
static int st_radix_sort_pairs(annchor_ctx *c, uint32_t *cnt, unsigned long long *keys_a, unsigned long long *keys_b, uint32_t *vals_a,
                               uint32_t *vals_b, int64_t n, int end_bit, int *where)
{
    const int nblk = (int)((n + RS_TILE - 1) / RS_TILE);
    int cur = 0;
    for (int shift = 0; shift < end_bit; shift += 8) {
        unsigned long long *ki = cur ? keys_b : keys_a, *ko = cur ? keys_a : keys_b;
        uint32_t *vi = cur ? vals_b : vals_a, *vo = cur ? vals_a : vals_b;
        k_rs_count<<<nblk, RS_THREADS, 0, c->stream>>>(ki, n, shift, nblk, cnt);
        k_rs_scan<<<256, 256, 0, c->stream>>>(cnt, nblk, cnt + (size_t)256 * nblk);
        k_rs_scatter<<<nblk, RS_THREADS, 0, c->stream>>>(ki, vi, n, shift, nblk, cnt, cnt + (size_t)256 * nblk, ko, vo);
        cur ^= 1;
    }
    ANN_CHECK_HIP(c, hipGetLastError());
    *where = cur;
    return ANNCHOR_OK;
}

// Order the bound rows into 128-row tiles of the k-d order and build the tile-ordered copies and the per-tile
// anchor-distance intervals.  Two steps, so that a row-sharded run orders only what it owns:
//
//   annchor_stream_order_begin(min_tiles, tile_begin, tile_count)   the level sorts, restricted at every level to the
//       segments that intersect the caller's tile range: level l cuts the order into 2^l equal position ranges, the
//       ranges are nested, so the rows of the caller's positions [tile_begin, tile_begin + tile_count) x 128 at level
//       l + 1 come out of the level-l segments that contain them and of nothing else.  A rank that owns 1 / G of the
//       tiles sorts n / G + (at most two segments: 2 n / 2^l) keys at level l instead of n: the sum over the levels is
//       ~(levels / G + 2) n keys instead of levels x n.  The positions of its range -- and only those -- hold the final
//       order; *order_local / *order_bytes describe that slice (uint32 [tile_count x 128], device) and *order_all the
//       all-gather target (uint32 [n_tiles x 128]) a multi-rank host gathers the slices into (rank r's slice at
//       r x tile_count x 128: the ranks own equal, contiguous tile ranges).  tile_count == n_tiles (one rank): the
//       whole order, nothing to gather.
//   annchor_stream_order_end()   gathers the rows into tile order (every rank holds every row as a column), the
//       squared norms, the global ids, the tiles' anchor-distance intervals and the fp16 split copy.
//
// The tile structure is a function of the data alone: the same whatever the number of ranks.
extern "C" int annchor_stream_order_begin(annchor_ctx *c, int32_t min_tiles, int32_t tile_begin, int32_t tile_count, void **order_local,
                                          void **order_all, int64_t *order_bytes)
{
    if (!c || !order_local || !order_all || !order_bytes) return ANNCHOR_EINVAL;
    StreamState *s = state_of(c, false);
    ANN_REQUIRE(c, s && s->na > 0, ANNCHOR_EINVAL, "anchor rounds not run");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    const int64_t n = s->n_local;
    s->nt = (int)((n + ST_T - 1) / ST_T);
    if (s->nt < min_tiles) s->nt = min_tiles;  // common tile count across ranks; extra tiles are pure padding
    s->n_pad = (int64_t)s->nt * ST_T;
    if (tile_count <= 0) { tile_begin = 0; tile_count = s->nt; }
    ANN_REQUIRE(c, tile_begin >= 0 && tile_begin + tile_count <= s->nt, ANNCHOR_EINVAL, "tile range [%d, %d) outside the %d tiles",
                tile_begin, tile_begin + tile_count, s->nt);
    ANN_REQUIRE(c, n < (1ll << 32), ANNCHOR_ELIMIT, "streamed ordering: %lld rows", (long long)n);
    ANN_TRY(sreserve(c, s->keys, 8 * (size_t)n));
    ANN_TRY(sreserve(c, s->keys2, 8 * (size_t)n));
    ANN_TRY(sreserve(c, s->vals, 4 * (size_t)s->n_pad));    // (n_pad: a rank's slice may reach past the last row)
    ANN_TRY(sreserve(c, s->vals2, 4 * (size_t)s->n_pad));
    ANN_TRY(sreserve(c, s->order_all, 4 * (size_t)s->n_pad));
    const int64_t pb = std::min<int64_t>((int64_t)tile_begin * ST_T, n), pe = std::min<int64_t>((int64_t)(tile_begin + tile_count) * ST_T, n);
    ProfScope ps(c, "stream_order_tiles", (double)(pe - pb) * (s->na * 8.0 + 40.0));
    int levels = 0;
    while (((int64_t)ST_T << levels) < n) ++levels;   // segments end up <= one tile long
    const int max_seg = 1 << levels;
    ANN_TRY(sreserve(c, s->red_val, sizeof(double) * 2 * (size_t)max_seg * s->na));
    ANN_TRY(sreserve(c, s->red_idx, sizeof(int32_t) * (size_t)max_seg));
    uint32_t *cur = s->vals.as<uint32_t>(), *nxt = s->vals2.as<uint32_t>();
    k_st_iota<<<ann_blocks(n, 256), 256, 0, c->stream>>>(cur, n);
    const int nap = (s->na + 3) & ~3;
    ANN_TRY(sreserve(c, s->Dt, sizeof(float) * (size_t)n * nap));
    k_st_transpose_D<<<ann_blocks(n * nap, 256), 256, 0, c->stream>>>(s->D.as<float>(), n, s->na, nap, s->Dt.as<float>());
    ANN_TRY(sreserve(c, s->cubtmp, sizeof(uint32_t) * 256 * ((size_t)((n + RS_TILE - 1) / RS_TILE) + 1)));
    for (int level = 0; level < levels && pe > pb; ++level) {
        // the segments of this level that hold the positions [pb, pe): positions [lb, le)
        const int seg_lo = st_seg_of(pb, n, level), seg_hi = st_seg_of(pe - 1, n, level);
        const int64_t lb = ((int64_t)seg_lo * n + (1ll << level) - 1) >> level, le = ((int64_t)(seg_hi + 1) * n + (1ll << level) - 1) >> level;
        k_st_split_coord<<<seg_hi - seg_lo + 1, 1024, 0, c->stream>>>(s->Dt.as<float>(), nap, cur, n, s->na, level, seg_lo, s->red_idx.as<int32_t>());
        static const bool lds_levels = !getenv("ANNCHOR_ST_ORDER_RADIX_ONLY");   // (the switch: every level by the radix sort, as before round 6)
        if (lds_levels && ((n + (1ll << level) - 1) >> level) <= ST_SEG_LDS) {   // every segment of the level fits a workgroup's LDS
            k_st_level_sort_lds<<<seg_hi - seg_lo + 1, 256, 0, c->stream>>>(s->D.as<float>(), cur, n, level, s->red_idx.as<int32_t>(), seg_lo);
            continue;
        }
        k_st_level_keys<<<ann_blocks(le - lb, 256), 256, 0, c->stream>>>(s->D.as<float>(), cur, n, level, s->red_idx.as<int32_t>(),
                                                                        s->keys.as<unsigned long long>(), lb, le);
        // (segment, distance to the split anchor): stable sort = every segment ordered along its coordinate.  The sort runs on
        // the sub-arrays [lb, le): outside them the two order buffers drift apart, which no later level reads (nested ranges)
        int where = 0;
        ANN_TRY(st_radix_sort_pairs(c, s->cubtmp.as<uint32_t>(), s->keys.as<unsigned long long>() + lb, s->keys2.as<unsigned long long>() + lb,
                                    cur + lb, nxt + lb, le - lb, 32 + level, &where));
        if (where) { uint32_t *t = cur; cur = nxt; nxt = t; }
    }
    ANN_CHECK_HIP(c, hipGetLastError());
    s->order_cur = cur;
    s->order_tile_begin = tile_begin;
    s->order_tile_count = tile_count;
    *order_local = cur + (size_t)tile_begin * ST_T;
    *order_all = s->order_all.p;
    *order_bytes = (int64_t)sizeof(uint32_t) * tile_count * ST_T;
    return ANNCHOR_OK;
}

extern "C" int annchor_stream_order_end(annchor_ctx *c, void **Xs, void **rs, void **perm, void **lo, void **hi, void **mid, int64_t *n_pad,
                                        int32_t *n_tiles, int32_t *dim_padded)
{
    if (!c || !Xs || !rs || !perm || !lo || !hi || !mid || !n_pad || !n_tiles || !dim_padded) return ANNCHOR_EINVAL;
    StreamState *s = state_of(c, false);
    ANN_REQUIRE(c, s && s->na > 0 && s->order_cur, ANNCHOR_ESTATE, "annchor_stream_order_begin first");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    const int64_t n = s->n_local;
    // the complete order: the context's own buffer (one rank: its range was everything) or the gathered slices
    const uint32_t *order = s->order_tile_count == s->nt ? s->order_cur : s->order_all.as<uint32_t>();
    s->order_cur = nullptr;
    ANN_TRY(sreserve(c, s->Xs, sizeof(float) * (size_t)s->n_pad * s->dimp));
    ANN_TRY(sreserve(c, s->rs, sizeof(float) * (size_t)s->n_pad));
    ANN_TRY(sreserve(c, s->perm, sizeof(int64_t) * (size_t)s->n_pad));
    ANN_TRY(sreserve(c, s->lo, sizeof(float) * (size_t)s->na * s->nt));
    ANN_TRY(sreserve(c, s->hi, sizeof(float) * (size_t)s->na * s->nt));
    ANN_TRY(sreserve(c, s->mid, sizeof(float) * (size_t)s->na * s->nt));
    ANN_TRY(ann_comm_side_join(c));   // the gathered rows (their all-gather may have run beside the anchor rounds and the ordering)
    {
        // every rank holds every row as a column: this part is per rank whatever the number of ranks
        ProfScope ps(c, "stream_order_gather_rows", (double)n * (s->dim * 4.0 + s->dimp * 8.0 + s->na * 4.0 + 16.0));
        k_st_gather<<<ann_blocks(s->n_pad * 16, 256), 256, 0, c->stream>>>(s->X.as<float>(), order, n, s->n_pad, s->dim, s->dimp, s->base,
                                                                          s->Xs.as<float>(), s->rs.as<float>(), s->perm.as<int64_t>());
        k_st_intervals<<<s->nt, ST_T, 0, c->stream>>>(s->Dt.as<float>(), (s->na + 3) & ~3, order, n, s->na, s->nt, s->lo.as<float>(),
                                                     s->hi.as<float>(), s->mid.as<float>());
        ANN_TRY(ann_stream_split_rows(c, s));   // the fp16 hi / lo copy the tile kernels stream (knnbf.hip, knnbk.hip)
    }
    ANN_CHECK_HIP(c, hipGetLastError());
    ANN_CHECK_HIP(c, ann_stream_wait(c, __func__));
    *Xs = s->Xs.p; *rs = s->rs.p; *perm = s->perm.p; *lo = s->lo.p; *hi = s->hi.p; *mid = s->mid.p;
    *n_pad = s->n_pad; *n_tiles = s->nt; *dim_padded = s->dimp;
    return ANNCHOR_OK;
}

// both steps for a context that orders everything itself (one rank; queries)
extern "C" int annchor_stream_order(annchor_ctx *c, int32_t min_tiles, void **Xs, void **rs, void **perm, void **lo, void **hi,
                                    void **mid, int64_t *n_pad, int32_t *n_tiles, int32_t *dim_padded)
{
    if (!c || !Xs || !rs || !perm || !lo || !hi || !mid || !n_pad || !n_tiles || !dim_padded) return ANNCHOR_EINVAL;
    void *ol = nullptr, *oa = nullptr;
    int64_t ob = 0;
    ANN_TRY(annchor_stream_order_begin(c, min_tiles, 0, 0, &ol, &oa, &ob));
    return annchor_stream_order_end(c, Xs, rs, perm, lo, hi, mid, n_pad, n_tiles, dim_padded);
}

// ------------------------------------------------------------------ k-NN

// Lists of more than 64 entries (n_neighbors 66 .. 128): 128 rows x 129 entries x 8 bytes do not fit beside the operand buffers, so
// a row tile is taken by TWO workgroups, each with the lists of 64 of its rows (the other 64 rows never accept a candidate: their
// thresholds are -1, like padding rows); both stream and multiply everything -- half the throughput, for list lengths the
// reference takes (n_neighbors is unbounded there: annchor.py:136) and the split-fp16 kernels do not.
template <int KMAX> struct KnnHalf {
    static constexpr bool ON = KMAX > ST_KMAX_BIG;
    static constexpr int LROWS = ON ? ST_T / 2 : ST_T;
    __device__ static __forceinline__ int lr(int row) { return ON ? (row & (ST_T / 2 - 1)) : row; }   // list slot of a row
};
template <int DIM, int KMAX> struct KnnShared {
    float Bs[ST_SLAB][DIM + 1];
    float rsJ[ST_SLAB];
    // (row strides that are not multiples of the 32 banks: lanes own different rows and touch the same slot)
    float cand_d[ST_T][ST_SLAB + 1];
    uint8_t cand_c[ST_T][ST_SLAB + 4];   // column inside the slab
    float list_d[KnnHalf<KMAX>::LROWS][KMAX + 1];
    int32_t list_c[KnnHalf<KMAX>::LROWS][KMAX + 1];
    float thr[ST_T];
    int cnt[ST_T];
    float loI[64], hiI[64], midI[64];
    float surv_lb[ST_SURV];   // rank key (mid-point bound)
    float surv_vb[ST_SURV];   // valid interval lower bound
    int32_t surv_j[ST_SURV];
    float wave_thr[ST_THREADS / 64];
    int wave_ins[ST_THREADS / 64];   // list insertions so far, per wave (cumulative)
    uint32_t slab_id[2][ST_SLAB];   // join passes: ordered column index of each staged column (slab parity)
    int nsurv;
    int sel_bin;
    uint32_t sel_before;
};

// One column tile against the workgroup's row tile.  Rows [32 w, 32 w + 32) belong to wave w
// for the whole kernel -- operand registers, thresholds, candidate slots and result lists --
// so everything after the dot products is wave-private and needs no workgroup barrier:
//   slab s:  [barrier] stage registers -> LDS  [barrier]  global loads of slab s+1 in flight
//            MFMA block of slab s, with the threshold test of slab s-1's accumulators issued
//            between the MFMAs (VALU work in the shadow of the matrix pipe)
//            merge of slab s-1's survivors into the wave's row lists
// The two barriers only hand the single LDS operand buffer from its readers to its writers.
// Register staging of one 32-column slab; survives from one tile to the next so that the
// first slab of the next tile is already in flight while the current tile finishes.
// Which float4 of a 32-column slab thread item q stages: column colr (0..31), first dimension k4.  Inside every
// group of 32 consecutive items the column varies over 4 and the float4 index over 8, so the four scalar LDS
// writes of the group land in 32 different banks (row stride DIM + 1 = 1 mod 32: bank = colr + k4 + j); the
// plain row-major assignment put 32 lanes on 8 banks (a third of the kernel's LDS-busy cycles were conflicts).
template <int DIM> __device__ __forceinline__ void stage_map(int q, int &colr, int &k4)
{
    const int g = q >> 5, r = q & 31;
    colr = (g & 7) * 4 + (r & 3);
    k4 = ((g >> 3) * 8 + (r >> 2)) * 4;
}

template <int DIM> struct SlabStage {
    float4 v[ST_SLAB * DIM / 4 / ST_THREADS];
    float r;
    uint32_t id; // join passes: ordered column index of the staged column (threads < ST_SLAB)
    int ins;     // join passes: list insertions made by this thread's rows
    int J;   // tile whose slab 0 is held (-1: none)
};

// Returns the worst k-th squared distance over the row tile after this column tile.
// GATHER (join passes): "tile" J is the J-th run of 128 entries of the row tile's candidate list
// a.ucand (ordered column indices, 0xffffffff = padding) instead of 128 consecutive columns.
template <int DIM, int KMAX, bool GATHER = false>
__device__ __forceinline__ float knn_process_tile(KnnShared<DIM, KMAX> &sh, const KnnArgs &a, int J, int Jnext, SlabStage<DIM> &st,
                                                  const float (&areg)[DIM / 2], const float (&ri)[16], int rowbase_wave,
                                                  int64_t grow0, int K, long long *pf_ext, const uint32_t *ulist = nullptr)
{
#ifdef ST_PROFILE
    long long pf_t = clock64();
    long long pf_none[8];
    long long *pf = pf_ext ? pf_ext : pf_none;
#endif
    const int lane = threadIdx.x & 63;
    constexpr int NLD = ST_SLAB * DIM / 4 / ST_THREADS;   // float4 loads per thread per slab
    constexpr int NSLAB = ST_T / ST_SLAB;
    constexpr int MF = DIM / 32;                           // MFMAs per accumulator row group (16 groups)
    float4 (&stage)[NLD] = st.v;
    float &stage_r = st.r;
    auto slab_load = [&](int Jl, int slab) {
        const int64_t c0 = (int64_t)Jl * ST_T + slab * ST_SLAB;
        if constexpr (GATHER) {
            uint32_t ids[NLD];
#pragma unroll
            for (int u = 0; u < NLD; ++u) {
                int colr, k4;
                stage_map<DIM>(u * ST_THREADS + threadIdx.x, colr, k4);
                ids[u] = ulist[c0 + colr];
            }
#pragma unroll
            for (int u = 0; u < NLD; ++u) {
                int colr, k4;
                stage_map<DIM>(u * ST_THREADS + threadIdx.x, colr, k4);
                const uint32_t src = ids[u] == 0xffffffffu ? 0u : ids[u];
                stage[u] = *reinterpret_cast<const float4 *>(a.Xs + (size_t)src * DIM + k4);
            }
            if (threadIdx.x < ST_SLAB) {
                const uint32_t id = ulist[c0 + threadIdx.x];
                st.id = id;
                stage_r = id == 0xffffffffu ? INFINITY : a.rs[id];
            }
        } else {
#pragma unroll
            for (int u = 0; u < NLD; ++u) {
                int colr, k4;
                stage_map<DIM>(u * ST_THREADS + threadIdx.x, colr, k4);
                stage[u] = *reinterpret_cast<const float4 *>(a.Xs + (size_t)(c0 + colr) * DIM + k4);
            }
            if (threadIdx.x < ST_SLAB) stage_r = a.rs[c0 + threadIdx.x];
        }
    };
    const int col = lane & 31;
    const int rowq = rowbase_wave + 4 * (lane >> 5);   // C layout: row = rowq + (r & 3) + 8 (r >> 2), col = lane & 31
    const bool self_tile = !GATHER && !a.query && (int64_t)J * ST_T == grow0;
    // merge of a slab's survivors into the sorted per-row lists: lane l < 32 owns row 32 w + l
    auto merge = [&](int64_t col0, int par) {
        wave_fence_lds();
        if (lane < 32) {
            const int row = rowbase_wave + lane;
            const int lr = KnnHalf<KMAX>::lr(row);   // (rows without lists -- the other workgroup's half -- never have candidates)
            const int nc = sh.cnt[row];
            if (nc) {
                for (int q = 0; q < nc; ++q) {
                    const float d = sh.cand_d[row][q];
                    int32_t cc;
                    if constexpr (GATHER) {
                        // gathered columns: the point itself and columns the row already lists may come by
                        cc = (int32_t)sh.slab_id[par][sh.cand_c[row][q]];
                        bool skip = (int64_t)cc == grow0 + row;
                        for (int e = 0; e < K && !skip; ++e) skip = sh.list_c[lr][e] == cc;
                        if (skip) continue;
                    } else {
                        cc = (int32_t)(col0 + sh.cand_c[row][q]);
                    }
                    // insertion by (d, col); list is padded with +inf
                    if (d < sh.list_d[lr][K - 1] || (d == sh.list_d[lr][K - 1] && cc < sh.list_c[lr][K - 1])) {
                        ++st.ins;   // (the yield of the tile / pass: early stop of the tile phase, extra join passes)
                        int p = K - 1;
                        while (p > 0 && (d < sh.list_d[lr][p - 1] || (d == sh.list_d[lr][p - 1] && cc < sh.list_c[lr][p - 1]))) {
                            sh.list_d[lr][p] = sh.list_d[lr][p - 1];
                            sh.list_c[lr][p] = sh.list_c[lr][p - 1];
                            --p;
                        }
                        sh.list_d[lr][p] = d;
                        sh.list_c[lr][p] = cc;
                    }
                }
                sh.cnt[row] = 0;
                sh.thr[row] = sh.list_d[lr][K - 1];
            }
        }
        wave_fence_lds();
    };
    f32x16 acc_prev;
    float rj_prev = 0.f;
    if (st.J != J) slab_load(J, 0);   // not prefetched by the previous tile
    for (int slab = 0; slab < NSLAB; ++slab) {
        // every wave is done reading the previous slab's operands (slab 0: the caller's last
        // barrier -- tile end or candidate scan -- already guarantees it)
        if (slab > 0) __syncthreads();
        ST_PROF(0)
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            int colr, k4;
            stage_map<DIM>(u * ST_THREADS + threadIdx.x, colr, k4);
            sh.Bs[colr][k4] = stage[u].x; sh.Bs[colr][k4 + 1] = stage[u].y; sh.Bs[colr][k4 + 2] = stage[u].z; sh.Bs[colr][k4 + 3] = stage[u].w;
        }
        if (threadIdx.x < ST_SLAB) {
            sh.rsJ[threadIdx.x] = stage_r;
            if constexpr (GATHER) sh.slab_id[slab & 1][threadIdx.x] = st.id;
        }
        ST_PROF(1)
        __syncthreads();
        ST_PROF(2)
        if (slab + 1 < NSLAB) slab_load(J, slab + 1);
        else if (Jnext >= 0) slab_load(Jnext, 0);   // speculative: the next ranked tile is almost never pruned
        const float rj = sh.rsJ[col];
        // ---- 32x32 block of dot products per wave: DIM/2 MFMAs of K = 2, the previous slab's
        // threshold tests issued in their shadow (row group g after the g-th bundle of MFMAs)
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float *bp = &sh.Bs[col][lane >> 5];
        // operands of bundle g+1 and the thresholds of the lane's 16 rows are requested ahead of
        // use: nothing in the MFMA stream waits for an LDS round trip
        float thr_r[16];
        if (slab > 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t4 = *reinterpret_cast<const float4 *>(&sh.thr[rowq + 8 * q]);
                thr_r[4 * q] = t4.x; thr_r[4 * q + 1] = t4.y; thr_r[4 * q + 2] = t4.z; thr_r[4 * q + 3] = t4.w;
            }
        }
        float breg[2][MF];
#pragma unroll
        for (int u = 0; u < MF; ++u) breg[0][u] = bp[2 * u];
        // Straight-line MFMA stream; in the shadow of bundle g the previous slab's accumulator
        // row g is turned into a distance and compared with its row threshold.  Only the
        // outcome bit is kept: survivors are rare once the lists have warmed up, and they are
        // inserted after the stream (one branch per slab instead of one per row).
        uint32_t pass = 0;
        float d2r[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (g + 1 < 16) {
#pragma unroll
                for (int u = 0; u < MF; ++u) breg[(g + 1) & 1][u] = bp[2 * ((g + 1) * MF + u)];
            }
#pragma unroll
            for (int u = 0; u < MF; ++u)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[g * MF + u], breg[g & 1][u], acc, 0, 0, 0);
            if (slab > 0) {
                d2r[g] = fmaxf(ri[g] + rj_prev - 2.f * acc_prev[g], 0.f);
                pass |= (d2r[g] < thr_r[g] ? 1u : 0u) << g;
            }
        }
        ST_PROF(3)
        if (slab > 0 && pass) {
            if (self_tile) {   // a point is not its own neighbour
                const int dcol = (slab - 1) * ST_SLAB + col - rowq;   // row offset inside the lane's row set that equals its column
                if (dcol >= 0 && dcol < 32 && (dcol & 4) == 0) pass &= ~(1u << ((dcol & 3) + 4 * (dcol >> 3)));
            }
            while (pass) {
                const int g = __builtin_ctz(pass);
                pass &= pass - 1;
                const int rowl = rowq + (g & 3) + 8 * (g >> 2);
                float d2 = d2r[0];
#pragma unroll
                for (int t = 1; t < 16; ++t) d2 = g == t ? d2r[t] : d2;
                const int slot = atomicAdd(&sh.cnt[rowl], 1);
                sh.cand_d[rowl][slot] = d2;
                sh.cand_c[rowl][slot] = (uint8_t)col;
            }
        }
        ST_PROF(4)
        if (slab > 0) merge((int64_t)J * ST_T + (slab - 1) * ST_SLAB, (slab - 1) & 1);
        ST_PROF(5)
        acc_prev = acc;
        rj_prev = rj;
    }
    // last slab: nothing left to hide it behind
    {
        uint32_t pass = 0;
        float d2r[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 t4 = *reinterpret_cast<const float4 *>(&sh.thr[rowq + 8 * q]);
            const float tq[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int g = 4 * q + e;
                d2r[g] = fmaxf(ri[g] + rj_prev - 2.f * acc_prev[g], 0.f);
                pass |= (d2r[g] < tq[e] ? 1u : 0u) << g;
            }
        }
        if (pass) {
            if (self_tile) {
                const int dcol = (NSLAB - 1) * ST_SLAB + col - rowq;
                if (dcol >= 0 && dcol < 32 && (dcol & 4) == 0) pass &= ~(1u << ((dcol & 3) + 4 * (dcol >> 3)));
            }
            while (pass) {
                const int g = __builtin_ctz(pass);
                pass &= pass - 1;
                const int rowl = rowq + (g & 3) + 8 * (g >> 2);
                float d2 = d2r[0];
#pragma unroll
                for (int t = 1; t < 16; ++t) d2 = g == t ? d2r[t] : d2;
                const int slot = atomicAdd(&sh.cnt[rowl], 1);
                sh.cand_d[rowl][slot] = d2;
                sh.cand_c[rowl][slot] = (uint8_t)col;
            }
        }
    }
    merge((int64_t)J * ST_T + (NSLAB - 1) * ST_SLAB, (NSLAB - 1) & 1);
    st.J = Jnext;
    // worst k-th squared distance of the row tile (padding rows have thr = -1): wave maxima,
    // one barrier (which is also the operand-buffer hand-over for the next tile)
    float t = lane < 32 ? sh.thr[rowbase_wave + lane] : -1.f;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) t = fmaxf(t, __shfl_xor(t, off));
    int wins = lane < 32 ? st.ins : 0;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) wins += __shfl_xor(wins, off);
    if (lane == 0) { sh.wave_thr[threadIdx.x >> 6] = t; sh.wave_ins[threadIdx.x >> 6] = wins; }
    __syncthreads();
    ST_PROF(6)
    return fmaxf(fmaxf(sh.wave_thr[0], sh.wave_thr[1]), fmaxf(sh.wave_thr[2], sh.wave_thr[3]));
}

template <int DIM, int KMAX> __global__ __launch_bounds__(ST_THREADS, (DIM <= 128 ? 2 : 1)) void k_st_knn(KnnArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    KnnShared<DIM, KMAX> &sh = *reinterpret_cast<KnnShared<DIM, KMAX> *>(smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // XCD-banded row-tile assignment (block b runs on XCD b % 8): the workgroups resident on
    // one XCD own consecutive row tiles of the k-d order, whose column-tile lists overlap, so
    // the streamed column tiles are shared through that XCD's L2
    using Half = KnnHalf<KMAX>;
    const int hsel = Half::ON ? (int)(blockIdx.x & 1) : 0;            // which 64 rows' lists this workgroup keeps (lists beyond 64 entries)
    const int bid = Half::ON ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;
    auto own = [&](int row) { return !Half::ON || (row >> 6) == hsel; };
    int bt;
    {
        const int nb_ = Half::ON ? (int)(gridDim.x >> 1) : (int)gridDim.x, q = nb_ >> 3, r = nb_ & 7, x = bid & 7, y = bid >> 3;
        bt = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    }
    const int I = a.tile_begin + bt;
    const int64_t grow0 = (int64_t)I * ST_T;
    const int K = a.K;
    // ---- per-wave row operand in registers: lane holds row (lane & 31), dims of parity (lane >> 5)
    float areg[DIM / 2];
    {
        const float *xr = a.Rs + (size_t)(grow0 + wave * 32 + (lane & 31)) * DIM + (lane >> 5);
#pragma unroll
        for (int s = 0; s < DIM / 2; ++s) areg[s] = xr[2 * s];
    }
    float ri[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ri[r] = a.rr[grow0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
    if (threadIdx.x < ST_T) {
        const int row = threadIdx.x;
        const bool real = a.rr[grow0 + row] < INFINITY;
        sh.thr[row] = real && own(row) ? INFINITY : -1.f;  // padding rows (and the other workgroup's half) never accept candidates
        sh.cnt[row] = 0;
        if (own(row))
            for (int q = 0; q < KMAX; ++q) { sh.list_d[Half::lr(row)][q] = INFINITY; sh.list_c[Half::lr(row)][q] = 0x7fffffff; }
    }
    if ((int)threadIdx.x < a.na) {
        sh.loI[threadIdx.x] = a.rlo[(size_t)threadIdx.x * a.nt_r + I];
        sh.hiI[threadIdx.x] = a.rhi[(size_t)threadIdx.x * a.nt_r + I];
        sh.midI[threadIdx.x] = a.rmid[(size_t)threadIdx.x * a.nt_r + I];
    }
    if (threadIdx.x == 0) sh.nsurv = 0;
    float thrmax = INFINITY;   // worst k-th squared distance of the row tile (uniform)
    int processed = 0;         // column tiles evaluated so far (uniform)
    SlabStage<DIM> st;
    st.J = -1;
    st.ins = 0;
    int win_start = 0, win_ins = 0;   // early stop: tiles / insertions at the start of the current window
    bool dried = false;
    long long *pf_ptr = nullptr;
    ST_PROF_DECL
#ifdef ST_PROFILE
    pf_ptr = pf;
#endif
    __syncthreads();

    // ---- phase A: the row tile against itself (gives every row k finite candidates); query
    // rows are not part of the data set and start from the ranked tiles directly
    if (!a.query) {
        thrmax = knn_process_tile<DIM, KMAX>(sh, a, I, -1, st, areg, ri, wave * 32, grow0, K, pf_ptr);
        ++processed;
    }
    // column tiles this row tile has evaluated (a join pass skips candidates inside them): this
    // workgroup is the only writer of its bitmap row
    // (lists split over two workgroups: each half has its own threshold, early stop and budget, hence its own set of
    // evaluated tiles and its own bitmap row -- the join passes skip a candidate only where BOTH halves evaluated its tile)
    uint32_t *ebits = a.eval_bits ? a.eval_bits + ((size_t)bt * (Half::ON ? 2 : 1) + hsel) * a.eval_words : nullptr;
    if (ebits && !a.query && threadIdx.x == 0) atomicOr(&ebits[I >> 5], 1u << (I & 31));

    // ---- phase B: all other column tiles.  A tile is ELIGIBLE while its interval bound lb
    // (a valid lower bound of every pair distance) is below the worst k-th distance of the
    // row tile; eligible tiles are RANKED by the distance between the tiles' mean
    // anchor-distance vectors (the anchors embed the data; this is the tile analogue of
    // ranking pairs by a distance predicted from anchor features, annchor.py:345-380).  Rounds: scan all tiles keeping the ST_KEEP best-ranked not yet considered,
    // evaluate them in rank order, repeat until nothing is eligible or the budget is spent.
    // Implementation: (1) once per row tile, the rank key and the valid bound of every column
    // tile go to a scratch row in global memory (two floats per tile; at N = 8M that is 62 500
    // tiles, far more than LDS holds); (2) each round selects the next ST_KEEP tiles in
    // (key, J) order among the still eligible ones with a 3-level radix selection on the key
    // bits (12 + 12 + 8; non-negative floats order like their bit patterns), collects them and
    // sorts them once.  A round costs four sweeps over the scratch row and one 1024-entry
    // bitonic sort, independent of how the keys are distributed.
    float *skey = a.scr_key + (size_t)bt * a.nt_all;
    float *slb = a.scr_lb + (size_t)bt * a.nt_all;
    if (!a.pre_ranked)
    for (int J = threadIdx.x; J < a.nt_all; J += ST_THREADS) {
        float lb = 0.f, lbc = 0.f;
        for (int an = 0; an < a.na; ++an) {
            const float lj = a.lo[(size_t)an * a.nt_all + J], hj = a.hi[(size_t)an * a.nt_all + J];
            const float gap = fmaxf(sh.loI[an] - hj, lj - sh.hiI[an]);
            // slack for the float32 rounding of D (bounds must stay valid lower bounds)
            lb = fmaxf(lb, gap - 4e-6f * (fabsf(hj) + fabsf(sh.hiI[an])));
            const float dm = a.mid[(size_t)an * a.nt_all + J] - sh.midI[an];
            lbc += dm * dm;   // rank key: squared L2 distance between the tiles' mean anchor vectors
        }
        skey[J] = ((J == I && !a.query) || !(lbc < INFINITY)) ? INFINITY : lbc;   // +inf: never a candidate
        slb[J] = lb;
    }
    __syncthreads();   // block-scope visibility of the scratch row (same CU)
    uint32_t *hist = reinterpret_cast<uint32_t *>(&sh.cand_d[0][0]);   // 4096 bins; cand_d is idle between tiles
    static_assert(sizeof(sh.cand_d) >= 4096 * sizeof(uint32_t), "histogram does not fit");
    uint32_t done_bits = 0;   // (done_bits, done_j): key bits / index of the last tile already considered
    int done_j = -1;          // (nothing considered yet: every key is > (0, -1))
    for (;;) {
        // ---- selection: bits of the ST_KEEP-th smallest remaining key
        uint32_t prefix = 0;       // key bits fixed so far (high part)
        uint32_t want = ST_KEEP;   // rank still to be located inside the current prefix class
        bool all = false;          // fewer than ST_KEEP candidates remain: take them all
        for (int level = 0; level < 3 && !all; ++level) {
            const int shift = level == 0 ? 20 : level == 1 ? 8 : 0;
            const int nbins = level == 2 ? 256 : 4096;
            const uint32_t pmask = level == 0 ? 0u : level == 1 ? 0xfff00000u : 0xffffff00u;
            for (int q = threadIdx.x; q < nbins; q += ST_THREADS) hist[q] = 0;
            __syncthreads();
            for (int J = threadIdx.x; J < a.nt_all; J += ST_THREADS) {
                const uint32_t kb = __float_as_uint(skey[J]);
                const float lb = slb[J];
                const bool after_done = kb > done_bits || (kb == done_bits && J > done_j);
                if (kb < 0x7f800000u && after_done && lb * lb < thrmax && (kb & pmask) == prefix)
                    atomicAdd(&hist[(kb >> shift) & (nbins - 1)], 1u);
            }
            __syncthreads();
            // first bin whose cumulative count reaches `want`: thread t owns bins [per t, per (t+1));
            // exclusive scan of the per-thread sums (wave shuffles + 4 wave totals), then the
            // one thread whose range holds the crossing walks its own bins
            const int per = nbins / ST_THREADS;
            uint32_t mine = 0;
            for (int q = 0; q < per; ++q) mine += hist[threadIdx.x * per + q];
            uint32_t incl = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t up = __shfl_up(incl, off);
                if (lane >= off) incl += up;
            }
            uint32_t *wtot = reinterpret_cast<uint32_t *>(&sh.surv_lb[0]);   // 4 wave totals (surv_lb is idle here)
            if (threadIdx.x == 0) sh.sel_bin = -1;
            if (lane == 63) wtot[wave] = incl;
            __syncthreads();
            uint32_t before = incl - mine;
            for (int w2 = 0; w2 < wave; ++w2) before += wtot[w2];
            if (before < want && before + mine >= want) {
                uint32_t acc = before;
                int q = threadIdx.x * per;
                for (;; ++q) { if (acc + hist[q] >= want) break; acc += hist[q]; }
                sh.sel_bin = q;
                sh.sel_before = acc;
            }
            __syncthreads();
            if (sh.sel_bin < 0) all = true;
            else { prefix |= (uint32_t)sh.sel_bin << shift; want -= sh.sel_before; }
            __syncthreads();
        }
        const uint32_t cut_bits = all ? 0x7f7fffffu : prefix;   // take keys <= cut (ties resolved by the sort below)
        // ---- collect (at most ST_SURV; more than ST_SURV - ST_KEEP ties on one key value would be dropped)
        if (threadIdx.x == 0) sh.nsurv = 0;
        __syncthreads();
        for (int J = threadIdx.x; J < a.nt_all; J += ST_THREADS) {
            const uint32_t kb = __float_as_uint(skey[J]);
            const float lb = slb[J];
            const bool after_done = kb > done_bits || (kb == done_bits && J > done_j);
            if (kb < 0x7f800000u && after_done && lb * lb < thrmax && kb <= cut_bits) {
                const int slot = atomicAdd(&sh.nsurv, 1);
                if (slot < ST_SURV) { sh.surv_lb[slot] = __uint_as_float(kb); sh.surv_vb[slot] = lb; sh.surv_j[slot] = J; }
            }
        }
        __syncthreads();
        int ns = min(sh.nsurv, ST_SURV);
        if (ns == 0) break;
        {   // sort by (rank key, J): bitonic over ST_SURV slots
            for (int q = threadIdx.x; q < ST_SURV; q += ST_THREADS)
                if (q >= ns) { sh.surv_lb[q] = INFINITY; sh.surv_j[q] = 0x7fffffff; }
            __syncthreads();
            for (int k2 = 2; k2 <= ST_SURV; k2 <<= 1)
                for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
                    for (int q = threadIdx.x; q < ST_SURV; q += ST_THREADS) {
                        const int p2 = q ^ j2;
                        if (p2 > q) {
                            const bool up = (q & k2) == 0;
                            const float lq = sh.surv_lb[q], lp = sh.surv_lb[p2];
                            const int jq = sh.surv_j[q], jp = sh.surv_j[p2];
                            const bool gt = lq > lp || (lq == lp && jq > jp);
                            if (gt == up) {
                                sh.surv_lb[q] = lp; sh.surv_lb[p2] = lq; sh.surv_j[q] = jp; sh.surv_j[p2] = jq;
                                const float t = sh.surv_vb[q]; sh.surv_vb[q] = sh.surv_vb[p2]; sh.surv_vb[p2] = t;
                            }
                        }
                    }
                    __syncthreads();
                }
        }
        const bool more = !all;          // the selection was cut at ST_KEEP: later tiles remain
        if (ns > ST_KEEP && more) ns = ST_KEEP;
        const uint32_t round_last_bits = __float_as_uint(sh.surv_lb[ns - 1]);
        const int round_last_j = sh.surv_j[ns - 1];
        __syncthreads();   // hist (cand_d) and the partial sums (surv_lb) are idle again: tiles may run
        for (int q = 0; q < ns; ++q) {
            if (processed >= a.max_tiles) break;
            const int J = sh.surv_j[q];
            const float lb = sh.surv_vb[q];  // re-check against the current, tighter threshold
            if (lb * lb < thrmax) {
                const int Jn = (q + 1 < ns && processed + 1 < a.max_tiles) ? sh.surv_j[q + 1] : -1;
#ifdef ST_PROFILE
                pf[7] += clock64() - pf_t;
#endif
                thrmax = knn_process_tile<DIM, KMAX>(sh, a, J, Jn, st, areg, ri, wave * 32, grow0, K, pf_ptr);
                ++processed;
                if (ebits && threadIdx.x == 0) atomicOr(&ebits[J >> 5], 1u << (J & 31));
                if (a.early_window > 0 && processed - win_start >= a.early_window) {
                    // the ranked tiles have stopped improving the lists: the rest of the budget would buy (almost)
                    // nothing that the join passes do not find for a fraction of the cost
                    const int cur = sh.wave_ins[0] + sh.wave_ins[1] + sh.wave_ins[2] + sh.wave_ins[3];
                    if (cur - win_ins < (Half::ON ? (a.early_tau + 1) / 2 : a.early_tau)) { dried = true; break; }   // (the yield of 64 rows)
                    win_start = processed; win_ins = cur;
                }
#ifdef ST_PROFILE
                pf_t = clock64();
#endif
            }
        }
        done_bits = round_last_bits;
        done_j = round_last_j;
        __syncthreads();
        if (dried) break;
        if (processed >= a.max_tiles) break;
        if (!more) break;   // the selection saw every eligible tile
    }
    __syncthreads();
    // ---- write the lists
    for (int q = threadIdx.x; q < ST_T * K; q += ST_THREADS) {
        const int row = q / K, e = q - row * K;
        if (!own(row)) continue;
        a.out_d2[((size_t)bt * ST_T + row) * K + e] = sh.list_d[Half::lr(row)][e];
        a.out_col[((size_t)bt * ST_T + row) * K + e] = sh.list_c[Half::lr(row)][e];
    }
    if (threadIdx.x == 0) atomicAdd(a.evals, (unsigned long long)processed);
#ifdef ST_PROFILE
    pf[7] += clock64() - pf_t;
    if (lane == 0 && a.prof)
        for (int i = 0; i < 8; ++i) atomicAdd(a.prof + i, (unsigned long long)pf[i]);
#endif
}


// ------------------------------------------------------------------ join passes
// The streamed analogue of update_anchor_points (reference annchor/annchor.py:475-512,
// utils.py:304-352): there, pairs (i, j) that share an already-computed neighbour c get the bounds
// |d_ic - d_jc| <= d_ij <= d_ic + d_jc and the next iteration refines the promising ones.  Here the
// computed neighbours of a row are its current k-NN list, so the pairs with a common computed
// neighbour are (i, j) with j in list(c), c in list(i): every row TILE collects that set for its
// 128 rows (minus the columns inside column tiles it has already evaluated), sorts / deduplicates
// it (k_st_join_cands) and evaluates all 128 x |U| pairs exactly as gathered tile GEMMs
// (k_st_join) -- the whole candidate set of a tile costs a few dozen tile evaluations.
#define JN_CAP 8192      // candidate columns per row tile (LDS sort buffer)
#define JN_B1 8192       // first-hop ids per row tile (128 rows x (K + JN_RK) <= 6144)
#define JN_RK 15         // reverse neighbours kept per point (the closest by list position)
#define JN_EB_WORDS 4096  // evaluated-tile bitmap words k_st_join_cands keeps in LDS (row tiles up to 131 072)
#define JN_THREADS 1024  // threads of k_st_join_cands: its time is bitonic stages (~340 per row tile) -- 16 waves make a stage a quarter as long as 4 did
#define ANNCHOR_JOIN_YIELD 0.01   // extra join passes run while a pass still replaces more than this share of the list entries
// The tile phase's early stop: a row tile stops when a window of ST_EARLY_WINDOW ranked tiles replaced fewer than this share of its
// 128 x K list entries.  1 % until the join passes' overflow handling was repaired (round 5: recall at C3 0.99715 -> 0.99852, at
// C5 0.9943 -> 0.9954 at the same cost); 1.25 % spends part of that -- tile phase -5 % (C3) / -8 % (C5) at recall 0.9984 / 0.9946,
// no lower than before the repair at either size.  (C3, 10 000 rows against the tile kernel's own truth, and C5, 1000 rows against
// float64: 1 % 0.99852 / 0.9954, 1.5 % 0.99826 / 0.99353, 2 % 0.99787 / 0.99193, 3 % 0.99659 / 0.98573; ANNCHOR_ST_EARLY_TAU sets
// the count directly.)
#define ANNCHOR_TILE_YIELD 0.0125

// ascending bitonic sort of P (a power of two, JN_THREADS <= P <= 8 JN_THREADS) uint32 keys in LDS by the workgroup.
// Thread t keeps elements [t E, t E + E) in registers (E = P / JN_THREADS): compare-exchange distances below E stay inside the
// thread, distances below 64 E inside the wave (one shuffle per element), and only the longer ones go through LDS with a
// barrier on either side -- 10 of the 91 stages at P = 8192.  (Every stage through LDS with a barrier behind it made the two
// sorts two thirds of k_st_join_cands' time.)  The network is the same, so is the result.
template <int E> __device__ __forceinline__ void jn_sort_reg(uint32_t *v)
{
    constexpr int P = E * JN_THREADS;
    const int t = threadIdx.x;
    uint32_t r[E];
#pragma unroll
    for (int e = 0; e < E; ++e) r[e] = v[t * E + e];
    for (int k2 = 2; k2 <= P; k2 <<= 1) {
        int j2 = k2 >> 1;
        for (; j2 >= 64 * E; j2 >>= 1) {
            __syncthreads();   // (the previous stage's readers are done with the array)
#pragma unroll
            for (int e = 0; e < E; ++e) v[t * E + e] = r[e];
            __syncthreads();
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = t * E + e;
                const uint32_t o = v[i ^ j2];
                const bool keep_min = ((i & k2) == 0) == ((i & j2) == 0);
                r[e] = keep_min ? min(r[e], o) : max(r[e], o);
            }
        }
        for (; j2 >= E; j2 >>= 1) {
            const int lx = j2 / E;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = t * E + e;
                const uint32_t o = __shfl_xor(r[e], lx);
                const bool keep_min = ((i & k2) == 0) == ((i & j2) == 0);
                r[e] = keep_min ? min(r[e], o) : max(r[e], o);
            }
        }
#pragma unroll
        for (int jj = E / 2; jj > 0; jj >>= 1) {
            if (jj <= j2) {   // (uniform: the stages of this k2 that are left)
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if ((e & jj) == 0) {
                        const bool up = (((t * E + e) & k2) == 0);
                        const uint32_t a = r[e], b = r[e | jj];
                        r[e] = up ? min(a, b) : max(a, b);
                        r[e | jj] = up ? max(a, b) : min(a, b);
                    }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) v[t * E + e] = r[e];
    __syncthreads();
}
__device__ __forceinline__ void jn_sort(uint32_t *v, int P)
{
    switch (P / JN_THREADS) {
    case 1: jn_sort_reg<1>(v); break;
    case 2: jn_sort_reg<2>(v); break;
    case 4: jn_sort_reg<4>(v); break;
    default: jn_sort_reg<8>(v); break;
    }
}

// in-place compaction of the distinct keys != 0xffffffff of a sorted array; returns their number
// (uniform).  Every thread owns a contiguous run, reads it to registers, then writes behind a
// workgroup scan of the run counts.
// cnt_out != NULL: also the multiplicity of every distinct key (run length in the sorted input), same order
template <int PER> __device__ __forceinline__ int jn_unique(uint32_t *v, int P, uint32_t *wsum /*[5]*/, uint32_t *cnt_out = nullptr)
{
    const int per = P / JN_THREADS;   // <= PER
    const int b = threadIdx.x * per;
    uint32_t r[PER];
    uint16_t rc[PER];
    uint32_t prev = b > 0 ? v[b - 1] : 0xffffffffu;
    int mine = 0;
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        uint32_t x = 0xffffffffu;
        if (e < per) x = v[b + e];
        const bool keep = e < per && x != 0xffffffffu && (x != prev || (b + e) == 0);
        r[e] = keep ? x : 0xffffffffu;
        rc[e] = 0;
        if (keep && cnt_out) {   // end of the run: first position > b + e whose key differs (binary search, input still intact)
            int lo = b + e + 1, hi = P;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (v[mid] == x) lo = mid + 1; else hi = mid; }
            rc[e] = (uint16_t)min(lo - (b + e), 65535);
        }
        mine += keep;
        if (e < per) prev = x;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int up = __shfl_up(inc, off); if (lane >= off) inc += up; }
    __syncthreads();   // every run is in registers
    if (lane == 63) wsum[wave] = (uint32_t)inc;
    __syncthreads();
    int base = inc - mine, tot = 0;
    for (int w2 = 0; w2 < JN_THREADS / 64; ++w2) { if (w2 < wave) base += (int)wsum[w2]; tot += (int)wsum[w2]; }
#pragma unroll
    for (int e = 0; e < PER; ++e)
        if (r[e] != 0xffffffffu) {
            if (cnt_out) cnt_out[base] = rc[e];
            v[base++] = r[e];
        }
    __syncthreads();
    return tot;
}

// ascending bitonic sort of P (a power of two) 64-bit keys in LDS
__device__ __forceinline__ void jn_sort64(unsigned long long *v, int P)
{
    for (int k2 = 2; k2 <= P; k2 <<= 1)
        for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
            for (int t = threadIdx.x; t < (P >> 1); t += JN_THREADS) {
                const int q = ((t & ~(j2 - 1)) << 1) | (t & (j2 - 1));
                const int p2 = q | j2;
                const unsigned long long x = v[q], y = v[p2];
                const bool up = (q & k2) == 0;
                if ((x > y) == up) { v[q] = y; v[p2] = x; }
            }
            __syncthreads();
        }
}

// ---- reverse neighbour lists: rev[c] = the (at most JN_RK) rows that list c, those that rank it
// highest first (key = position in the lister's list, then the lister's index: deterministic).
// Built for the columns [col0, col0 + ncols) only: every edge of every list is read (coalesced), the atomics, the
// edge records and the selection are those of the own columns -- a rank of a row-sharded run builds the reverse
// lists of the column range it owns and the ranks all-gather the slices (annchor_stream_join_rev_begin).
__global__ void k_st_rev_count(const int32_t *__restrict__ lists_all, int64_t n_edges, int64_t col0, int64_t ncols, int32_t *__restrict__ cnt)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_edges) return;
    const int64_t dst = (int64_t)lists_all[t] - col0;
    if (dst >= 0 && dst < ncols) atomicAdd(&cnt[dst], 1);      // (0x7fffffff = no entry: outside every range)
}

__global__ void k_st_rev_fill(const int32_t *__restrict__ lists_all, int64_t n_edges, int K, int64_t col0, int64_t ncols,
                              const int64_t *__restrict__ ptr, int32_t *__restrict__ cursor, unsigned long long *__restrict__ edges)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_edges) return;
    const int64_t dst = (int64_t)lists_all[t] - col0;
    if (dst < 0 || dst >= ncols) return;
    const int64_t src = t / K;
    const int e = (int)(t - src * K);
    const int pos = atomicAdd(&cursor[dst], 1);
    edges[ptr[dst] + pos] = ((unsigned long long)e << 32) | (unsigned long long)src;
}

__global__ void k_st_rev_select(const int64_t *__restrict__ ptr, const unsigned long long *__restrict__ edges, int64_t ncols,
                                int32_t *__restrict__ rev)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncols) return;
    const int64_t b = ptr[c], e = ptr[c + 1];
    // the JN_RK smallest keys, ascending, in registers: every edge is read once and bubbles in by compare-exchange (keys are
    // distinct: position << 32 | lister); JN_RK rounds over the list re-read every edge JN_RK times
    unsigned long long best[JN_RK];
#pragma unroll
    for (int r = 0; r < JN_RK; ++r) best[r] = ~0ull;
    for (int64_t q = b; q < e; ++q) {
        unsigned long long k = edges[q];
        if (k >= best[JN_RK - 1]) continue;
#pragma unroll
        for (int r = 0; r < JN_RK; ++r) {
            const unsigned long long lo = k < best[r] ? k : best[r], hi = k < best[r] ? best[r] : k;
            best[r] = lo;
            k = hi;
        }
    }
#pragma unroll
    for (int r = 0; r < JN_RK; ++r) rev[c * JN_RK + r] = best[r] == ~0ull ? 0x7fffffff : (int32_t)(best[r] & 0xffffffffull);
}

// -DJN_PROFILE: cycle sums of k_st_join_cands' phases (first-hop gather, sort, unique, second hop, sort, unique, selection, output), printed per launch
static unsigned long long *jn_prof_buffer(annchor_ctx *c)
{
#ifdef JN_PROFILE
    static unsigned long long *buf = nullptr;
    if (!buf) { (void)hipMalloc(&buf, 128); (void)hipMemset(buf, 0, 128); }
    else {
        unsigned long long h[16];
        (void)hipMemcpy(h, buf, 128, hipMemcpyDeviceToHost);
        double tot = 0; for (int i = 0; i < 8; ++i) tot += (double)h[i];
        fprintf(stderr, "[jn-prof]"); for (int i = 0; i < 8; ++i) fprintf(stderr, " %5.1f%%", 100.0 * (double)h[i] / tot); fprintf(stderr, "  (Mcycles %.1f)\n", tot / 1e6);
        if (h[11]) fprintf(stderr, "[jn-prof] row tiles %llu, overflowing %llu, mean first-hop ids %.0f, mean survivors of the full second hop %.0f\n", h[11], h[8], (double)h[9] / (double)h[11], (double)h[10] / (double)h[11]);
        (void)hipMemset(buf, 0, 128);
    }
    return buf;
#else
    (void)c;
    return nullptr;
#endif
}

// Candidate columns of one row tile: first hop = the current neighbours and reverse neighbours of
// its 128 rows; second hop = their neighbours and reverse neighbours (and the first hop itself),
// minus everything inside column tiles this row tile has already evaluated; sorted, distinct.
__global__ __launch_bounds__(JN_THREADS) void k_st_join_cands(const int32_t *__restrict__ lists_all, const int32_t *__restrict__ rev,
                                                             int K, int tile_begin, const uint32_t *__restrict__ eval_bits,
                                                             int eval_words, int eval_halves, int max_cols, uint32_t *__restrict__ ucand,
                                                             int32_t *__restrict__ ucount, unsigned long long *__restrict__ jprof)
{
#ifdef JN_PROFILE
    long long jt = clock64();
#define JP(i) { __syncthreads(); const long long n_ = clock64(); if (threadIdx.x == 0 && jprof) atomicAdd(jprof + i, (unsigned long long)(n_ - jt)); jt = n_; }
#else
#define JP(i)
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char jsm[];
    uint32_t *buf = reinterpret_cast<uint32_t *>(jsm);   // [JN_CAP]
    uint32_t *b1 = buf + JN_CAP;                         // [JN_B1]
    uint32_t *wsum = b1 + JN_B1;                         // [JN_THREADS / 64 + 1]
    __shared__ int nsurv_s;
    const int bt = blockIdx.x, I = tile_begin + bt;
    const int64_t grow0 = (int64_t)I * ST_T;
    const int KK = K + (rev ? JN_RK : 0);
    auto base_of = [&](uint32_t c, int e) -> int32_t {
        return e < K ? lists_all[(size_t)c * K + e] : rev[(size_t)c * JN_RK + (e - K)];
    };
    // ---- first hop: at most JN_B1 / 128 = 64 entries per row -- the reverse neighbours and the closest K0 list entries (beyond
    // n_neighbors = 50 the lists are longer than that; before the cap their first hop overran b1 and what the hardware drops
    // from an out-of-range LDS write was silently missing from the candidates)
    const int K0 = min(K, JN_B1 / ST_T - (rev ? JN_RK : 0)), KK0 = K0 + (rev ? JN_RK : 0);
    const int n0 = ST_T * KK0;
    int P0 = JN_THREADS;
    while (P0 < n0) P0 <<= 1;
    for (int t = threadIdx.x; t < P0; t += JN_THREADS) {
        int32_t id = 0x7fffffff;
        if (t < n0) {
            const int r = t / KK0, e = t % KK0;
            id = base_of((uint32_t)(grow0 + r), e < K0 ? e : K + (e - K0));
        }
        b1[t] = id == 0x7fffffff ? 0xffffffffu : (uint32_t)id;
    }
    __syncthreads();
    JP(0)
    jn_sort(b1, P0);
    JP(1)
    int n1 = jn_unique<JN_B1 / JN_THREADS>(b1, P0, wsum);
    JP(2)
    // ---- second hop, filtered, appended in any order (sorted below)
    // the row tile's evaluated-tile bitmap (1 KB at N = 10^6, 8 KB at 8 x 10^6) next to the sort buffers: the second hop's filter
    // was the second of two dependent global reads per entry
    // (eval_halves = 2: the lists of the row tile were kept by two workgroups of 64 rows, each with its own bitmap row; a tile
    // counts as evaluated where both rows say so)
    const uint32_t *eb = eval_bits + (size_t)bt * eval_halves * eval_words;
    int eb_and = eval_halves == 2 ? eval_words : 0;   // offset of the word to AND with (0: the word itself)
    if (eval_words <= JN_EB_WORDS) {
        uint32_t *ebl = wsum + JN_THREADS / 64 + 8;
        for (int t = threadIdx.x; t < eval_words; t += JN_THREADS) ebl[t] = eb[t] & eb[t + eb_and];
        eb = ebl;
        eb_and = 0;
        __syncthreads();
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (threadIdx.x == 0) nsurv_s = 0;
        __syncthreads();
        // (second attempt, after an overflow: every first-hop id still takes part, with itself and only its closest W2 - 1 list
        // entries -- n1 W2 <= JN_CAP fits whatever the filter lets through.  The first form kept a PREFIX of the sorted first-hop
        // ids: at n_neighbors = 62 the join passes then found nothing the tile phase had not)
        const int W2 = attempt == 0 ? KK + 1 : max(1, min(JN_CAP / max(n1, 1), K + 1));
        const int n2 = n1 * W2;
        // Eight entries per thread and step, their two dependent global reads (the neighbour's list entry, then the evaluated-tile
        // word of its tile) issued as batches: with one entry per step every iteration waited out both round trips alone (two waves
        // per SIMD: nothing to hide them behind) -- 5.5 ms per pass at C3, most of it this loop.  Survivors are appended with one
        // LDS atomic per wave and step (the order of the list is arbitrary; it is sorted below).
        const float inv_kk1 = 1.0f / (float)W2;
        const int lane_j = threadIdx.x & 63;
        for (int t0 = threadIdx.x; t0 - (int)threadIdx.x < n2; t0 += JN_THREADS * 8) {
            int32_t id[8];
            uint32_t ew[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = min(t0 + u * JN_THREADS, n2 - 1);
                const int q = (int)(((float)t + 0.5f) * inv_kk1);      // t / W2, exact for t < 2^22
                const int e = t - q * W2;
                const uint32_t c = b1[q];
                const int32_t *src;
                if (attempt == 0) src = e < K ? lists_all + (size_t)c * K + e : (rev && e < KK ? rev + (size_t)c * JN_RK + (e - K) : nullptr);
                else src = e < W2 - 1 ? lists_all + (size_t)c * K + e : nullptr;
                id[u] = src ? *src : (int32_t)c;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int J = id[u] == 0x7fffffff ? 0 : id[u] >> 7;   // ST_T = 128
                ew[u] = eb[J >> 5] & eb[(J >> 5) + eb_and];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool in = t0 + u * JN_THREADS < n2 && id[u] != 0x7fffffff;
                const int J = in ? id[u] >> 7 : 0;
                const bool keep = in && !((ew[u] >> (J & 31)) & 1u);
                const unsigned long long kb = __ballot(keep);
                if (kb) {
                    int base = 0;
                    if (lane_j == 0) base = atomicAdd(&nsurv_s, __popcll(kb));
                    base = __shfl(base, 0);
                    const int pos = base + __popcll(kb & ((1ull << lane_j) - 1ull));
                    if (keep && pos < JN_CAP) buf[pos] = (uint32_t)id[u];
                }
            }
        }
        __syncthreads();
#ifdef JN_PROFILE
        if (threadIdx.x == 0 && jprof && attempt == 0) {
            atomicAdd(jprof + 11, 1ull); atomicAdd(jprof + 9, (unsigned long long)n1); atomicAdd(jprof + 10, (unsigned long long)nsurv_s);
            if (nsurv_s > JN_CAP) atomicAdd(jprof + 8, 1ull);
        }
#endif
        if (nsurv_s <= JN_CAP) break;
        // overflow (which entries an atomic append drops is arbitrary): once more, narrower (above)
        __syncthreads();
    }
    JP(3)
    const int ns = min(nsurv_s, JN_CAP);
    int P = JN_THREADS;
    while (P < ns) P <<= 1;
    for (int t = ns + threadIdx.x; t < P; t += JN_THREADS) buf[t] = 0xffffffffu;
    __syncthreads();
    jn_sort(buf, P);
    JP(4)
    int nu = jn_unique<JN_CAP / JN_THREADS>(buf, P, wsum, b1);   // b1 (free by now) receives the multiplicities
    JP(5)
    if (nu > max_cols) {
        // More candidates than this pass may evaluate: keep the max_cols reached over the most
        // two-hop paths from the tile's rows (the tile analogue of ranking pairs by their
        // probability and refining the top of the list, annchor.py:444-457), ties to the smaller index.
        constexpr int PER64 = JN_CAP / JN_THREADS;
        uint32_t ids[PER64], cn[PER64];
#pragma unroll
        for (int e = 0; e < PER64; ++e) {
            const int t = e * JN_THREADS + threadIdx.x;
            ids[e] = t < nu ? buf[t] : 0xffffffffu;
            cn[e] = t < nu ? b1[t] : 0u;
        }
        __syncthreads();
        unsigned long long *k64 = reinterpret_cast<unsigned long long *>(jsm);   // [JN_CAP] over buf + b1
        int P64 = JN_THREADS;
        while (P64 < nu) P64 <<= 1;
#pragma unroll
        for (int e = 0; e < PER64; ++e) {
            const int t = e * JN_THREADS + threadIdx.x;
            if (t < P64) k64[t] = t < nu ? ((unsigned long long)(0xffffffffu - cn[e]) << 32) | ids[e] : ~0ull;
        }
        __syncthreads();
        jn_sort64(k64, P64);
        uint32_t keep[PER64];
#pragma unroll
        for (int e = 0; e < PER64; ++e) {
            const int t = e * JN_THREADS + threadIdx.x;
            keep[e] = t < max_cols ? (uint32_t)(k64[t] & 0xffffffffull) : 0xffffffffu;
        }
        __syncthreads();
        int P2 = JN_THREADS;
        while (P2 < max_cols) P2 <<= 1;
#pragma unroll
        for (int e = 0; e < PER64; ++e) {
            const int t = e * JN_THREADS + threadIdx.x;
            if (t < P2) buf[t] = keep[e];
        }
        __syncthreads();
        jn_sort(buf, P2);   // back to index order (row gathers of neighbouring indices share pages)
        nu = max_cols;
    }
    JP(6)
    const int padded = (nu + ST_T - 1) / ST_T * ST_T;
    uint32_t *dst = ucand + (size_t)bt * JN_CAP;
    for (int t = threadIdx.x; t < padded; t += JN_THREADS) dst[t] = t < nu ? buf[t] : 0xffffffffu;
    if (threadIdx.x == 0) ucount[bt] = nu;
    JP(7)
#undef JP
}

template <int DIM, int KMAX> __global__ __launch_bounds__(ST_THREADS, (DIM <= 128 ? 2 : 1)) void k_st_join(KnnArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    KnnShared<DIM, KMAX> &sh = *reinterpret_cast<KnnShared<DIM, KMAX> *>(smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    using Half = KnnHalf<KMAX>;
    const int hsel = Half::ON ? (int)(blockIdx.x & 1) : 0;
    const int bid = Half::ON ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;
    auto own = [&](int row) { return !Half::ON || (row >> 6) == hsel; };
    int bt;
    {
        const int nb_ = Half::ON ? (int)(gridDim.x >> 1) : (int)gridDim.x, q = nb_ >> 3, r = nb_ & 7, x = bid & 7, y = bid >> 3;
        bt = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    }
    const int I = a.tile_begin + bt;
    const int64_t grow0 = (int64_t)I * ST_T;
    const int K = a.K;
    float areg[DIM / 2];
    {
        const float *xr = a.Rs + (size_t)(grow0 + wave * 32 + (lane & 31)) * DIM + (lane >> 5);
#pragma unroll
        for (int s = 0; s < DIM / 2; ++s) areg[s] = xr[2 * s];
    }
    float ri[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ri[r] = a.rr[grow0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
    // the lists as the previous phase left them
    for (int q = threadIdx.x; q < ST_T * KMAX; q += ST_THREADS) {
        const int row = q / KMAX, e = q - row * KMAX;
        if (!own(row)) continue;
        sh.list_d[Half::lr(row)][e] = e < K ? a.out_d2[((size_t)bt * ST_T + row) * K + e] : INFINITY;
        sh.list_c[Half::lr(row)][e] = e < K ? a.lists_all[((size_t)grow0 + row) * K + e] : 0x7fffffff;
    }
    __syncthreads();
    if (threadIdx.x < ST_T) {
        const int row = threadIdx.x;
        const bool real = a.rr[grow0 + row] < INFINITY;
        sh.thr[row] = real && own(row) ? sh.list_d[Half::lr(row)][K - 1] : -1.f;
        sh.cnt[row] = 0;
    }
    const int nu = a.ucount[bt];
    const int nchunks = (nu + ST_T - 1) / ST_T;
    const uint32_t *ulist = a.ucand + (size_t)bt * a.ucap;
    SlabStage<DIM> st;
    st.J = -1;
    st.ins = 0;
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch)
        knn_process_tile<DIM, KMAX, true>(sh, a, ch, ch + 1 < nchunks ? ch + 1 : -1, st, areg, ri, wave * 32, grow0, K, nullptr, ulist);
    __syncthreads();
    for (int q = threadIdx.x; q < ST_T * K; q += ST_THREADS) {
        const int row = q / K, e = q - row * K;
        if (!own(row)) continue;
        a.out_d2_new[((size_t)bt * ST_T + row) * K + e] = sh.list_d[Half::lr(row)][e];
        a.out_col_new[((size_t)bt * ST_T + row) * K + e] = sh.list_c[Half::lr(row)][e];
    }
    if (threadIdx.x == 0) atomicAdd(a.evals + 2, (unsigned long long)nchunks);   // slot 2: join chunks (0: tile phase, 1: pass yield)
    int ins = st.ins;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ins += __shfl_xor(ins, off);
    if (lane == 0 && ins) atomicAdd(a.updates, (unsigned long long)ins);
}

// exact float32 distances of the selected neighbours + final per-row ordering
__global__ __launch_bounds__(256) void k_st_finalize(const float *__restrict__ Xs, const float *__restrict__ Rs,
                                                    const int64_t *__restrict__ perm_all, int dimp,
                                                    int64_t row_begin, int64_t rows, int K, const int32_t *__restrict__ col,
                                                    int64_t *__restrict__ oidx, float *__restrict__ odist)
{
    // 16 lanes per (row, entry)
    const int sub = threadIdx.x & 15;
    const int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (t >= rows * K) return;
    const int64_t r = t / K;
    const int e = (int)(t - r * K);
    const int32_t cc = col[t];
    double acc = 0;
    const bool ok = cc != 0x7fffffff;
    if (ok) {
        const float *x = Rs + (size_t)(row_begin + r) * dimp, *y = Xs + (size_t)cc * dimp;
        for (int k = sub; k < dimp; k += 16) { double d = (double)x[k] - (double)y[k]; acc += d * d; }
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 16);
    if (sub == 0) {
        odist[r * K + e] = ok ? (float)sqrt(acc) : INFINITY;
        oidx[r * K + e] = ok ? perm_all[cc] : -1;
    }
}

// The split-fp16 kernels hand on EXACT float32 sums (x - y)^2 of the original rows (their re-ranking epilogue): the distances
// are their square roots -- no second pass over 2 x 512 bytes per kept entry (19 ms at N = 8 x 10^6).  float32 accumulation, like
// the reference's np.linalg.norm on float32 rows (distances.py:8-13); ANNCHOR_ST_FINALIZE_F64=1 keeps the float64 pass.
__global__ void k_st_finalize_from_d2(const int64_t *__restrict__ perm_all, int64_t n, const float *__restrict__ d2, const int32_t *__restrict__ col,
                                      int64_t *__restrict__ oidx, float *__restrict__ odist)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int32_t cc = col[t];
    const bool ok = cc != 0x7fffffff;
    odist[t] = ok ? sqrtf(d2[t]) : INFINITY;
    oidx[t] = ok ? perm_all[cc] : -1;
}

__global__ void k_st_rowsort(int64_t rows, int K, int64_t *__restrict__ oidx, float *__restrict__ odist)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    int64_t *ii = oidx + r * K;
    float *dd = odist + r * K;
    for (int a = 1; a < K; ++a) {  // insertion sort by (distance, id): K <= 32
        const float d = dd[a];
        const int64_t id = ii[a];
        int p = a;
        while (p > 0 && (dd[p - 1] > d || (dd[p - 1] == d && ii[p - 1] > id))) { dd[p] = dd[p - 1]; ii[p] = ii[p - 1]; --p; }
        dd[p] = d; ii[p] = id;
    }
}

// final graph rows in the shard's own row order: row perm[r] - base gets (self, 0.0) followed by
// the K ordered neighbours, as int64 / float64 (the reference's neighbor_graph dtypes)
__global__ void k_st_emit(const int64_t *__restrict__ perm, int64_t row_begin, int64_t rows, int K, int64_t base, int64_t n_local,
                          const int64_t *__restrict__ idx, const float *__restrict__ dist, int64_t *__restrict__ oidx,
                          double *__restrict__ odist)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int k = K + 1;
    if (t >= rows * k) return;
    const int64_t r = t / k;
    const int e = (int)(t - r * k);
    const int64_t g = perm[row_begin + r];
    if (g < 0) return;   // padding row
    const int64_t loc = g - base;
    if (loc < 0 || loc >= n_local) return;
    oidx[loc * k + e] = e == 0 ? g : idx[r * K + e - 1];
    odist[loc * k + e] = e == 0 ? 0.0 : (double)dist[r * K + e - 1];
}

// query results in the queries' own order: row perm_q[r] gets its K ordered neighbours (no self column)
__global__ void k_st_emit_query(const int64_t *__restrict__ perm_q, int64_t rows, int K, int64_t nq, const int64_t *__restrict__ idx,
                                const float *__restrict__ dist, int64_t *__restrict__ oidx, double *__restrict__ odist)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * K) return;
    const int64_t r = t / K;
    const int e = (int)(t - r * K);
    const int64_t g = perm_q[r];
    if (g < 0 || g >= nq) return;   // padding row
    oidx[g * K + e] = idx[t];
    odist[g * K + e] = (double)dist[t];
}

template <int DIM, int KMAX> static int launch_knn2(annchor_ctx *c, const KnnArgs &a, bool join)
{
    const size_t lds = sizeof(KnnShared<DIM, KMAX>);
    ANN_REQUIRE(c, lds <= 160 * 1024, ANNCHOR_ELIMIT, "streamed k-NN needs %zu B of LDS", lds);
    if (join) {
        ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)k_st_join<DIM, KMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_st_join<DIM, KMAX><<<a.tile_count * (KnnHalf<KMAX>::ON ? 2 : 1), ST_THREADS, lds, c->stream>>>(a);
    } else {
        ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)k_st_knn<DIM, KMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_st_knn<DIM, KMAX><<<a.tile_count * (KnnHalf<KMAX>::ON ? 2 : 1), ST_THREADS, lds, c->stream>>>(a);
    }
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}

template <int DIM> static int launch_knn(annchor_ctx *c, const KnnArgs &a, bool join)
{
    // list capacity by n_neighbors: 16 / 32 entries per row (two workgroups per CU), 64 (the lists alone are 66 KB: one per CU)
    return a.K <= 16 ? launch_knn2<DIM, 16>(c, a, join) : a.K <= ST_KMAX ? launch_knn2<DIM, ST_KMAX>(c, a, join) : a.K <= ST_KMAX_BIG ? launch_knn2<DIM, ST_KMAX_BIG>(c, a, join) : launch_knn2<DIM, ST_KMAX_HUGE>(c, a, join);
}

static int launch_by_dim(annchor_ctx *c, const KnnArgs &a, int dim_padded, bool join, bool exact = false)
{
    {
        // ANNCHOR_ST_KERNEL=4wave: the exact-f32 kernels below for every shape (A/B runs, tests); default: the split-fp16
        // kernel (knnbf.hip) where the shape fits it -- the join passes follow the tile phase's kernel (a tile phase that fell
        // back to the exact kernel is followed by exact join passes)
        static const char *kern = getenv("ANNCHOR_ST_KERNEL");
        StreamState *st = state_of(c, false);
        if (!join && st) st->last_kernel = 0;
        const bool want_split = !exact && (!kern || strcmp(kern, "4wave")) && (!join || (st && st->last_kernel == 1));
        if (want_split) {
            bool handled = false;
#ifdef ST_PAIR_KERNEL
            // experiment (tools/experiments/knnbf2.hip, not part of the library): two adjacent row tiles per workgroup on one
            // column stream; ANNCHOR_ST_KERNEL=bf4 keeps one row tile per workgroup
            if (!join && !(kern && !strcmp(kern, "bf4"))) {
                ANN_TRY(ann_stream_launch_knnbf2(c, a, dim_padded, &handled));
                if (handled) {
                    if (st) st->last_kernel = 1;
                    return ANNCHOR_OK;
                }
            }
#endif
            if (!join) {   // (two-stage form: knnh.hip)
                ANN_TRY(ann_stream_launch_knnh(c, a, dim_padded, &handled, ann_stream_launch_knnbf));
                if (st) st->last_two_stage = handled;
            }
            if (!handled && !(kern && !strcmp(kern, "bk") && dim_padded == 128)) ANN_TRY(ann_stream_launch_knnbf(c, a, dim_padded, &handled, join));
            if (!handled) ANN_TRY(ann_stream_launch_knnbk(c, a, dim_padded, &handled, join));   // padded dim 256 .. 1024: k-blocked
            if (handled) {
                if (!join && st) st->last_kernel = 1;
                return ANNCHOR_OK;
            }
        }
    }
    ANN_REQUIRE(c, dim_padded <= 256, ANNCHOR_ELIMIT, "padded dim %d: beyond 256 dimensions only the split-fp16 kernel exists (n_neighbors <= 63)%s",
                dim_padded, exact ? "; the data is too ill-conditioned for it (rows far from the anchors' centre with neighbours very close together)" : "");
    switch (dim_padded) {
    case 32: return launch_knn<32>(c, a, join);
    case 64: return launch_knn<64>(c, a, join);
    case 128: return launch_knn<128>(c, a, join);
    case 256: return launch_knn<256>(c, a, join);
    default: ann_set_err(c, "unsupported padded dim %d", dim_padded); return ANNCHOR_ELIMIT;
    }
}

// The p_work budget of one row tile -- ceil(p_work * n_tiles) tile evaluations, the streamed
// form's share of ITS brute force (n_tiles per row tile) -- split between the tile phase and the
// join passes: every pass may evaluate up to per_pass runs of 128 gathered columns (an eighth of
// the budget, at most 24), the tile phase gets the rest.
extern "C" int annchor_stream_budget(int32_t n_tiles, double p_work, int32_t join_passes, int32_t *total, int32_t *tile_phase,
                                     int32_t *per_pass)
{
    if (n_tiles < 1 || join_passes < 0 || !total || !tile_phase || !per_pass) return ANNCHOR_EINVAL;
    const double mt = p_work >= 1.0 ? (double)n_tiles : std::ceil(p_work * (double)n_tiles);
    const int T = (int)std::max(1.0, std::min(mt, (double)n_tiles));
    static const int div_ = getenv("ANNCHOR_JOIN_DIV") ? atoi(getenv("ANNCHOR_JOIN_DIV")) : 8;
    static const int pp_max = getenv("ANNCHOR_JOIN_PP_MAX") ? atoi(getenv("ANNCHOR_JOIN_PP_MAX")) : 24;   // runs of 128 gathered columns per pass and row tile
    int pp = std::min(pp_max, std::max(1, T / div_));
    if (join_passes == 0) pp = 0;
    int tp = T - pp * join_passes;
    if (tp < 1) tp = 1;
    if (T >= n_tiles) tp = n_tiles;   // the full budget evaluates every tile: nothing is left to join
    *total = T; *tile_phase = tp; *per_pass = pp;
    return ANNCHOR_OK;
}

void ann_stream_free_run(StreamState *s)
{
    delete s->run;
    s->run = nullptr;
    s->run_finished = false;
}

// The graph build of row tiles [tile_begin, tile_begin + tile_count) against ALL column tiles runs
// in three steps shared by the one-call form (annchor_stream_knn), the multi-rank form
// (annchor_stream_knn_begin / _join / _end: the ranks all-gather their lists between the steps)
// and queries:
//   knn_tile_phase  budgeted tile evaluation (k_st_knn); the lists stay on the device
//   knn_join_pass   one pass over the neighbours' neighbours (k_st_join_cands + k_st_join)
//   knn_finish      exact float32 distances of the kept neighbours, final order, ids
// Rank key and valid bound of EVERY (row tile, column tile) pair of a launch, ahead of the tile kernel:
//   scr_lb[I][J]  = max over anchors of the gap between the two tiles' intervals (the triangle bound of utils.py:274-301 on
//                   intervals, with the slack for the float32 rounding of the anchor distances),
//   scr_key[I][J] = squared distance between the tiles' mean anchor vectors (+inf: never a candidate).
// The tile kernels used to do this themselves, each workgroup for its row tile against all column tiles: na x 12 bytes per
// pair from the tables, O(n_tiles^2) -- at N = 8 x 10^6 (62 500 tiles) 24 MB per row tile and more than half of the tile
// kernel's wave cycles (-DST_PROFILE), for ~660 tiles evaluated per row tile.  Pruning it does not work: the box of 64
// neighbouring tiles' means bounds nothing in a 32-anchor embedding of 8-dimensional data (58-96 % of the runs would have to be
// ranked to prove a selection round's cut), and ranking runs by their centroid costs recall (C3 0.9972 -> 0.990: the early
// stop fires on the worse order).  So the work stays O(n_tiles^2) but becomes a tiled pass: a workgroup takes RK_I row tiles x
// RK_J column tiles, a thread keeps ONE column tile's 3 x na table entries in registers (coalesced reads) and meets the row
// tiles' entries as LDS broadcasts -- 12 bytes of table traffic per pair instead of 384, the same arithmetic in the same order
// (the keys and bounds, hence the selection rounds and the graph, are bit for bit what the in-kernel ranking produced).
#define RK_I 32
#define RK_J 256
// (round 6) Branch-free and two anchors per instruction: anchors beyond a.na are zeros on both sides (a zero interval against a zero
// interval: gap 0, slack 0, difference of the means 0 -- lb = max(lb, 0) and lbc + 0 change nothing), so the anchor loop has no
// `an < na` test (it was a scalar compare + branch per anchor and pair); the subtractions, the slack and the squares are packed
// float32 operations (v_pk_add_f32 / v_pk_mul_f32: the same IEEE results lane by lane), the running maximum and the running sum
// stay scalar in the original order -- keys and bounds bit for bit as before.  N = 8 x 10^6 (3.9 x 10^9 pairs x 32 anchors): see
// NOTEBOOK.md, round 6.
typedef float rk_f2 __attribute__((ext_vector_type(2)));
template <int NA> __global__ __launch_bounds__(RK_J) void k_st_rank_pairs(KnnArgs a)
{
    static_assert(NA % 2 == 0, "anchors in pairs");
    __shared__ __attribute__((aligned(8))) float rowt[RK_I][4][NA];   // [row tile]{lo, hi, mid, |hi|}[anchor] of the block's row tiles
    const int J = blockIdx.x * RK_J + threadIdx.x;
    const int i0 = blockIdx.y * RK_I, ni = min(RK_I, a.tile_count - i0);
    for (int t = threadIdx.x; t < ni * NA; t += RK_J) {
        const int i = t / NA, an = t - i * NA;
        const int I = a.tile_begin + i0 + i;
        const bool ok = an < a.na;
        const float h = ok ? a.rhi[(size_t)an * a.nt_r + I] : 0.f;
        rowt[i][0][an] = ok ? a.rlo[(size_t)an * a.nt_r + I] : 0.f;
        rowt[i][1][an] = h;
        rowt[i][2][an] = ok ? a.rmid[(size_t)an * a.nt_r + I] : 0.f;
        rowt[i][3][an] = fabsf(h);
    }
    rk_f2 lj[NA / 2], hj[NA / 2], mj[NA / 2], ahj[NA / 2];
    const bool have = J < a.nt_all;
#pragma unroll
    for (int an = 0; an < NA; ++an) {
        const bool ok = have && an < a.na;
        const float l = ok ? a.lo[(size_t)an * a.nt_all + J] : 0.f, h = ok ? a.hi[(size_t)an * a.nt_all + J] : 0.f;
        lj[an >> 1][an & 1] = l;
        hj[an >> 1][an & 1] = h;
        mj[an >> 1][an & 1] = ok ? a.mid[(size_t)an * a.nt_all + J] : 0.f;
        ahj[an >> 1][an & 1] = fabsf(h);
    }
    __syncthreads();
    if (!have) return;
    for (int i = 0; i < ni; ++i) {
        const int I = a.tile_begin + i0 + i;
        float lb = 0.f, lbc = 0.f;
#pragma unroll
        for (int q = 0; q < NA / 2; ++q) {
            const rk_f2 loI = *reinterpret_cast<const rk_f2 *>(&rowt[i][0][2 * q]), hiI = *reinterpret_cast<const rk_f2 *>(&rowt[i][1][2 * q]);
            const rk_f2 midI = *reinterpret_cast<const rk_f2 *>(&rowt[i][2][2 * q]), ahiI = *reinterpret_cast<const rk_f2 *>(&rowt[i][3][2 * q]);
            const rk_f2 g1 = loI - hj[q], g2 = lj[q] - hiI;
            // slack for the float32 rounding of D (bounds must stay valid lower bounds)
            const rk_f2 u = 4e-6f * (ahj[q] + ahiI);
            const rk_f2 v = rk_f2{fmaxf(g1.x, g2.x), fmaxf(g1.y, g2.y)} - u;
            lb = fmaxf(fmaxf(lb, v.x), v.y);
            const rk_f2 dm = mj[q] - midI;
            const rk_f2 d2 = dm * dm;
            lbc += d2.x;   // rank key: squared L2 distance between the tiles' mean anchor vectors (summed in anchor order)
            lbc += d2.y;
        }
        a.scr_key[(size_t)(i0 + i) * a.nt_all + J] = ((J == I && !a.query) || !(lbc < INFINITY)) ? INFINITY : lbc;   // +inf: never a candidate
        a.scr_lb[(size_t)(i0 + i) * a.nt_all + J] = lb;
    }
}

static int knn_tile_phase(annchor_ctx *c, StreamState *s, KnnArgs &a, int dim_padded, int tile_budget, bool record_tiles)
{
    const int K = a.K;
    const int64_t rows = (int64_t)a.tile_count * ST_T;
    ANN_TRY(sreserve(c, s->out_d2, sizeof(float) * (size_t)rows * K));
    ANN_TRY(sreserve(c, s->out_col, sizeof(int32_t) * (size_t)rows * K));
    ANN_TRY(sreserve(c, s->evals, 64));
    ANN_CHECK_HIP(c, hipMemsetAsync(s->evals.p, 0, 64, c->stream));
    a.max_tiles = std::max(1, std::min(tile_budget, a.nt_all));
    a.out_d2 = s->out_d2.as<float>(); a.out_col = s->out_col.as<int32_t>();
    ANN_TRY(sreserve(c, s->scr_key, sizeof(float) * (size_t)a.tile_count * (size_t)a.nt_all));
    ANN_TRY(sreserve(c, s->scr_lb, sizeof(float) * (size_t)a.tile_count * (size_t)a.nt_all));
    a.scr_key = s->scr_key.as<float>(); a.scr_lb = s->scr_lb.as<float>();
    a.scr_cl = nullptr;
    static const int cl_min = getenv("ANNCHOR_ST_SHORT_LIST_MIN") ? atoi(getenv("ANNCHOR_ST_SHORT_LIST_MIN")) : 2 * ST_CL_CAP;   // (tests lower it)
    if (a.nt_all > cl_min && !getenv("ANNCHOR_ST_NO_SHORT_LIST")) {   // (short rows are swept as they are)
        ANN_TRY(sreserve(c, s->scr_cl, sizeof(uint32_t) * 3 * ST_CL_CAP * (size_t)a.tile_count));
        a.scr_cl = s->scr_cl.as<uint32_t>();
    }
    a.evals = s->evals.as<unsigned long long>();
    a.eval_bits = nullptr;
    a.eval_words = (a.nt_all + 31) / 32;
    a.eval_halves = (a.K > ST_KMAX_BIG) ? 2 : 1;   // (KnnHalf: two workgroups per row tile, a bitmap row each)
    if (record_tiles || ann_stream_knnh_fits(a, dim_padded)) {   // (the two-stage kernel resumes from the warm-up's record)
        const size_t nb = sizeof(uint32_t) * (size_t)a.tile_count * a.eval_halves * a.eval_words;
        ANN_TRY(sreserve(c, s->eval_bits, nb));
        ANN_CHECK_HIP(c, hipMemsetAsync(s->eval_bits.p, 0, nb, c->stream));
        a.eval_bits = s->eval_bits.as<uint32_t>();
    }
    a.lists_all = nullptr; a.ucand = nullptr; a.ucount = nullptr; a.ucap = JN_CAP; a.out_d2_new = nullptr; a.out_col_new = nullptr;
    ANN_TRY(sreserve(c, s->guard_tiles, sizeof(uint32_t) * 4 * (size_t)a.tile_count));
    ANN_CHECK_HIP(c, hipMemsetAsync(s->guard_tiles.p, 0, sizeof(uint32_t) * 4 * (size_t)a.tile_count, c->stream));
    a.guard_tiles = s->guard_tiles.as<uint32_t>();
    a.updates = nullptr;
    {
        // A row tile stops spending its tile budget when ST_EARLY_WINDOW consecutive ranked tiles replaced fewer
        // than ANNCHOR_TILE_YIELD of its 128 x K list entries -- the yield rule that ends the join
        // passes.  Only builds followed by join passes stop early (the passes pick up what the tail of the
        // ranking would have found: C3 0.292 -> ~0.25 s at recall 0.9989 -> ~0.9985); the budget stays an upper bound.
        const char *ew = getenv("ANNCHOR_ST_EARLY_WINDOW"), *et = getenv("ANNCHOR_ST_EARLY_TAU");
        a.early_window = record_tiles ? (ew ? atoi(ew) : ST_EARLY_WINDOW) : 0;
        a.early_tau = et ? atoi(et) : std::max(1, (int)std::lround(ANNCHOR_TILE_YIELD * ST_T * a.K));
    }
    a.prof = nullptr;
#ifdef ST_PROFILE
    static unsigned long long *d_prof = nullptr;
    if (!d_prof) (void)hipMalloc(&d_prof, 128);
    (void)hipMemsetAsync(d_prof, 0, 128, c->stream);
    a.prof = d_prof;
#endif
    ANN_CHECK_HIP(c, hipEventRecord(c->call_a, c->stream));
    a.pre_ranked = 0;
    if (a.na <= 64 && !getenv("ANNCHOR_ST_RANK_IN_KERNEL")) {   // (the switch: every workgroup ranks its own row tile, as before round 5)
        ProfScope ps(c, "stream_rank_tile_pairs", (double)a.tile_count * a.nt_all * 8.0);
        const dim3 grid((unsigned)((a.nt_all + RK_J - 1) / RK_J), (unsigned)((a.tile_count + RK_I - 1) / RK_I));
        if (a.na <= 32) k_st_rank_pairs<32><<<grid, RK_J, 0, c->stream>>>(a);
        else k_st_rank_pairs<64><<<grid, RK_J, 0, c->stream>>>(a);
        ANN_CHECK_HIP(c, hipGetLastError());
        a.pre_ranked = 1;
    }
    {
        // algorithmic flops are data dependent (tiles that survive the bound): reported by the caller from tile_evals
        ProfScope ps(c, "stream_tile_gemm_topk", 0.0);
        ANN_TRY(launch_by_dim(c, a, dim_padded, false));
    }
    s->last_guard_rows = 0;
    s->last_repaired = false;
    if (s->last_kernel != 0) {
        // The split-fp16 kernel keeps K + 2 columns per row by a distance that is off by ~2^-22 |x||y| and re-ranks them exactly;
        // its epilogue counts the rows whose K-th exact distance comes within the MEASURED error of the list's last approximate
        // entry -- rows where a neighbour may have been left outside the list.  Well-conditioned data flags (almost) none; when
        // more than 1 row in 200 is flagged (tight clusters far from the centre: |x|^2 >> d^2) the tile phase runs again on
        // the exact-f32 kernel.
        unsigned long long flagged = 0;
        unsigned long long fl2[2] = {0, 0};
        ANN_TRY(ann_d2h(c, fl2, a.evals + 3, 16));
        flagged = fl2[0];
        s->last_fetched_tiles = (int64_t)fl2[1];   // slot 4: column tiles the paired kernel fetched (0: one row tile per workgroup)
        if (getenv("ANNCHOR_ST_VERBOSE")) fprintf(stderr, "annchor: tile phase fetched %llu column tiles\n", fl2[1]);
        s->last_guard_rows = (int64_t)flagged;
        static const bool no_fallback = getenv("ANNCHOR_ST_NO_FALLBACK") != nullptr;
        // (round 6) Every row tile that holds a flagged row is done again with float32 DIFFERENCES (repair.hip: the reference's own
        // arithmetic, annchor/distances.py:8-13) over the column tiles it evaluated: exact whatever the conditioning and the
        // dimension -- before, <= 1 row in 200 was let through, beyond that the phase was repeated on the exact-f32 MFMA kernel (the
        // same expanded form in float32: not exact on such data either) and beyond 256 dimensions only a warning was printed.
        // ANNCHOR_ST_FALLBACK=rerun keeps the old behaviour for A/B runs.
        static const bool rerun = getenv("ANNCHOR_ST_FALLBACK") && !strcmp(getenv("ANNCHOR_ST_FALLBACK"), "rerun");
        if (flagged > 0 && !no_fallback && !rerun) {
            if ((int64_t)flagged > std::max<int64_t>(8, rows / 200))
                fprintf(stderr, "annchor: streamed tile phase: %llu of %lld rows have neighbours closer together than float32-grade products of |x|^2 "
                                "resolve (|x|^2 >> d^2); their row tiles are evaluated again with float32 differences (exact, slower)\n",
                        flagged, (long long)rows);
            ProfScope ps(c, "stream_tile_exact_repair", 0.0);
            ANN_TRY(ann_stream_repair_flagged(c, s, a, dim_padded, a.guard_tiles, (int64_t)flagged));
            s->last_repaired = true;
        } else if ((int64_t)flagged > std::max<int64_t>(8, rows / 200) && !no_fallback && dim_padded > 256) {
            // (the exact-f32 tile kernel stops at padded dim 256: the caller sees the count through annchor_stream_last_kernel)
            fprintf(stderr, "annchor: streamed tile phase: %llu of %lld rows have neighbours closer together than float32-grade products of |x|^2 "
                            "resolve (|x|^2 >> d^2); beyond 256 dimensions there is no exact-f32 tile kernel to repeat the phase on -- the lists "
                            "of those rows may miss a neighbour (reported distances stay exact)\n", flagged, (long long)rows);
        } else if ((int64_t)flagged > std::max<int64_t>(8, rows / 200) && !no_fallback) {
            fprintf(stderr, "annchor: streamed tile phase: %llu of %lld rows have neighbours closer together than float32-grade products of |x|^2 "
                            "resolve (|x|^2 >> d^2); running the exact float32 tile kernel instead\n", flagged, (long long)rows);
            ANN_CHECK_HIP(c, hipMemsetAsync(s->evals.p, 0, 64, c->stream));
            if (a.eval_bits) ANN_CHECK_HIP(c, hipMemsetAsync(s->eval_bits.p, 0, sizeof(uint32_t) * (size_t)a.tile_count * a.eval_halves * a.eval_words, c->stream));
            ProfScope ps(c, "stream_tile_gemm_topk_exact_rerun", 0.0);
            ANN_TRY(launch_by_dim(c, a, dim_padded, false, true));
        }
    }
    ANN_CHECK_HIP(c, hipEventRecord(c->call_b, c->stream));
    c->call_timed = true;
#ifdef ST_PROFILE
    {
        unsigned long long hp[8];
        ANN_TRY(ann_d2h(c, hp, a.prof, 64));
        static const char *names4[8] = {"barrier before stash", "stash (incl. global-load wait)", "barrier after stash",
                                        "MFMA stream + thresholds", "survivor inserts", "merge", "tile end", "candidate scan + rest"};
        static const char *names8[8] = {"MFMA stream", "barrier after stream", "next-tile choice", "test + inserts", "LDS-DMA requests",
                                        "barrier after requests", "merge (+ publish / prologue / tail)", "ranking + selection + rest"};
        const bool four = getenv("ANNCHOR_ST_KERNEL") && !strcmp(getenv("ANNCHOR_ST_KERNEL"), "4wave");
        static const char *names2[8] = {"operand reads + MFMA stream", "survivor inserts + merge", "waits in front of a slab", "next-tile choice",
                                        "selection rounds", "merge of the round lists", "prologue + ranking", "epilogue + rest"};
        const char **names = s->last_fetched_tiles > 0 ? names2 : (four || dim_padded > 128 || a.K > ST_KMAX) ? names4 : names8;
        double tot = 0;
        for (int i = 0; i < 8; ++i) tot += (double)hp[i];
        for (int i = 0; i < 8; ++i) fprintf(stderr, "[st-prof] %-32s %6.2f %%\n", names[i], 100.0 * (double)hp[i] / tot);
        {   // knn8.hip's extra slots 8..15 (sub-segments; already contained in the eight above)
            unsigned long long hx[8];
            ANN_TRY(ann_d2h(c, hx, a.prof + 8, 64));
            for (int i = 0; i < 8; ++i)
                if (hx[i]) fprintf(stderr, "[st-prof]    sub %d %36s %6.2f %%\n", i + 8, "", 100.0 * (double)hx[i] / tot);
        }
    }
#endif
    return ANNCHOR_OK;
}

// reverse lists of the columns [col0, col0 + ncols) from the lists of every ordered row: count, scan, fill, select (the
// lists are short: K on average) -> out [col0 .. col0 + ncols)[JN_RK] of an int32 [n_all][JN_RK] buffer
static int knn_reverse_lists(annchor_ctx *c, StreamState *s, const int32_t *lists_all, int64_t n_all, int K, int64_t col0, int64_t ncols,
                             int32_t *dst /* [ncols][JN_RK]: the first own column's entries */)
{
    ProfScope ps(c, "stream_join_reverse_lists", (double)n_all * K * 8.0 + (double)ncols * K * 24.0);
    const int64_t n_edges = n_all * K;
    if (ncols <= 0) return ANNCHOR_OK;
    ANN_TRY(sreserve(c, s->rev_cnt, sizeof(int32_t) * (size_t)(ncols + 1)));
    ANN_TRY(sreserve(c, s->rev_ptr, sizeof(int64_t) * (size_t)(ncols + 1)));
    // (room for every edge: the own columns receive n_edges x ncols / n_all of them on average, a rank that owns hubs more --
    // knowing how many would take a host wait per pass)
    ANN_TRY(sreserve(c, s->rev_edges, sizeof(unsigned long long) * (size_t)std::max<int64_t>(n_edges, 1)));
    ANN_CHECK_HIP(c, hipMemsetAsync(s->rev_cnt.p, 0, sizeof(int32_t) * (size_t)(ncols + 1), c->stream));
    k_st_rev_count<<<ann_blocks(n_edges, 256), 256, 0, c->stream>>>(lists_all, n_edges, col0, ncols, s->rev_cnt.as<int32_t>());
    ANN_TRY(ann_exclusive_scan_i32_to_i64(c, s->rev_cnt.as<int32_t>(), s->rev_ptr.as<int64_t>(), ncols));   // ptr has ncols + 1 entries
    ANN_CHECK_HIP(c, hipMemsetAsync(s->rev_cnt.p, 0, sizeof(int32_t) * (size_t)(ncols + 1), c->stream));
    k_st_rev_fill<<<ann_blocks(n_edges, 256), 256, 0, c->stream>>>(lists_all, n_edges, K, col0, ncols, s->rev_ptr.as<int64_t>(),
                                                                  s->rev_cnt.as<int32_t>(), s->rev_edges.as<unsigned long long>());
    k_st_rev_select<<<ann_blocks(ncols, 256), 256, 0, c->stream>>>(s->rev_ptr.as<int64_t>(), s->rev_edges.as<unsigned long long>(), ncols,
                                                                  dst);
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}

// lists_all: current lists of every ordered row of every rank, int32 [n_all][K] (single rank: the
// context's own list buffer).  The new lists replace the context's own (a.out_col / a.out_d2).
static int knn_join_pass(annchor_ctx *c, StreamState *s, KnnArgs &a, int dim_padded, const int32_t *lists_all, int per_pass,
                         int64_t *updates)
{
    const int K = a.K;
    const int64_t rows = (int64_t)a.tile_count * ST_T;
    const int64_t n_all = (int64_t)a.nt_all * ST_T;
    ANN_REQUIRE(c, a.eval_bits != nullptr, ANNCHOR_ESTATE, "join pass without a recorded tile phase");
    ANN_TRY(sreserve(c, s->ucand, sizeof(uint32_t) * (size_t)a.tile_count * JN_CAP));
    ANN_TRY(sreserve(c, s->ucount, sizeof(int32_t) * (size_t)a.tile_count));
    const bool to_b = a.out_col == s->out_col.as<int32_t>();
    DevBuf &nd = to_b ? s->out_d2b : s->out_d2, &nc = to_b ? s->out_colb : s->out_col;
    ANN_TRY(sreserve(c, nd, sizeof(float) * (size_t)rows * K));
    ANN_TRY(sreserve(c, nc, sizeof(int32_t) * (size_t)rows * K));
    a.lists_all = lists_all;
    a.ucand = s->ucand.as<uint32_t>(); a.ucount = s->ucount.as<int32_t>(); a.ucap = JN_CAP;
    a.out_d2_new = nd.as<float>(); a.out_col_new = nc.as<int32_t>();
    ANN_TRY(sreserve(c, s->rev_all, sizeof(int32_t) * (size_t)n_all * JN_RK));
    if (!s->rev_gathered)      // one rank (or a host that does not gather the slices): the reverse lists of every column, here
        ANN_TRY(knn_reverse_lists(c, s, lists_all, n_all, K, 0, n_all, s->rev_all.as<int32_t>()));
    s->rev_gathered = false;   // (they belong to THESE lists: the next pass builds its own)
    {
        ProfScope ps(c, "stream_join_candidates", (double)rows * (K + JN_RK) * 4.0 * (K + JN_RK + 1));
        const size_t lds = sizeof(uint32_t) * (JN_CAP + JN_B1 + JN_THREADS / 64 + 8 + (a.eval_words <= JN_EB_WORDS ? a.eval_words : 0));
        ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)k_st_join_cands, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int max_cols = std::max(1, std::min(per_pass * ST_T, JN_CAP));
        k_st_join_cands<<<a.tile_count, JN_THREADS, lds, c->stream>>>(lists_all, s->rev_all.as<int32_t>(), K, a.tile_begin, a.eval_bits,
                                                                      a.eval_words, a.eval_halves, max_cols, s->ucand.as<uint32_t>(),
                                                                      s->ucount.as<int32_t>(), jn_prof_buffer(c));
    }
    a.updates = s->evals.as<unsigned long long>() + 1;
    ANN_CHECK_HIP(c, hipMemsetAsync(a.updates, 0, 8, c->stream));
#ifdef ST_PROFILE
    static unsigned long long *dj_prof = nullptr;
    if (!dj_prof) (void)hipMalloc(&dj_prof, 128);
    (void)hipMemsetAsync(dj_prof, 0, 128, c->stream);
    a.prof = dj_prof;
#endif
    {
        ProfScope ps(c, "stream_join_gemm_topk", 0.0);
        ANN_TRY(launch_by_dim(c, a, dim_padded, true));
    }
    ANN_CHECK_HIP(c, hipGetLastError());
#ifdef ST_PROFILE
    {
        unsigned long long hp[16];
        (void)hipMemcpy(hp, dj_prof, 128, hipMemcpyDeviceToHost);
        double tot = 0;
        for (int i = 0; i < 8; ++i) tot += (double)hp[i];
        static const char *nm[8] = {"MFMA stream", "barrier after stream", "next-tile choice", "test + inserts", "LDS-DMA requests", "barrier after requests", "merge (+ publish / prologue / tail)", "the rest"};
        if (tot > 0) for (int i = 0; i < 8; ++i) fprintf(stderr, "[st-prof join] %-36s %6.2f %%\n", nm[i], 100.0 * (double)hp[i] / tot);
        if (tot > 0) for (int i = 8; i < 16; ++i) if (hp[i]) fprintf(stderr, "[st-prof join]    sub %d %6.2f %%\n", i, 100.0 * (double)hp[i] / tot);
        a.prof = nullptr;
    }
#endif
    a.out_d2 = a.out_d2_new; a.out_col = a.out_col_new;
    if (updates) {
        unsigned long long u = 0;
        ANN_TRY(ann_d2h(c, &u, a.updates, 8));
        *updates = (int64_t)u;
    }
    return ANNCHOR_OK;
}

int ann_stream_knn_finish(annchor_ctx *c, StreamState *s, KnnArgs &a, const void *perm_all, int dim_padded, int64_t **d_idx_out,
                          float **d_dist_out, int64_t *tile_evals)
{
    const int K = a.K;
    const int64_t rows = (int64_t)a.tile_count * ST_T;
    // exact distances of the selected neighbours, final order, ids
    ANN_TRY(sreserve(c, s->keys, sizeof(int64_t) * (size_t)rows * K));   // reuse: ids
    ANN_TRY(sreserve(c, s->keys2, sizeof(float) * (size_t)rows * K));    // reuse: distances
    int64_t *d_idx = s->keys.as<int64_t>();
    float *d_dist = s->keys2.as<float>();
    {
        ProfScope ps(c, "stream_finalize", (double)rows * K * (2.0 * dim_padded * 4 + 16));
        static const bool f64_pass = getenv("ANNCHOR_ST_FINALIZE_F64") != nullptr;
        if (s->last_kernel == 1 && !f64_pass)
            k_st_finalize_from_d2<<<ann_blocks(rows * K, 256), 256, 0, c->stream>>>((const int64_t *)perm_all, rows * K, a.out_d2, a.out_col, d_idx, d_dist);
        else
        k_st_finalize<<<ann_blocks(rows * K * 16, 256), 256, 0, c->stream>>>(a.Xs, a.Rs, (const int64_t *)perm_all, dim_padded,
                                                                             (int64_t)a.tile_begin * ST_T, rows, K, a.out_col, d_idx,
                                                                             d_dist);
        k_st_rowsort<<<ann_blocks(rows, 256), 256, 0, c->stream>>>(rows, K, d_idx, d_dist);
    }
    ANN_CHECK_HIP(c, hipGetLastError());
    *d_idx_out = d_idx;
    *d_dist_out = d_dist;
    if (tile_evals) {
        unsigned long long ev[3] = {0, 0, 0};
        ANN_TRY(ann_d2h(c, ev, s->evals.p, 24));
        if (a.K > ST_KMAX_BIG) { ev[0] /= 2; ev[2] /= 2; }   // (two workgroups per row tile, 64 rows' lists each: both count what they stream)
        *tile_evals = (int64_t)(ev[0] + ev[2]);
        s->last_tile_evals = (int64_t)ev[0];
        s->last_join_chunks = (int64_t)ev[2];
    }
    return ANNCHOR_OK;
}

// tile evaluations of the last build's tile phase and 128-column runs of its join passes
extern "C" int annchor_stream_last_kernel(annchor_ctx *c, int32_t *kind, int64_t *guard_rows)
{
    if (!c || !kind || !guard_rows) return ANNCHOR_EINVAL;
    StreamState *s = state_of(c, false);
    ANN_REQUIRE(c, s != nullptr, ANNCHOR_ESTATE, "no streamed build on this context");
    *kind = s->last_kernel;
    *guard_rows = s->last_guard_rows;
    return ANNCHOR_OK;
}

// *two_stage: the last tile phase ran the two-stage kernel k_st_knnh (fp16 hi-only filter + float32 differences) behind
// k_st_knnbf's warm-up; *repaired: rows flagged by the split kernels' guard were evaluated again exactly (k_st_repair)
extern "C" int annchor_stream_last_tile_kernels(annchor_ctx *c, int32_t *two_stage, int32_t *repaired)
{
    if (!c || !two_stage || !repaired) return ANNCHOR_EINVAL;
    StreamState *s = state_of(c, false);
    ANN_REQUIRE(c, s, ANNCHOR_ESTATE, "no streamed build on this context");
    *two_stage = s->last_two_stage ? 1 : 0;
    *repaired = s->last_repaired ? 1 : 0;
    return ANNCHOR_OK;
}

extern "C" int annchor_stream_last_counts(annchor_ctx *c, int64_t *tile_phase_evals, int64_t *join_chunks)
{
    if (!c || !tile_phase_evals || !join_chunks) return ANNCHOR_EINVAL;
    StreamState *s = state_of(c, false);
    ANN_REQUIRE(c, s != nullptr, ANNCHOR_ESTATE, "no streamed build on this context");
    *tile_phase_evals = s->last_tile_evals;
    *join_chunks = s->last_join_chunks;
    return ANNCHOR_OK;
}

static int knn_download_graph(annchor_ctx *c, StreamState *s, const KnnArgs &a, const void *perm_all, int64_t *d_idx, float *d_dist,
                              int64_t *row_ids, int64_t *ng_idx, double *ng_dist)
{
    const int K = a.K, k = K + 1;
    const int64_t rows = (int64_t)a.tile_count * ST_T;
    if (!row_ids) {
        const int64_t n_local = s->n_local;
        ANN_TRY(sreserve(c, s->emit_idx, sizeof(int64_t) * (size_t)n_local * k));
        ANN_TRY(sreserve(c, s->emit_dist, sizeof(double) * (size_t)n_local * k));
        k_st_emit<<<ann_blocks(rows * k, 256), 256, 0, c->stream>>>((const int64_t *)perm_all, (int64_t)a.tile_begin * ST_T, rows, K, s->base,
                                                                   n_local, d_idx, d_dist, s->emit_idx.as<int64_t>(),
                                                                   s->emit_dist.as<double>());
        ANN_CHECK_HIP(c, hipGetLastError());
        ANN_TRY(ann_d2h(c, ng_idx, s->emit_idx.p, sizeof(int64_t) * (size_t)n_local * k));
        ANN_TRY(ann_d2h(c, ng_dist, s->emit_dist.p, sizeof(double) * (size_t)n_local * k));
    } else {
        std::vector<int64_t> hidx((size_t)rows * K);
        std::vector<float> hd((size_t)rows * K);
        ANN_TRY(ann_d2h(c, hidx.data(), d_idx, sizeof(int64_t) * hidx.size()));
        ANN_TRY(ann_d2h(c, hd.data(), d_dist, sizeof(float) * hd.size()));
        ANN_TRY(ann_d2h(c, row_ids, (const int64_t *)perm_all + (size_t)a.tile_begin * ST_T, sizeof(int64_t) * (size_t)rows));
        for (int64_t r = 0; r < rows; ++r) {
            ng_idx[r * k] = row_ids[r];
            ng_dist[r * k] = 0.0;
            for (int e = 0; e < K; ++e) {
                ng_idx[r * k + 1 + e] = hidx[(size_t)r * K + e];
                ng_dist[r * k + 1 + e] = (double)hd[(size_t)r * K + e];
            }
        }
    }
    return ANNCHOR_OK;
}

static int knn_args_graph(annchor_ctx *c, KnnArgs &a, const void *Xs_all, const void *rs_all, const void *lo_all, const void *hi_all,
                          const void *mid_all, int64_t n_all, int32_t nt_all, int32_t n_anchors, int32_t tile_begin,
                          int32_t tile_count, int32_t k)
{
    ANN_REQUIRE(c, k >= 2 && k - 1 < ST_KMAX_HUGE, ANNCHOR_ELIMIT, "streamed form supports 2 <= n_neighbors <= %d (beyond 256 dimensions: <= %d)", ST_KMAX_HUGE,
                ST_KMAX_BIG - 1);
    ANN_REQUIRE(c, n_all == (int64_t)nt_all * ST_T && tile_begin >= 0 && tile_begin + tile_count <= nt_all, ANNCHOR_EINVAL,
                "tile range out of bounds");
    ANN_REQUIRE(c, n_all < (1ll << 31), ANNCHOR_ELIMIT, "n_all exceeds 2^31");
    a.Xs = (const float *)Xs_all; (void)ann_stream_split_of(Xs_all, &a.Xb, &a.rsb, &a.cvec); a.rs = (const float *)rs_all; a.lo = (const float *)lo_all; a.hi = (const float *)hi_all; a.mid = (const float *)mid_all;
    a.Rs = a.Xs; a.rr = a.rs; a.rlo = a.lo; a.rhi = a.hi; a.rmid = a.mid; a.nt_r = nt_all; a.query = 0;
    a.nt_all = nt_all; a.na = n_anchors; a.tile_begin = tile_begin; a.tile_count = tile_count; a.K = k - 1;
    return ANNCHOR_OK;
}

// k nearest neighbours (self included as column 0, like get_ann of annchor.py:514-530) of
// the row tiles [tile_begin, tile_begin + tile_count) against ALL column tiles.  The array
// arguments are DEVICE pointers laid out as annchor_stream_order produces them.  join_passes > 0
// (only when the launch covers every row tile, i.e. one rank: the passes read every row's list)
// runs that many neighbour-of-neighbour passes after the tile phase.  Outputs are HOST arrays:
// ng_idx int64 [rows, k] (global ids), ng_dist float64 [rows, k], in tile order;
// row_ids int64 [rows] gives the global id of each output row (-1 = padding row).
// With row_ids == NULL the outputs are instead [n_local, k] arrays in the bound shard's own row
// order (row = global id - global_base of annchor_stream_bind), written by a device kernel and
// copied out in one piece -- no host-side reordering.  tile_evals counts 128 x 128 pair blocks
// (tile phase + join passes).
// annchor_stream_knn = annchor_stream_knn_run (everything on the device; the graph stays there) + annchor_stream_knn_fetch
// (the download): a host that prepares its result arrays on another thread while the GPU works (1.9 GB to allocate and
// touch at N = 8 x 10^6) calls the halves itself and needs the arrays only for the second.
extern "C" int annchor_stream_knn_run(annchor_ctx *c, const void *Xs_all, const void *rs_all, const void *perm_all,
                                      const void *lo_all, const void *hi_all, const void *mid_all, int64_t n_all, int32_t nt_all, int32_t n_anchors,
                                      int32_t dim_padded, int32_t tile_begin, int32_t tile_count, int32_t k, double p_work,
                                      int32_t join_passes, int32_t join_extra, int64_t *tile_evals)
{
    if (!c || !Xs_all || !rs_all || !perm_all || !lo_all || !hi_all || !mid_all) return ANNCHOR_EINVAL;
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    StreamState *s = state_of(c, true);
    ann_stream_free_run(s);
    KnnArgs a;
    ANN_TRY(knn_args_graph(c, a, Xs_all, rs_all, lo_all, hi_all, mid_all, n_all, nt_all, n_anchors, tile_begin, tile_count, k));
    ANN_REQUIRE(c, join_passes >= 0 && join_extra >= 0 && (join_passes + join_extra == 0 || tile_count == nt_all), ANNCHOR_EINVAL,
                "join passes need every row's list: use annchor_stream_knn_begin / _join / _end across ranks");
    int T = 0, tp = 0, pp = 0;
    ANN_TRY(annchor_stream_budget(nt_all, p_work, join_passes, &T, &tp, &pp));
    ANN_TRY(knn_tile_phase(c, s, a, dim_padded, tp, join_passes + join_extra > 0 && tp < nt_all));
    // join_passes passes always; up to join_extra more while a pass still replaces more than
    // ANNCHOR_JOIN_YIELD (1 %) of all list entries and the budget has room for another pass
    const double floor_updates = ANNCHOR_JOIN_YIELD * (double)tile_count * ST_T * (k - 1);
    for (int p = 0; tp < nt_all && p < join_passes + join_extra; ++p) {
        if (p >= join_passes && tp + (p + 1) * std::max(pp, 1) > T) break;
        int64_t upd = 0;
        ANN_TRY(knn_join_pass(c, s, a, dim_padded, a.out_col, std::max(pp, 1), &upd));
        if (p + 1 >= join_passes && (double)upd <= floor_updates) break;
    }
    ANN_TRY(ann_stream_knn_finish(c, s, a, perm_all, dim_padded, &s->fin_idx, &s->fin_dist, tile_evals));
    s->run = new KnnArgs(a);
    s->run_perm = perm_all;
    s->run_dimp = dim_padded;
    s->run_finished = true;
    return ANNCHOR_OK;
}

extern "C" int annchor_stream_knn_fetch(annchor_ctx *c, int64_t *row_ids, int64_t *ng_idx, double *ng_dist)
{
    if (!c || !ng_idx || !ng_dist) return ANNCHOR_EINVAL;
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    StreamState *s = state_of(c, false);
    ANN_REQUIRE(c, s && s->run && s->run_finished, ANNCHOR_ESTATE, "annchor_stream_knn_run not called");
    const int rc = knn_download_graph(c, s, *s->run, s->run_perm, s->fin_idx, s->fin_dist, row_ids, ng_idx, ng_dist);
    ann_stream_free_run(s);
    return rc;
}

extern "C" int annchor_stream_knn(annchor_ctx *c, const void *Xs_all, const void *rs_all, const void *perm_all,
                                  const void *lo_all, const void *hi_all, const void *mid_all, int64_t n_all, int32_t nt_all, int32_t n_anchors,
                                  int32_t dim_padded, int32_t tile_begin, int32_t tile_count, int32_t k, double p_work,
                                  int32_t join_passes, int32_t join_extra, int64_t *row_ids, int64_t *ng_idx, double *ng_dist,
                                  int64_t *tile_evals)
{
    if (!c || !ng_idx || !ng_dist) return ANNCHOR_EINVAL;
    ANN_TRY(annchor_stream_knn_run(c, Xs_all, rs_all, perm_all, lo_all, hi_all, mid_all, n_all, nt_all, n_anchors, dim_padded, tile_begin,
                                   tile_count, k, p_work, join_passes, join_extra, tile_evals));
    return annchor_stream_knn_fetch(c, row_ids, ng_idx, ng_dist);
}

// The same build in steps, for row-sharded runs: after _begin (tile phase) and after every _join
// the caller all-gathers the ranks' list buffers (*lists_local: int32 [tile_count * 128][k - 1],
// device memory, ordered column indices; every rank has the same tile_count) into
// lists_all [n_all][k - 1] and hands that to the next _join; _end finishes and downloads.
extern "C" int annchor_stream_knn_begin(annchor_ctx *c, const void *Xs_all, const void *rs_all, const void *perm_all,
                                        const void *lo_all, const void *hi_all, const void *mid_all, int64_t n_all, int32_t nt_all,
                                        int32_t n_anchors, int32_t dim_padded, int32_t tile_begin, int32_t tile_count, int32_t k,
                                        int32_t tile_budget, void **lists_local, int64_t *lists_bytes)
{
    if (!c || !Xs_all || !rs_all || !perm_all || !lo_all || !hi_all || !mid_all || !lists_local || !lists_bytes) return ANNCHOR_EINVAL;
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    StreamState *s = state_of(c, true);
    ann_stream_free_run(s);
    KnnArgs *a = new KnnArgs();
    s->run = a;
    int rc = knn_args_graph(c, *a, Xs_all, rs_all, lo_all, hi_all, mid_all, n_all, nt_all, n_anchors, tile_begin, tile_count, k);
    if (rc == ANNCHOR_OK) rc = knn_tile_phase(c, s, *a, dim_padded, tile_budget, true);
    if (rc != ANNCHOR_OK) { ann_stream_free_run(s); return rc; }
    s->run_perm = perm_all;
    s->run_dimp = dim_padded;
    ANN_CHECK_HIP(c, ann_stream_wait(c, __func__));
    *lists_local = a->out_col;
    *lists_bytes = (int64_t)sizeof(int32_t) * tile_count * ST_T * (k - 1);
    return ANNCHOR_OK;
}

extern "C" int annchor_stream_knn_join(annchor_ctx *c, const void *lists_all, int32_t per_pass, void **lists_local, int64_t *updates)
{
    if (!c || !lists_all || !lists_local || !updates) return ANNCHOR_EINVAL;
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    StreamState *s = state_of(c, false);
    ANN_REQUIRE(c, s && s->run, ANNCHOR_ESTATE, "annchor_stream_knn_begin not called");
    ANN_REQUIRE(c, per_pass >= 1, ANNCHOR_EINVAL, "per_pass must be >= 1");
    ANN_TRY(knn_join_pass(c, s, *s->run, s->run_dimp, (const int32_t *)lists_all, per_pass, updates));
    ANN_CHECK_HIP(c, ann_stream_wait(c, __func__));
    *lists_local = s->run->out_col;
    return ANNCHOR_OK;
}

// Row-sharded join pass, first half: the reverse neighbour lists of THIS rank's columns (its tile range of the global
// order) from the all-gathered lists.  *rev_local: int32 [tile_count x 128][JN_RK] (device; the rank's slice), *rev_all: int32 [n_all][JN_RK], *rev_bytes: bytes per rank.  The host all-gathers the slices into
// *rev_all (rank r's at r x *rev_bytes) and calls annchor_stream_knn_join with the same lists_all; without that call
// the join pass builds every column's reverse list itself.
extern "C" int annchor_stream_join_rev_begin(annchor_ctx *c, const void *lists_all, void **rev_local, void **rev_all, int64_t *rev_bytes)
{
    if (!c || !lists_all || !rev_local || !rev_all || !rev_bytes) return ANNCHOR_EINVAL;
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    StreamState *s = state_of(c, false);
    ANN_REQUIRE(c, s && s->run, ANNCHOR_ESTATE, "annchor_stream_knn_begin not called");
    const KnnArgs &a = *s->run;
    const int64_t n_all = (int64_t)a.nt_all * ST_T, col0 = (int64_t)a.tile_begin * ST_T, ncols = (int64_t)a.tile_count * ST_T;
    // (the slice has its own buffer: an all-gather whose input lies inside its output is legal for RCCL, but nothing is gained
    // by depending on it)
    ANN_TRY(sreserve(c, s->rev_all, sizeof(int32_t) * (size_t)n_all * JN_RK));
    ANN_TRY(sreserve(c, s->rev_slice, sizeof(int32_t) * (size_t)std::max<int64_t>(ncols, 1) * JN_RK));
    ANN_TRY(knn_reverse_lists(c, s, (const int32_t *)lists_all, n_all, a.K, col0, ncols, s->rev_slice.as<int32_t>()));
    ANN_CHECK_HIP(c, ann_stream_wait(c, __func__));
    s->rev_gathered = true;
    *rev_local = s->rev_slice.p;
    *rev_all = s->rev_all.p;
    *rev_bytes = (int64_t)sizeof(int32_t) * ncols * JN_RK;
    return ANNCHOR_OK;
}

extern "C" int annchor_stream_knn_end(annchor_ctx *c, int64_t *row_ids, int64_t *ng_idx, double *ng_dist, int64_t *tile_evals)
{
    if (!c || !ng_idx || !ng_dist) return ANNCHOR_EINVAL;
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    StreamState *s = state_of(c, false);
    ANN_REQUIRE(c, s && s->run, ANNCHOR_ESTATE, "annchor_stream_knn_begin not called");
    int64_t *d_idx = nullptr;
    float *d_dist = nullptr;
    int rc = ann_stream_knn_finish(c, s, *s->run, s->run_perm, s->run_dimp, &d_idx, &d_dist, tile_evals);
    if (rc == ANNCHOR_OK) rc = knn_download_graph(c, s, *s->run, s->run_perm, d_idx, d_dist, row_ids, ng_idx, ng_dist);
    ann_stream_free_run(s);
    return rc;
}

// Queries against a fitted (ordered) data set: the context holds the QUERY rows -- bound with
// annchor_stream_bind (global_base 0), given the data set's anchor vectors through
// annchor_stream_anchor_round, ordered with annchor_stream_order -- and the six column arrays
// are the data set's (device pointers from ITS annchor_stream_order / all-gather).  For every
// query the nn nearest data rows: out_idx int64 [nq, nn] (global ids), out_dist float64 [nq, nn],
// in the queries' own order.  Replaces Annchor.query (annchor.py:643-683 ->
// query_functions.py:183-212) for data sets too large for the pair-list form.
extern "C" int annchor_stream_query(annchor_ctx *c, const void *Xs_all, const void *rs_all, const void *perm_all, const void *lo_all,
                                    const void *hi_all, const void *mid_all, int64_t n_all, int32_t nt_all, int32_t n_anchors,
                                    int32_t dim_padded, int32_t nn, double p_work, int64_t *out_idx, double *out_dist,
                                    int64_t *tile_evals)
{
    if (!c || !Xs_all || !rs_all || !perm_all || !lo_all || !hi_all || !mid_all || !out_idx || !out_dist) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, nn >= 1 && nn < ST_KMAX_HUGE, ANNCHOR_ELIMIT, "streamed query supports 1 <= nn <= %d (beyond 256 dimensions: <= %d)", ST_KMAX_HUGE - 1, ST_KMAX_BIG - 2);
    ANN_REQUIRE(c, n_all == (int64_t)nt_all * ST_T && n_all < (1ll << 31), ANNCHOR_EINVAL, "column arrays out of range");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    StreamState *s = state_of(c, false);
    ANN_REQUIRE(c, s && s->nt > 0 && s->Xs.p && s->na == n_anchors && s->dimp == dim_padded, ANNCHOR_ESTATE,
                "queries are not ordered (bind, anchor rounds with the data set's anchors, order) or do not match the data set");
    KnnArgs a;
    a.Xs = (const float *)Xs_all; (void)ann_stream_split_of(Xs_all, &a.Xb, &a.rsb, &a.cvec); a.rs = (const float *)rs_all; a.lo = (const float *)lo_all; a.hi = (const float *)hi_all; a.mid = (const float *)mid_all;
    a.Rs = s->Xs.as<float>(); a.rr = s->rs.as<float>(); a.rlo = s->lo.as<float>(); a.rhi = s->hi.as<float>(); a.rmid = s->mid.as<float>();
    a.nt_r = s->nt; a.query = 1;
    a.nt_all = nt_all; a.na = n_anchors; a.tile_begin = 0; a.tile_count = s->nt; a.K = nn;
    int64_t *d_idx = nullptr;
    float *d_dist = nullptr;
    int T = 0, tp = 0, pp = 0;
    ANN_TRY(annchor_stream_budget(nt_all, p_work, 0, &T, &tp, &pp));
    ANN_TRY(knn_tile_phase(c, s, a, dim_padded, tp, false));
    ANN_TRY(ann_stream_knn_finish(c, s, a, perm_all, dim_padded, &d_idx, &d_dist, tile_evals));
    const int64_t rows = (int64_t)s->nt * ST_T, nq = s->n_local;
    ANN_TRY(sreserve(c, s->emit_idx, sizeof(int64_t) * (size_t)nq * nn));
    ANN_TRY(sreserve(c, s->emit_dist, sizeof(double) * (size_t)nq * nn));
    k_st_emit_query<<<ann_blocks(rows * nn, 256), 256, 0, c->stream>>>(s->perm.as<int64_t>(), rows, nn, nq, d_idx, d_dist,
                                                                      s->emit_idx.as<int64_t>(), s->emit_dist.as<double>());
    ANN_CHECK_HIP(c, hipGetLastError());
    ANN_TRY(ann_d2h(c, out_idx, s->emit_idx.p, sizeof(int64_t) * (size_t)nq * nn));
    return ann_d2h(c, out_dist, s->emit_dist.p, sizeof(double) * (size_t)nq * nn);
}

// interval tables after an all-gather: [world][na][nt] (rank-major, as the collective delivers
// them) -> [na][world * nt] (the tile axis joined across ranks, as the kernels index them)
__global__ void k_st_join_tables(const float *__restrict__ src, int world, int na, int nt, float *__restrict__ dst)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per = (int64_t)na * nt;
    if (t >= per * world) return;
    const int r = (int)(t / per);
    const int64_t rem = t - (int64_t)r * per;
    const int a = (int)(rem / nt), j = (int)(rem - (int64_t)a * nt);
    dst[(size_t)a * world * nt + (size_t)r * nt + j] = src[t];
}

extern "C" int annchor_stream_join_tables(annchor_ctx *c, const void *gathered, int32_t world, int32_t n_anchors, int32_t n_tiles,
                                          void *joined)
{
    if (!c || !gathered || !joined || world < 1 || n_anchors < 1 || n_tiles < 1) return ANNCHOR_EINVAL;
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    const int64_t tot = (int64_t)world * n_anchors * n_tiles;
    k_st_join_tables<<<ann_blocks(tot, 256), 256, 0, c->stream>>>((const float *)gathered, world, n_anchors, n_tiles, (float *)joined);
    ANN_CHECK_HIP(c, hipGetLastError());
    ANN_CHECK_HIP(c, ann_stream_wait(c, __func__));
    return ANNCHOR_OK;
}

// ------------------------------------------------ raw device memory for host-staged gathers
extern "C" int annchor_device_alloc(annchor_ctx *c, int64_t bytes, void **dptr)
{
    if (!c || !dptr || bytes <= 0) return ANNCHOR_EINVAL;
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    ANN_CHECK_HIP(c, hipMalloc(dptr, (size_t)bytes));
    return ANNCHOR_OK;
}

extern "C" int annchor_device_free(annchor_ctx *c, void *dptr)
{
    if (!c) return ANNCHOR_EINVAL;
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    if (dptr) ANN_CHECK_HIP(c, hipFree(dptr));
    return ANNCHOR_OK;
}

extern "C" int annchor_device_copy(annchor_ctx *c, void *dst, const void *src, int64_t bytes, int32_t kind)
{
    if (!c || !dst || !src || bytes < 0) return ANNCHOR_EINVAL;
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    hipMemcpyKind kk = kind == 1 ? hipMemcpyHostToDevice : kind == 2 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    ANN_CHECK_HIP(c, hipMemcpyAsync(dst, src, (size_t)bytes, kk, c->stream));
    ANN_CHECK_HIP(c, ann_stream_wait(c, __func__));
    return ANNCHOR_OK;
}
