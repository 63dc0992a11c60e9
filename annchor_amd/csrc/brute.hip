// brute.hip -- BruteForce.fit (reference annchor/annchor.py:1004-1023): the metric on all
// nx(nx-1)/2 pairs, then every row sorted; returns the first k columns of
// (argsort(D, axis=1), sort(D, axis=1)) with the stable tie rule (distance, then index).
// Also the ground-truth generator behind every recall number.
#include "common.h"
#include "rowsel.h"

__device__ __forceinline__ int64_t bf_pos(int64_t i, int64_t j, int64_t nx)  // i < j
{
    return i * (2 * nx - i - 1) / 2 + (j - i - 1);
}

__global__ void k_bf_pairs(int64_t nx, int2 *__restrict__ ij)
{
    // block per row i, threads over j > i
    const int64_t i = blockIdx.x;
    const int64_t base = bf_pos(i, i + 1, nx);
    for (int64_t j = i + 1 + threadIdx.x; j < nx; j += blockDim.x) ij[base + (j - i - 1)] = make_int2((int)i, (int)j);
}

__global__ __launch_bounds__(ROW_THREADS) void k_bf_rows(const double *__restrict__ vals, int64_t nx, int k,
                                                        int64_t *__restrict__ oidx, double *__restrict__ odist, int cap)
{
    __shared__ RowSelShared sh;
    __shared__ uint32_t cnt_lt;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    uint64_t *keys = reinterpret_cast<uint64_t *>(dyn);      // [cap]
    uint64_t *lkey = keys + cap;                             // [k]
    int32_t *lslot = reinterpret_cast<int32_t *>(lkey + k);  // [k]
    const int64_t i = row_of_block(gridDim.x);
    const int len = (int)nx;
    const bool in_lds = len <= cap;
    auto key_of = [&](int s) -> uint64_t {
        if (s == i) return ann_key_asc(0.0);  // the diagonal of D
        const int64_t a = s < i ? s : i, b = s < i ? i : s;
        return ann_key_asc(vals[bf_pos(a, b, nx)]);
    };
    if (threadIdx.x == 0) cnt_lt = 0;
    if (in_lds)
        for (int s = threadIdx.x; s < len; s += ROW_THREADS) keys[s] = key_of(s);
    __syncthreads();
    auto kf = [&](int s) -> uint64_t { return in_lds ? keys[s] : key_of(s); };
    const int want = min(k, len);
    const uint64_t t = row_kth_key(sh, len, (uint32_t)(want - 1), kf);
    for (int s = threadIdx.x; s < len; s += ROW_THREADS) {
        const uint64_t kk = kf(s);
        if (kk < t) { const uint32_t o = atomicAdd(&cnt_lt, 1u); lkey[o] = kk; lslot[o] = s; }
    }
    __syncthreads();
    uint32_t run = cnt_lt;
    for (int base = 0; base < len && run < (uint32_t)want; base += ROW_THREADS) {
        const int s = base + threadIdx.x;
        const uint32_t f = (s < len && kf(s) == t) ? 1u : 0u;
        uint32_t tot;
        const uint32_t ex = row_block_scan(f, sh.wsum, &tot);
        if (f && run + ex < (uint32_t)want) { lkey[run + ex] = t; lslot[run + ex] = s; }
        run += tot;
        __syncthreads();
    }
    __syncthreads();
    for (int e = threadIdx.x; e < want; e += ROW_THREADS) {
        const uint64_t ke = lkey[e];
        const int32_t se = lslot[e];
        int r = 0;
        for (int o = 0; o < want; ++o) r += (lkey[o] < ke) || (lkey[o] == ke && lslot[o] < se);
        oidx[i * k + r] = se;
        odist[i * k + r] = ann_key_asc_inv(ke);
    }
}

extern "C" int annchor_brute_force(annchor_ctx *c, int32_t k, int64_t *ng_idx, double *ng_dist)
{
    if (!c || !ng_idx || !ng_dist) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->metric != ANNCHOR_METRIC_NONE, ANNCHOR_EINVAL, "no device metric bound to this context");
    const int64_t nx = c->nx;
    ANN_REQUIRE(c, k >= 1 && k <= nx, ANNCHOR_EINVAL, "k=%d out of range 1..%lld", k, (long long)nx);
    ANN_REQUIRE(c, (size_t)k * 12 <= 100 * 1024, ANNCHOR_ELIMIT, "k=%d: at most 8533 columns per row", k);
    const int64_t n = nx * (nx - 1) / 2;
    ANN_REQUIRE(c, n < (1ll << 30), ANNCHOR_ELIMIT, "brute force over %lld points exceeds the 2^30 pair limit", (long long)nx);
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    ANN_TRY(ann_reserve(c, c->ij, sizeof(int2) * (size_t)n));
    ANN_TRY(ann_reserve(c, c->RA, sizeof(double) * (size_t)n));
    c->n = 0; c->have_bitmap = false; c->have_features = c->have_RA = false; c->sel_prepared = false;  // the pair-list state of a previous fit is gone
    const size_t cells = (size_t)nx * k;
    ANN_TRY(ann_reserve(c, c->stage_out, cells * 16));
    int64_t *d_i = c->stage_out.as<int64_t>();
    double *d_d = reinterpret_cast<double *>(d_i + cells);
    k_bf_pairs<<<(int)nx, 256, 0, c->stream>>>(nx, c->ij.as<int2>());
    PairSource src;
    src.ij = c->ij.as<int2>();
    src.n = n;
    ANN_CHECK_HIP(c, hipEventRecord(c->call_a, c->stream));
    ANN_TRY(ann_metric_launch(c, src, c->RA.as<double>(), nullptr, nullptr));
    ANN_CHECK_HIP(c, hipEventRecord(c->call_b, c->stream));
    c->call_timed = true;
    const size_t tail = (((size_t)k * 12) + 15) & ~(size_t)15;
    ANN_REQUIRE(c, tail < ROW_LDS_LIMIT / 2, ANNCHOR_ELIMIT, "n_neighbors %d too large for the row kernel", k);
    const int cap = row_lds_cap(nx, tail);
    const size_t dyn = (size_t)cap * 8 + tail;
    ANN_TRY(row_lds_prepare(c, k_bf_rows, dyn));
    {
        ProfScope ps(c, "brute_force_row_sort", (double)n * 2 * 8.0 + (double)cells * 16.0);
        k_bf_rows<<<(int)nx, ROW_THREADS, dyn, c->stream>>>(c->RA.as<double>(), nx, k, d_i, d_d, cap);
    }
    ANN_CHECK_HIP(c, hipGetLastError());
    ANN_TRY(ann_d2h(c, ng_idx, d_i, cells * 8));
    return ann_d2h(c, ng_dist, d_d, cells * 8);
}
