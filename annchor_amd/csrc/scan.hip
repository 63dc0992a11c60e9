// scan.hip -- small device primitives shared by the pipeline stages:
//   * exclusive prefix sum int32 -> int64,
//   * exact k-th order statistic of float64 keys under a byte mask (MSB-first
//     8-bit radix select; up to 4 ranks resolved in the same passes).
// The second one implements np.partition(x[mask], k)[k] as used by
// SimpleStratifiedSampler.get_partition (reference annchor/samplers.py:119-140).
#include "common.h"

#define SCAN_THREADS 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_THREADS * SCAN_ITEMS)

__device__ __forceinline__ int64_t block_exclusive_scan_i64(int64_t v, int64_t *total)
{
    // returns exclusive prefix of v across the block; *total = block sum (all threads)
    __shared__ int64_t wsum[SCAN_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int64_t o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int64_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; ++w) {
        if (w < wave) base += wsum[w];
        tot += wsum[w];
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_block_sums(const int32_t *__restrict__ in, int64_t n,
                                                                 int64_t *__restrict__ bsum)
{
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
    int64_t s = 0;
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        int64_t t = base + (int64_t)k * SCAN_THREADS + threadIdx.x;
        if (t < n) s += in[t];
    }
    int64_t tot;
    block_exclusive_scan_i64(s, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_of_sums(int64_t *__restrict__ bsum, int nb, int64_t *__restrict__ grand)
{
    int64_t carry = 0;
    for (int base = 0; base < nb; base += SCAN_THREADS) {
        int t = base + threadIdx.x;
        int64_t v = t < nb ? bsum[t] : 0;
        int64_t tot;
        int64_t ex = block_exclusive_scan_i64(v, &tot);
        if (t < nb) bsum[t] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *grand = carry;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const int32_t *__restrict__ in, int64_t n,
                                                            const int64_t *__restrict__ bsum, int64_t *__restrict__ out)
{
    // items are assigned thread-contiguously so that the scan order equals the index order
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int32_t v[SCAN_ITEMS];
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        s += v[k];
    }
    int64_t tot;
    int64_t ex = block_exclusive_scan_i64(s, &tot) + bsum[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < n) out[base + k] = ex;
        ex += v[k];
    }
}

int ann_exclusive_scan_i32_to_i64(annchor_ctx *c, const int32_t *in, int64_t *out, int64_t n)
{
    int nb = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
    if (nb < 1) nb = 1;
    ANN_TRY(ann_reserve(c, c->scan_tmp, sizeof(int64_t) * (size_t)(nb + 1)));
    int64_t *bsum = c->scan_tmp.as<int64_t>();
    ProfScope ps(c, "exclusive_scan", (double)n * 12);
    k_scan_block_sums<<<nb, SCAN_THREADS, 0, c->stream>>>(in, n, bsum);
    k_scan_of_sums<<<1, SCAN_THREADS, 0, c->stream>>>(bsum, nb, out + n);
    k_scan_apply<<<nb, SCAN_THREADS, 0, c->stream>>>(in, n, bsum, out);
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}

// ------------------------------------------------------------- radix select
#define SEL_MAXQ 4
struct SelState {
    uint64_t prefix[SEL_MAXQ];  // resolved high bits
    int64_t k[SEL_MAXQ];        // remaining rank inside the current bucket
    int nq;
    int pass;
    unsigned long long vor, vand;  // OR / AND of all (flagged) keys: equal bytes are uniform and skipped
};

__global__ __launch_bounds__(256) void k_sel_orand(const double *__restrict__ vals, const uint8_t *__restrict__ flag, int64_t n,
                                                  SelState *__restrict__ st)
{
    unsigned long long o = 0, a = ~0ull;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        if (flag && !flag[t]) continue;
        const unsigned long long key = ann_key_asc(vals[t]);
        o |= key; a &= key;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { o |= __shfl_xor(o, off); a &= __shfl_xor(a, off); }
    __shared__ unsigned long long so[4], sa[4];
    if ((threadIdx.x & 63) == 0) { so[threadIdx.x >> 6] = o; sa[threadIdx.x >> 6] = a; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicOr(&st->vor, so[0] | so[1] | so[2] | so[3]);
        atomicAnd(&st->vand, sa[0] & sa[1] & sa[2] & sa[3]);
    }
}

__global__ __launch_bounds__(256) void k_sel_hist(const double *__restrict__ vals, const uint8_t *__restrict__ flag,
                                                 int64_t n, const SelState *__restrict__ st, uint32_t *__restrict__ hist)
{
    __shared__ uint32_t lh[SEL_MAXQ * 256];
    for (int t = threadIdx.x; t < SEL_MAXQ * 256; t += blockDim.x) lh[t] = 0;
    __syncthreads();
    const int nq = st->nq, pass = st->pass;
    const int shift = 56 - 8 * pass;
    if ((((st->vor ^ st->vand) >> shift) & 0xffull) == 0) return;  // uniform byte: k_sel_step fills it in
    uint64_t pre[SEL_MAXQ];
    for (int q = 0; q < SEL_MAXQ; ++q) pre[q] = st->prefix[q];
    const uint64_t himask = pass == 0 ? 0ull : (~0ull << (shift + 8));
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        if (flag && !flag[t]) continue;
        uint64_t key = ann_key_asc(vals[t]);
        uint32_t d = (uint32_t)(key >> shift) & 0xffu;
        for (int q = 0; q < nq; ++q)
            if ((key & himask) == pre[q]) atomicAdd(&lh[q * 256 + d], 1u);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < nq * 256; t += blockDim.x)
        if (lh[t]) atomicAdd(&hist[t], lh[t]);
}

__global__ __launch_bounds__(256) void k_sel_step(SelState *st, uint32_t *hist)
{
    // thread d owns digit d; one block-wide scan per query finds the bucket holding rank k
    __shared__ uint32_t wsum[4];
    __shared__ int64_t newk[SEL_MAXQ];
    __shared__ int newd[SEL_MAXQ];
    const int d = threadIdx.x, lane = d & 63, wave = d >> 6;
    const int shift = 56 - 8 * st->pass;
    const int nq = st->nq;
    if ((((st->vor ^ st->vand) >> shift) & 0xffull) == 0) {
        __syncthreads();
        if (d < nq) st->prefix[d] |= st->vor & (0xffull << shift);
        if (d == 0) st->pass += 1;
        return;
    }
    for (int q = 0; q < nq; ++q) {
        const uint32_t h = hist[q * 256 + d];
        uint32_t inc = h;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
        }
        __syncthreads();
        if (lane == 63) wsum[wave] = inc;
        if (d == 0) { newd[q] = 255; newk[q] = 0; }  // rank beyond the population clamps to the maximum
        __syncthreads();
        uint32_t base = 0;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        const int64_t ex = (int64_t)base + inc - h, k = st->k[q];
        if (h != 0 && k >= ex && k < ex + h) { newd[q] = d; newk[q] = k - ex; }
        hist[q * 256 + d] = 0;
    }
    __syncthreads();
    if (d < nq) {
        st->prefix[d] |= (uint64_t)newd[d] << shift;
        st->k[d] = newk[d];
    }
    if (d == 0) st->pass += 1;
}

int ann_kth_smallest(annchor_ctx *c, const double *vals, const uint8_t *flag, int64_t n, const int64_t *ks, int nk,
                     double *h_out)
{
    ANN_REQUIRE(c, nk >= 1 && nk <= SEL_MAXQ, ANNCHOR_EINVAL, "kth_smallest: 1..%d ranks per call", SEL_MAXQ);
    ANN_TRY(ann_reserve(c, c->sel_state, sizeof(SelState)));
    ANN_TRY(ann_reserve(c, c->sel_hist, sizeof(uint32_t) * SEL_MAXQ * 256));
    SelState h;
    memset(&h, 0, sizeof h);
    h.nq = nk;
    for (int q = 0; q < nk; ++q) h.k[q] = ks[q];
    h.vor = 0; h.vand = ~0ull;
    ANN_TRY(ann_h2d(c, c->sel_state.p, &h, sizeof h));
    ANN_CHECK_HIP(c, hipMemsetAsync(c->sel_hist.p, 0, sizeof(uint32_t) * SEL_MAXQ * 256, c->stream));
    int blocks = (int)((n + 256 * 8 - 1) / (256 * 8));
    if (blocks > c->prop.multiProcessorCount * 4) blocks = c->prop.multiProcessorCount * 4;
    if (blocks < 1) blocks = 1;
    {
        ProfScope ps(c, "radix_select_f64", (double)n * 9 * 8);
        k_sel_orand<<<blocks, 256, 0, c->stream>>>(vals, flag, n, c->sel_state.as<SelState>());
        for (int pass = 0; pass < 8; ++pass) {
            k_sel_hist<<<blocks, 256, 0, c->stream>>>(vals, flag, n, c->sel_state.as<SelState>(), c->sel_hist.as<uint32_t>());
            k_sel_step<<<1, 256, 0, c->stream>>>(c->sel_state.as<SelState>(), c->sel_hist.as<uint32_t>());
        }
    }
    ANN_CHECK_HIP(c, hipGetLastError());
    ANN_TRY(ann_d2h(c, &h, c->sel_state.p, sizeof h));
    for (int q = 0; q < nk; ++q) {
        uint64_t key = h.prefix[q];
        uint64_t u = (key & 0x8000000000000000ull) ? (key & 0x7fffffffffffffffull) : ~key;
        memcpy(&h_out[q], &u, sizeof(double));
    }
    return ANNCHOR_OK;
}
