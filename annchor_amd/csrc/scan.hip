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

// One radix step for every query from the histogram of byte `pass` (thread d owns digit d):
// st_in -> (prefix, k) after the step, returned in registers to all threads through `sh`.
struct SelStepShared {
    uint32_t wsum[4];
    uint64_t prefix[SEL_MAXQ];
    int64_t k[SEL_MAXQ];
};
__device__ __forceinline__ void sel_step(const SelState *st_in, const uint32_t *hist, int pass, SelStepShared &sh)
{
    const int d = threadIdx.x, lane = d & 63, wave = d >> 6;
    const int shift = 56 - 8 * pass;
    const int nq = st_in->nq;
    const bool uniform = (((st_in->vor ^ st_in->vand) >> shift) & 0xffull) == 0;
    if (d < nq) {
        sh.prefix[d] = st_in->prefix[d] | (uniform ? (st_in->vor & (0xffull << shift)) : ((uint64_t)255 << shift));
        sh.k[d] = uniform ? st_in->k[d] : 0;   // non-uniform default: rank beyond the population clamps to the maximum
    }
    __syncthreads();
    if (uniform) return;
    for (int q = 0; q < nq; ++q) {
        const uint32_t h = hist[q * 256 + d];
        uint32_t inc = h;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
        }
        __syncthreads();
        if (lane == 63) sh.wsum[wave] = inc;
        __syncthreads();
        uint32_t base = 0;
        for (int w = 0; w < wave; ++w) base += sh.wsum[w];
        const int64_t ex = (int64_t)base + inc - h, k = st_in->k[q];
        if (h != 0 && k >= ex && k < ex + h) {
            sh.prefix[q] = st_in->prefix[q] | ((uint64_t)d << shift);
            sh.k[q] = k - ex;
        }
    }
    __syncthreads();
}

// Pass `pass` of the selection in ONE launch: every block first replays the previous pass's
// step from that pass's (now complete) histogram -- 256 bins, a microsecond -- then histograms
// byte `pass` of the keys under the resulting prefixes.  Block 0 also publishes the state for
// the next launch.  (A separate one-block step kernel per pass doubled the launch count of a
// selection that is launch bound: 17 launches of ~7 us.)  hist holds one table per pass.
__global__ __launch_bounds__(256) void k_sel_pass(const double *__restrict__ vals, const uint8_t *__restrict__ flag, int64_t n,
                                                 SelState *st, uint32_t *hist_all, int pass)
{
    __shared__ uint32_t lh[SEL_MAXQ * 256];
    __shared__ SelStepShared sh;
    // st[pass] = state before this pass's step is known; st[pass] is written by block 0 of this launch
    const SelState *st_prev = st + (pass > 0 ? pass - 1 : 0);
    const int nq = st_prev->nq;
    if (pass > 0) sel_step(st_prev, hist_all + (size_t)(pass - 1) * SEL_MAXQ * 256, pass - 1, sh);
    else {
        if ((int)threadIdx.x < nq) { sh.prefix[threadIdx.x] = st_prev->prefix[threadIdx.x]; sh.k[threadIdx.x] = st_prev->k[threadIdx.x]; }
        __syncthreads();
    }
    if (blockIdx.x == 0 && pass > 0) {   // publish (prefix, k) after step pass-1 for the next launch
        SelState *o = st + pass;
        if ((int)threadIdx.x < nq) { o->prefix[threadIdx.x] = sh.prefix[threadIdx.x]; o->k[threadIdx.x] = sh.k[threadIdx.x]; }
        if (threadIdx.x == 0) { o->nq = nq; o->vor = st_prev->vor; o->vand = st_prev->vand; }
    }
    if (pass == 8) return;   // final launch: only the last step
    const int shift = 56 - 8 * pass;
    if ((((st_prev->vor ^ st_prev->vand) >> shift) & 0xffull) == 0) return;  // uniform byte: the next step fills it in
    uint32_t *hist = hist_all + (size_t)pass * SEL_MAXQ * 256;
    for (int t = threadIdx.x; t < SEL_MAXQ * 256; t += blockDim.x) lh[t] = 0;
    uint64_t pre[SEL_MAXQ];
    for (int q = 0; q < SEL_MAXQ; ++q) pre[q] = q < nq ? sh.prefix[q] : 0;
    __syncthreads();
    const uint64_t himask = pass == 0 ? 0ull : (~0ull << (shift + 8));
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        if (flag && !flag[t]) continue;
        uint64_t key = ann_key_asc(vals[t]);
        uint32_t d = (uint32_t)(key >> shift) & 0xffu;
        for (int q = 0; q < nq; ++q)
            if ((key & himask) == (pre[q] & himask)) atomicAdd(&lh[q * 256 + d], 1u);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < nq * 256; t += blockDim.x)
        if (lh[t]) atomicAdd(&hist[t], lh[t]);
}

int ann_kth_smallest(annchor_ctx *c, const double *vals, const uint8_t *flag, int64_t n, const int64_t *ks, int nk,
                     double *h_out)
{
    ANN_REQUIRE(c, nk >= 1 && nk <= SEL_MAXQ, ANNCHOR_EINVAL, "kth_smallest: 1..%d ranks per call", SEL_MAXQ);
    ANN_TRY(ann_reserve(c, c->sel_state, sizeof(SelState) * 9));   // state before pass 0 .. after pass 7
    ANN_TRY(ann_reserve(c, c->sel_hist, sizeof(uint32_t) * 8 * SEL_MAXQ * 256));
    SelState h;
    memset(&h, 0, sizeof h);
    h.nq = nk;
    for (int q = 0; q < nk; ++q) h.k[q] = ks[q];
    h.vor = 0; h.vand = ~0ull;
    ANN_TRY(ann_h2d(c, c->sel_state.p, &h, sizeof h));
    ANN_CHECK_HIP(c, hipMemsetAsync(c->sel_hist.p, 0, sizeof(uint32_t) * 8 * SEL_MAXQ * 256, c->stream));
    int blocks = (int)((n + 256 * 8 - 1) / (256 * 8));
    if (blocks > c->prop.multiProcessorCount * 4) blocks = c->prop.multiProcessorCount * 4;
    if (blocks < 1) blocks = 1;
    {
        ProfScope ps(c, "radix_select_f64", (double)n * 9 * 8);
        k_sel_orand<<<blocks, 256, 0, c->stream>>>(vals, flag, n, c->sel_state.as<SelState>());
        for (int pass = 0; pass < 8; ++pass)
            k_sel_pass<<<blocks, 256, 0, c->stream>>>(vals, flag, n, c->sel_state.as<SelState>(), c->sel_hist.as<uint32_t>(), pass);
        k_sel_pass<<<1, 256, 0, c->stream>>>(vals, flag, n, c->sel_state.as<SelState>(), c->sel_hist.as<uint32_t>(), 8);
    }
    ANN_CHECK_HIP(c, hipGetLastError());
    ANN_TRY(ann_d2h(c, &h, c->sel_state.as<SelState>() + 8, sizeof h));
    for (int q = 0; q < nk; ++q) {
        uint64_t key = h.prefix[q];
        uint64_t u = (key & 0x8000000000000000ull) ? (key & 0x7fffffffffffffffull) : ~key;
        memcpy(&h_out[q], &u, sizeof(double));
    }
    return ANNCHOR_OK;
}
