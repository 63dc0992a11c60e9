// scan.hip -- small device primitives shared by the pipeline stages:
//   * exclusive prefix sum int32 -> int64,
//   * exact k-th order statistic of float64 keys under a byte mask (filter-then-finish radix
//     selection, up to 4 ranks per call; the MSB-first 8-bit radix select is its fallback).
// The second one implements np.partition(x[mask], k)[k] as used by
// SimpleStratifiedSampler.get_partition (reference annchor/samplers.py:119-140).
#include "common.h"
#include "selstate.h"

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
    int32_t v[SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) v[k] = ann_ldc(in, base + (int64_t)k * SCAN_THREADS + threadIdx.x, n);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + (int64_t)k * SCAN_THREADS + threadIdx.x < n) s += v[k];
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
    for (int k = 0; k < SCAN_ITEMS; ++k) v[k] = ann_ldc(in, base + k, n);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k >= n) v[k] = 0;
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

// short inputs (the per-row arrays of a small data set): the whole scan in one workgroup, one launch
#define SCAN1_THREADS 1024
#define SCAN1_ITEMS 8
__global__ __launch_bounds__(SCAN1_THREADS) void k_scan_one(const int32_t *__restrict__ in, int64_t n, int64_t *__restrict__ out)
{
    __shared__ int64_t wsum[SCAN1_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)threadIdx.x * SCAN1_ITEMS;   // thread-contiguous: scan order = index order
    int32_t v[SCAN1_ITEMS];
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN1_ITEMS; ++k) v[k] = ann_ldc(in, base + k, n);
#pragma unroll
    for (int k = 0; k < SCAN1_ITEMS; ++k) {
        if (base + k >= n) v[k] = 0;
        s += v[k];
    }
    int64_t inc = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int64_t o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int64_t pre = 0, tot = 0;
    for (int w = 0; w < SCAN1_THREADS / 64; ++w) { const int64_t x = wsum[w]; if (w < wave) pre += x; tot += x; }
    int64_t ex = pre + inc - s;
#pragma unroll
    for (int k = 0; k < SCAN1_ITEMS; ++k) {
        if (base + k < n) out[base + k] = ex;
        ex += v[k];
    }
    if (threadIdx.x == 0) out[n] = tot;
}

int ann_exclusive_scan_i32_to_i64(annchor_ctx *c, const int32_t *in, int64_t *out, int64_t n)
{
    if (n >= 1 && n <= SCAN1_THREADS * SCAN1_ITEMS) {
        ProfScope ps(c, "exclusive_scan", (double)n * 12);
        k_scan_one<<<1, SCAN1_THREADS, 0, c->stream>>>(in, n, out);
        ANN_CHECK_HIP(c, hipGetLastError());
        return ANNCHOR_OK;
    }
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
                                                 SelState *st, uint32_t *hist_all, int pass, int first)
{
    __shared__ uint32_t lh[SEL_MAXQ * 256];
    __shared__ SelStepShared sh;
    // st[pass] = state before this pass's step is known; st[pass] is written by block 0 of this launch
    // (first = the pass the selection starts at: its input state is st[first])
    const SelState *st_prev = st + (pass > first ? pass - 1 : first);
    const int nq = st_prev->nq;
    if (pass > first) sel_step(st_prev, hist_all + (size_t)(pass - 1) * SEL_MAXQ * 256, pass - 1, sh);
    else {
        if ((int)threadIdx.x < nq) { sh.prefix[threadIdx.x] = st_prev->prefix[threadIdx.x]; sh.k[threadIdx.x] = st_prev->k[threadIdx.x]; }
        __syncthreads();
    }
    if (blockIdx.x == 0 && pass > first) {   // publish (prefix, k) after step pass-1 for the next launch
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
    const uint64_t himask = shift == 56 ? 0ull : (~0ull << (shift + 8));
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

// ------------------------------------------------- filter-then-finish selection
// The byte-at-a-time selection above reads all n keys nine times in ten launches.  This one
// reads them twice: a histogram of the top 11 key bits, then one pass that keeps only the keys
// in the bucket(s) holding the wanted ranks while it histograms their next 11 bits.  A third
// launch narrows the candidates by 10 more bits, and a single workgroup finishes the last 32
// bits from LDS.  A bucket that is one repeated value (tied distances / probabilities) is
// recognised from the OR and AND of its keys; a mixed bucket too long for LDS falls back to
// the byte passes above.
//
// Atomics on ONE global address cost ~12.5 ns each on this part, serialised, returning or not
// (tools/microbench/atomics.hip: 625 workgroups -> 8.8 us, 4096 -> 50 us), so the layout avoids
// them: few fat workgroups (1024 threads, <= S2_MAXWG of them) flush their LDS histograms once,
// and the candidate lists are segmented -- tile i's survivors go to slots [i*S2_TILE, ...) with
// their count in segcnt[i] -- instead of being appended through a shared counter.
#define S2_T 1024
#define S2_ITEMS 8
#define S2_TILE (S2_T * S2_ITEMS)
#define S2_MAXWG 1024
#define S2_NB0 2048   // bits 63..53
#define S2_NB1 2048   // bits 52..42
#define S2_NB2 1024   // bits 41..32
#define S2_NB3 2048   // bits 31..21 (long lists only)
#define S2_CAP 4096   // candidates the finishing workgroup holds in LDS
#define S3_SAMPLE_MIN (1ll << 23)   // lists this long bracket the wanted ranks from a sample first (one read instead of two)
#define S2_LEVEL3_MIN (1ll << 22)   // lists this long filter a third time (smooth keys leave ~n / 2^11 candidates per level)
struct Sel2State {
    uint64_t prefix[SEL_MAXQ];
    int64_t k[SEL_MAXQ];
    int nq, unfinished;
    unsigned long long cnt;
};
// Sampled bracket (long lists): what the sample kernel decided and what the bracket pass counted
struct Sel3Info {
    uint64_t lo[SEL_MAXQ], hi[SEL_MAXQ];   // key brackets (lo, hi] holding each wanted rank (with overwhelming probability)
    long long lt[SEL_MAXQ], eq[SEL_MAXQ];  // flagged keys < lo[q] / == lo[q] (a heavy tie on the bracket's edge is counted, not copied)
    long long drop[SEL_MAXQ];              // flagged keys <= lo[q] outside every bracket: what the wanted rank moves down by
    long long kept;                        // keys inside the union of the brackets
};
struct Sel2Tables {
    uint32_t hist0[S2_NB0];
    uint32_t hist1[SEL_MAXQ * S2_NB1];
    uint32_t hist2[SEL_MAXQ * S2_NB2];
    uint32_t hist3[SEL_MAXQ * S2_NB3];
    // OR of key / OR of ~key over each query's candidates after level 2 ([0]) and level 3 ([1]): a bucket whose
    // keys are one repeated value shows vor == ~vnand on the open bits
    unsigned long long vor[2][SEL_MAXQ], vnand[2][SEL_MAXQ];
    // ---- not part of the zeroed region
    Sel2State st1, st2, st3, out;
    Sel3Info s3;   // (directly behind `out`: one download brings both)
    // the sample of the bracket path: keys (~0 = not flagged) gathered by S3_WG workgroups, the last one to finish reduces them
    uint64_t skey[8192];
    uint32_t sdone;
};
#define S2_ZERO_BYTES offsetof(Sel2Tables, st1)
struct Sel2Sh {
    uint32_t wsum[S2_T / 64];
    int digit;
    int64_t krem;
};

// rank k_in among NB bins (thread t owns bins t*PER..): the bin holding it and the rank inside
// that bin.  A rank beyond the population clamps to the largest key.
template <int NB> __device__ __forceinline__ void sel2_step(const uint32_t *hist, int64_t k_in, Sel2Sh &sh, int &digit, int64_t &krem)
{
    constexpr int PER = NB >= S2_T ? NB / S2_T : 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t cb[PER], s = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) { cb[j] = tid * PER + j < NB ? hist[tid * PER + j] : 0u; s += cb[j]; }
    uint32_t inc = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    __syncthreads();
    if (lane == 63) sh.wsum[wave] = inc;
    if (tid == 0) { sh.digit = NB - 1; sh.krem = 0; }
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (int w = 0; w < S2_T / 64; ++w) { const uint32_t x = sh.wsum[w]; if (w < wave) base += x; total += x; }
    int64_t k = k_in;
    if (k >= (int64_t)total) k = (int64_t)total - 1;
    int64_t ex = (int64_t)base + inc - s;
    if (s != 0 && k >= ex && k < ex + s) {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (k >= ex && k < ex + cb[j]) { sh.digit = tid * PER + j; sh.krem = k - ex; }
            ex += cb[j];
        }
    }
    __syncthreads();
    digit = sh.digit;
    krem = sh.krem;
}

// (segcnt != nullptr: `vals` is a segmented candidate list -- what the bracket pass kept)
__global__ __launch_bounds__(S2_T) void k_sel2_hist0(const double *__restrict__ vals, const uint8_t *__restrict__ flag, int64_t n,
                                                    Sel2Tables *__restrict__ tb, const uint32_t *__restrict__ segcnt)
{
    __shared__ uint32_t lh[S2_NB0];
    for (int t = threadIdx.x; t < S2_NB0; t += S2_T) lh[t] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t ntiles = (n + S2_TILE - 1) / S2_TILE;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t lim = segcnt ? segcnt[tile] : (uint32_t)S2_TILE;
        if (lim == 0) break;   // (segmented input comes packed from the bracket pass: nothing behind an empty segment)
        double v[S2_ITEMS];
        uint8_t f[S2_ITEMS];
#pragma unroll
        for (int j = 0; j < S2_ITEMS; ++j) {   // keys and flags in flight together
            const int64_t t = tile * S2_TILE + j * S2_T + threadIdx.x;
            v[j] = ann_ldc(vals, t, n);
            f[j] = flag ? ann_ldc(flag, t, n) : (uint8_t)1;
            if (t >= n || (uint32_t)(j * S2_T + threadIdx.x) >= lim) f[j] = 0;
        }
#pragma unroll
        for (int j = 0; j < S2_ITEMS; ++j) {
            // sign + exponent: a handful of distinct digits per wave, so the first lane's digit is
            // counted once for everyone who shares it
            const bool act = f[j] != 0;
            const uint32_t d = (uint32_t)(ann_key_asc(v[j]) >> 53);
            const unsigned long long am = __ballot(act);
            if (!am) continue;
            const int lead = __ffsll((long long)am) - 1;
            const uint32_t d0 = __shfl(d, lead);
            const unsigned long long m = __ballot(act && d == d0);
            if (lane == lead) atomicAdd(&lh[d0], (uint32_t)__popcll(m));
            if (act && d != d0) atomicAdd(&lh[d], 1u);
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < S2_NB0; t += S2_T)
        if (lh[t]) atomicAdd(&tb->hist0[t], lh[t]);
}

struct Sel2Init { int nq; int64_t k[SEL_MAXQ]; };

// every query's bucket is non-empty and one repeated value on the bits below `open`
__device__ __forceinline__ bool sel2_tied(const unsigned long long *vor, const unsigned long long *vnand, int nq, int open)
{
    const uint64_t low = (1ull << open) - 1;
    bool tied = true;
#pragma unroll
    for (int q = 0; q < SEL_MAXQ; ++q)
        if (q < nq && ((vor[q] ^ ~vnand[q]) & low)) tied = false;
    return tied;
}

// LEVEL 1: all keys -> candidates A (top 11 bits match a wanted bucket), histogram of bits 52..42.
// LEVEL 2: candidates A -> candidates B (top 22 bits match), histogram of bits 41..32.
// LEVEL 3: candidates B -> candidates C (top 32 bits match), histogram of bits 31..21.
// The candidate lists are segmented by tile (see above); segcnt_in is the view of the input list.
// Levels 2 and 3 also gather the tie statistics of their survivors; level 3 does nothing when
// level 2 already found every bucket to be one repeated value.
// rank of query q inside the kept list.  A query whose answer is its bracket's lower edge (the rank falls into the keys
// counted as == lo) needs no selection: it borrows the rank of the first query that does, so that it adds no candidates.
__device__ __forceinline__ bool sel3_on_edge(const Sel3Info &s, const Sel2Init &init, int q)
{
    return init.k[q] >= s.lt[q] && init.k[q] < s.lt[q] + s.eq[q];
}
__device__ __forceinline__ int64_t sel3_rank(const Sel3Info &s, const Sel2Init &init, int q)
{
    int p = q;
    if (sel3_on_edge(s, init, q)) {
        p = -1;
        for (int o = 0; o < init.nq; ++o)
            if (p < 0 && !sel3_on_edge(s, init, o)) p = o;
        if (p < 0) return 0;
    }
    return max((int64_t)0, init.k[p] - (int64_t)s.drop[p]);
}

template <int LEVEL, bool SEG = false> __global__ __launch_bounds__(S2_T, LEVEL == 1 ? 8 : 4) void k_sel2_filter(const double *__restrict__ vals,
                                                                          const uint8_t *__restrict__ flag, int64_t n,
                                                                          const uint32_t *__restrict__ segcnt_in, Sel2Init init,
                                                                          Sel2Tables *__restrict__ tb, double *__restrict__ dst,
                                                                          uint32_t *__restrict__ segcnt_out, int packed)
{
    constexpr int NBP = LEVEL == 1 ? S2_NB0 : LEVEL == 2 ? S2_NB1 : S2_NB2;
    constexpr int NBN = LEVEL == 1 ? S2_NB1 : LEVEL == 2 ? S2_NB2 : S2_NB3;
    constexpr int SHP = LEVEL == 1 ? 53 : LEVEL == 2 ? 42 : 32, SHN = LEVEL == 1 ? 42 : LEVEL == 2 ? 32 : 21;
    constexpr bool LAST = LEVEL >= 2;   // gathers tie statistics
    constexpr int SLOT = LEVEL == 3 ? 1 : 0;
    __shared__ uint32_t lh[SEL_MAXQ * NBN];
    __shared__ unsigned long long lor[SEL_MAXQ], lnand[SEL_MAXQ];
    __shared__ Sel2Sh sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Sel2State *si = LEVEL == 2 ? &tb->st1 : &tb->st2;   // (LEVEL 1 starts from `init`)
    const int nq = LEVEL == 1 ? init.nq : si->nq;
    if (LEVEL == 3 && sel2_tied(tb->vor[0], tb->vnand[0], nq, 42)) return;   // level 2 resolved everything
    uint64_t pre[SEL_MAXQ];
    int64_t kk[SEL_MAXQ];
#pragma unroll
    for (int q = 0; q < SEL_MAXQ; ++q) {
        pre[q] = 0; kk[q] = 0;
        if (q < nq) {
            int digit; int64_t krem;
            const uint32_t *hp = LEVEL == 1 ? tb->hist0 : LEVEL == 2 ? tb->hist1 + q * S2_NB1 : tb->hist2 + q * S2_NB2;
            // (SEG: the input is what the bracket pass kept; the wanted rank moves down by the keys it dropped below the bracket)
            const int64_t k1 = SEG ? sel3_rank(tb->s3, init, q) : init.k[q];
            sel2_step<NBP>(hp, LEVEL == 1 ? k1 : si->k[q], sh, digit, krem);
            pre[q] = (LEVEL == 1 ? 0ull : si->prefix[q]) | ((uint64_t)digit << SHP);
            // (workgroup-uniform: keep it in scalar registers)
            pre[q] = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(pre[q] >> 32)) << 32) |
                     (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)pre[q]);
            kk[q] = krem;
        }
    }
    Sel2State *so = LEVEL == 1 ? &tb->st1 : LEVEL == 2 ? &tb->st2 : &tb->st3;
    if (blockIdx.x == 0 && tid == 0) {
        for (int q = 0; q < SEL_MAXQ; ++q) { so->prefix[q] = pre[q]; so->k[q] = kk[q]; }
        so->nq = nq; so->unfinished = 0; so->cnt = 0;
    }
    for (int t = tid; t < nq * NBN; t += S2_T) lh[t] = 0;
    if (tid < SEL_MAXQ) { lor[tid] = 0; lnand[tid] = 0; }
    __syncthreads();
    uint32_t *hnext = LEVEL == 1 ? tb->hist1 : LEVEL == 2 ? tb->hist2 : tb->hist3;
    unsigned long long vo[SEL_MAXQ] = {0, 0, 0, 0}, vn[SEL_MAXQ] = {0, 0, 0, 0};
    const int64_t ntiles = (n + S2_TILE - 1) / S2_TILE;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t lim = (LEVEL == 1 && !SEG) ? min((int64_t)S2_TILE, n - tile * S2_TILE) : (int64_t)segcnt_in[tile];
        if (lim == 0) {   // an empty segment stays empty
            if (packed) {   // (the bracket pass packs a workgroup's keys into its first segments: nothing behind an empty one)
                for (int64_t o = tile + (int64_t)tid * gridDim.x; o < ntiles; o += (int64_t)S2_T * gridDim.x) segcnt_out[o] = 0;
                break;
            }
            if (tid == 0) segcnt_out[tile] = 0;
            continue;
        }
        double v[S2_ITEMS];
        uint8_t f[S2_ITEMS];
#pragma unroll
        for (int j = 0; j < S2_ITEMS; ++j) {
            const int o = j * S2_T + tid;
            const int64_t t = tile * S2_TILE + o;
            // (clamped, unconditional: the whole batch stays in flight; a tile always has lim >= 1 or is skipped)
            const int64_t tc = o < lim ? t : tile * S2_TILE;
            v[j] = 0.0; f[j] = 0;
            if ((LEVEL == 1 && !SEG) || j * S2_T < lim) {   // (uniform) a short segment fills only its first slices
                v[j] = vals[tc];
                f[j] = (LEVEL == 1 && flag) ? flag[tc] : (uint8_t)1;
                if (o >= lim) f[j] = 0;
            }
        }
        uint32_t keep = 0;
#pragma unroll
        for (int j = 0; j < S2_ITEMS; ++j) {
            if ((LEVEL != 1 || SEG) && j * S2_T >= lim) break;
            const uint64_t key = ann_key_asc(v[j]);
#pragma unroll
            for (int q = 0; q < SEL_MAXQ; ++q) {
                if (q >= nq) continue;
                const bool hit = f[j] && (key >> SHP) == (pre[q] >> SHP);
                // ties put whole waves on one bin: the first hit's digit is counted once for all who share it
                const unsigned long long hm = __ballot(hit);
                if (!hm) continue;
                const uint32_t d = (uint32_t)((key >> SHN) & (NBN - 1));
                const int lead = __ffsll((long long)hm) - 1;
                const uint32_t d0 = __shfl(d, lead);
                const unsigned long long m = __ballot(hit && d == d0);
                if (lane == lead) atomicAdd(&lh[q * NBN + d0], (uint32_t)__popcll(m));
                if (hit && d != d0) atomicAdd(&lh[q * NBN + d], 1u);
                if (hit) {
                    keep |= 1u << j;
                    if (LAST) { vo[q] |= key; vn[q] |= ~key; }
                }
            }
        }
        // this tile's own segment, wave by wave, item by item: the survivors of one 64-wide load
        // go to consecutive slots (coalesced stores); the order inside a segment is irrelevant
        uint32_t wtot = 0;
        uint32_t slot[S2_ITEMS];
#pragma unroll
        for (int j = 0; j < S2_ITEMS; ++j) {
            const unsigned long long km = __ballot((keep >> j) & 1u);
            slot[j] = wtot + (uint32_t)__popcll(km & ((1ull << lane) - 1ull));
            wtot += (uint32_t)__popcll(km);
        }
        __syncthreads();
        if (lane == 0) sh.wsum[wave] = wtot;
        __syncthreads();
        uint32_t wbase = 0, total = 0;
        for (int w = 0; w < S2_T / 64; ++w) { const uint32_t x = sh.wsum[w]; if (w < wave) wbase += x; total += x; }
        if (tid == 0) segcnt_out[tile] = total;
        double *seg = dst + tile * S2_TILE + wbase;
#pragma unroll
        for (int j = 0; j < S2_ITEMS; ++j)
            if (keep & (1u << j)) seg[slot[j]] = v[j];
    }
    if (LAST) {
        // which low bits all of a query's candidates share: a bucket that is one value repeated
        // (tied distances, tied probabilities) is then resolved without another pass
#pragma unroll
        for (int q = 0; q < SEL_MAXQ; ++q) {
            if (q >= nq) continue;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { vo[q] |= __shfl_xor(vo[q], off); vn[q] |= __shfl_xor(vn[q], off); }
            if (lane == 0 && (vo[q] | vn[q])) { atomicOr(&lor[q], vo[q]); atomicOr(&lnand[q], vn[q]); }
        }
    }
    __syncthreads();
    for (int t = tid; t < nq * NBN; t += S2_T)
        if (lh[t]) atomicAdd(&hnext[t], lh[t]);
    if (LAST && tid < nq && (lor[tid] | lnand[tid])) { atomicOr(&tb->vor[SLOT][tid], lor[tid]); atomicOr(&tb->vnand[SLOT][tid], lnand[tid]); }
}

// ------------------------------------------------- sampled bracket (long lists)
// The top 11 key bits are a double's sign and most of its exponent: on real columns (distances of one order of
// magnitude, probabilities) one or two buckets hold nearly every key, so level 1 above copies nearly the whole list and the
// selection costs two reads and a write of it.  For long lists a sample decides first where the wanted ranks lie:
// S3_M keys at a fixed stride, sorted by one workgroup; the sample rank of the k-th flagged key is Binomial(k, S3_M / n) --
// mean r = k S3_M / n, deviation <= sqrt(r) -- so the sample's order statistics r -+ (5 sqrt(r + 1) + 4) bracket it except
// with probability < 10^-6.  ONE pass over the list then keeps the keys inside the brackets (a few percent), counts what it
// drops below each bracket and histograms the top bits of what it keeps; levels 1..3 and the finishing workgroup run on that
// short list with the ranks moved down accordingly.  The host checks the answer against the bracket (and the moved rank
// against the kept count) when it downloads it and repeats the selection the plain way if the sample misled -- the result
// is exact either way.
#define S3_M 8192
#define S3_NB 4096
__device__ __forceinline__ uint32_t sel3_hash(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// One workgroup: S3_M keys, one per stratum of n / S3_M consecutive entries at a hashed offset inside it (a fixed stride
// aliases with the row structure of a pair list: at 16 000 points the stride is about one row), and two 4096-bin histograms
// of them -- linear in the key (a log-like scale: good for values over many binades) and linear in the value (good for a
// bounded range such as probabilities).  The bins holding the sample ranks r -+ w give each bracket edge twice; the tighter
// one counts.  A lower edge that IS a heavily repeated value (probability 0) excludes that value: its keys are counted.
#define S3_WG 8   // workgroups gathering the sample (8192 dependent-free but TLB-missing loads: 44 us from one CU)
__global__ __launch_bounds__(S2_T) void k_sel3_sample(const double *__restrict__ vals, const uint8_t *__restrict__ flag, int64_t n,
                                                     Sel2Init init, Sel2Tables *__restrict__ tb)
{
    __shared__ uint32_t hK[S3_NB], hV[S3_NB];
    __shared__ unsigned long long kmin_sh, kmax_sh;
    __shared__ int ms_sh, last_sh;
    __shared__ uint32_t wsum[2][S2_T / 64];
    __shared__ int found[SEL_MAXQ][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    static_assert(S3_M == S3_WG * S2_T && S3_M == 8192, "one sample per thread of the gathering workgroups");
    {
        const int s = blockIdx.x * S2_T + tid;
        const int64_t b0 = (int64_t)(((unsigned long long)s * (unsigned long long)n) / S3_M);
        const int64_t b1 = (int64_t)(((unsigned long long)(s + 1) * (unsigned long long)n) / S3_M);
        const int64_t pos = b0 + (int64_t)(sel3_hash((uint32_t)s * 2654435761u + 12345u) % (uint32_t)max((int64_t)1, b1 - b0));
        const bool f = flag ? flag[pos] != 0 : true;
        const uint64_t k = ann_key_asc(vals[pos]);
        tb->skey[s] = f ? k : ~0ull;
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) last_sh = atomicAdd(&tb->sdone, 1u) == (uint32_t)(gridDim.x - 1);
    __syncthreads();
    if (!last_sh) return;
    __threadfence();
    if (tid == 0) { kmin_sh = ~0ull; kmax_sh = 0ull; ms_sh = 0; tb->sdone = 0; }
    for (int t2 = tid; t2 < S3_NB; t2 += S2_T) { hK[t2] = 0; hV[t2] = 0; }
    __syncthreads();
    constexpr int PER = S3_M / S2_T;
    uint64_t key[PER];
    bool fl[PER];
    uint64_t mn = ~0ull, mx = 0ull;
    int mine = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        key[j] = __builtin_nontemporal_load(&tb->skey[j * S2_T + tid]);
        fl[j] = key[j] != ~0ull;
        if (fl[j]) { mn = min(mn, key[j]); mx = max(mx, key[j]); ++mine; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mn = min(mn, (uint64_t)__shfl_xor((unsigned long long)mn, off));
        mx = max(mx, (uint64_t)__shfl_xor((unsigned long long)mx, off));
        mine += __shfl_xor(mine, off);
    }
    if (lane == 0 && mine) { atomicMin(&kmin_sh, (unsigned long long)mn); atomicMax(&kmax_sh, (unsigned long long)mx); atomicAdd(&ms_sh, mine); }
    __syncthreads();
    const int ms = ms_sh;
    const uint64_t kmin = kmin_sh, kmax = kmax_sh, range = ms ? kmax - kmin : 0ull;
    const int sK = range ? max(0, 64 - (int)__clzll((long long)range) - 12) : 0;   // (range >> sK) < 4096
    const double vmin = ann_key_asc_inv(kmin), vmax = ann_key_asc_inv(kmax);
    const double span = vmax - vmin;
    const double scale = (ms && span > 0.0 && span < 1e300) ? (double)(S3_NB - 1) / span : 0.0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        // (a heavy tie -- most probabilities are exactly 0 -- puts whole waves on one bin: the leader counts for all who share its bin)
        const uint32_t bk = fl[j] ? (uint32_t)((key[j] - kmin) >> sK) : 0u;
        const double x = (ann_key_asc_inv(key[j]) - vmin) * scale;
        const uint32_t bv = (fl[j] && x >= 0.0) ? (x < (double)(S3_NB - 1) ? (uint32_t)x : (uint32_t)(S3_NB - 1)) : 0u;
        const unsigned long long am = __ballot(fl[j]);
        if (!am) continue;
        const int lead = __ffsll((long long)am) - 1;
        const uint32_t k0 = __shfl(bk, lead), v0 = __shfl(bv, lead);
        const unsigned long long mk = __ballot(fl[j] && bk == k0), mv = __ballot(fl[j] && bv == v0);
        if (lane == lead) { atomicAdd(&hK[k0], (uint32_t)__popcll(mk)); atomicAdd(&hV[v0], (uint32_t)__popcll(mv)); }
        if (fl[j] && bk != k0) atomicAdd(&hK[bk], 1u);
        if (fl[j] && bv != v0) atomicAdd(&hV[bv], 1u);
    }
    __syncthreads();
    {   // inclusive scans of both histograms, in place (4 bins per thread)
        constexpr int BP = S3_NB / S2_T;
        uint32_t a[BP], b[BP], sa = 0, sb = 0;
#pragma unroll
        for (int e = 0; e < BP; ++e) { sa += hK[tid * BP + e]; a[e] = sa; sb += hV[tid * BP + e]; b[e] = sb; }
        uint32_t ia = sa, ib = sb;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t oa = __shfl_up(ia, off), ob = __shfl_up(ib, off);
            if (lane >= off) { ia += oa; ib += ob; }
        }
        if (lane == 63) { wsum[0][wave] = ia; wsum[1][wave] = ib; }
        __syncthreads();
        uint32_t ba = 0, bb = 0;
        for (int w = 0; w < wave; ++w) { ba += wsum[0][w]; bb += wsum[1][w]; }
        ba += ia - sa; bb += ib - sb;
#pragma unroll
        for (int e = 0; e < BP; ++e) { hK[tid * BP + e] = ba + a[e]; hV[tid * BP + e] = bb + b[e]; }
    }
    __syncthreads();
    if (tid < 4 * SEL_MAXQ) {
        const int q = tid >> 2, which = tid & 3;   // 0: key map, lower  1: key map, upper  2: value map, lower  3: value map, upper
        int bin = -1;                              // -1: no bound
        if (q < init.nq && ms > 0) {
            const int64_t r = (int64_t)(((unsigned long long)init.k[q] * S3_M) / (unsigned long long)n);
            const int64_t w = (int64_t)(5.0 * sqrt((double)r + 1.0)) + 4;
            const int64_t tr = (which & 1) ? r + w : r - w;
            if (tr >= 0 && tr < ms) {
                const uint32_t *cum = (which & 2) ? hV : hK;
                int lo2 = 0, hi2 = S3_NB - 1;      // smallest bin with cum[bin] > tr
                while (lo2 < hi2) { const int mid = (lo2 + hi2) >> 1; if (cum[mid] > (uint32_t)tr) hi2 = mid; else lo2 = mid + 1; }
                bin = lo2;
            }
        }
        found[q][which] = bin;
    }
    __syncthreads();
    if (tid < SEL_MAXQ) {
        uint64_t lo = 0, hi = ~0ull;
        if (tid < init.nq && ms > 0) {
            const int bKl = found[tid][0], bKh = found[tid][1], bVl = found[tid][2], bVh = found[tid][3];
            if (bKl >= 0) lo = kmin + ((uint64_t)bKl << sK);
            if (bVl >= 0 && scale > 0.0) {   // (one bin of slack for the rounding of the edge)
                const uint64_t e = bVl <= 1 ? kmin : ann_key_asc(vmin + (double)(bVl - 1) / scale);
                lo = max(lo, min(e, kmax));
            }
            if (bKh >= 0) {
                const uint64_t top = ((uint64_t)(bKh + 1) << sK) - 1;
                hi = (sK + 12 >= 64 || kmin + top < kmin) ? ~0ull : kmin + top;
            }
            if (bVh >= 0 && scale > 0.0 && bVh + 2 < S3_NB - 1) hi = min(hi, max(kmin, ann_key_asc(vmin + (double)(bVh + 2) / scale)));
            if (hi < lo) { lo = 0; hi = ~0ull; }   // (cannot happen with consistent edges; never trade exactness of the check for it)
        }
        tb->s3.lo[tid] = lo; tb->s3.hi[tid] = hi; tb->s3.lt[tid] = 0; tb->s3.eq[tid] = 0; tb->s3.drop[tid] = 0;
    }
    if (tid == 0) tb->s3.kept = 0;
}

// The bracket pass.  Counters are per wave (ballot + popcount: scalar adds), not per thread, and there is no barrier inside
// the tile loop (a workgroup of 16 waves that meets twice per tile exposes every tile's load latency: 0.57 ms against 0.28 ms
// for the barrier-free histogram pass over the same bytes): a wave reserves room for its kept keys with one LDS atomic on the
// workgroup's running count g; slot g lives in the workgroup's (g / S2_TILE)-th output segment (segment ids = the tile ids
// it reads), so that the levels that follow find a few full segments per workgroup instead of a nearly empty one per tile.
// (A workgroup never keeps more keys than it reads: the segment exists.)
template <int NQ> __global__ __launch_bounds__(S2_T, 2) void k_sel3_bracket(const double *__restrict__ vals, const uint8_t *__restrict__ flag,
                                                                           int64_t n, Sel2Tables *__restrict__ tb, double *__restrict__ dst,
                                                                           uint32_t *__restrict__ segcnt_out)
{
    __shared__ unsigned long long bsum[3 * SEL_MAXQ];
    __shared__ uint32_t wg_fill;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < 3 * SEL_MAXQ) bsum[tid] = 0;
    if (tid == 0) wg_fill = 0;
    uint64_t lo[NQ], hi[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { lo[q] = tb->s3.lo[q]; hi[q] = tb->s3.hi[q]; }
    __syncthreads();
    uint32_t c_lt[NQ], c_le[NQ], c_kle[NQ];   // (per lane) flagged keys < lo / <= lo / kept keys <= lo
#pragma unroll
    for (int q = 0; q < NQ; ++q) c_lt[q] = c_le[q] = c_kle[q] = 0;
    const int64_t ntiles = (n + S2_TILE - 1) / S2_TILE;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        double v[S2_ITEMS];
        uint32_t f[S2_ITEMS];
        static_assert(S2_ITEMS % 2 == 0, "two consecutive keys per load");
#pragma unroll
        for (int j = 0; j < S2_ITEMS / 2; ++j) {
            // two consecutive keys (16 bytes) and their two flags per lane and load: half the load instructions of the
            // one-key form.  Clamped addresses, values masked afterwards (ann_ldc's reason); the lone last key of an odd
            // list arrives as the second half of the pair before it.
            const int64_t e0 = tile * S2_TILE + 2 * ((int64_t)j * S2_T + tid);
            const bool pair = e0 + 1 < n, lone = e0 + 1 == n;
            const int64_t pc = pair ? e0 : n - 2;
            const double2 vv = *reinterpret_cast<const double2 *>(vals + pc);
            uint32_t ff = 0x0101u;
            if (flag) ff = *reinterpret_cast<const unsigned short *>(flag + pc);
            v[2 * j] = pair ? vv.x : vv.y;
            v[2 * j + 1] = vv.y;
            f[2 * j] = pair ? (ff & 0xffu) : lone ? (ff >> 8) : 0u;
            f[2 * j + 1] = pair ? (ff >> 8) : 0u;
            asm volatile("" : "+v"(f[2 * j]), "+v"(f[2 * j + 1]));   // (values in registers here, not compare masks kept alive in scalar registers)
        }
        // one 64-wide slice at a time; the counters take the compare results as carries (per-lane counters, summed at the end)
        uint32_t keep = 0;   // (per lane: bit j = item j is kept)
        uint32_t wtot = 0;
        uint32_t slot[S2_ITEMS];
#pragma unroll
        for (int j = 0; j < S2_ITEMS; ++j) {
            const uint32_t vh = (uint32_t)__double2hiint(v[j]), vl = (uint32_t)__double2loint(v[j]);
            const uint32_t sm = (uint32_t)((int32_t)vh >> 31);   // ann_key_asc: negative values flip every bit, the others the sign bit
            const uint64_t key = ((uint64_t)(vh ^ (sm | 0x80000000u)) << 32) | (uint64_t)(vl ^ sm);
            const bool act = f[j] != 0;
            bool in = false;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const bool le = act && key <= lo[q];
                c_lt[q] += (act && key < lo[q]) ? 1u : 0u;
                c_le[q] += le ? 1u : 0u;
                // (pinned: left alone the compiler sums the eight slices' compare masks at the end of the tile and spills them all)
                asm volatile("" : "+v"(c_lt[q]), "+v"(c_le[q]));
                in = in || (act && !le && key <= hi[q]);
            }
            if (NQ > 1) {   // (one bracket: a kept key is never <= its lower edge)
#pragma unroll
                for (int q = 0; q < NQ; ++q) { c_kle[q] += (in && key <= lo[q]) ? 1u : 0u; asm volatile("" : "+v"(c_kle[q])); }
            }
            const unsigned long long km = __ballot(in);
            slot[j] = wtot + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
            wtot += (uint32_t)__popcll(km);
            keep |= in ? 1u << j : 0u;
            asm volatile("" : "+v"(keep), "+v"(slot[j]));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (wtot) {   // (wave-uniform)
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&wg_fill, wtot);
            base = __shfl(base, 0);
#pragma unroll
            for (int j = 0; j < S2_ITEMS; ++j)
                if (keep & (1u << j)) {
                    const uint32_t g = base + slot[j];
                    dst[((int64_t)blockIdx.x + (int64_t)(g / S2_TILE) * gridDim.x) * S2_TILE + g % S2_TILE] = v[j];
                }
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            c_lt[q] += __shfl_xor(c_lt[q], off); c_le[q] += __shfl_xor(c_le[q], off); c_kle[q] += __shfl_xor(c_kle[q], off);
        }
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (c_lt[q]) atomicAdd(&bsum[q], (unsigned long long)c_lt[q]);
            if (c_le[q] - c_lt[q]) atomicAdd(&bsum[SEL_MAXQ + q], (unsigned long long)(c_le[q] - c_lt[q]));
            if (c_le[q] - c_kle[q]) atomicAdd(&bsum[2 * SEL_MAXQ + q], (unsigned long long)(c_le[q] - c_kle[q]));
        }
    }
    __syncthreads();
    if (tid < 3 * SEL_MAXQ && bsum[tid]) {
        long long *dstc = tid < SEL_MAXQ ? &tb->s3.lt[tid] : tid < 2 * SEL_MAXQ ? &tb->s3.eq[tid - SEL_MAXQ] : &tb->s3.drop[tid - 2 * SEL_MAXQ];
        atomicAdd(reinterpret_cast<unsigned long long *>(dstc), bsum[tid]);
    }
    if (tid == 0) {
        uint32_t left = wg_fill;
        if (left) atomicAdd(reinterpret_cast<unsigned long long *>(&tb->s3.kept), (unsigned long long)left);
        for (int64_t o = blockIdx.x; o < ntiles; o += gridDim.x) {
            const uint32_t c = left < (uint32_t)S2_TILE ? left : (uint32_t)S2_TILE;
            segcnt_out[o] = c;
            left -= c;
        }
    }
}

// levels = filter levels run (2: hist2 splits bits 41..32, 32 bits left; 3: hist3 splits bits 31..21, 21 left)
__global__ __launch_bounds__(S2_T) void k_sel2_finish(Sel2Tables *__restrict__ tb, const double *__restrict__ cand,
                                                     const uint32_t *__restrict__ segcnt, int64_t nseg, int levels, Sel2Epilogue ep)
{
    __shared__ uint64_t keys[S2_CAP];
    __shared__ uint32_t lh[SEL_MAXQ * 256];
    __shared__ uint32_t sbase[S2_T], scnt[S2_T];
    __shared__ Sel2Sh sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Sel2State *si = levels == 2 ? &tb->st2 : &tb->st3;
    const int nq = tb->st2.nq;   // (st3 is stale when level 3 had nothing to do)
    const int rem0 = levels == 2 ? 32 : 21;          // key bits still open after this step
    const int shp = levels == 2 ? 42 : 32;           // the candidates agree with their query on bits >= shp
    uint64_t pre[SEL_MAXQ];
    int64_t kk[SEL_MAXQ];
#pragma unroll
    for (int q = 0; q < SEL_MAXQ; ++q) {
        pre[q] = 0; kk[q] = 0;
        if (q < nq) {
            int digit; int64_t krem;
            if (levels == 2) sel2_step<S2_NB2>(tb->hist2 + q * S2_NB2, si->k[q], sh, digit, krem);
            else sel2_step<S2_NB3>(tb->hist3 + q * S2_NB3, si->k[q], sh, digit, krem);
            pre[q] = si->prefix[q] | ((uint64_t)digit << rem0);
            kk[q] = krem;
        }
    }
    // how many candidates in all
    unsigned long long cnt = 0;
    {
        uint32_t s = 0;
        for (int64_t g = tid; g < nseg; g += S2_T) s += segcnt[g];
        unsigned long long x = s;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        __syncthreads();
        if (lane == 0) { sh.wsum[wave] = (uint32_t)x; }   // < 2^32 candidates (pair positions are int32)
        __syncthreads();
        for (int w = 0; w < S2_T / 64; ++w) cnt += sh.wsum[w];
    }
    // resolved at once if each query's bucket is one repeated value (tied distances / probabilities);
    // with three levels the second one may already have seen that (the third then did nothing)
    bool tied = false;
    if (levels == 3 && sel2_tied(tb->vor[0], tb->vnand[0], nq, 42)) {
        tied = true;
#pragma unroll
        for (int q = 0; q < SEL_MAXQ; ++q)
            if (q < nq) { pre[q] = tb->st2.prefix[q] | (tb->vor[0][q] & ((1ull << 42) - 1)); kk[q] = 0; }
    } else if (sel2_tied(tb->vor[levels - 2], tb->vnand[levels - 2], nq, shp)) {
        tied = true;
#pragma unroll
        for (int q = 0; q < SEL_MAXQ; ++q)
            if (q < nq) { pre[q] = si->prefix[q] | (tb->vor[levels - 2][q] & ((1ull << shp) - 1)); kk[q] = 0; }
    }
    const bool fits = !tied && cnt <= S2_CAP;   // else: the last 32 bits from LDS, or (mixed bucket too long) unfinished
    if (fits) {
        // gather the segments into LDS: 1024 segments at a time, a wave per segment
        uint32_t filled = 0;
        for (int64_t g0 = 0; g0 < nseg; g0 += S2_T) {
            const uint32_t c = g0 + tid < nseg ? segcnt[g0 + tid] : 0u;
            uint32_t inc = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                uint32_t o = __shfl_up(inc, off);
                if (lane >= off) inc += o;
            }
            __syncthreads();
            if (lane == 63) sh.wsum[wave] = inc;
            __syncthreads();
            uint32_t wbase = 0, total = 0;
            for (int w = 0; w < S2_T / 64; ++w) { const uint32_t x = sh.wsum[w]; if (w < wave) wbase += x; total += x; }
            if (total == 0) continue;   // (uniform; packed candidate lists leave most segments empty)
            sbase[tid] = filled + wbase + inc - c;
            scnt[tid] = c;
            __syncthreads();
            for (int sidx = wave; sidx < S2_T; sidx += S2_T / 64) {
                const uint32_t cs = scnt[sidx];
                for (uint32_t e = lane; e < cs; e += 64)
                    keys[sbase[sidx] + e] = ann_key_asc(cand[(g0 + sidx) * S2_TILE + e]);
            }
            filled += total;
        }
        for (int rem = rem0; rem > 0;) {
            const int w = rem < 8 ? rem : 8, shift = rem - w;
            for (int t = tid; t < nq * 256; t += S2_T) lh[t] = 0;
            __syncthreads();
            for (int t = tid; t < (int)cnt; t += S2_T) {
                const uint64_t key = keys[t];
#pragma unroll
                for (int q = 0; q < SEL_MAXQ; ++q)
                    if (q < nq && (key >> rem) == (pre[q] >> rem))
                        atomicAdd(&lh[q * 256 + (int)((key >> shift) & ((1u << w) - 1))], 1u);
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < SEL_MAXQ; ++q)
                if (q < nq) {
                    int digit; int64_t krem;
                    sel2_step<256>(lh + q * 256, kk[q], sh, digit, krem);
                    pre[q] |= (uint64_t)digit << shift;
                    kk[q] = krem;
                }
            rem = shift;
        }
    }
    __syncthreads();
    if (tid == 0) {
        for (int q = 0; q < SEL_MAXQ; ++q) { tb->out.prefix[q] = pre[q]; tb->out.k[q] = kk[q]; }
        tb->out.nq = nq; tb->out.unfinished = (fits || tied) ? 0 : 1; tb->out.cnt = cnt;
        if (ep.kind) sel2_epilogue(ep, pre, (fits || tied) ? 0 : 1);   // the chained consumer's state (selstate.h)
    }
    // leave the tables zeroed for the next selection (saves a memset launch per call)
    uint4 *z = reinterpret_cast<uint4 *>(tb);
    for (int t = tid; t < (int)(S2_ZERO_BYTES / 16); t += S2_T) z[t] = make_uint4(0, 0, 0, 0);
}

// The plain (two-read) selection enqueued without a host wait, for callers that chain device work on the answer
// (annchor_sampler_stats): the keys of the answers and the "unfinished" flag stay in the tables; the caller downloads the flag
// with its own results and calls ann_kth_async_done().  Lists long enough for the sampled bracket (whose check is the host's)
// are not taken here: *d_prefix stays null and the caller uses ann_kth_smallest.
int ann_kth_async(annchor_ctx *c, const double *vals, const uint8_t *flag, int64_t n, const int64_t *ks, int nk,
                  const unsigned long long **d_prefix, const int **d_unfinished, const Sel2Epilogue *epi)
{
    *d_prefix = nullptr; *d_unfinished = nullptr;
    ANN_REQUIRE(c, nk >= 1 && nk <= SEL_MAXQ, ANNCHOR_EINVAL, "kth_smallest: 1..%d ranks per call", SEL_MAXQ);
    ANN_REQUIRE(c, n > 0, ANNCHOR_EINVAL, "kth_smallest: empty list");
    const char *smin = getenv("ANNCHOR_SEL_SAMPLE_MIN");
    if (n >= (smin ? atoll(smin) : S3_SAMPLE_MIN) && n >= S3_M) return ANNCHOR_OK;
    const int64_t ntiles = std::max<int64_t>((n + S2_TILE - 1) / S2_TILE, 1);
    ANN_TRY(ann_reserve(c, c->sel2, sizeof(Sel2Tables)));
    ANN_TRY(ann_reserve(c, c->sel_bufA, sizeof(double) * (size_t)ntiles * S2_TILE));
    ANN_TRY(ann_reserve(c, c->sel_bufB, sizeof(double) * (size_t)ntiles * S2_TILE));
    ANN_TRY(ann_reserve(c, c->sel_seg, sizeof(uint32_t) * 2 * (size_t)ntiles));
    Sel2Tables *tb = c->sel2.as<Sel2Tables>();
    uint32_t *segA = c->sel_seg.as<uint32_t>(), *segB = segA + ntiles;
    if (c->sel2_clean != (const void *)tb) {
        ANN_CHECK_HIP(c, hipMemsetAsync(tb, 0, sizeof(Sel2Tables), c->stream));
        c->sel2_clean = nullptr;
    }
    Sel2Init init;
    memset(&init, 0, sizeof init);
    init.nq = nk;
    for (int q = 0; q < nk; ++q) init.k[q] = ks[q];
    const int grid = (int)(ntiles <= 256 ? ntiles : std::min<int64_t>(S2_MAXWG, std::max<int64_t>(256, ntiles / 4)));
    const char *l3 = getenv("ANNCHOR_SEL_LEVEL3_MIN");
    const bool three = n >= (l3 ? atoll(l3) : S2_LEVEL3_MIN);
    {
        ProfScope ps(c, "radix_select_f64", (double)n * 9.0);
        double *A = c->sel_bufA.as<double>(), *B = c->sel_bufB.as<double>();
        k_sel2_hist0<<<grid, S2_T, 0, c->stream>>>(vals, flag, n, tb, nullptr);
        k_sel2_filter<1><<<grid, S2_T, 0, c->stream>>>(vals, flag, n, nullptr, init, tb, A, segA, 0);
        k_sel2_filter<2><<<grid, S2_T, 0, c->stream>>>(A, nullptr, n, segA, init, tb, B, segB, 0);
        Sel2Epilogue ep;
        if (epi) ep = *epi; else memset(&ep, 0, sizeof ep);
        if (!three) k_sel2_finish<<<1, S2_T, 0, c->stream>>>(tb, B, segB, ntiles, 2, ep);
        else {
            k_sel2_filter<3><<<grid, S2_T, 0, c->stream>>>(B, nullptr, n, segB, init, tb, A, segA, 0);
            k_sel2_finish<<<1, S2_T, 0, c->stream>>>(tb, A, segA, ntiles, 3, ep);
        }
    }
    ANN_CHECK_HIP(c, hipGetLastError());
    c->sel2_clean = nullptr;
    *d_prefix = reinterpret_cast<const unsigned long long *>(&tb->out.prefix[0]);
    *d_unfinished = &tb->out.unfinished;
    return ANNCHOR_OK;
}
size_t ann_sel2_table_bytes() { return sizeof(Sel2Tables); }
void ann_kth_async_done(annchor_ctx *c) { c->sel2_clean = (const void *)c->sel2.p; }   // (the finishing workgroup left the tables zeroed)

int ann_kth_smallest(annchor_ctx *c, const double *vals, const uint8_t *flag, int64_t n, const int64_t *ks, int nk,
                     double *h_out)
{
    ANN_REQUIRE(c, nk >= 1 && nk <= SEL_MAXQ, ANNCHOR_EINVAL, "kth_smallest: 1..%d ranks per call", SEL_MAXQ);
    ANN_REQUIRE(c, n > 0, ANNCHOR_EINVAL, "kth_smallest: empty list");
    static_assert(S2_ZERO_BYTES % 16 == 0, "zeroed region is cleared with 16-byte stores");
    const int64_t ntiles = std::max<int64_t>((n + S2_TILE - 1) / S2_TILE, 1);
    ANN_TRY(ann_reserve(c, c->sel2, sizeof(Sel2Tables)));
    ANN_TRY(ann_reserve(c, c->sel_bufA, sizeof(double) * (size_t)ntiles * S2_TILE));
    ANN_TRY(ann_reserve(c, c->sel_bufB, sizeof(double) * (size_t)ntiles * S2_TILE));
    ANN_TRY(ann_reserve(c, c->sel_seg, sizeof(uint32_t) * 2 * (size_t)ntiles));
    Sel2Tables *tb = c->sel2.as<Sel2Tables>();
    uint32_t *segA = c->sel_seg.as<uint32_t>(), *segB = segA + ntiles;
    if (c->sel2_clean != (const void *)tb) {
        ANN_CHECK_HIP(c, hipMemsetAsync(tb, 0, sizeof(Sel2Tables), c->stream));
        c->sel2_clean = nullptr;
    }
    Sel2Init init;
    memset(&init, 0, sizeof init);
    init.nq = nk;
    for (int q = 0; q < nk; ++q) init.k[q] = ks[q];
    // few fat workgroups: every one ends with atomics on shared histogram bins (~12.5 ns each, serialised)
    const int grid = (int)(ntiles <= 256 ? ntiles : std::min<int64_t>(S2_MAXWG, std::max<int64_t>(256, ntiles / 4)));
    Sel2Epilogue no_ep;
    memset(&no_ep, 0, sizeof no_ep);
    struct { Sel2State out; Sel3Info s3; } dl;
    static_assert(offsetof(Sel2Tables, s3) == offsetof(Sel2Tables, out) + sizeof(Sel2State), "s3 sits directly behind out");
    Sel2State &out = dl.out;
    const char *l3 = getenv("ANNCHOR_SEL_LEVEL3_MIN");   // tests reach the three-level route on short lists
    const char *smin = getenv("ANNCHOR_SEL_SAMPLE_MIN");  // ... and the sampled bracket
    const bool three = n >= (l3 ? atoll(l3) : S2_LEVEL3_MIN);
    bool sampled = n >= (smin ? atoll(smin) : S3_SAMPLE_MIN) && n >= S3_M;
    const int grid_full = grid;
    for (;;) {
        // (sampled: every workgroup resident at once -- two per CU -- and half as many fixed costs in the short passes that follow)
        const int grid = sampled ? std::min(grid_full, 2 * c->prop.multiProcessorCount) : grid_full;
        {
            // algorithmic bytes: one read of the keys and their flags
            ProfScope ps(c, "radix_select_f64", (double)n * 9.0);
            double *A = c->sel_bufA.as<double>(), *B = c->sel_bufB.as<double>();
            uint32_t *sa = segA, *sb = segB;
            if (sampled) {
                k_sel3_sample<<<S3_WG, S2_T, 0, c->stream>>>(vals, flag, n, init, tb);
                switch (nk) {
                case 1: k_sel3_bracket<1><<<grid, S2_T, 0, c->stream>>>(vals, flag, n, tb, B, segB); break;
                case 2: k_sel3_bracket<2><<<grid, S2_T, 0, c->stream>>>(vals, flag, n, tb, B, segB); break;
                case 3: k_sel3_bracket<3><<<grid, S2_T, 0, c->stream>>>(vals, flag, n, tb, B, segB); break;
                default: k_sel3_bracket<4><<<grid, S2_T, 0, c->stream>>>(vals, flag, n, tb, B, segB); break;
                }
                k_sel2_hist0<<<grid, S2_T, 0, c->stream>>>(B, nullptr, n, tb, segB);
                k_sel2_filter<1, true><<<grid, S2_T, 0, c->stream>>>(B, nullptr, n, segB, init, tb, A, segA, 1);
            } else {
                k_sel2_hist0<<<grid, S2_T, 0, c->stream>>>(vals, flag, n, tb, nullptr);
                k_sel2_filter<1><<<grid, S2_T, 0, c->stream>>>(vals, flag, n, nullptr, init, tb, A, segA, 0);
            }
            k_sel2_filter<2><<<grid, S2_T, 0, c->stream>>>(A, nullptr, n, sa, init, tb, B, sb, sampled);
            if (!three) {
                k_sel2_finish<<<1, S2_T, 0, c->stream>>>(tb, B, sb, ntiles, 2, no_ep);
            } else {   // (C reuses A's slots: A is dead once B exists)
                k_sel2_filter<3><<<grid, S2_T, 0, c->stream>>>(B, nullptr, n, sb, init, tb, A, sa, sampled);
                k_sel2_finish<<<1, S2_T, 0, c->stream>>>(tb, A, sa, ntiles, 3, no_ep);
            }
        }
        ANN_CHECK_HIP(c, hipGetLastError());
        c->sel2_clean = nullptr;
        ANN_TRY(ann_d2h(c, &dl, &tb->out, sizeof dl));
        c->sel2_clean = (const void *)tb;
        if (!sampled) break;
        // per query: the rank falls into the keys counted as == lo (answer: lo), or above (answer: the selection inside the kept
        // list, checked against the bracket), or below lo (the sample misled)
        bool ok = true, need_sel = false;
        uint64_t ans[SEL_MAXQ];
        for (int q = 0; q < nk; ++q) {
            const long long k = ks[q], lt = dl.s3.lt[q], eq = dl.s3.eq[q];
            if (k >= lt && k < lt + eq) { ans[q] = dl.s3.lo[q]; continue; }
            need_sel = true;
            const long long kp = k - dl.s3.drop[q];
            ans[q] = out.prefix[q];
            ok = ok && k >= lt + eq && kp >= 0 && kp < dl.s3.kept &&
                 (out.unfinished || (out.prefix[q] > dl.s3.lo[q] && out.prefix[q] <= dl.s3.hi[q]));
        }
        if (getenv("ANNCHOR_SEL_DEBUG"))
            for (int q = 0; q < nk; ++q)
                fprintf(stderr, "sel3 n=%lld q=%d k=%lld lt=%lld eq=%lld drop=%lld kept=%lld lo=%016llx hi=%016llx got=%016llx unfinished=%d ok=%d\n",
                        (long long)n, q, (long long)ks[q], dl.s3.lt[q], dl.s3.eq[q], dl.s3.drop[q], dl.s3.kept, (unsigned long long)dl.s3.lo[q],
                        (unsigned long long)dl.s3.hi[q], (unsigned long long)out.prefix[q], out.unfinished, (int)ok);
        if (need_sel && out.unfinished && ok) break;   // (a mixed bucket too long for the finishing workgroup: the byte passes below)
        if (ok) {
            for (int q = 0; q < nk; ++q) out.prefix[q] = ans[q];
            out.unfinished = 0;
            break;
        }
        sampled = false;   // the sample misled (adversarial order): the plain two-read selection
    }
    if (out.unfinished) {
        // a mixed bucket longer than the finishing workgroup's LDS: the byte passes over all keys
        ANN_TRY(ann_reserve(c, c->sel_pass_state, sizeof(SelState) * 9));
        ANN_TRY(ann_reserve(c, c->sel_hist, sizeof(uint32_t) * 8 * SEL_MAXQ * 256));
        SelState h;
        memset(&h, 0, sizeof h);
        h.nq = nk;
        for (int q = 0; q < nk; ++q) h.k[q] = ks[q];
        h.vor = ~0ull; h.vand = 0ull;   // no byte is known to be uniform
        ANN_TRY(ann_h2d(c, c->sel_pass_state.p, &h, sizeof h));
        ANN_CHECK_HIP(c, hipMemsetAsync(c->sel_hist.p, 0, sizeof(uint32_t) * 8 * SEL_MAXQ * 256, c->stream));
        int blocks = (int)std::min<int64_t>((n + 256 * 8 - 1) / (256 * 8), c->prop.multiProcessorCount * 4);
        if (blocks < 1) blocks = 1;
        {
            ProfScope ps(c, "radix_select_f64_bytewise", (double)n * 9.0);
            for (int pass = 0; pass < 8; ++pass)
                k_sel_pass<<<blocks, 256, 0, c->stream>>>(vals, flag, n, c->sel_pass_state.as<SelState>(), c->sel_hist.as<uint32_t>(), pass, 0);
            k_sel_pass<<<1, 256, 0, c->stream>>>(vals, flag, n, c->sel_pass_state.as<SelState>(), c->sel_hist.as<uint32_t>(), 8, 0);
        }
        ANN_CHECK_HIP(c, hipGetLastError());
        ANN_TRY(ann_d2h(c, &h, c->sel_pass_state.as<SelState>() + 8, sizeof h));
        for (int q = 0; q < nk; ++q) out.prefix[q] = h.prefix[q];
    }
    for (int q = 0; q < nk; ++q) {
        uint64_t key = out.prefix[q];
        uint64_t u = (key & 0x8000000000000000ull) ? (key & 0x7fffffffffffffffull) : ~key;
        memcpy(&h_out[q], &u, sizeof(double));
    }
    return ANNCHOR_OK;
}
