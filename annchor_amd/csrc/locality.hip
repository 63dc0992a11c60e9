// locality.hip -- candidate-pair generation (permutation / shared-nearest-anchor prefilter).
//
// Replaces Annchor.get_locality (reference annchor/annchor.py:208-256) with
// get_check / adjust_check / create_IJs / get_IJs_from_check
// (annchor/utils.py:437-540) and check_locality_size (utils.py:592-597).
//
// The reference materialises a dense one-hot matrix and Python dicts; here
//   * sid[i]   = the `locality` nearest anchors of point i as a bit mask (1, 2 or 4 64-bit words: up to 256 anchors)
//                (ties resolve to the smaller anchor index),
//   * c_ij     = popcount(sid[i] & sid[j])            (= sum(A[sid[i], :])[j]),
//   * thr_i    = min(loc_thresh, (loc_min+1)-th largest c_i.)   (utils.py:472-480),
//   * keep_ij  = c_ij >= min(thr_i, thr_j)            (adjust_check symmetrisation),
// and keep is stored as a symmetric bitmap with per-word prefix counts, from which
// the sorted pair list IJs and the CSR index I are written without atomics:
// rank queries are O(1) popcounts.  I[i] lists pair positions by ascending other
// endpoint (the reference's order inside a group is arbitrary -- unstable argsort).
#include "common.h"

template <int NW> __global__ void k_sid(const double *__restrict__ Dt, int64_t nx, int na, int locality, uint64_t *__restrict__ sid,
                                        int32_t *__restrict__ cA)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nx) return;
    Sid<NW> mask = sid_zero<NW>();
    int first = 0;
    for (int r = 0; r < locality; ++r) {
        double best = INFINITY;
        int ba = -1;
        for (int a = 0; a < na; ++a) {
            if ((mask.w[NW == 1 ? 0 : a >> 6] >> (a & 63)) & 1ull) continue;
            double v = Dt[(size_t)a * nx + i];
            if (ba < 0 || v < best) { best = v; ba = a; }  // strict <: smaller index wins ties
        }
        if (ba < 0) break;
        if (r == 0) first = ba;
#pragma unroll
        for (int w = 0; w < NW; ++w)
            if (w == (ba >> 6)) mask.w[w] |= 1ull << (ba & 63);
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) sid[i * NW + w] = mask.w[w];
    cA[i] = first;  // np.argmin(D[i]) -- first minimal index (utils.py:375)
}

#define LOC_THREADS 256
// one block per row: histogram of c_ij over all j, then the (loc_min+1)-th largest
template <int NW>
__global__ __launch_bounds__(LOC_THREADS) void k_loc_thresh(const uint64_t *__restrict__ sid, int64_t nx, int loc_thresh,
                                                           int loc_min, int32_t *__restrict__ thr)
{
    __shared__ uint32_t hist[ANN_MAX_ANCHORS + 1];
    for (int t = threadIdx.x; t < ANN_MAX_ANCHORS + 1; t += blockDim.x) hist[t] = 0;
    __syncthreads();
    const int64_t i = blockIdx.x;
    const Sid<NW> mi = sid_ld<NW>(sid, i);
    for (int64_t j = threadIdx.x; j < nx; j += blockDim.x) atomicAdd(&hist[sid_common<NW>(mi, sid_ld<NW>(sid, j))], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t lm = loc_min < nx - 1 ? loc_min : nx - 1;
        int64_t cum = 0;
        int v = ANN_MAX_ANCHORS;
        for (; v >= 0; --v) {
            cum += hist[v];
            if (cum >= lm + 1) break;
        }
        if (v < 0) v = 0;
        thr[i] = v < loc_thresh ? v : loc_thresh;
    }
}

// Register-counter form for small thresholds (loc_thresh <= 8, the usual 1-5): the histogram above funnels nx LDS atomics per
// row into the handful of bins popcounts of 5-of-60 masks can take -- 19 ms at 100 000 points.  thr = the largest t <= loc_thresh
// with #{j : shared(i, j) >= t} >= loc_min + 1 (the same cut: the cumulative histogram from the top), so T running counts per
// thread do, reduced once per row.
template <int T, int NW>
__global__ __launch_bounds__(LOC_THREADS) void k_loc_thresh_small(const uint64_t *__restrict__ sid, int64_t nx, int loc_min, int32_t *__restrict__ thr)
{
    __shared__ uint32_t part[LOC_THREADS / 64][T];
    const int64_t i = blockIdx.x;
    const Sid<NW> mi = sid_ld<NW>(sid, i);
    uint32_t c[T];
#pragma unroll
    for (int t = 0; t < T; ++t) c[t] = 0;
    for (int64_t j0 = threadIdx.x; j0 < nx; j0 += 4 * LOC_THREADS) {
        Sid<NW> mj[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) mj[e] = sid_ld<NW>(sid, min(j0 + e * LOC_THREADS, nx - 1));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int cc = j0 + e * LOC_THREADS < nx ? sid_common<NW>(mi, mj[e]) : 0;
#pragma unroll
            for (int t = 0; t < T; ++t) c[t] += cc > t;
        }
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c[t] += __shfl_xor(c[t], off);
    }
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int t = 0; t < T; ++t) part[threadIdx.x >> 6][t] = c[t];
    __syncthreads();
    if (threadIdx.x == 0) {
        const int64_t lm = loc_min < nx - 1 ? loc_min : nx - 1;
        int v = 0;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            uint32_t g = 0;   // #{j : shared >= t + 1}
            for (int w = 0; w < LOC_THREADS / 64; ++w) g += part[w][t];
            if ((int64_t)g >= lm + 1) v = t + 1;
        }
        thr[i] = v;
    }
}

// keep bits: wave per (row, 64-column word) item, lane = column -- the 64 sid / thr reads of a word are one
// line each and the word is the wave's ballot (a thread per word walking its 64 columns read 64 scattered
// 8-byte pieces per load instruction)
template <int NW>
__global__ __launch_bounds__(256) void k_keep_bits(const uint64_t *__restrict__ sid, const int32_t *__restrict__ thr, int64_t nx, int kw,
                                                  uint64_t *__restrict__ K)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t wave_count = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t items = nx * kw;
    // a wave takes a run of consecutive words of (mostly) one row: the row's sid / thr stay in scalar registers
    const int64_t per = (items + wave_count - 1) / wave_count;
    const int64_t t0 = wave_global * per, t1 = min(t0 + per, items);
    for (int64_t t = t0; t < t1; ++t) {
        const int64_t i = t / kw;
        const int w = (int)(t - i * kw);
        const Sid<NW> mi = sid_ld<NW>(sid, i);
        const int ti = thr[i];
        const int64_t j = (int64_t)w * 64 + lane;
        bool keep = false;
        if (j < nx && j != i) {
            const int cc = sid_common<NW>(mi, sid_ld<NW>(sid, j));
            const int tj = thr[j];
            keep = cc >= (ti < tj ? ti : tj);
        }
        const unsigned long long bits = __ballot(keep);
        if (lane == 0) K[t] = bits;
    }
}

// Column-stationary form for long tables: k_keep_bits spends ~110 vector instructions per (row, word) -- a 64-bit division, the
// columns' sid / thr re-read for every row -- and was 80 % VALU-busy for 35 ms at 100 000 points (PMC).  Here a wave owns one word's
// 64 columns (their sid / thr in registers) and walks a block of 64 rows, whose sid / thr are contiguous scalar reads; the 64
// ballots are collected one per lane and stored together.  The four waves of a workgroup take neighbouring words of
// the same rows, so their 8-byte stores fill the same lines.
template <int NW>
__global__ __launch_bounds__(256) void k_keep_bits_cols(const uint64_t *__restrict__ sid, const int32_t *__restrict__ thr, int64_t nx, int kw,
                                                       uint64_t *__restrict__ K)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t task = (int64_t)blockIdx.x * 4 + wave;
    const int64_t rb = task / kw;
    const int w = (int)(task - rb * kw);
    if (rb * 64 >= nx) return;
    const int64_t i0 = rb * 64, i_base = min(i0, nx - 64);   // (the last block re-walks the 64 rows that end at nx; nx >= 64 here)
    const int64_t j = (int64_t)w * 64 + lane;
    const bool jv = j < nx;
    const Sid<NW> mj = jv ? sid_ld<NW>(sid, j) : sid_zero<NW>();
    const int tj = jv ? thr[j] : 0;
    int r_lo = 0, r_hi = 0;
#pragma unroll 16
    for (int r = 0; r < 64; ++r) {
        const int64_t i = i_base + r;            // uniform
        const Sid<NW> mi = sid_ld<NW>(sid, i);
        const int ti = thr[i];
        const int cc = sid_common<NW>(mi, mj);
        const bool keep = jv && j != i && cc >= (ti < tj ? ti : tj);
        const unsigned long long bits = __ballot(keep);
        r_lo = lane == r ? (int)(uint32_t)bits : r_lo;
        r_hi = lane == r ? (int)(uint32_t)(bits >> 32) : r_hi;
    }
    const int64_t i = i_base + lane;
    if (i >= i0) K[i * kw + w] = ((uint64_t)(uint32_t)r_hi << 32) | (uint32_t)r_lo;
}

// one block per row: exclusive prefix of popcounts over the row's words
__global__ __launch_bounds__(LOC_THREADS) void k_row_prefix(const uint64_t *__restrict__ K, int64_t nx, int kw,
                                                           uint32_t *__restrict__ pref, int32_t *__restrict__ deg,
                                                           int32_t *__restrict__ low, int32_t *__restrict__ up)
{
    __shared__ uint32_t wsum[LOC_THREADS / 64];
    __shared__ uint32_t carry_s;
    const int64_t i = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < kw; base += LOC_THREADS) {
        int w = base + threadIdx.x;
        uint32_t v = w < kw ? (uint32_t)__popcll(K[i * kw + w]) : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t b = carry_s;
        for (int ww = 0; ww < wave; ++ww) b += wsum[ww];
        if (w < kw) pref[i * kw + w] = b + inc - v;
        __syncthreads();
        if (threadIdx.x == LOC_THREADS - 1) carry_s = b + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        uint32_t d = carry_s;
        int wi = (int)(i >> 6);
        uint32_t l = pref[i * kw + wi] + (uint32_t)__popcll(K[i * kw + wi] & ((1ull << (i & 63)) - 1ull));
        deg[i] = (int32_t)d;
        low[i] = (int32_t)l;
        up[i] = (int32_t)(d - l);
    }
}

__device__ __forceinline__ uint32_t keep_rank(const uint64_t *K, const uint32_t *pref, int kw, int64_t row, int64_t col)
{
    int w = (int)(col >> 6);
    return pref[row * kw + w] + (uint32_t)__popcll(K[row * kw + w] & ((1ull << (col & 63)) - 1ull));
}

// wave per (row, 64-column word), lane = column bit: consecutive kept columns are consecutive
// positions of the pair list and of the CSR index, so both are written in whole lines (one thread per
// word walking its bits wrote 64 scattered 8-byte pieces per store instruction)
__global__ __launch_bounds__(256) void k_emit_pairs(const uint64_t *__restrict__ K, const uint32_t *__restrict__ pref, int64_t nx, int kw,
                                                   const int32_t *__restrict__ low, const int64_t *__restrict__ rowstart,
                                                   const int64_t *__restrict__ Iptr, int2 *__restrict__ ij, int32_t *__restrict__ Iidx,
                                                   int stream, int rows_only)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t wave_count = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t items = nx * kw;
    for (int64_t t = wave_global; t < items; t += wave_count) {
        const int64_t i = t / kw;
        const int w = (int)(t - i * kw);
        const uint64_t bits = K[t];          // wave-uniform
        if (bits == 0) continue;
        const bool mine = (bits >> lane) & 1ull;
        if (!mine) continue;
        const uint32_t r = pref[t] + (uint32_t)__popcll(bits & ((1ull << lane) - 1ull));   // rank of this column among row i's partners
        const int64_t j = (int64_t)w * 64 + lane;
        int64_t pos;
        if (j > i) {
            pos = rowstart[i] + ((int64_t)r - low[i]);
            ann_store(reinterpret_cast<long long *>(ij) + pos, (long long)(((unsigned long long)(uint32_t)j << 32) | (uint32_t)i), stream);
        } else {
            if (rows_only) continue;   // column-like entries come from k_emit_cols
            pos = rowstart[j] + ((int64_t)keep_rank(K, pref, kw, j, i) - low[j]);
        }
        ann_store(Iidx + Iptr[i] + r, (int32_t)pos, stream);
    }
}

// Long-table form of the row half (rows_only): k_emit_pairs pays a 64-bit division and four dependent uniform reads for every
// (row, word) it looks at -- 20 ms on the 1.6 x 10^8 words of 100 000 points, most of them for words that hold no pair.  Here a
// wave takes RUNS of 64 consecutive words: their bitmap and prefix words come in one lane-parallel read each, empty words cost
// nothing, the row's three table entries are re-read only when the row changes.
__global__ __launch_bounds__(256) void k_emit_rows_run(const uint64_t *__restrict__ K, const uint32_t *__restrict__ pref, int64_t nx, int kw,
                                                      const int32_t *__restrict__ low, const int64_t *__restrict__ rowstart,
                                                      const int64_t *__restrict__ Iptr, int2 *__restrict__ ij, int32_t *__restrict__ Iidx,
                                                      int stream)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t wave_count = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t items = nx * kw;
    int64_t cur_i = -1, rs = 0, ip = 0;
    int lo = 0;
    for (int64_t base = wave_global * 64; base < items; base += wave_count * 64) {
        const int64_t tt = base + lane;
        const uint64_t bitsL = tt < items ? K[tt] : 0ull;
        const uint32_t prefL = tt < items ? pref[tt] : 0u;
        unsigned long long active = __ballot(bitsL != 0ull);
        if (!active) continue;
        const int64_t i0 = base / kw;
        const int w0 = (int)(base - i0 * kw);
        const uint32_t b_lo = (uint32_t)bitsL, b_hi = (uint32_t)(bitsL >> 32);
        while (active) {
            const int q = __ffsll((long long)active) - 1;   // uniform
            active &= active - 1;
            int64_t i = i0;
            int w = w0 + q;
            while (w >= kw) { w -= kw; ++i; }
            const uint64_t bits = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)b_hi, q) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)b_lo, q);
            const uint32_t pr = (uint32_t)__builtin_amdgcn_readlane((int)prefL, q);
            if (i != cur_i) { cur_i = i; rs = rowstart[i]; lo = low[i]; ip = Iptr[i]; }
            const int64_t j = (int64_t)w * 64 + lane;
            if (((bits >> lane) & 1ull) && j > i) {
                const uint32_t r = pr + (uint32_t)__popcll(bits & ((1ull << lane) - 1ull));   // rank of this column among row i's partners
                const int64_t pos = rs + ((int64_t)r - lo);
                ann_store(reinterpret_cast<long long *>(ij) + pos, (long long)(((unsigned long long)(uint32_t)j << 32) | (uint32_t)i), stream);
                ann_store(Iidx + ip + r, (int32_t)pos, stream);
            }
        }
    }
}

// The column-like half of the CSR index: entry (j, i), j < i, of row i is the position of pair (j, i) in
// row j's run of the pair list.  Computed where it is cheap -- in row j, from row j's bitmap words -- and
// handed to row i through a 64 x 64 LDS tile, so that both the bitmap reads and the index writes are whole
// lines (looked up from row i's side it is two scattered table reads per entry: 4.1 ms at 127 M pairs).
#define EC_T 64
#define EC_SUPER 16
__global__ __launch_bounds__(256) void k_emit_cols(const uint64_t *__restrict__ K, const uint32_t *__restrict__ pref, int kw,
                                                  const int32_t *__restrict__ low, const int64_t *__restrict__ rowstart,
                                                  const int64_t *__restrict__ Iptr, int64_t nx, int32_t *__restrict__ Iidx, int stream, int supertiles)
{
    __shared__ int32_t tp[EC_T][EC_T + 1];
    // tile (jb, ib), jb <= ib: 16 x 16 super-tiles dealt to the XCDs (id % 8), as in k_transpose_cols (refine.hip) -- the column
    // side's table lines are shared by 16 (32) tiles that row-major order ran a whole tile row apart
    const int nb = kw;
    int jb, ib;
    if (supertiles) {
        const int S = (nb + EC_SUPER - 1) / EC_SUPER;
        const int64_t k_in = (int64_t)blockIdx.x >> 3;
        const int64_t st = (k_in / (EC_SUPER * EC_SUPER)) * 8 + (blockIdx.x & 7);
        if (st >= (int64_t)S * (S + 1) / 2) return;
        int JB = (int)((2.0 * S + 1.0 - sqrt((2.0 * S + 1.0) * (2.0 * S + 1.0) - 8.0 * (double)st)) * 0.5);
        while ((int64_t)JB * S - (int64_t)JB * (JB - 1) / 2 > st) --JB;
        while ((int64_t)(JB + 1) * S - (int64_t)(JB + 1) * JB / 2 <= st) ++JB;
        const int IB = JB + (int)(st - ((int64_t)JB * S - (int64_t)JB * (JB - 1) / 2));
        const int w_in = (int)(k_in % (EC_SUPER * EC_SUPER));
        jb = JB * EC_SUPER + w_in / EC_SUPER; ib = IB * EC_SUPER + w_in % EC_SUPER;
        if (jb > ib || ib >= nb) return;
    } else {
        // small point sets (the tables are cache-resident anyway): row-major over the upper triangle of tiles, no idle workgroups
        const int64_t t = blockIdx.x;
        jb = (int)((2.0 * nb + 1.0 - sqrt((2.0 * nb + 1.0) * (2.0 * nb + 1.0) - 8.0 * (double)t)) * 0.5);
        while ((int64_t)jb * nb - (int64_t)jb * (jb - 1) / 2 > t) --jb;
        while ((int64_t)(jb + 1) * nb - (int64_t)(jb + 1) * jb / 2 <= t) ++jb;
        ib = jb + (int)(t - ((int64_t)jb * nb - (int64_t)jb * (jb - 1) / 2));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // ---- row side: wave handles 16 rows j, lane = column i
    const int64_t i_r = (int64_t)ib * 64 + lane;
#pragma unroll 4
    for (int q = 0; q < 16; ++q) {
        const int64_t j = (int64_t)jb * 64 + wave * 16 + q;
        int32_t pos = 0;
        if (j < nx) {
            const uint64_t bits = K[j * kw + ib];
            if (i_r > j && ((bits >> lane) & 1ull))
                pos = (int32_t)(rowstart[j] + ((int64_t)pref[j * kw + ib] + __popcll(bits & ((1ull << lane) - 1ull)) - low[j]));
        }
        tp[wave * 16 + q][lane] = pos;
    }
    __syncthreads();
    // ---- column side: wave handles 16 columns i, lane = row j; the columns' table words lane-parallel first (lane q reads column q's)
    const int64_t j_w = (int64_t)jb * 64 + lane;
    uint32_t cb_lo = 0, cb_hi = 0, cd_lo = 0, cd_hi = 0;
    {
        const int64_t i = (int64_t)ib * 64 + wave * 16 + (lane & 15);
        if (i < nx) {
            const uint64_t bits = K[i * kw + jb];   // symmetric bitmap: bit j of row i <=> pair (j, i) kept
            const int64_t dst0 = Iptr[i] + (int64_t)pref[i * kw + jb];
            cb_lo = (uint32_t)bits; cb_hi = (uint32_t)(bits >> 32);
            cd_lo = (uint32_t)dst0; cd_hi = (uint32_t)((uint64_t)dst0 >> 32);
        }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int il = wave * 16 + q;
        const int64_t i = (int64_t)ib * 64 + il;
        const uint64_t bits = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)cb_hi, q) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)cb_lo, q);
        if (j_w < i && ((bits >> lane) & 1ull)) {   // (columns >= nx hold no bits)
            const int64_t dst0 = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)cd_hi, q) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)cd_lo, q));
            ann_store(Iidx + dst0 + __popcll(bits & ((1ull << lane) - 1ull)), tp[lane][il], stream);
        }
    }
}

__global__ void k_min_i32(const int32_t *__restrict__ v, int64_t n, int32_t *__restrict__ out)
{
    // single block
    int m = 0x7fffffff;
    for (int64_t t = threadIdx.x; t < n; t += blockDim.x) m = min(m, v[t]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = min(m, __shfl_xor(m, off));
    __shared__ int s[16];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = min(m, s[w]);
        out[0] = m;
        out[1] = 0; out[2] = 0; out[3] = 0;   // (the 64-bit counter k_count_anchor_pairs adds to lives in out[2..3])
    }
}

// Candidate pairs with an anchor endpoint: the pairs compute_features marks computed (annchor.py:286-289), counted here so
// that the number of not-computed pairs after the feature stage rides with this stage's download instead of costing the
// first sampling step a host wait.  Thread per (anchor slot, point); a pair of two anchors counts from its smaller point, a
// repeated anchor from its last slot.
__global__ __launch_bounds__(256) void k_count_anchor_pairs(const int32_t *__restrict__ A, int nA, const int32_t *__restrict__ anchorRank,
                                                           int64_t nx, const uint64_t *__restrict__ K, int kw,
                                                           unsigned long long *__restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool hit = false;
    if (t < (int64_t)nA * nx) {
        const int r = (int)(t / nx);
        const int64_t a = A[r], o = t % nx;
        if (anchorRank[a] == r && a != o && !(anchorRank[o] >= 0 && o < a)) {
            const int64_t i = a < o ? a : o, j = a < o ? o : a;
            hit = (K[i * kw + (j >> 6)] >> (j & 63)) & 1ull;
        }
    }
    const unsigned long long m = __ballot(hit);
    __shared__ uint32_t s;
    if (threadIdx.x == 0) s = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&s, (uint32_t)__popcll(m));
    __syncthreads();
    if (threadIdx.x == 0 && s) atomicAdd(out, (unsigned long long)s);
}

extern "C" int annchor_build_locality(annchor_ctx *c, int32_t locality, int32_t loc_thresh, int32_t loc_min,
                                      int64_t *n_pairs, int64_t *min_row_len)
{
    if (!c || !n_pairs || !min_row_len) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->na > 0, ANNCHOR_EINVAL, "anchors not set");
    ANN_REQUIRE(c, locality >= 1, ANNCHOR_EINVAL, "locality must be >= 1");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    const int64_t nx = c->nx;
    const int kw = (int)((nx + 63) / 64);
    ANN_REQUIRE(c, (double)nx * kw * 12.0 < 64e9, ANNCHOR_ELIMIT,
                "nx=%lld is too large for the pair-list form (use the streamed form)", (long long)nx);
    if (locality > c->na) locality = c->na;
    const int nw = c->sid_nw = ann_sid_words(c->na);
    ANN_TRY(ann_reserve(c, c->sid, sizeof(uint64_t) * (size_t)nx * nw));
    ANN_TRY(ann_reserve(c, c->cA, sizeof(int32_t) * (size_t)nx));
    ANN_TRY(ann_reserve(c, c->thr, sizeof(int32_t) * (size_t)nx));
    ANN_TRY(ann_reserve(c, c->Kbits, sizeof(uint64_t) * (size_t)nx * kw));
    ANN_TRY(ann_reserve(c, c->Kpref, sizeof(uint32_t) * (size_t)nx * kw));
    ANN_TRY(ann_reserve(c, c->deg, sizeof(int32_t) * (size_t)nx));
    ANN_TRY(ann_reserve(c, c->low, sizeof(int32_t) * (size_t)nx));
    ANN_TRY(ann_reserve(c, c->tmp1, sizeof(int32_t) * (size_t)nx));  // up[]
    ANN_TRY(ann_reserve(c, c->rowstart, sizeof(int64_t) * (size_t)(nx + 1)));
    ANN_TRY(ann_reserve(c, c->Iptr, sizeof(int64_t) * (size_t)(nx + 1)));
    ANN_TRY(ann_reserve(c, c->tmp2, sizeof(int32_t) * 4));
    ANN_CHECK_HIP(c, hipEventRecord(c->call_a, c->stream));
    {
        ProfScope ps(c, "locality_sid", (double)nx * (c->na * 8.0 + 12));
#define SID_CALL(NW) k_sid<NW><<<ann_blocks(nx, 256), 256, 0, c->stream>>>(c->Dt.as<double>(), nx, c->na, locality, c->sid.as<uint64_t>(), c->cA.as<int32_t>())
        ANN_SID_DISPATCH(nw, SID_CALL);
#undef SID_CALL
    }
    {
        ProfScope ps(c, "locality_keep_bitmap", (double)nx * kw * 12.0);
        static const bool hist_form = getenv("ANNCHOR_LOC_THRESH_HIST") != nullptr;   // tests: the histogram form at any threshold
        if (loc_thresh >= 1 && loc_thresh <= 8 && !hist_form) {
            switch (loc_thresh) {
#define LT_NW(NW) k_loc_thresh_small<LT_T, NW><<<(int)nx, LOC_THREADS, 0, c->stream>>>(c->sid.as<uint64_t>(), nx, loc_min, c->thr.as<int32_t>())
#define LT_CASE(T) case T: { constexpr int LT_T = T; ANN_SID_DISPATCH(nw, LT_NW); } break;
                LT_CASE(1) LT_CASE(2) LT_CASE(3) LT_CASE(4) LT_CASE(5) LT_CASE(6) LT_CASE(7) LT_CASE(8)
#undef LT_CASE
#undef LT_NW
            }
        } else {
#define LTH_CALL(NW) k_loc_thresh<NW><<<(int)nx, LOC_THREADS, 0, c->stream>>>(c->sid.as<uint64_t>(), nx, loc_thresh, loc_min, c->thr.as<int32_t>())
            ANN_SID_DISPATCH(nw, LTH_CALL);
#undef LTH_CALL
        }
        static const long long cols_min = getenv("ANNCHOR_KEEP_COLS_MIN") ? atoll(getenv("ANNCHOR_KEEP_COLS_MIN")) : (1ll << 20);   // bitmap words (N = 16 000: 0.60 -> 0.27 ms)
        if (nx >= 64 && nx * kw >= cols_min) {
#define KBC_CALL(NW) k_keep_bits_cols<NW><<<(unsigned)((((nx + 63) / 64) * kw + 3) / 4), 256, 0, c->stream>>>(c->sid.as<uint64_t>(), c->thr.as<int32_t>(), nx, kw, c->Kbits.as<uint64_t>())
            ANN_SID_DISPATCH(nw, KBC_CALL);
#undef KBC_CALL
        } else {
#define KB_CALL(NW) k_keep_bits<NW><<<(int)std::min<int64_t>(ann_blocks(nx * kw * 64, 256), (int64_t)c->prop.multiProcessorCount * 32), 256, 0, c->stream>>>(c->sid.as<uint64_t>(), c->thr.as<int32_t>(), nx, kw, c->Kbits.as<uint64_t>())
            ANN_SID_DISPATCH(nw, KB_CALL);
#undef KB_CALL
        }
        k_row_prefix<<<(int)nx, LOC_THREADS, 0, c->stream>>>(c->Kbits.as<uint64_t>(), nx, kw, c->Kpref.as<uint32_t>(),
                                                            c->deg.as<int32_t>(), c->low.as<int32_t>(),
                                                            c->tmp1.as<int32_t>());
        k_min_i32<<<1, 1024, 0, c->stream>>>(c->deg.as<int32_t>(), nx, c->tmp2.as<int32_t>());   // (also zeroes out[1..3]: the counter below)
        if (c->nA > 0 && c->anchorRank.p)
            k_count_anchor_pairs<<<ann_blocks((int64_t)c->nA * nx, 256), 256, 0, c->stream>>>(
                c->A.as<int32_t>(), c->nA, c->anchorRank.as<int32_t>(), nx, c->Kbits.as<uint64_t>(), kw,
                reinterpret_cast<unsigned long long *>(c->tmp2.as<int32_t>() + 2));
    }
    ANN_TRY(ann_exclusive_scan_i32_to_i64(c, c->tmp1.as<int32_t>(), c->rowstart.as<int64_t>(), nx));
    ANN_TRY(ann_exclusive_scan_i32_to_i64(c, c->deg.as<int32_t>(), c->Iptr.as<int64_t>(), nx));
    int64_t n = 0;
    struct { int32_t mn, pad; long long anchor_pairs; } st = {0, 0, 0};
    ANN_TRY(ann_d2h2(c, &n, c->rowstart.as<int64_t>() + nx, sizeof n, &st, c->tmp2.p, sizeof st));
    const int32_t mn = st.mn;
    c->n_unc_after_features = (c->nA > 0 && c->anchorRank.p) ? n - st.anchor_pairs : -1;
    ANN_REQUIRE(c, n < (1ll << 30), ANNCHOR_ELIMIT,
                "%lld candidate pairs exceed the pair-list limit of 2^30 (int32 positions): raise loc_thresh / lower locality", (long long)n);
    if (n > (1ll << 27)) {   // ~130 B per pair over the stages that follow: refuse here, not in the middle of a fit
        size_t fr = 0, tot = 0;
        ANN_CHECK_HIP(c, hipMemGetInfo(&fr, &tot));
        int64_t parked = 0;   // blocks this process keeps in its pool count as free (ann_dev_alloc hands them out or flushes them)
        (void)annchor_parked_bytes(c->device, &parked);
        const double have = (double)fr + (double)parked + (double)c->ij.cap + (double)c->Iidx.cap + (double)c->lb.cap + (double)c->ub.cap + (double)c->dad.cap + (double)c->RA.cap + (double)c->prob.cap;
        ANN_REQUIRE(c, (double)n * 130.0 <= have, ANNCHOR_ELIMIT,
                    "%lld candidate pairs need ~%.0f GB of device memory, %.0f GB are free: raise loc_thresh / lower locality",
                    (long long)n, (double)n * 130.0 / 1e9, have / 1e9);
    }
    ANN_TRY(ann_reserve(c, c->ij, sizeof(int2) * (size_t)n));
    ANN_TRY(ann_reserve(c, c->Iidx, sizeof(int32_t) * 2 * (size_t)n));
    {
        ProfScope ps(c, "locality_emit_pairs", (double)n * 16 + (double)nx * kw * 12.0);
        static const long long tiled_min = getenv("ANNCHOR_EMIT_TILED_MIN") ? atoll(getenv("ANNCHOR_EMIT_TILED_MIN")) : 0;
        const int stream_hint = n >= ANN_STREAM_MIN_PAIRS, tiled = n >= tiled_min;
        static const long long run_min = getenv("ANNCHOR_EMIT_RUN_MIN") ? atoll(getenv("ANNCHOR_EMIT_RUN_MIN")) : (1ll << 20);   // bitmap words (N = 16 000: 1.20 -> 0.71 ms)
        if (tiled && nx * kw >= run_min)
            k_emit_rows_run<<<(int)std::min<int64_t>(ann_blocks(nx * kw, 256), (int64_t)c->prop.multiProcessorCount * 32), 256, 0, c->stream>>>(
                c->Kbits.as<uint64_t>(), c->Kpref.as<uint32_t>(), nx, kw, c->low.as<int32_t>(), c->rowstart.as<int64_t>(),
                c->Iptr.as<int64_t>(), c->ij.as<int2>(), c->Iidx.as<int32_t>(), stream_hint);
        else
        k_emit_pairs<<<(int)std::min<int64_t>(ann_blocks(nx * kw * 64, 256), (int64_t)c->prop.multiProcessorCount * 64), 256, 0, c->stream>>>(
            c->Kbits.as<uint64_t>(), c->Kpref.as<uint32_t>(), nx, kw, c->low.as<int32_t>(), c->rowstart.as<int64_t>(),
            c->Iptr.as<int64_t>(), c->ij.as<int2>(), c->Iidx.as<int32_t>(), stream_hint, tiled);
        if (tiled) {
            const int64_t S = (kw + EC_SUPER - 1) / EC_SUPER;
            static const int super_min = getenv("ANNCHOR_EMIT_SUPER_MIN") ? atoi(getenv("ANNCHOR_EMIT_SUPER_MIN")) : 4 * EC_SUPER;   // bitmap words per row
            const int supertiles = kw >= super_min;
            const int64_t grid = supertiles ? ((S * (S + 1) / 2 + 7) / 8) * 8 * EC_SUPER * EC_SUPER : (int64_t)kw * (kw + 1) / 2;
            k_emit_cols<<<(unsigned)grid, 256, 0, c->stream>>>(
                c->Kbits.as<uint64_t>(), c->Kpref.as<uint32_t>(), kw, c->low.as<int32_t>(), c->rowstart.as<int64_t>(),
                c->Iptr.as<int64_t>(), nx, c->Iidx.as<int32_t>(), stream_hint, supertiles);
        }
    }
    ANN_CHECK_HIP(c, hipEventRecord(c->call_b, c->stream));
    c->call_timed = true;
    ANN_CHECK_HIP(c, hipGetLastError());
    c->n = n;
    c->have_bitmap = true;
    c->have_features = c->have_RA = false; c->sel_prepared = false;
    *n_pairs = n;
    *min_row_len = mn;
    return ANNCHOR_OK;
}

// ------------------------------------------------------------------ query locality
// get_query_locality + the pair list of get_query_features (reference
// annchor/query_functions.py:18-62): the data set bound to the context is X (rows
// [0, nx_base)) followed by the query set Q (rows [nx_base, nx)); candidates of query j are
// the points i < nx_base sharing >= loc_thresh of their `locality` nearest anchors with it
// (no loc_min widening on the query side).  Pairs are (i, nx_base + j), sorted by (j, i): the
// entries of a query row are contiguous, so I is the identity over each row's range and
// rows of X are empty.
template <int NW>
__global__ __launch_bounds__(LOC_THREADS) void k_qloc_count(const uint64_t *__restrict__ sid, int64_t nxb, int loc_thresh,
                                                           int32_t *__restrict__ cnt)
{
    __shared__ uint32_t acc;
    if (threadIdx.x == 0) acc = 0;
    __syncthreads();
    const int64_t q = nxb + blockIdx.x;
    const Sid<NW> mq = sid_ld<NW>(sid, q);
    uint32_t s = 0;
    for (int64_t i = threadIdx.x; i < nxb; i += blockDim.x) s += sid_common<NW>(mq, sid_ld<NW>(sid, i)) >= loc_thresh;
    if (s) atomicAdd(&acc, s);
    __syncthreads();
    if (threadIdx.x == 0) cnt[q] = (int32_t)acc;
}

template <int NW>
__global__ __launch_bounds__(LOC_THREADS) void k_qloc_emit(const uint64_t *__restrict__ sid, int64_t nxb, int loc_thresh,
                                                          const int64_t *__restrict__ Iptr, int2 *__restrict__ ij,
                                                          int32_t *__restrict__ Iidx)
{
    __shared__ uint32_t wsum[LOC_THREADS / 64];
    __shared__ uint32_t run_s;
    const int64_t q = nxb + blockIdx.x;
    const Sid<NW> mq = sid_ld<NW>(sid, q);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) run_s = 0;
    __syncthreads();
    const int64_t base = Iptr[q];
    for (int64_t b0 = 0; b0 < nxb; b0 += LOC_THREADS) {
        const int64_t i = b0 + threadIdx.x;
        const uint32_t f = (i < nxb && sid_common<NW>(mq, sid_ld<NW>(sid, min(i, nxb - 1))) >= loc_thresh) ? 1u : 0u;
        const unsigned long long m = __ballot(f);
        if (lane == 0) wsum[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t pre = run_s;
        for (int w = 0; w < wave; ++w) pre += wsum[w];
        if (f) {
            const int64_t pos = base + pre + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            ij[pos] = make_int2((int)i, (int)q);
            Iidx[pos] = (int32_t)pos;
        }
        __syncthreads();
        if (threadIdx.x == 0) run_s += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
}

__global__ void k_zero_i32(int32_t *p, int64_t n)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) p[t] = 0;
}

extern "C" int annchor_build_query_locality(annchor_ctx *c, int64_t nx_base, int32_t locality, int32_t loc_thresh,
                                            int64_t *n_pairs, int64_t *min_row_len)
{
    if (!c || !n_pairs || !min_row_len) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->na > 0, ANNCHOR_EINVAL, "anchors not set");
    ANN_REQUIRE(c, nx_base >= 1 && nx_base < c->nx, ANNCHOR_EINVAL, "nx_base=%lld outside (0, nx)", (long long)nx_base);
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    const int64_t nx = c->nx, nq = nx - nx_base;
    if (locality > c->na) locality = c->na;
    const int nw = c->sid_nw = ann_sid_words(c->na);
    ANN_TRY(ann_reserve(c, c->sid, sizeof(uint64_t) * (size_t)nx * nw));
    ANN_TRY(ann_reserve(c, c->cA, sizeof(int32_t) * (size_t)nx));
    ANN_TRY(ann_reserve(c, c->deg, sizeof(int32_t) * (size_t)nx));
    ANN_TRY(ann_reserve(c, c->Iptr, sizeof(int64_t) * (size_t)(nx + 1)));
    ANN_TRY(ann_reserve(c, c->tmp2, sizeof(int32_t) * 4));
#define SID_CALL(NW) k_sid<NW><<<ann_blocks(nx, 256), 256, 0, c->stream>>>(c->Dt.as<double>(), nx, c->na, locality, c->sid.as<uint64_t>(), c->cA.as<int32_t>())
    ANN_SID_DISPATCH(nw, SID_CALL);
#undef SID_CALL
    k_zero_i32<<<ann_blocks(nx, 256), 256, 0, c->stream>>>(c->deg.as<int32_t>(), nx);
    {
        ProfScope ps(c, "query_locality", (double)nq * nx_base * 8.0);
#define QC_CALL(NW) k_qloc_count<NW><<<(int)nq, LOC_THREADS, 0, c->stream>>>(c->sid.as<uint64_t>(), nx_base, loc_thresh, c->deg.as<int32_t>())
        ANN_SID_DISPATCH(nw, QC_CALL);
#undef QC_CALL
    }
    k_min_i32<<<1, 1024, 0, c->stream>>>(c->deg.as<int32_t>() + nx_base, nq, c->tmp2.as<int32_t>());
    ANN_TRY(ann_exclusive_scan_i32_to_i64(c, c->deg.as<int32_t>(), c->Iptr.as<int64_t>(), nx));
    int64_t n = 0;
    int32_t mn = 0;
    ANN_TRY(ann_d2h(c, &n, c->Iptr.as<int64_t>() + nx, sizeof n));
    ANN_TRY(ann_d2h(c, &mn, c->tmp2.p, sizeof mn));
    ANN_REQUIRE(c, n < (1ll << 30), ANNCHOR_ELIMIT, "%lld query pairs exceed the pair-list limit", (long long)n);
    ANN_TRY(ann_reserve(c, c->ij, sizeof(int2) * (size_t)(n + 1)));
    ANN_TRY(ann_reserve(c, c->Iidx, sizeof(int32_t) * (size_t)(n + 1)));
    if (n > 0) {
#define QE_CALL(NW) k_qloc_emit<NW><<<(int)nq, LOC_THREADS, 0, c->stream>>>(c->sid.as<uint64_t>(), nx_base, loc_thresh, c->Iptr.as<int64_t>(), c->ij.as<int2>(), c->Iidx.as<int32_t>())
        ANN_SID_DISPATCH(nw, QE_CALL);
#undef QE_CALL
    }
    ANN_CHECK_HIP(c, hipGetLastError());
    c->n_unc_after_features = -1;
    c->n = n;
    c->have_bitmap = false;   // query form: rows are contiguous already, no bitmap
    c->have_features = c->have_RA = false; c->sel_prepared = false;
    *n_pairs = n;
    *min_row_len = mn;
    return ANNCHOR_OK;
}
