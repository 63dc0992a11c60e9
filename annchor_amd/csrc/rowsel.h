// rowsel.h -- block-level selection helpers shared by select.hip and refine.hip.
// One 256-thread block owns one row of the CSR index I; the row's float64 keys are
// mapped to order-preserving uint64 and the k-th smallest is found by an MSB-first
// 8-bit radix descent on an LDS histogram (exact, tie-safe, no sort).
#pragma once
#include "common.h"
#include <mutex>
#include <vector>

#define ROW_THREADS 256

// Row owned by this workgroup.  Workgroups are dispatched round-robin over the 8 XCDs
// (block b -> XCD b % 8, a speed heuristic only): give each XCD a contiguous band of rows
// so that the column-like half of a row's gather (pairs (j, i), j < i, one 8-byte value per
// 64-byte line) is shared through that XCD's own L2 by the neighbouring rows.
__device__ __forceinline__ int64_t row_of_block(int64_t nrows)
{
    const int64_t b = blockIdx.x, q = nrows >> 3, r = nrows & 7, x = b & 7, y = b >> 3;
    // bijective for any nrows: XCD x owns q (+1 if x < r) rows
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
}
// Row keys live in DYNAMIC shared memory sized per launch: rows up to `cap` entries are gathered
// once and every later pass (byte census, radix passes, collection) reads LDS; longer rows fall
// back to re-gathering.  cap = longest possible row, bounded by what one workgroup may hold.
#define ROW_LDS_LIMIT (156 * 1024)   // dynamic LDS a row kernel may ask for (static shared stays < 4 KiB)
static inline int row_lds_cap(int64_t max_row_len, size_t other_dyn_bytes)
{
    int64_t cap = (int64_t)((ROW_LDS_LIMIT - other_dyn_bytes) / 8);
    if (cap > max_row_len) cap = max_row_len;
    if (cap < 1) cap = 1;
    return (int)((cap + 1) & ~1ll);   // keep what follows 16-byte aligned
}
template <typename K> static inline int row_lds_prepare(annchor_ctx *c, K kernel, size_t dyn_bytes)
{
    if (dyn_bytes > 64 * 1024)
        ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_bytes));
    return ANNCHOR_OK;
}

// The LDS copy of the row serves the fallback passes only (the sampled fast path streams the row once), and it can cost a
// launch its residency: 1600 rows of 1599 entries are 28 KB per workgroup -- five per CU, 1280 of the 1600 resident, a second
// round of workgroups for the rest (row_kth 38 us, guarantee_nmin lists 74 us, the graph 73 us at C2).  When the launch's
// workgroups are not all resident with the copy and more of them are without it, the copy is dropped (cap = 2: a fallback row
// re-reads global memory, as the streamed rows of the large lists always do).  ANNCHOR_ROW_LDS_COPY=1 keeps it.
template <typename K> static inline int row_pick_cap(annchor_ctx *c, K kernel, int64_t nblocks, int64_t max_row_len, size_t other_dyn_bytes, int *cap)
{
    const int full = row_lds_cap(max_row_len, other_dyn_bytes);
    *cap = full;
    const char *e_keep = getenv("ANNCHOR_ROW_LDS_COPY");   // (read per call: A/B runs inside one process)
    const bool keep = e_keep && atoi(e_keep) == 1;
    if (keep || full <= 2) return ANNCHOR_OK;
    struct Memo { const void *k; size_t dyn; int per_cu; };
    static std::vector<Memo> memo;
    static std::mutex mu;
    auto resident = [&](size_t dyn, int *out) -> int {
        {
            std::lock_guard<std::mutex> lk(mu);
            for (auto &m : memo) if (m.k == (const void *)kernel && m.dyn == dyn) { *out = m.per_cu; return ANNCHOR_OK; }
        }
        ANN_TRY(row_lds_prepare(c, kernel, dyn));
        int per_cu = 0;
        ANN_CHECK_HIP(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, ROW_THREADS, dyn));
        std::lock_guard<std::mutex> lk(mu);
        memo.push_back({(const void *)kernel, dyn, per_cu});
        *out = per_cu;
        return ANNCHOR_OK;
    };
    int with_copy = 0, without = 0;
    ANN_TRY(resident((size_t)full * 8 + other_dyn_bytes, &with_copy));
    if ((int64_t)with_copy * c->prop.multiProcessorCount >= nblocks) return ANNCHOR_OK;
    ANN_TRY(resident((size_t)2 * 8 + other_dyn_bytes, &without));
    if (without > with_copy) *cap = 2;
    return ANNCHOR_OK;
}

struct RowSelShared {
    uint32_t hist[256];
    uint32_t wsum[ROW_THREADS / 64];
    uint64_t prefix;
    uint32_t k;
    uint32_t digit;
    unsigned long long diff;  // OR over the row of (key ^ key[0]): bytes that are zero here are shared by all keys
};

// Exclusive prefix (in thread order) of `v` over the block; *total = block sum.
__device__ __forceinline__ uint32_t row_block_scan(uint32_t v, uint32_t *wsum, uint32_t *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    __syncthreads();
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < ROW_THREADS / 64; ++w) {
        if (w < wave) base += wsum[w];
        tot += wsum[w];
    }
    *total = tot;
    return base + inc - v;
}

// k-th smallest (0-based) key of the row.  KeyFn(s) returns the uint64 key of row
// slot s (s in [0, len)).  All threads return the same key.  k is clamped to len-1.
template <typename KeyFn> __device__ uint64_t row_kth_key(RowSelShared &sh, int len, uint32_t k, KeyFn key)
{
    if (k >= (uint32_t)len) k = (uint32_t)len - 1;
    const uint64_t key0 = key(0);
    if (threadIdx.x == 0) { sh.prefix = 0; sh.k = k; sh.diff = 0; }
    __syncthreads();
    {
        // which key bytes vary at all inside this row?  (integer-valued or narrowly ranged
        // distances share most of their 8 bytes: those radix passes are skipped)
        uint64_t d = 0;
        for (int s = threadIdx.x; s < len; s += ROW_THREADS) d |= key(s) ^ key0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) d |= __shfl_xor(d, off);
        if ((threadIdx.x & 63) == 0 && d) atomicOr(&sh.diff, (unsigned long long)d);
        __syncthreads();
    }
    const uint64_t diff = sh.diff;
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 56 - 8 * pass;
        if (((diff >> shift) & 0xffull) == 0) {   // uniform byte: nothing to select on
            if (threadIdx.x == 0) sh.prefix |= key0 & (0xffull << shift);
            __syncthreads();
            continue;
        }
        const uint64_t himask = pass == 0 ? 0ull : (~0ull << (shift + 8));
        sh.hist[threadIdx.x] = 0;  // ROW_THREADS == 256
        __syncthreads();
        const uint64_t pre = sh.prefix;
        for (int s = threadIdx.x; s < len; s += ROW_THREADS) {
            const uint64_t kk = key(s);
            if ((kk & himask) == pre) atomicAdd(&sh.hist[(uint32_t)(kk >> shift) & 0xffu], 1u);
        }
        __syncthreads();
        const uint32_t h = sh.hist[threadIdx.x];
        uint32_t tot;
        const uint32_t ex = row_block_scan(h, sh.wsum, &tot);
        const uint32_t kk = sh.k;
        __syncthreads();
        if (h != 0 && kk >= ex && kk < ex + h) {
            sh.digit = threadIdx.x;
            sh.k = kk - ex;
            sh.prefix = pre | ((uint64_t)threadIdx.x << shift);
        }
        __syncthreads();
    }
    return sh.prefix;
}

// ---- fast path of the row selections: candidates below a sampled threshold.
// The radix descent above walks the whole row once per key byte that varies (float64 distances:
// seven or eight passes with LDS histogram atomics).  The rows here want their ~16-40 smallest of up
// to ~10^4 entries, so: take 256 evenly spaced entries, let t0 be the sample's r-th smallest key
// with r three times what the wanted count corresponds to, stream the row ONCE keeping every entry
// with key <= t0 (a few dozen), and finish exactly among those.  If fewer than `want` or more than
// ROWC_CAP entries qualify (ties, adversarial order) the caller falls back to the radix descent.
#define ROWC_CAP 1024
#ifndef ROWC_U
#define ROWC_U 8
#endif
struct RowCand {
    uint64_t key[ROWC_CAP];
    int32_t slot[ROWC_CAP];
    uint64_t samp[ROW_THREADS];
    uint64_t t0;
    uint32_t count;
};

// Collects the ELIGIBLE entries with key <= t0 into rc (any order); returns their number, or -1 when
// it exceeds ROWC_CAP.  Uniform result.  key(s) -> uint64 key of entry s; elig(s) -> bool (entries that
// may be selected at all, e.g. the not-computed ones); visit(s, key, eligible) is called once for every
// entry, so callers count / reduce what they need in the same pass.
template <typename KeyFn, typename EligFn, typename Visit>
__device__ int row_candidates(RowCand &rc, int len, int want, KeyFn key, EligFn elig, Visit visit)
{
    if (threadIdx.x == 0) { rc.count = 0; rc.t0 = ~0ull; }
    __syncthreads();
    if (len > ROWC_CAP) {
        const int s0 = (int)(((int64_t)threadIdx.x * len) / ROW_THREADS);
        const uint64_t mine = elig(s0) ? key(s0) : ~0ull;   // ineligible entries sort last in the sample
        rc.samp[threadIdx.x] = mine;
        __syncthreads();
        int r = (int)((3ll * want * ROW_THREADS + len - 1) / len) + 3;
        if (r < ROW_THREADS) {
            int less = 0;
            for (int o = 0; o < ROW_THREADS; ++o) {
                const uint64_t ko = rc.samp[o];
                less += (ko < mine) || (ko == mine && o < (int)threadIdx.x);
            }
            if (less == r) rc.t0 = mine;   // ranks are a permutation: exactly one thread
        }
        __syncthreads();
    }
    const uint64_t t0 = rc.t0;
    // ROWC_U entries per thread per trip, their loads issued together (clamped index: no branch around a
    // load): a workgroup keeps ROWC_U x 2 KB of the row in flight
    for (int s0 = threadIdx.x; s0 < len; s0 += ROWC_U * ROW_THREADS) {
        uint64_t kk[ROWC_U];
        bool el[ROWC_U];
#pragma unroll
        for (int u = 0; u < ROWC_U; ++u) {
            const int s = min(s0 + u * ROW_THREADS, len - 1);
            kk[u] = key(s);
            el[u] = elig(s);
        }
#pragma unroll
        for (int u = 0; u < ROWC_U; ++u) {
            const int s = s0 + u * ROW_THREADS;
            if (s < len) {
                visit(s, kk[u], el[u]);
                if (el[u] && kk[u] <= t0) {
                    const uint32_t o = atomicAdd(&rc.count, 1u);
                    if (o < ROWC_CAP) { rc.key[o] = kk[u]; rc.slot[o] = s; }
                }
            }
        }
    }
    __syncthreads();
    const uint32_t c = rc.count;
    return c > ROWC_CAP ? -1 : (int)c;
}

// Second cut, inside LDS: the callers rank their candidates against each other (c^2 / 256 LDS reads per
// thread), and a 16 000-entry row leaves ~400 of them for the 15-40 that are wanted.  The same sampling
// idea once more: 64 evenly spaced candidates, t1 = their r2-th smallest key with r2 three times the wanted
// share, keep the candidates with key <= t1 (all ties included).  Every candidate outside the kept set has
// a larger key than each kept one, so if at least `want` are kept the `want` smallest by (key, slot) are
// among them.  Returns the new count (uniform); < want means the cut was too tight (caller falls back).
__device__ __forceinline__ int row_cand_shrink(RowCand &rc, int c, int want, int min_c)
{
    if (c <= min_c || c > ROWC_CAP) return c;
    __syncthreads();
    if (threadIdx.x < 64) rc.samp[threadIdx.x] = rc.key[(int)(((int64_t)threadIdx.x * c) >> 6)];
    if (threadIdx.x == 0) rc.t0 = ~0ull;
    __syncthreads();
    int r2 = (int)((3ll * want * 64 + c - 1) / c) + 2;
    if (threadIdx.x < 64 && r2 < 64) {
        const uint64_t mine = rc.samp[threadIdx.x];
        int less = 0;
        for (int o = 0; o < 64; ++o) {
            const uint64_t ko = rc.samp[o];
            less += (ko < mine) || (ko == mine && o < (int)threadIdx.x);
        }
        if (less == r2) rc.t0 = mine;
    }
    __syncthreads();
    const uint64_t t1 = rc.t0;
    uint64_t kk[ROWC_CAP / ROW_THREADS];
    int32_t ss[ROWC_CAP / ROW_THREADS];
#pragma unroll
    for (int u = 0; u < ROWC_CAP / ROW_THREADS; ++u) {
        const int e = threadIdx.x + u * ROW_THREADS;
        kk[u] = e < c ? rc.key[e] : ~0ull;
        ss[u] = e < c ? rc.slot[e] : 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) rc.count = 0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < ROWC_CAP / ROW_THREADS; ++u) {
        const int e = threadIdx.x + u * ROW_THREADS;
        if (e < c && kk[u] <= t1) {
            const uint32_t o = atomicAdd(&rc.count, 1u);
            rc.key[o] = kk[u]; rc.slot[o] = ss[u];
        }
    }
    __syncthreads();
    return (int)rc.count;
}

// Where a row kernel takes the values of row i from.  Row i of the CSR index lists first its
// "column-like" entries -- pairs (j, i), j < i: one 8-byte value in each of `low` different rows of the
// pair list, i.e. one 64-byte line per value -- then its "row-like" entries (i, j), j > i, which are
// contiguous in the pair list.  For large pair lists the column-like halves are first transposed into
// the column-ordered copy T (ann_transpose_columns: tiled, coalesced on both sides), so that both
// halves stream; small lists (everything L2 / MALL resident) are gathered through Iidx as before.
struct RowSrc {
    const double *RA;
    const uint8_t *ncm;
    const int32_t *Iidx;
    const double *T;        // nullptr: gather through Iidx
    const uint8_t *Tm;
    const int64_t *rowstart;
    const int32_t *low;
    int shrink_min;         // candidate sets larger than this get the second cut (row_cand_shrink)
};
#define ROWC_SHRINK_MIN 128

struct RowView {
    const double *RA, *Tv;
    const uint8_t *ncm, *Tm;
    const int32_t *idx;
    int low;
    bool direct;
    __device__ __forceinline__ double val(int s) const
    {
        if (!direct) return RA[idx[s]];
        const double *p = s < low ? Tv + s : RA + (s - low);   // select the address, one load
        return *p;
    }
    __device__ __forceinline__ bool unc(int s) const
    {
#ifdef ROW_NO_MASK   // timing experiment only: what the byte-mask loads cost
        return (s & 15) != 0;
#endif
        if (!direct) return ncm[idx[s]] != 0;
        const uint8_t *p = s < low ? Tm + s : ncm + (s - low);
        return *p != 0;
    }
};

__device__ __forceinline__ RowView row_view(const RowSrc &src, int64_t i, int64_t b /* Iptr[i] */)
{
    RowView v;
    v.idx = src.Iidx + b;
    v.direct = src.T != nullptr;
    v.low = 0;
    v.RA = src.RA; v.ncm = src.ncm; v.Tv = nullptr; v.Tm = nullptr;
    if (v.direct) {
        const int64_t rs = src.rowstart[i];
        v.low = src.low[i];
        v.Tv = src.T + (b - rs);     // colbase(i) = Iptr[i] - rowstart[i]
        v.Tm = src.Tm + (b - rs);
        v.RA = src.RA + rs;
        v.ncm = src.ncm + rs;
    }
    return v;
}

int ann_transpose_columns(annchor_ctx *c, RowSrc *src, bool with_values = true);   // refine.hip: fills T / Tm when the list is large, sets *src
