// rowsel.h -- block-level selection helpers shared by select.hip and refine.hip.
// One 256-thread block owns one row of the CSR index I; the row's float64 keys are
// mapped to order-preserving uint64 and the k-th smallest is found by an MSB-first
// 8-bit radix descent on an LDS histogram (exact, tie-safe, no sort).
#pragma once
#include "common.h"

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
