// refine.hip -- bound tightening from computed distances, and the final k-NN graph.
//
// Replaces Annchor.update_anchor_points + update_bounds / get_bounds_alt (reference
// annchor/annchor.py:475-512, annchor/utils.py:304-352) and Annchor.get_ann + get_nn
// (annchor/annchor.py:514-530, annchor/utils.py:383-429).
//
// update_bounds: for every lookahead pair (i, j), over the points c whose distance
// to BOTH i and j is already computed: ub = min(ub, d_ic + d_jc),
// lb = max(lb, |d_ic - d_jc|).  The reference builds per-point sorted Python lists
// and merges them; here the computed neighbours of every point are compacted into a
// CSR (already sorted, because I[] is ordered by the other endpoint), and one
// wavefront per pair intersects the two lists by binary search.  All chunks are
// processed (the reference's 10 s wall-clock cut-off is a speed guard, not semantics).
#include "common.h"
#include "rowsel.h"

// ---- computed-neighbour CSR
__global__ __launch_bounds__(ROW_THREADS) void k_comp_count(const int64_t *__restrict__ Iptr, RowSrc src, int32_t *__restrict__ cnt)
{
    __shared__ uint32_t acc;
    if (threadIdx.x == 0) acc = 0;
    __syncthreads();
    const int64_t i = row_of_block(gridDim.x), b = Iptr[i];
    const int len = (int)(Iptr[i + 1] - b);
    const RowView rv = row_view(src, i, b);
    uint32_t s = 0;
    for (int k0 = threadIdx.x; k0 < len; k0 += 4 * ROW_THREADS) {
        bool u[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = rv.unc(min(k0 + e * ROW_THREADS, len - 1));
#pragma unroll
        for (int e = 0; e < 4; ++e) s += (k0 + e * ROW_THREADS < len) && !u[e];
    }
    if (s) atomicAdd(&acc, s);
    __syncthreads();
    if (threadIdx.x == 0) cnt[i] = (int32_t)acc;
}

__global__ __launch_bounds__(ROW_THREADS) void k_comp_fill(const int64_t *__restrict__ Iptr, RowSrc src, const int2 *__restrict__ ij,
                                                          const int64_t *__restrict__ cptr,
                                                          int32_t *__restrict__ cidx, double *__restrict__ cval)
{
    __shared__ uint32_t wsum[ROW_THREADS / 64];
    const int64_t i = row_of_block(gridDim.x), b = Iptr[i];
    const int len = (int)(Iptr[i + 1] - b);
    const RowView rv = row_view(src, i, b);
    const int32_t *Iidx = src.Iidx;
    const double *RA = src.RA;
    int64_t w = cptr[i];
    for (int base = 0; base < len; base += ROW_THREADS) {
        const int k = base + threadIdx.x;
        uint32_t f = 0;
        if (k < len) f = !rv.unc(k);
        uint32_t tot;
        const uint32_t ex = row_block_scan(f, wsum, &tot);
        if (f) {
            const int32_t p = Iidx[b + k];   // only the computed entries (a few per cent) look their pair up
            const int2 q = ij[p];
            cidx[w + ex] = q.x == (int)i ? q.y : q.x;
            cval[w + ex] = RA[p];
        }
        w += tot;
        __syncthreads();
    }
}

// The two CSR kernels over the column-ordered copy (long lists: a row's flags are two contiguous byte runs, the column-like half in
// Tm and the row half in ncm): 16 flags per load and one block scan per 4096 entries instead of a byte per thread and a scan per 256
// (1.17 ms at 127 M pairs, 14 % of HBM, for 0.25 GB of flags).  Flags are 0 / 1 bytes: the zero bytes of a word are 4 - popcount.
__device__ __forceinline__ uint32_t comp_zero_bytes(uint4 v) { return 16u - (uint32_t)(__popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w)); }
#define COMP_ONES make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u)

__global__ __launch_bounds__(ROW_THREADS) void k_comp_count_direct(const int64_t *__restrict__ Iptr, RowSrc src, int32_t *__restrict__ cnt)
{
    __shared__ uint32_t acc;
    if (threadIdx.x == 0) acc = 0;
    __syncthreads();
    const int64_t i = row_of_block(gridDim.x), b = Iptr[i];
    const int len = (int)(Iptr[i + 1] - b);
    const RowView rv = row_view(src, i, b);
    const int nA = min(rv.low, len);
    uint32_t s = 0;
    for (int part = 0; part < 2; ++part) {
        const uint8_t *p = part ? rv.ncm : rv.Tm;
        const int n = part ? len - nA : nA;
        const int head = min(n, (int)((16u - (uint32_t)(reinterpret_cast<uintptr_t>(p) & 15u)) & 15u));
        if ((int)threadIdx.x < head) s += p[threadIdx.x] == 0;
        const uint4 *q = reinterpret_cast<const uint4 *>(p + head);
        const int n16 = (n - head) >> 4;
        for (int t = threadIdx.x; t < n16; t += ROW_THREADS) s += comp_zero_bytes(q[t]);
        for (int k = head + (n16 << 4) + threadIdx.x; k < n; k += ROW_THREADS) s += p[k] == 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(&acc, s);
    __syncthreads();
    if (threadIdx.x == 0) cnt[i] = (int32_t)acc;
}

__global__ __launch_bounds__(ROW_THREADS) void k_comp_fill_direct(const int64_t *__restrict__ Iptr, RowSrc src, const int2 *__restrict__ ij,
                                                                 const int64_t *__restrict__ cptr,
                                                                 int32_t *__restrict__ cidx, double *__restrict__ cval)
{
    __shared__ uint32_t wsum[ROW_THREADS / 64];
    const int64_t i = row_of_block(gridDim.x), b = Iptr[i];
    const int len = (int)(Iptr[i + 1] - b);
    const RowView rv = row_view(src, i, b);
    const int32_t *Iidx = src.Iidx + b;
    const double *RA = src.RA;
    const int nA = min(rv.low, len);
    int64_t w = cptr[i];
    // one step: every thread brings up to 16 flags (missing ones as 1) of entries k0, k0 + 1, ...; computed entries are written in order
    auto step = [&](uint4 v, int k0) {
        const uint32_t z = comp_zero_bytes(v);
        uint32_t tot;
        uint32_t ex = row_block_scan(z, wsum, &tot);
        if (z) {
            const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (((wd[e >> 2] >> (8 * (e & 3))) & 0xFFu) == 0u) {
                    const int32_t p = Iidx[k0 + e];   // only the computed entries (a few per cent) look their pair up
                    const int2 q = ij[p];
                    cidx[w + ex] = q.x == (int)i ? q.y : q.x;
                    cval[w + ex] = RA[p];
                    ++ex;
                }
        }
        w += tot;
        __syncthreads();
    };
    for (int part = 0; part < 2; ++part) {
        const uint8_t *p = part ? rv.ncm : rv.Tm;
        const int n = part ? len - nA : nA, kofs = part ? nA : 0;
        const int head = min(n, (int)((16u - (uint32_t)(reinterpret_cast<uintptr_t>(p) & 15u)) & 15u));
        if (head) {   // (uniform)
            uint4 v = COMP_ONES;
            if ((int)threadIdx.x < head) v.x = 0x01010100u | p[threadIdx.x];
            step(v, kofs + (int)threadIdx.x);
        }
        const uint4 *q = reinterpret_cast<const uint4 *>(p + head);
        const int n16 = (n - head) >> 4;
        for (int t0 = 0; t0 < n16; t0 += ROW_THREADS) {
            const int t = t0 + threadIdx.x;
            step(t < n16 ? q[t] : COMP_ONES, kofs + head + (t << 4));
        }
        const int tail0 = head + (n16 << 4);
        if (tail0 < n) {   // fewer than 16 entries
            uint4 v = COMP_ONES;
            if (tail0 + (int)threadIdx.x < n) v.x = 0x01010100u | p[tail0 + threadIdx.x];
            step(v, kofs + tail0 + (int)threadIdx.x);
        }
    }
}

// one wavefront per lookahead pair
#define UB_STAGE 1024   // keys of the searched list a wave keeps in LDS
// G lanes per pair: 64, or 32 / 16 for short lists (two / four pairs per wave: the kernel is a chain of ~6 dependent reads per pair,
// so what counts at C2 -- 375 000 pairs, ~100-entry lists -- is how many pairs are in flight, not lanes per list: 142 us with 64
// lanes per pair, 97 with 32, 76 with 16, 110 with 8)
template <int G>
__global__ __launch_bounds__(256) void k_update_bounds(const int32_t *__restrict__ next, int64_t nnext,
                                                      const int2 *__restrict__ ij, const int64_t *__restrict__ cptr,
                                                      const int32_t *__restrict__ cidx, const double *__restrict__ cval,
                                                      double *__restrict__ lb, double *__restrict__ ub)
{
    constexpr int GROUPS = 256 / G, STAGE = UB_STAGE * G / 64;   // keys of the searched list a group keeps in LDS
    __shared__ int32_t stage[GROUPS][STAGE];
    const int lane = threadIdx.x & (G - 1), grp = threadIdx.x / G;
    const int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const bool live = t < nnext;
    const int32_t p = live ? next[t] : 0;
    const int2 q = live ? ij[p] : make_int2(0, 0);
    int64_t a0 = 0, a1 = 0, b0 = 0, b1 = 0;
    if (live) { a0 = cptr[q.x]; a1 = cptr[q.x + 1]; b0 = cptr[q.y]; b1 = cptr[q.y + 1]; }
    if (a1 - a0 > b1 - b0) { int64_t x = a0; a0 = b0; b0 = x; x = a1; a1 = b1; b1 = x; }  // walk the shorter list
    double nl = 0.0, nu = INFINITY;
    const int nbk = (int)(b1 - b0);
    if (nbk <= STAGE) {
        // the searched list's keys go to LDS (one coalesced pass): ~10 dependent probes per element
        // at LDS latency instead of L2 latency
        int32_t *sk = stage[grp];
        for (int e = lane; e < nbk; e += G) sk[e] = cidx[b0 + e];
        // (a wave's LDS writes are visible to its own later reads -- in-order LDS queue -- the
        // fence only keeps the compiler from moving the reads above the writes)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int64_t e = a0 + lane; e < a1; e += G) {
            const int32_t key = cidx[e];
            int lo = 0, hi = nbk;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (sk[mid] < key) lo = mid + 1; else hi = mid;
            }
            if (lo < nbk && sk[lo] == key) {
                const double x = cval[e], y = cval[b0 + lo];
                nu = fmin(nu, x + y);
                nl = fmax(nl, fabs(x - y));
            }
        }
    } else {
        for (int64_t e = a0 + lane; e < a1; e += G) {
            const int32_t key = cidx[e];
            int64_t lo = b0, hi = b1;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (cidx[mid] < key) lo = mid + 1; else hi = mid;
            }
            if (lo < b1 && cidx[lo] == key) {
                const double x = cval[e], y = cval[lo];
                nu = fmin(nu, x + y);
                nl = fmax(nl, fabs(x - y));
            }
        }
    }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) {
        nu = fmin(nu, __shfl_xor(nu, off));
        nl = fmax(nl, __shfl_xor(nl, off));
    }
    if (lane == 0 && live) {
        lb[p] = fmax(nl, lb[p]);  // annchor.py:503-510
        ub[p] = fmin(nu, ub[p]);
    }
}

// Row-grouped form.  The lookahead list is in pair-list order, so consecutive entries share their first point i: a workgroup takes a
// run of UBB_CHUNK entries, keeps "is c in L_i, and where" for the current i in LDS and every wave streams the other points' lists
// L_j past it -- one coalesced key read and one LDS lookup per entry instead of ~10 dependent binary-search probes for every entry
// of the shorter list.  The table is rebuilt (old entries cleared, new ones written) when i changes inside the run; any order of the
// list is handled, the sorted one just rebuilds least.
// This is the second design of the round.  The first kept a uint16 slot per point (or a 64-bit word + running count per 64 points)
// and paid ~40 vector instructions per list entry -- 64-bit shifts and popcounts, two value loads issued whether the entry matched
// or not; the PMC pass over the N = 100 000 Levenshtein fit (2.5 x 10^8 lookahead pairs x ~1000-entry lists) showed VALU 77 % busy
// AND 0.95 TB of HBM reads (4-byte keys), 308 ms.  Only ~3 % of the entries match.  Now:
//  * membership of the current first point's list is one 8-byte LDS word per 32 points {bits, members before the word}; an entry
//    costs a key load, one LDS read and a bit test; everything else happens for MATCHES only: a wave-uniform branch on the ballot,
//    then rank (= slot in the first point's list) and position go to a small ring in LDS;
//  * the ring is drained 64 matches at a time (two value loads + two LDS atomics on per-pair accumulators; the sums and differences
//    are non-negative doubles, so their order is the order of their bit patterns and 64-bit integer min / max do) -- the value loads'
//    latency is paid once per 64 matches, not once per pair;
//  * a wave takes the metadata of 64 pairs at a time lane-parallel (next -> ij -> cptr: three dependent reads once per 64 pairs
//    instead of once per pair) and writes the 64 results back together;
//  * K16: the partner lists' keys are streamed from a 2-byte copy (low 16 bits; the lists are sorted, so the high bit of a key is
//    "position >= the list's first entry with key >= 65 536", one number per list) -- half the HBM bytes.  Lists start 8-byte aligned
//    in that copy (list j at (cptr[j] & ~3) + 4 j: no extra offset array), a lane reads four keys per load.  nx <= 131 072; beyond
//    that the 4-byte keys are read as before.
#define UBB_THREADS 512  // eight waves share one table (25 KB at 100 000 points: three workgroups = 24 waves per CU)
#define UBB_CHUNK 2048   // lookahead entries per workgroup
#define UBB_RING 128     // pending matches per wave (drained whenever 64 are waiting)
#define UBB_GAP 4        // list j of the 2-byte copy starts at (cptr[j] & ~3) + UBB_GAP j: 8-byte aligned, no overlap, no offset array
#define UBB_K16_WORDS 4096   // K16 table size: entries read past a list's end (masked) still index inside it whatever their 17 bits are
__device__ __forceinline__ size_t ubb_off16(int64_t c0, int64_t j) { return (size_t)((c0 & ~(int64_t)3) + UBB_GAP * j); }

// the 2-byte key copy + the per-list position of the first key >= 65 536: one wave per list
__global__ __launch_bounds__(256) void k_comp_narrow(const int64_t *__restrict__ cptr, const int32_t *__restrict__ cidx, int64_t nx,
                                                    uint16_t *__restrict__ c16, uint32_t *__restrict__ cbnd)
{
    const int lane = threadIdx.x & 63;
    const int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (j >= nx) return;
    const int64_t c0 = cptr[j], len = cptr[j + 1] - c0;
    uint16_t *out = c16 + ubb_off16(c0, j);
    uint32_t below = 0;
    for (int64_t e = lane; e < len; e += 64) {
        const int32_t k = cidx[c0 + e];
        out[e] = (uint16_t)k;
        below += k < 65536;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) below += __shfl_xor(below, off);
    if (lane == 0) cbnd[j] = below;
}

// NK = keys per lane and step (K16: NK / 4 eight-byte reads of four keys, else NK four-byte reads), NT threads share a table, CH lookahead
// entries per workgroup: <8, 512, 2048>
template <bool K16, int NK, int NT, int CH>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT == 512 ? 6 : 4, NT == 512 ? 6 : 8))) void k_update_bounds_bits(const int32_t *__restrict__ next, int64_t nnext,
                                                           const int2 *__restrict__ ij, const int64_t *__restrict__ cptr,
                                                           const int32_t *__restrict__ cidx, const uint16_t *__restrict__ c16,
                                                           const uint32_t *__restrict__ cbnd, const double *__restrict__ cval,
                                                           double *__restrict__ lb, double *__restrict__ ub, int nx)
{
    // all of the workgroup's LDS is the dynamic block, the table FIRST: its byte offsets are then plain functions of the key bits
    // (no base to add per lookup).  [TW] table, then per wave the ring and the two accumulator rows, then the scan / run scratch
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    uint2 *tabw = reinterpret_cast<uint2 *>(dyn);   // [TW] {members of the current first point's list in points 32w .. 32w+31, members before}
    const int W = (nx + 31) / 32;
    constexpr int NW = NT / 64, EPI = 64 * NK;
    static_assert(!K16 || NK % 4 == 0, "2-byte keys come four per read");
    const int TW = K16 ? UBB_K16_WORDS : W;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned char *after = dyn + (size_t)TW * 8;
    uint2 *ring = reinterpret_cast<uint2 *>(after) + wave * UBB_RING;   // {slot in the first point's list | pair-in-batch << 20, position in cval}
    unsigned long long *accU = reinterpret_cast<unsigned long long *>(after + NW * UBB_RING * 8) + wave * 64;
    unsigned long long *accL = reinterpret_cast<unsigned long long *>(after + NW * UBB_RING * 8 + NW * 64 * 8) + wave * 64;
    uint32_t *scan_w = reinterpret_cast<uint32_t *>(after + NW * UBB_RING * 8 + 2 * NW * 64 * 8);
    int &first_other = *reinterpret_cast<int *>(scan_w + NW);
    for (int c = threadIdx.x; c < W; c += NT) tabw[c] = make_uint2(0u, 0u);
    const int64_t t0 = (int64_t)blockIdx.x * CH, t1 = min(t0 + CH, nnext);
    int cur = -1;
    int64_t ca0 = 0, ca1 = 0;
    __syncthreads();
    for (int64_t t = t0; t < t1;) {
        const int i = ij[next[t]].x;   // uniform
        // how many consecutive entries from t share this first point?
        if (threadIdx.x == 0) first_other = (int)(t1 - t);
        __syncthreads();
        for (int64_t tt = t + threadIdx.x; tt < t1; tt += NT)
            if (ij[next[tt]].x != i) { atomicMin(&first_other, (int)(tt - t)); break; }
        __syncthreads();
        const int seg = first_other;
        if (cur != i) {
            for (int64_t e = ca0 + threadIdx.x; e < ca1; e += NT) tabw[cidx[e] >> 5] = make_uint2(0u, 0u);
            ca0 = cptr[i]; ca1 = cptr[i + 1];
            __syncthreads();
            for (int64_t e = ca0 + threadIdx.x; e < ca1; e += NT) {
                const int cc = cidx[e];
                atomicOr(&tabw[cc >> 5].x, 1u << (cc & 31));
            }
            __syncthreads();
            // members before each word: blocked over the threads (consecutive words each) + a block scan
            const int per = (W + NT - 1) / NT, w0 = threadIdx.x * per, w1 = min(w0 + per, W);
            uint32_t mine = 0;
            for (int w = w0; w < w1; ++w) mine += (uint32_t)__popc(tabw[w].x);
            uint32_t inc = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(inc, off); if (lane >= off) inc += o; }
            if (lane == 63) scan_w[wave] = inc;
            __syncthreads();
            uint32_t base = inc - mine;
            for (int w = 0; w < wave; ++w) base += scan_w[w];
            for (int w = w0; w < w1; ++w) { tabw[w].y = base; base += (uint32_t)__popc(tabw[w].x); }
            cur = i;
            __syncthreads();
        }
        // this wave's pairs of the run: t + wave + NW m, 64 of them (one per lane) at a time
        for (int m0 = 0; wave + NW * m0 < seg; m0 += 64) {
            const int q = wave + NW * (m0 + lane);
            const bool have = q < seg;
            int32_t p = 0;
            int64_t c0 = 0;
            uint32_t len = 0, bd = 0;
            size_t o16 = 0;
            if (have) {
                p = next[t + q];
                const int j = ij[p].y;
                c0 = cptr[j];
                len = (uint32_t)(cptr[j + 1] - c0);
                if (K16) { bd = cbnd[j]; o16 = ubb_off16(c0, j); }
            }
            accU[lane] = 0x7FF0000000000000ull;   // +inf
            accL[lane] = 0ull;
            const int npair = min(64, (seg - wave - NW * m0 + NW - 1) / NW);
            int qhead = 0, qcount = 0;   // uniform
            auto drain = [&](int nf) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (lane < nf) {
                    const uint2 e = ring[(qhead + lane) & (UBB_RING - 1)];
                    const double x = cval[ca0 + (e.x & 0xFFFFFu)], y = cval[e.y];
                    const int kk = (int)(e.x >> 20);
                    atomicMin(&accU[kk], (unsigned long long)__double_as_longlong(x + y));
                    atomicMax(&accL[kk], (unsigned long long)__double_as_longlong(fabs(x - y)));
                }
                qhead = (qhead + nf) & (UBB_RING - 1);
                qcount -= nf;
            };
            // the partner lists as ONE stream of (pair, offset) steps: the keys of the next step are requested before the current
            // step is looked at (a wave that waited out every step's read alone kept 1 KB in flight: 2.4 TB/s over the chip)
            uint32_t kv[NK];   // K16: kv[0..3] = the two 8-byte reads; else eight keys
            auto request = [&](int kk, uint32_t off, uint32_t *dst) {
                const uint32_t L = (uint32_t)__builtin_amdgcn_readlane((int)len, kk);
                const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(K16 ? (uint64_t)o16 : (uint64_t)c0), kk);
                const uint32_t bhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((K16 ? (uint64_t)o16 : (uint64_t)c0) >> 32), kk);
                const size_t base = ((size_t)bhi << 32) | blo;
                if (K16) {
                    const uint2 *src = reinterpret_cast<const uint2 *>(c16 + base);   // 8-byte aligned: four keys per lane and read
                    const uint32_t last4 = (max(L, 1u) - 1) >> 2;
#pragma unroll
                    for (int u = 0; u < NK / 4; ++u) {
                        const uint2 v = src[min((off >> 2) + 64 * u + lane, last4)];
                        dst[2 * u] = v.x; dst[2 * u + 1] = v.y;
                    }
                } else {
                    const int32_t *src = cidx + base;
                    const uint32_t last = max(L, 1u) - 1;
#pragma unroll
                    for (int u = 0; u < NK; ++u) dst[u] = (uint32_t)src[min(off + 64 * u + lane, last)];
                }
            };
            int kk = 0;
            uint32_t off = 0;
            if (npair > 0) request(0, 0u, kv);
            while (kk < npair) {
                const uint32_t L = (uint32_t)__builtin_amdgcn_readlane((int)len, kk);
                int nk = kk;
                uint32_t noff = off + EPI;
                if (noff >= L) { nk = kk + 1; noff = 0; }
                uint32_t nv[NK];
#pragma unroll
                for (int u = 0; u < NK; ++u) nv[u] = 0;
                if (nk < npair) request(nk, noff, nv);
                const uint32_t c0lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)c0, kk);   // (positions in cval fit 32 bits: <= 2^31 entries)
                const uint32_t Lr = L > off ? L - off : 0u;   // entries of this list from off on
                uint32_t key[NK], lpos[NK];   // key and position (relative to off) of the lane's entries
                uint2 w[NK];
                uint32_t mask = 0;
                if (K16) {
                    const uint32_t B = (uint32_t)__builtin_amdgcn_readlane((int)bd, kk);
                    const uint32_t Br = B > off ? B - off : 0u;   // entries from Br on carry the high bit
                    // the table word's byte offset straight from the packed half (bits 5..15 of the key -> bits 3..13, the high bit
                    // -> 16 384), the tested bit from its low five bits; which of the lane's entries exist is ONE mask per step (the
                    // lane's entries of a read are four consecutive positions), not a compare per entry
                    uint32_t bitidx[NK];
#pragma unroll
                    for (int u = 0; u < NK; ++u) {
                        const uint32_t half = kv[u >> 1];
                        lpos[u] = 4u * (64u * (u >> 2) + lane) + (u & 3);
                        const uint32_t at = (((u & 1) ? half >> 18 : half >> 2) & 0x3FF8u) | (lpos[u] >= Br ? 16384u : 0u);
                        w[u] = *reinterpret_cast<const uint2 *>(dyn + at);
                        bitidx[u] = (u & 1) ? half >> 16 : half;
                        key[u] = 0;   // (rebuilt for the matches only, below)
                    }
                    uint32_t vm = 0;
#pragma unroll
                    for (int g = 0; g < NK / 4; ++g) {
                        const int nv = min(max((int)Lr - (int)(4u * (64u * g + lane)), 0), 4);
                        vm |= ((1u << nv) - 1u) << (4 * g);
                    }
#pragma unroll
                    for (int u = 0; u < NK; ++u) mask |= __builtin_amdgcn_ubfe(w[u].x, bitidx[u], 1u) << u;   // (v_bfe_u32 takes the low five bits of the offset)
                    mask &= vm;
                } else {
#pragma unroll
                    for (int u = 0; u < NK; ++u) { key[u] = kv[u]; lpos[u] = 64u * u + lane; }
#pragma unroll
                    for (int u = 0; u < NK; ++u) w[u] = tabw[key[u] >> 5];
#pragma unroll
                    for (int u = 0; u < NK; ++u) mask |= (lpos[u] < Lr ? (w[u].x >> (key[u] & 31)) & 1u : 0u) << u;
                }
                // matches are few (3 % of the entries at 100 000 clustered strings, ~15 per step; one step in two has none): everything
                // below runs for the lanes that have one
                unsigned long long bal = __ballot(mask != 0);
                while (bal) {
                    if (mask) {
                        const int u = __ffs(mask) - 1;
                        mask &= mask - 1;
                        uint32_t k, lp;
                        if (K16) {
                            uint32_t half = kv[0];
#pragma unroll
                            for (int e = 1; e < NK / 2; ++e) half = (u >> 1) == e ? kv[e] : half;
                            lp = 4u * (64u * (uint32_t)(u >> 2) + lane) + (uint32_t)(u & 3);
                            const uint32_t B = (uint32_t)__builtin_amdgcn_readlane((int)bd, kk);
                            k = ((half >> ((u & 1) * 16)) & 0xFFFFu) | (off + lp >= B ? 65536u : 0u);
                        } else {
                            k = key[0]; lp = lpos[0];
#pragma unroll
                            for (int e = 1; e < NK; ++e) { k = u == e ? key[e] : k; lp = u == e ? lpos[e] : lp; }
                        }
                        const uint2 ww = tabw[k >> 5];
                        const uint32_t rank = ww.y + (uint32_t)__popc(ww.x & ((1u << (k & 31)) - 1u));
                        const int at = qcount + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                        ring[(qhead + at) & (UBB_RING - 1)] = make_uint2(rank | ((uint32_t)kk << 20), c0lo + off + lp);
                    }
                    qcount += __popcll(bal);
                    if (qcount >= 64) drain(64);
                    bal = __ballot(mask != 0);
                }
#pragma unroll
                for (int u = 0; u < NK; ++u) kv[u] = nv[u];
                kk = nk; off = noff;
            }
            if (qcount) drain(qcount);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (have) {
                lb[p] = fmax(__longlong_as_double((long long)accL[lane]), lb[p]);  // annchor.py:503-510
                ub[p] = fmin(__longlong_as_double((long long)accU[lane]), ub[p]);
            }
        }
        t += seg;
        __syncthreads();
    }
}

extern "C" int annchor_update_bounds(annchor_ctx *c)
{
    if (!c) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->have_RA, ANNCHOR_EINVAL, "RefineApprox not initialised");
    if (c->nnext == 0) return ANNCHOR_OK;
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    const int64_t nx = c->nx;
    ANN_TRY(ann_reserve(c, c->tmp1, sizeof(int32_t) * (size_t)nx));
    ANN_TRY(ann_reserve(c, c->cptr, sizeof(int64_t) * (size_t)(nx + 1)));
    ANN_CHECK_HIP(c, hipEventRecord(c->call_a, c->stream));
    int64_t total = 0;
    RowSrc rsrc;
    ANN_TRY(ann_transpose_columns(c, &rsrc, false));   // the mask alone (one byte per pair)
    {
        ProfScope ps(c, "computed_neighbour_csr", (double)c->n * 2 * 5.0);
        static const bool comp_generic = getenv("ANNCHOR_COMP_GENERIC") != nullptr;   // tests: the element-wise kernels on the column-ordered copy too
        const bool comp_direct = rsrc.T != nullptr && !comp_generic;
        if (comp_direct) k_comp_count_direct<<<(int)nx, ROW_THREADS, 0, c->stream>>>(c->Iptr.as<int64_t>(), rsrc, c->tmp1.as<int32_t>());
        else
        k_comp_count<<<(int)nx, ROW_THREADS, 0, c->stream>>>(c->Iptr.as<int64_t>(), rsrc, c->tmp1.as<int32_t>());
        ANN_TRY(ann_exclusive_scan_i32_to_i64(c, c->tmp1.as<int32_t>(), c->cptr.as<int64_t>(), nx));
        // how many computed entries (both directions)?  Small lists: room for the worst case (every pair
        // computed) instead of a host wait for the exact number; the profile's byte count then uses the
        // host's running count of computed pairs
        if ((size_t)c->n * 2 * 12 <= ((size_t)256 << 20)) {
            total = 2 * c->n;
        } else {
            ANN_TRY(ann_d2h(c, &total, c->cptr.as<int64_t>() + nx, sizeof total));
        }
        ANN_TRY(ann_reserve(c, c->cidx, sizeof(int32_t) * (size_t)(total + 1)));
        ANN_TRY(ann_reserve(c, c->cval, sizeof(double) * (size_t)(total + 1)));
        if (comp_direct) k_comp_fill_direct<<<(int)nx, ROW_THREADS, 0, c->stream>>>(c->Iptr.as<int64_t>(), rsrc, c->ij.as<int2>(), c->cptr.as<int64_t>(),
                                                           c->cidx.as<int32_t>(), c->cval.as<double>());
        else
        k_comp_fill<<<(int)nx, ROW_THREADS, 0, c->stream>>>(c->Iptr.as<int64_t>(), rsrc, c->ij.as<int2>(), c->cptr.as<int64_t>(),
                                                           c->cidx.as<int32_t>(), c->cval.as<double>());
    }
    {
        const int64_t counted = c->n_unc >= 0 ? 2 * (c->n - c->n_unc) : total;   // computed entries, both directions
        const double avg = nx > 0 ? (double)counted / (double)nx : 0.0;
        const char *ube = getenv("ANNCHOR_UPDATE_BOUNDS");   // "pairs" / "rows" force a form (tests compare the two)
        // long lists only: with ~100 entries per list (C2) the table rebuilds and the 512-entry strides cost more
        // than they save (0.28 vs 0.14 ms); at 800 entries per list 12.8 vs 18.1 ms
        // the row-grouped bit-table form ("bits16" / "bits32" force a variant, "pairs" the wave-per-pair form): long lists on
        // 2-byte keys (4-byte beyond 131 072 points), 512-entry steps; short lists (C2: ~100 entries) on 4-byte keys, 128-entry steps
        const bool long_form = ube ? strncmp(ube, "bits", 4) == 0 : avg >= 256.0;
        // (a row-grouped variant sized for short lists -- 128-entry steps, 256-pair chunks -- measured 156 us against the wave-per-pair
        // form's 141 us at C2 and 0.34 against 0.22 ms at N = 3000: retired in round 5; short lists keep the wave-per-pair form)
        const bool k16 = long_form && nx <= 131072 && !(ube && strcmp(ube, "bits32") == 0);
        const int ub_waves = UBB_THREADS / 64;
        // (table + per wave a 128-entry ring and two 64-entry accumulator rows + scan / run scratch)
        const size_t bits_bytes = (k16 ? (size_t)UBB_K16_WORDS : ((size_t)nx + 31) / 32) * 8 + (size_t)ub_waves * (UBB_RING * 8 + 2 * 64 * 8 + 4) + 16;
        const bool rows_form = long_form && bits_bytes <= 128 * 1024 && nx < (1 << 20);
        // Algorithmic bytes.  Wave-per-pair form: both computed lists of every lookahead pair, 12 B per entry (key + value).
        // Row-grouped form: the lookahead list is in pair order, so a first point's list (key + value) is read once per RUN of pairs
        // (<= one per point and per workgroup chunk); of the partners' lists what HAS to be read is the keys, at the width the kernel
        // streams them (2 B, 4 B beyond 131 072 points) -- values are touched for the few per cent of entries that match, a count the
        // host does not know, so they are left out (the fraction is a lower bound; round 2 priced both lists per pair at 12 B per
        // entry and the first row-grouped kernel already came out above 1).
        const int chunk = UBB_CHUNK;
        const double runs = (double)std::min<int64_t>(c->nnext, nx + (c->nnext + chunk - 1) / chunk);
        const double alg = rows_form ? (double)c->nnext * (avg * (k16 ? 2.0 : 4.0) + 36.0) + runs * avg * 12.0
                                     : (double)c->nnext * (2.0 * avg * 12.0 + 36.0);
        ProfScope ps(c, "update_bounds_intersect", alg);
#define UBB_ARGS c->next.as<int32_t>(), c->nnext, c->ij.as<int2>(), c->cptr.as<int64_t>(), c->cidx.as<int32_t>(), k16 ? c->c16.as<uint16_t>() : nullptr, \
                 k16 ? c->cbnd.as<uint32_t>() : nullptr, c->cval.as<double>(), c->lb.as<double>(), c->ub.as<double>(), (int)nx
        if (rows_form) {
            if (k16) {
                ANN_TRY(ann_reserve(c, c->c16, sizeof(uint16_t) * (size_t)(total + UBB_GAP * nx + 16)));
                ANN_TRY(ann_reserve(c, c->cbnd, sizeof(uint32_t) * (size_t)nx));
                k_comp_narrow<<<ann_blocks(nx * 64, 256), 256, 0, c->stream>>>(c->cptr.as<int64_t>(), c->cidx.as<int32_t>(), nx,
                                                                             c->c16.as<uint16_t>(), c->cbnd.as<uint32_t>());
            }
            if (bits_bytes > 32 * 1024) {
                if (k16) ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)k_update_bounds_bits<true, 8, UBB_THREADS, UBB_CHUNK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bits_bytes));
                else ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)k_update_bounds_bits<false, 8, UBB_THREADS, UBB_CHUNK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bits_bytes));
            }
            if (k16) k_update_bounds_bits<true, 8, UBB_THREADS, UBB_CHUNK><<<ann_blocks(c->nnext, UBB_CHUNK), UBB_THREADS, bits_bytes, c->stream>>>(UBB_ARGS);
            else k_update_bounds_bits<false, 8, UBB_THREADS, UBB_CHUNK><<<ann_blocks(c->nnext, UBB_CHUNK), UBB_THREADS, bits_bytes, c->stream>>>(UBB_ARGS);
        } else
#undef UBB_ARGS
        if (ube ? strcmp(ube, "pairs16") == 0 : avg < 192.0)   // ("pairs" / "pairs32" / "pairs16" force the group width)
            k_update_bounds<16><<<ann_blocks(c->nnext * 16, 256), 256, 0, c->stream>>>(
            c->next.as<int32_t>(), c->nnext, c->ij.as<int2>(), c->cptr.as<int64_t>(), c->cidx.as<int32_t>(),
            c->cval.as<double>(), c->lb.as<double>(), c->ub.as<double>());
        else if (ube ? strcmp(ube, "pairs32") == 0 : avg < 256.0)
            k_update_bounds<32><<<ann_blocks(c->nnext * 32, 256), 256, 0, c->stream>>>(
            c->next.as<int32_t>(), c->nnext, c->ij.as<int2>(), c->cptr.as<int64_t>(), c->cidx.as<int32_t>(),
            c->cval.as<double>(), c->lb.as<double>(), c->ub.as<double>());
        else
        k_update_bounds<64><<<ann_blocks(c->nnext * 64, 256), 256, 0, c->stream>>>(
            c->next.as<int32_t>(), c->nnext, c->ij.as<int2>(), c->cptr.as<int64_t>(), c->cidx.as<int32_t>(),
            c->cval.as<double>(), c->lb.as<double>(), c->ub.as<double>());
    }
    ANN_CHECK_HIP(c, hipEventRecord(c->call_b, c->stream));
    c->call_timed = true;
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}


// ------------------------------------------------------- column-half transpose (rowsel.h: RowSrc)
// 64 x 64 tiles of the (j, i), j < i, triangle through LDS: the read side walks row j of the pair
// list (consecutive kept columns = consecutive positions), the write side walks column i of the copy
// (consecutive kept rows = consecutive positions of T): both sides in whole cache lines, where a
// row kernel gathering its column-like half directly touches one line per 8-byte value.
#define TR_T 64
#define TR_SUPER 16   // tiles per side of a super-tile (one XCD's L2 holds its table lines: 2 x 1024 rows x (128 + 64) B)
template <bool VALUES> __global__ __launch_bounds__(256) void k_transpose_cols(const uint64_t *__restrict__ K, const uint32_t *__restrict__ pref, int kw,
                                                       const int32_t *__restrict__ low, const int64_t *__restrict__ rowstart,
                                                       const int64_t *__restrict__ Iptr, const double *__restrict__ RA,
                                                       const uint8_t *__restrict__ ncm, int64_t nx, double *__restrict__ T,
                                                       uint8_t *__restrict__ Tm)
{
    __shared__ double tv[TR_T][TR_T + 1];
    __shared__ uint8_t tm[TR_T][TR_T + 4];
    // tile (jb, ib), jb <= ib.  The read side walks bitmap / prefix words of 64 ROWS j at word ib, the write side those of 64 rows i at
    // word jb: a 128-byte line of either table serves 16 (32) neighbouring tiles, but row-major over the triangle the tiles that share
    // a write-side line are a whole tile row apart -- 38 GB of L2 fills per launch at 100 000 points for 10 GB of values (PMC).  So
    // tiles go in 16 x 16 SUPER-TILES, and consecutive workgroup ids land on different XCDs (id % 8), each with its own L2, so a
    // super-tile is dealt to ONE XCD: id -> (xcd = id % 8, k = id / 8), super-tile (k / 256) * 8 + xcd, tile k % 256 inside it.
    const int nb = kw;   // 64-wide blocks per side == bitmap words per row
    const int S = (nb + TR_SUPER - 1) / TR_SUPER;
    const int64_t k_in = (int64_t)blockIdx.x >> 3;
    const int64_t st = (k_in / (TR_SUPER * TR_SUPER)) * 8 + (blockIdx.x & 7);
    if (st >= (int64_t)S * (S + 1) / 2) return;
    // super-tile (JB, IB), JB <= IB, row-major over the triangle of super-tiles
    int JB = (int)((2.0 * S + 1.0 - sqrt((2.0 * S + 1.0) * (2.0 * S + 1.0) - 8.0 * (double)st)) * 0.5);
    while ((int64_t)JB * S - (int64_t)JB * (JB - 1) / 2 > st) --JB;
    while ((int64_t)(JB + 1) * S - (int64_t)(JB + 1) * JB / 2 <= st) ++JB;
    const int IB = JB + (int)(st - ((int64_t)JB * S - (int64_t)JB * (JB - 1) / 2));
    const int w_in = (int)(k_in % (TR_SUPER * TR_SUPER));
    const int jb = JB * TR_SUPER + w_in / TR_SUPER, ib = IB * TR_SUPER + w_in % TR_SUPER;
    if (jb > ib || ib >= nb) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // ---- read: wave handles 16 rows j, lane = column i
    const int64_t i_r = (int64_t)ib * 64 + lane;
    {
        // positions of the wave's 16 rows first (wave-uniform table reads), then all value loads together
        int64_t pos[16];
        bool ok[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int64_t j = min((int64_t)jb * 64 + wave * 16 + q, nx - 1);
            const uint64_t bits = K[j * kw + ib];
            ok[q] = (int64_t)jb * 64 + wave * 16 + q < nx && i_r > j && ((bits >> lane) & 1ull);
            const int64_t p = rowstart[j] + ((int64_t)pref[j * kw + ib] + __popcll(bits & ((1ull << lane) - 1ull)) - low[j]);
            pos[q] = ok[q] ? p : 0;
        }
        double v[16];
        uint8_t m[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) { v[q] = VALUES ? __builtin_nontemporal_load(&RA[pos[q]]) : 0.0; m[q] = __builtin_nontemporal_load(&ncm[pos[q]]); }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (VALUES) tv[wave * 16 + q][lane] = ok[q] ? v[q] : 0.0;
            tm[wave * 16 + q][lane] = ok[q] ? m[q] : (uint8_t)0;
        }
    }
    __syncthreads();
    // ---- write: wave handles 16 columns i, lane = row j.  The columns' table words first, lane-parallel (lane q reads column q's:
    // sixteen dependent scalar round trips one after the other were most of a tile's ~18 us), then handed round by readlane
    const int64_t j_w = (int64_t)jb * 64 + lane;
    uint32_t cb_lo = 0, cb_hi = 0, cd_lo = 0, cd_hi = 0;
    {
        const int64_t i = (int64_t)ib * 64 + wave * 16 + (lane & 15);
        if (i < nx) {
            const uint64_t bits = K[i * kw + jb];   // symmetric bitmap: bit j of row i <=> pair (j, i) kept
            const int64_t dst0 = (Iptr[i] - rowstart[i]) + (int64_t)pref[i * kw + jb];
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
            const int64_t dst = dst0 + __popcll(bits & ((1ull << lane) - 1ull));
            // (this kernel only runs on lists beyond ANN_STREAM_MIN_PAIRS)
            if (VALUES) __builtin_nontemporal_store(tv[lane][il], &T[dst]);
            __builtin_nontemporal_store(tm[lane][il], &Tm[dst]);
        }
    }
}

// Row kernels over a large pair list take the column-like halves from a column-ordered copy.
#define ANN_TRANSPOSE_MIN_PAIRS (8ll << 20)   // below this the per-pair arrays are L2 / MALL resident: gather directly
int ann_transpose_columns(annchor_ctx *c, RowSrc *src, bool with_values)
{
    src->RA = c->RA.as<double>(); src->ncm = c->ncm.as<uint8_t>(); src->Iidx = c->Iidx.as<int32_t>();
    src->T = nullptr; src->Tm = nullptr; src->rowstart = c->rowstart.as<int64_t>(); src->low = c->low.as<int32_t>();
    static const int shrink_min = getenv("ANNCHOR_ROWC_SHRINK_MIN") ? atoi(getenv("ANNCHOR_ROWC_SHRINK_MIN")) : ROWC_SHRINK_MIN;   // tests: small rows too
    src->shrink_min = shrink_min;
    static const long long min_pairs = getenv("ANNCHOR_TRANSPOSE_MIN") ? atoll(getenv("ANNCHOR_TRANSPOSE_MIN")) : ANN_TRANSPOSE_MIN_PAIRS;
    if (!c->have_bitmap || c->n < min_pairs) return ANNCHOR_OK;
    ANN_TRY(ann_reserve(c, c->colT, sizeof(double) * (size_t)c->n));
    ANN_TRY(ann_reserve(c, c->colM, (size_t)c->n));
    const int kw = (int)((c->nx + 63) / 64);
    // (grid: the triangle of 16 x 16 super-tiles, 256 workgroups each, dealt to the XCDs in groups of eight)
    const int64_t S = (kw + TR_SUPER - 1) / TR_SUPER;
    const int64_t tiles = ((S * (S + 1) / 2 + 7) / 8) * 8 * TR_SUPER * TR_SUPER;
    if (!with_values) {
        ProfScope ps(c, "transpose_column_half_mask", (double)c->n * 2.0);
        k_transpose_cols<false><<<(unsigned)tiles, 256, 0, c->stream>>>(c->Kbits.as<uint64_t>(), c->Kpref.as<uint32_t>(), kw, c->low.as<int32_t>(),
                                                                       c->rowstart.as<int64_t>(), c->Iptr.as<int64_t>(), c->RA.as<double>(),
                                                                       c->ncm.as<uint8_t>(), c->nx, c->colT.as<double>(), c->colM.as<uint8_t>());
    } else {
        // algorithmic bytes: every pair's value and mask read once, written once
        ProfScope ps(c, "transpose_column_half", (double)c->n * 18.0);
        k_transpose_cols<true><<<(unsigned)tiles, 256, 0, c->stream>>>(c->Kbits.as<uint64_t>(), c->Kpref.as<uint32_t>(), kw, c->low.as<int32_t>(),
                                                                c->rowstart.as<int64_t>(), c->Iptr.as<int64_t>(), c->RA.as<double>(),
                                                                c->ncm.as<uint8_t>(), c->nx, c->colT.as<double>(), c->colM.as<uint8_t>());
    }
    ANN_CHECK_HIP(c, hipGetLastError());
    src->T = c->colT.as<double>();
    src->Tm = c->colM.as<uint8_t>();
    return ANNCHOR_OK;
}

// -------------------------------------------------------------------- get_nn
// block per row: keys d' = RA (+ row max on not-computed entries); the nn-1 smallest
// by (d', slot); output values are the un-shifted RA (utils.py:417-428)
__global__ __launch_bounds__(ROW_THREADS) void k_get_nn(const int64_t *__restrict__ Iptr, RowSrc src,
                                                       const int2 *__restrict__ ij, int nn, int64_t *__restrict__ ngi,
                                                       double *__restrict__ ngd, int cap)
{
    __shared__ RowSelShared sh;
    __shared__ RowCand rc;
    __shared__ double wmax[ROW_THREADS / 64];
    __shared__ uint32_t cnt_lt;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    const int L = nn - 1;
    uint64_t *keys = reinterpret_cast<uint64_t *>(dyn);      // [cap] (fallback path only)
    uint64_t *lkey = keys + cap;                             // [L]
    int32_t *lslot = reinterpret_cast<int32_t *>(lkey + L);  // [L]
    const int64_t i = row_of_block(gridDim.x), b = Iptr[i];
    const int len = (int)(Iptr[i + 1] - b);
    if (threadIdx.x == 0) { ngi[i * nn] = i; ngd[i * nn] = 0.0; cnt_lt = 0; }
    // row maximum of RA (utils.py:418)
    const RowView rv = row_view(src, i, b);
    const int32_t *Iidx = src.Iidx;
    const double *RA = src.RA;
    // ---- first pass: row maximum (utils.py:418) and, at the same time, the candidates among the
    // COMPUTED entries alone.  Not-computed entries are keyed RA + max: with RA > 0 they sort behind
    // every computed entry, so when the row has nn-1 computed entries and no not-computed entry with
    // RA <= 0 (guarantee_nmin's -1 marks) the answer is here already and the second pass is not needed.
    __shared__ uint32_t ncomp_s, risky_s;
    if (threadIdx.x == 0) { ncomp_s = 0; risky_s = 0; }
    double mx = -INFINITY;
    uint32_t my_comp = 0, my_risky = 0;
    const int want0 = min(L, len);
    const int fast0 = want0 > 0 ? row_candidates(rc, len, want0, [&](int s) { return ann_key_asc(rv.val(s)); },
        [&](int s) { return !rv.unc(s); },
        [&](int, uint64_t kk, bool computed) {
            const double d = ann_key_asc_inv(kk);
            mx = fmax(mx, d);
            my_comp += computed;
            my_risky += !computed && !(d > 0.0);
        }) : -1;
    if (my_comp) atomicAdd(&ncomp_s, my_comp);
    if (my_risky) atomicAdd(&risky_s, my_risky);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3]));
    if (want0 > 0 && fast0 >= want0 && (int)ncomp_s >= want0 && risky_s == 0) {
        const int fast0s = row_cand_shrink(rc, fast0, want0, src.shrink_min);
        if (fast0s >= want0) {
        for (int e = threadIdx.x; e < fast0s; e += ROW_THREADS) {
            const uint64_t ke = rc.key[e];
            const int32_t se = rc.slot[e];
            int r = 0;
            for (int o = 0; o < fast0s; ++o) { const uint64_t ko = rc.key[o]; r += (ko < ke) || (ko == ke && rc.slot[o] < se); }
            if (r < want0) {
                const int32_t p = Iidx[b + se];
                const int2 q = ij[p];
                ngi[i * nn + 1 + r] = q.x == (int)i ? q.y : q.x;
                ngd[i * nn + 1 + r] = RA[p];
            }
        }
        for (int e = want0 + threadIdx.x; e < L; e += ROW_THREADS) { ngi[i * nn + 1 + e] = 0; ngd[i * nn + 1 + e] = 0.0; }
        return;
        }
    }
    __syncthreads();
    const bool in_lds = len <= cap;
    auto key_of = [&](int s) -> uint64_t {
        const double d = rv.val(s);
        const bool u = rv.unc(s);
        return ann_key_asc(u ? d + mx : d);
    };
    const int want = min(L, len);
    {
        // fast path: the nn-1 smallest by (key, slot) are among the entries below a sampled threshold
        const int fast = want > 0 ? row_candidates(rc, len, want, key_of, [](int) { return true; },
                                                   [&](int s, uint64_t kk, bool) { if (in_lds) keys[s] = kk; }) : -1;
        const int fasts = (fast >= want && want > 0) ? row_cand_shrink(rc, fast, want, src.shrink_min) : -1;
        if (fasts >= want && want > 0) {
            for (int e = threadIdx.x; e < fasts; e += ROW_THREADS) {
                const uint64_t ke = rc.key[e];
                const int32_t se = rc.slot[e];
                int r = 0;
                for (int o = 0; o < fasts; ++o) { const uint64_t ko = rc.key[o]; r += (ko < ke) || (ko == ke && rc.slot[o] < se); }
                if (r < want) {
                    const int32_t p = Iidx[b + se];
                    const int2 q = ij[p];
                    ngi[i * nn + 1 + r] = q.x == (int)i ? q.y : q.x;
                    ngd[i * nn + 1 + r] = RA[p];
                }
            }
            for (int e = want + threadIdx.x; e < L; e += ROW_THREADS) { ngi[i * nn + 1 + e] = 0; ngd[i * nn + 1 + e] = 0.0; }
            return;
        }
        if (want <= 0 && in_lds) { /* nothing staged, nothing to select */ }
    }
    auto kf = [&](int s) -> uint64_t { return in_lds ? keys[s] : key_of(s); };
    if (want > 0) {
        // t = np.partition(d, nn-1)[nn-1]; entries <= t, stably sorted, first nn-1.
        // Equivalent: the (nn-1) smallest by (key, slot).
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
            const int32_t p = Iidx[b + se];
            const int2 q = ij[p];
            ngi[i * nn + 1 + r] = q.x == (int)i ? q.y : q.x;
            ngd[i * nn + 1 + r] = RA[p];
        }
    }
    for (int e = want + threadIdx.x; e < L; e += ROW_THREADS) { ngi[i * nn + 1 + e] = 0; ngd[i * nn + 1 + e] = 0.0; }
}

extern "C" int annchor_neighbor_graph(annchor_ctx *c, int32_t nn, int64_t *ng_idx, double *ng_dist)
{
    if (!c || !ng_idx || !ng_dist) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->have_RA, ANNCHOR_EINVAL, "RefineApprox not initialised");
    ANN_REQUIRE(c, nn >= 2 && nn <= 1024, ANNCHOR_ELIMIT, "n_neighbors=%d: 2..1024 supported", nn);
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    const size_t cells = (size_t)c->nx * nn;
    // Small graphs are written by the kernel straight into the pinned download region (hipHostMalloc
    // memory is device addressable): the device-to-host copy and its dispatch (~100 us after the
    // kernel on this stack) disappear, only the wait remains.  ANNCHOR_NO_ZEROCOPY=1: staged copy.
    static const bool no_zc = getenv("ANNCHOR_NO_ZEROCOPY") != nullptr;
    const bool direct = c->pin && !no_zc && cells * 16 <= annchor_ctx::PIN_DL_BYTES;
    unsigned char *slot = c->pin ? c->pin + (size_t)annchor_ctx::PIN_SLOTS * annchor_ctx::PIN_SLOT_BYTES : nullptr;
    ANN_TRY(ann_reserve(c, c->stage_out, cells * 16));
    int64_t *d_i = direct ? reinterpret_cast<int64_t *>(slot) : c->stage_out.as<int64_t>();
    double *d_d = reinterpret_cast<double *>(d_i + cells);
    // the device-fitted model of the last iteration (coefficients, flags, residual lists) rides with the graph: its copies are
    // queued here, before the kernel, into the tail of the pinned region -- the fit then ends with ONE host wait
    size_t mp_used = 0;
    unsigned char *mp_at = nullptr;
    if (direct) {
        const size_t base = (cells * 16 + 255) & ~(size_t)255;
        if (base < annchor_ctx::PIN_DL_BYTES) {
            mp_at = slot + base;
            ANN_TRY(ann_model_prefetch_begin(c, mp_at, annchor_ctx::PIN_DL_BYTES - base, &mp_used));
        }
    }
    ANN_CHECK_HIP(c, hipEventRecord(c->call_a, c->stream));
    RowSrc rsrc;
    ANN_TRY(ann_transpose_columns(c, &rsrc));
    {
        ProfScope ps(c, "row_topk_graph", (double)c->n * 2 * 13.0 + (double)cells * 16.0);
        const size_t tail = (((size_t)(nn - 1) * 12) + 15) & ~(size_t)15;
        int cap = 2;
        if (!rsrc.T) ANN_TRY(row_pick_cap(c, k_get_nn, c->nx, c->nx, tail, &cap));
        ANN_TRY(row_lds_prepare(c, k_get_nn, (size_t)cap * 8 + tail));
        k_get_nn<<<(int)c->nx, ROW_THREADS, (size_t)cap * 8 + tail, c->stream>>>(
            c->Iptr.as<int64_t>(), rsrc, c->ij.as<int2>(), nn, d_i, d_d, cap);
    }
    ANN_CHECK_HIP(c, hipEventRecord(c->call_b, c->stream));
    c->call_timed = true;
    ANN_CHECK_HIP(c, hipGetLastError());
    if (direct) {
        ANN_CHECK_HIP(c, ann_sync(c, __func__));
        ann_model_prefetch_end(c, mp_at, mp_used);
        memcpy(ng_idx, slot, cells * 8);
        memcpy(ng_dist, slot + cells * 8, cells * 8);
        return ANNCHOR_OK;
    }
    if (c->pin && cells * 16 <= annchor_ctx::PIN_DL_BYTES) {
        // indices and distances sit back to back: one transfer into the pinned download region
        ANN_CHECK_HIP(c, hipMemcpyAsync(slot, d_i, cells * 16, hipMemcpyDeviceToHost, c->stream));
        ANN_CHECK_HIP(c, ann_sync(c, __func__));
        memcpy(ng_idx, slot, cells * 8);
        memcpy(ng_dist, slot + cells * 8, cells * 8);
        return ANNCHOR_OK;
    }
    ANN_TRY(ann_d2h(c, ng_idx, d_i, cells * 8));
    return ann_d2h(c, ng_dist, d_d, cells * 8);
}
