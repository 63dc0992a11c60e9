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

// one wavefront per lookahead pair
#define UB_STAGE 1024   // keys of the searched list a wave keeps in LDS
__global__ __launch_bounds__(256) void k_update_bounds(const int32_t *__restrict__ next, int64_t nnext,
                                                      const int2 *__restrict__ ij, const int64_t *__restrict__ cptr,
                                                      const int32_t *__restrict__ cidx, const double *__restrict__ cval,
                                                      double *__restrict__ lb, double *__restrict__ ub)
{
    __shared__ int32_t stage[4][UB_STAGE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (t >= nnext) return;
    const int32_t p = next[t];
    const int2 q = ij[p];
    int64_t a0 = cptr[q.x], a1 = cptr[q.x + 1], b0 = cptr[q.y], b1 = cptr[q.y + 1];
    if (a1 - a0 > b1 - b0) { int64_t x = a0; a0 = b0; b0 = x; x = a1; a1 = b1; b1 = x; }  // walk the shorter list
    double nl = 0.0, nu = INFINITY;
    const int nbk = (int)(b1 - b0);
    if (nbk <= UB_STAGE) {
        // the searched list's keys go to LDS (one coalesced pass): ~10 dependent probes per element
        // at LDS latency instead of L2 latency
        int32_t *sk = stage[wave];
        for (int e = lane; e < nbk; e += 64) sk[e] = cidx[b0 + e];
        // (a wave's LDS writes are visible to its own later reads -- in-order LDS queue -- the
        // fence only keeps the compiler from moving the reads above the writes)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int64_t e = a0 + lane; e < a1; e += 64) {
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
        for (int64_t e = a0 + lane; e < a1; e += 64) {
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
    for (int off = 32; off > 0; off >>= 1) {
        nu = fmin(nu, __shfl_xor(nu, off));
        nl = fmax(nl, __shfl_xor(nl, off));
    }
    if (lane == 0) {
        lb[p] = fmax(nl, lb[p]);  // annchor.py:503-510
        ub[p] = fmin(nu, ub[p]);
    }
}

// Row-grouped form.  The lookahead list is in pair-list order, so consecutive entries share their first
// point i: a workgroup takes a run of UBR_CHUNK entries, keeps "slot of c in L_i" for the current i in an LDS
// table over all points (uint16: L_i has < 65 536 entries; 2 B x nx <= 60 KB at the pair-list form's largest
// nx) and every wave streams the other point's list L_j past it -- one coalesced key read and one LDS
// lookup per entry, values fetched for the matches only -- instead of ~10 dependent binary-search probes for
// every entry of the shorter list.  The table is rebuilt (old entries cleared, new ones written) when i
// changes inside the run; any order of the list is handled, the sorted one just rebuilds least.
// BITMAP = true (point sets too large for the table: nx >= 65 536): "is c in L_i, and where" from a bit per point plus a
// running count per 64-bit word -- the lists are sorted by the other endpoint, so the slot of c is its rank among the set
// bits: count[c / 64] + popcount(bits[c / 64] below c).  0.19 B per point instead of 2 (19 KB at 100 000 points).
#define UBR_CHUNK 256
template <bool BITMAP>
__global__ __launch_bounds__(256) void k_update_bounds_rows(const int32_t *__restrict__ next, int64_t nnext,
                                                           const int2 *__restrict__ ij, const int64_t *__restrict__ cptr,
                                                           const int32_t *__restrict__ cidx, const double *__restrict__ cval,
                                                           double *__restrict__ lb, double *__restrict__ ub, int nx)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    uint16_t *tab = reinterpret_cast<uint16_t *>(dyn);   // [nx] slot + 1 of point c in the current row's list, 0 = absent
    const int W = (nx + 63) / 64;
    unsigned long long *bits = reinterpret_cast<unsigned long long *>(dyn);   // BITMAP: [W] members of the current row's list
    uint32_t *wcnt = reinterpret_cast<uint32_t *>(bits + W);                  //         [W] members in the words before (32 bits: a
                                                                              //         point may have 65 536 computed neighbours and more)
    __shared__ int first_other;
    __shared__ uint32_t scan_w[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (BITMAP) { for (int c = threadIdx.x; c < W; c += 256) bits[c] = 0ull; }
    else for (int c = threadIdx.x; c < (nx + 1) / 2; c += 256) reinterpret_cast<uint32_t *>(tab)[c] = 0u;
    const int64_t t0 = (int64_t)blockIdx.x * UBR_CHUNK, t1 = min(t0 + UBR_CHUNK, nnext);
    int cur = -1;
    int64_t ca0 = 0, ca1 = 0;
    __syncthreads();
    for (int64_t t = t0; t < t1;) {
        const int i = ij[next[t]].x;   // uniform
        // how many consecutive entries from t share this first point?
        if (threadIdx.x == 0) first_other = (int)(t1 - t);
        __syncthreads();
        {
            const int64_t tt = t + threadIdx.x;
            if (tt < t1 && ij[next[tt]].x != i) atomicMin(&first_other, (int)threadIdx.x);
        }
        __syncthreads();
        const int seg = first_other;
        if (cur != i) {
            if (BITMAP) {
                for (int64_t e = ca0 + threadIdx.x; e < ca1; e += 256) bits[cidx[e] >> 6] = 0ull;
                ca0 = cptr[i]; ca1 = cptr[i + 1];
                __syncthreads();
                for (int64_t e = ca0 + threadIdx.x; e < ca1; e += 256) {
                    const int cc = cidx[e];
                    atomicOr(&bits[cc >> 6], 1ull << (cc & 63));
                }
                __syncthreads();
                // exclusive counts per word: blocked over the threads (W / 256 consecutive words each) + a block scan
                const int per = (W + 255) / 256, w0 = threadIdx.x * per, w1 = min(w0 + per, W);
                uint32_t mine = 0;
                for (int w = w0; w < w1; ++w) mine += (uint32_t)__popcll(bits[w]);
                uint32_t inc = mine;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(inc, off); if (lane >= off) inc += o; }
                if (lane == 63) scan_w[wave] = inc;
                __syncthreads();
                uint32_t base = inc - mine;
                for (int w = 0; w < wave; ++w) base += scan_w[w];
                for (int w = w0; w < w1; ++w) { wcnt[w] = base; base += (uint32_t)__popcll(bits[w]); }
            } else {
                for (int64_t e = ca0 + threadIdx.x; e < ca1; e += 256) tab[cidx[e]] = 0;
                ca0 = cptr[i]; ca1 = cptr[i + 1];
                __syncthreads();
                for (int64_t e = ca0 + threadIdx.x; e < ca1; e += 256) tab[cidx[e]] = (uint16_t)(e - ca0 + 1);
            }
            cur = i;
            __syncthreads();
        }
        for (int q = wave; q < seg; q += 4) {
            const int32_t p = next[t + q];
            const int j = ij[p].y;
            const int64_t b0 = cptr[j], b1 = cptr[j + 1];
            double nl = 0.0, nu = INFINITY;
            // eight key reads in flight per lane (a wave alone keeps 2 KB of the list on its way: one read at a
            // time the kernel ran at memory latency), then the lookups, then the value reads of the matches
            constexpr int U = 8;
            for (int64_t e0 = b0 + lane; e0 < b1; e0 += 64 * U) {
                int32_t key[U];
#pragma unroll
                for (int u = 0; u < U; ++u) key[u] = cidx[min(e0 + 64 * u, b1 - 1)];
                uint32_t sl[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (BITMAP) {
                        const unsigned long long b = bits[key[u] >> 6];
                        const int sh = key[u] & 63;
                        const uint32_t r = wcnt[key[u] >> 6] + (uint32_t)__popcll(b & ((1ull << sh) - 1ull)) + 1u;
                        sl[u] = (e0 + 64 * u < b1 && ((b >> sh) & 1ull)) ? r : 0u;
                    } else
                        sl[u] = e0 + 64 * u < b1 ? (uint32_t)tab[key[u]] : 0u;
                }
                // both values of every entry are requested whether it matches or not (clamped addresses, all in
                // flight together): a branch per entry around two dependent reads serialised eight round trips
                double x[U], y[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    x[u] = cval[ca0 + (sl[u] ? sl[u] - 1 : 0)];
                    y[u] = cval[sl[u] ? e0 + 64 * u : ca0];   // no match: a line that is hot anyway (the partner lists miss L2)
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    nu = sl[u] ? fmin(nu, x[u] + y[u]) : nu;
                    nl = sl[u] ? fmax(nl, fabs(x[u] - y[u])) : nl;
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                nu = fmin(nu, __shfl_xor(nu, off));
                nl = fmax(nl, __shfl_xor(nl, off));
            }
            if (lane == 0) {
                lb[p] = fmax(nl, lb[p]);  // annchor.py:503-510
                ub[p] = fmin(nu, ub[p]);
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
        k_comp_fill<<<(int)nx, ROW_THREADS, 0, c->stream>>>(c->Iptr.as<int64_t>(), rsrc, c->ij.as<int2>(), c->cptr.as<int64_t>(),
                                                           c->cidx.as<int32_t>(), c->cval.as<double>());
    }
    {
        const int64_t counted = c->n_unc >= 0 ? 2 * (c->n - c->n_unc) : total;   // computed entries, both directions
        const double avg = nx > 0 ? (double)counted / (double)nx : 0.0;
        const char *ube = getenv("ANNCHOR_UPDATE_BOUNDS");   // "pairs" / "rows" force a form (tests compare the two)
        // long lists only: with ~100 entries per list (C2) the table rebuilds and the 512-entry strides cost more
        // than they save (0.28 vs 0.14 ms); at 800 entries per list 12.8 vs 18.1 ms
        const size_t bm_bytes = (((size_t)nx + 63) / 64) * 12 + 16;   // bit per point + uint32 count per word
        const bool force_bm = ube && strcmp(ube, "bitmap") == 0;
        const bool table_ok = nx < 65536 && (((size_t)nx + 1) / 2) * 4 <= 150 * 1024;
        const bool rows_bitmap = (force_bm || (!ube && avg >= 256.0 && !table_ok)) && bm_bytes <= 150 * 1024;
        const bool rows_form = rows_bitmap || ((ube ? strcmp(ube, "rows") == 0 : avg >= 256.0) && table_ok);
        // Algorithmic bytes (12 B per list entry: key + value).  Wave-per-pair form: both computed lists of every
        // lookahead pair.  Row-grouped form: the lookahead list is in pair order, so a first point's list is read once
        // per RUN of pairs (<= one per point and per workgroup chunk) and only the partners' lists once per pair --
        // pricing it with both lists per pair (round 2) put the fraction above 1.
        const double runs = (double)std::min<int64_t>(c->nnext, nx + (c->nnext + UBR_CHUNK - 1) / UBR_CHUNK);
        const double alg = rows_form ? (double)c->nnext * (avg * 12.0 + 36.0) + runs * avg * 12.0
                                     : (double)c->nnext * (2.0 * avg * 12.0 + 36.0);
        ProfScope ps(c, "update_bounds_intersect", alg);
        const size_t tab_bytes = (((size_t)nx + 1) / 2) * 4;
        if (rows_bitmap) {
            if (bm_bytes > 64 * 1024)
                ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)k_update_bounds_rows<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     (int)bm_bytes));
            k_update_bounds_rows<true><<<ann_blocks(c->nnext, UBR_CHUNK), 256, bm_bytes, c->stream>>>(
                c->next.as<int32_t>(), c->nnext, c->ij.as<int2>(), c->cptr.as<int64_t>(), c->cidx.as<int32_t>(),
                c->cval.as<double>(), c->lb.as<double>(), c->ub.as<double>(), (int)nx);
        } else if (rows_form) {
            if (tab_bytes > 64 * 1024)
                ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)k_update_bounds_rows<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     (int)tab_bytes));
            k_update_bounds_rows<false><<<ann_blocks(c->nnext, UBR_CHUNK), 256, tab_bytes, c->stream>>>(
                c->next.as<int32_t>(), c->nnext, c->ij.as<int2>(), c->cptr.as<int64_t>(), c->cidx.as<int32_t>(),
                c->cval.as<double>(), c->lb.as<double>(), c->ub.as<double>(), (int)nx);
        } else
        k_update_bounds<<<ann_blocks(c->nnext * 64, 256), 256, 0, c->stream>>>(
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
template <bool VALUES> __global__ __launch_bounds__(256) void k_transpose_cols(const uint64_t *__restrict__ K, const uint32_t *__restrict__ pref, int kw,
                                                       const int32_t *__restrict__ low, const int64_t *__restrict__ rowstart,
                                                       const int64_t *__restrict__ Iptr, const double *__restrict__ RA,
                                                       const uint8_t *__restrict__ ncm, int64_t nx, double *__restrict__ T,
                                                       uint8_t *__restrict__ Tm)
{
    __shared__ double tv[TR_T][TR_T + 1];
    __shared__ uint8_t tm[TR_T][TR_T + 4];
    // tile (jb, ib), jb <= ib, from the linear block index (row-major over the upper triangle of tiles)
    const int nb = kw;   // 64-wide blocks per side == bitmap words per row
    int64_t t = blockIdx.x;
    int jb = (int)((2.0 * nb + 1.0 - sqrt((2.0 * nb + 1.0) * (2.0 * nb + 1.0) - 8.0 * (double)t)) * 0.5);
    while ((int64_t)jb * nb - (int64_t)jb * (jb - 1) / 2 > t) --jb;
    while ((int64_t)(jb + 1) * nb - (int64_t)(jb + 1) * jb / 2 <= t) ++jb;
    const int ib = jb + (int)(t - ((int64_t)jb * nb - (int64_t)jb * (jb - 1) / 2));
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
    // ---- write: wave handles 16 columns i, lane = row j
    const int64_t j_w = (int64_t)jb * 64 + lane;
    for (int q = 0; q < 16; ++q) {
        const int il = wave * 16 + q;
        const int64_t i = (int64_t)ib * 64 + il;
        if (i >= nx) continue;
        const uint64_t bits = K[i * kw + jb];   // symmetric bitmap: bit j of row i <=> pair (j, i) kept
        if (j_w < i && ((bits >> lane) & 1ull)) {
            const int64_t dst = (Iptr[i] - rowstart[i]) + ((int64_t)pref[i * kw + jb] + __popcll(bits & ((1ull << lane) - 1ull)));
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
    const int64_t tiles = (int64_t)kw * (kw + 1) / 2;
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
    ANN_CHECK_HIP(c, hipEventRecord(c->call_a, c->stream));
    RowSrc rsrc;
    ANN_TRY(ann_transpose_columns(c, &rsrc));
    {
        ProfScope ps(c, "row_topk_graph", (double)c->n * 2 * 13.0 + (double)cells * 16.0);
        const size_t tail = (((size_t)(nn - 1) * 12) + 15) & ~(size_t)15;
        const int cap = rsrc.T ? 2 : row_lds_cap(c->nx, tail);
        ANN_TRY(row_lds_prepare(c, k_get_nn, (size_t)cap * 8 + tail));
        k_get_nn<<<(int)c->nx, ROW_THREADS, (size_t)cap * 8 + tail, c->stream>>>(
            c->Iptr.as<int64_t>(), rsrc, c->ij.as<int2>(), nn, d_i, d_d, cap);
    }
    ANN_CHECK_HIP(c, hipEventRecord(c->call_b, c->stream));
    c->call_timed = true;
    ANN_CHECK_HIP(c, hipGetLastError());
    if (direct) {
        ANN_CHECK_HIP(c, ann_sync(c, __func__));
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
