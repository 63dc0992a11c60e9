// features.hip -- per-pair features, sampler support, regression predict / clip / merge.
//
// Replaces get_bounds_njit_ijs, get_dad_ijs (reference annchor/utils.py:274-301,
// 355-380), Annchor.get_features_IJ (annchor/annchor.py:258-303), the device side
// of Sampler.sample (annchor/samplers.py:44-140, annchor/utils.py:543-578),
// SimpleStratifiedLinearRegression.predict + clip + RefineApprox merge
// (annchor/regressors.py:71-103, annchor/annchor.py:356-380) and
// SimpleStratifiedErrorRegression.predict (annchor/error_predictors.py:56-67).
//
// Layout: structure-of-arrays float64 lb[], ub[], dad[], RA[] + uint8 masks, all
// indexed by pair position; pairs are sorted by (i, j) so consecutive lanes share i
// (scalar broadcast of D[.][i]) and read D[.][j] from consecutive addresses of the
// anchor-major Dt.  All arithmetic is float64 and uses only |a-b|, a+b, min, max,
// /2 -- results are bit-identical to the reference's NumPy arithmetic.  The
// regression predict is evaluated as ((w0*lb + w1*ub) + w2*dad) + c with
// contraction disabled so that host and device agree bit for bit.
#include "common.h"
#include "selstate.h"
#include <mutex>

#pragma clang fp contract(off)

__global__ __launch_bounds__(256) void k_features(const int2 *__restrict__ ij, int64_t n, const double *__restrict__ Dt,
                                                 int64_t nx, int na, const int32_t *__restrict__ cA,
                                                 const int32_t *__restrict__ anchorRank, double *__restrict__ lb,
                                                 double *__restrict__ ub, double *__restrict__ dad,
                                                 uint8_t *__restrict__ anc, uint8_t *__restrict__ ncm)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int2 q = ij[p];
    const int i = q.x, j = q.y;
    double l = 0.0, u = INFINITY;
    // Eight anchors' loads in flight per step (the plain loop waited out one L2 round trip per
    // anchor); the tail repeats the last anchor, which max / min ignore.  The vector memory
    // pipeline is this kernel's limit at scale (48 loads per pair: tools/microbench/
    // features_probe.hip), and pairs are sorted by i, so a wave that sits inside one row -- all
    // but one wave per row -- reads its D[.][i] through the scalar cache instead.
    const int i0 = __builtin_amdgcn_readfirstlane(i);
    if (__all(i == i0)) {
        for (int a0 = 0; a0 < na; a0 += 8) {
            double di[8], dj[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const size_t row = (size_t)min(a0 + e, na - 1) * nx;
                di[e] = Dt[row + i0];
                dj[e] = Dt[row + j];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                l = fmax(l, fabs(di[e] - dj[e]));
                u = fmin(u, di[e] + dj[e]);
            }
        }
    } else {
        for (int a0 = 0; a0 < na; a0 += 8) {
            double di[8], dj[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const size_t row = (size_t)min(a0 + e, na - 1) * nx;
                di[e] = Dt[row + i];
                dj[e] = Dt[row + j];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                l = fmax(l, fabs(di[e] - dj[e]));
                u = fmin(u, di[e] + dj[e]);
            }
        }
    }
    lb[p] = l;
    ub[p] = u;
    dad[p] = (Dt[(size_t)cA[j] * nx + i] + Dt[(size_t)cA[i] * nx + j]) / 2;
    const uint8_t isa = (anchorRank[i] >= 0) | (anchorRank[j] >= 0);
    anc[p] = isa;
    ncm[p] = !isa;
}


// Tiled form for large pair lists.  k_features above gathers D[a][j] once per (pair, anchor): 48
// eight-byte loads per pair from L2, the kernel's limit at scale (21 % of HBM at 127 M pairs).  Here a
// wave owns 64 consecutive columns j and keeps THEIR anchor distances in registers (na doubles per
// lane, loaded once, coalesced), then walks FT_ROWS rows i: D[a][i] is wave-uniform (scalar loads),
// the pair position comes from the keep bitmap's popcount ranks (no read of ij[] at all), and the
// outputs of a row are consecutive addresses.  Per pair: 2 gathers (the two dad terms) + the stores.
#define FT_ROWS 32
template <int NA_MAX> __global__ __launch_bounds__(256) void k_features_tiled(
    const uint64_t *__restrict__ K, const uint32_t *__restrict__ pref, int kw, const int32_t *__restrict__ low,
    const int64_t *__restrict__ rowstart, const double *__restrict__ Dt, int64_t nx, int na, const int32_t *__restrict__ cA,
    const int32_t *__restrict__ anchorRank, double *__restrict__ lb, double *__restrict__ ub, double *__restrict__ dad,
    uint8_t *__restrict__ anc, uint8_t *__restrict__ ncm)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform for the compiler too: row data goes through scalar loads
    const int nrb = (int)((nx + FT_ROWS - 1) / FT_ROWS);
    // task = (row block rb, column word jb) with jb >= first word that can hold j > i; tasks are laid out
    // row-block major so that neighbouring waves share the rows' scalar loads through the scalar cache
    const int64_t task = (int64_t)blockIdx.x * 4 + wave;
    const int64_t rb = task / kw;
    const int jb = (int)(task - rb * kw);
    if (rb >= nrb) return;
    const int64_t i_lo = rb * FT_ROWS, i_hi = min(i_lo + FT_ROWS, nx);
    if ((int64_t)jb * 64 + 63 <= i_lo) return;   // no column of this word lies right of the block's first row
    const int64_t j = (int64_t)jb * 64 + lane;
    const int64_t jc = min(j, nx - 1);
    double dj[NA_MAX];
#pragma unroll
    for (int a = 0; a < NA_MAX; ++a) dj[a] = Dt[(size_t)min(a, na - 1) * nx + jc];
    const int caj = cA[jc];
    for (int64_t i = i_lo; i < i_hi; ++i) {
        const uint64_t bits = K[i * kw + jb];                       // wave-uniform
        const bool mine = j > i && j < nx && ((bits >> lane) & 1ull);
        if (!__any(mine)) continue;
        const int64_t pos = rowstart[i] + ((int64_t)pref[i * kw + jb] + __popcll(bits & ((1ull << lane) - 1ull)) - low[i]);
        double l = 0.0, u = INFINITY;
        double di[NA_MAX];
#pragma unroll
        for (int a = 0; a < NA_MAX; ++a) di[a] = Dt[(size_t)min(a, na - 1) * nx + i];   // uniform addresses: scalar loads, all in flight;
#pragma unroll                                                                       // the tail repeats the last anchor (max / min ignore it)
        for (int a = 0; a < NA_MAX; ++a) {
            l = fmax(l, fabs(di[a] - dj[a]));
            u = fmin(u, di[a] + dj[a]);
        }
        if (mine) {
            const int cai = cA[i];
            // streamed out: 24 B per pair that no later kernel finds in L2 anyway
            __builtin_nontemporal_store(l, &lb[pos]);
            __builtin_nontemporal_store(u, &ub[pos]);
            __builtin_nontemporal_store((Dt[(size_t)caj * nx + i] + Dt[(size_t)cai * nx + j]) / 2, &dad[pos]);
            // (the two byte masks are preset by memset and patched for the few anchor rows / columns by
            // k_anchor_flags: 64-byte pieces of a byte array written from different CUs are partial lines)
        }
    }
}

// Anchor-outer form of the tiled kernel (round 3).  k_features_tiled walks rows outside and anchors inside: a row's na anchor
// distances are na scalar loads from na different cache lines (more SGPRs than a wave has, so in several dependent batches), paid
// for every (row, 64-column word) that holds a pair -- 152 ms on the thinned list of 100 000 points (12 % of the words' bits set,
// VALU 42 % busy).  Here a wave owns a 32-row x 64-column tile, keeps the tile's 32 + 32 running bounds per lane in registers
// and walks the ANCHORS outside: per anchor one coalesced read of the columns' distances and the 32 rows' distances as one
// contiguous scalar read (anchor-major D: consecutive points).  The tile is evaluated densely -- all N^2 / 2 pairs of 100 000 points
// x 60 anchors are ~30 ms of fp64 vector work -- and tiles without a pair are skipped; the pairs' positions come from the
// keep bitmap's popcount ranks as before.
#define FD_ROWS 32
__global__ __launch_bounds__(256) void k_features_dense(
    const uint64_t *__restrict__ K, const uint32_t *__restrict__ pref, int kw, const int32_t *__restrict__ low,
    const int64_t *__restrict__ rowstart, const double *__restrict__ Dt, int64_t nx, int na, const int32_t *__restrict__ cA,
    double *__restrict__ lb, double *__restrict__ ub, double *__restrict__ dad)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nrb = (int)((nx + FD_ROWS - 1) / FD_ROWS);
    const int64_t task = (int64_t)blockIdx.x * 4 + wave;   // row-block major: the four waves of a workgroup share their rows' reads
    const int64_t rb = task / kw;
    const int jb = (int)(task - rb * kw);
    if (rb >= nrb) return;
    const int64_t i_lo = rb * FD_ROWS;
    const int64_t i_base = min(i_lo, nx - FD_ROWS);   // the last block reads the 32 rows that end at nx (rows below i_lo are another block's)
    if ((int64_t)jb * 64 + 63 <= i_lo) return;        // no column of this word lies right of the block's first row
    // the rows' bitmap words, list positions and closest anchors, lane-parallel (row = lane & 31)
    const int64_t ir = i_base + (lane & 31);
    const uint64_t raw = ir >= i_lo ? K[ir * kw + jb] : 0ull;
    const int64_t posb = rowstart[ir] + (int64_t)pref[ir * kw + jb] - low[ir];
    const int cai_r = cA[ir];
    {
        uint64_t live = raw;   // the bits that are pairs of this tile: column > row (columns >= nx are never set)
        const int64_t d = ir - (int64_t)jb * 64;
        if (d >= 63) live = 0ull;
        else if (d >= 0) live &= ~((2ull << d) - 1ull);
        if (!__any(live != 0ull)) return;
    }
    const int64_t j = (int64_t)jb * 64 + lane;
    const int64_t jc = min(j, nx - 1);
    double l[FD_ROWS], u[FD_ROWS];
#pragma unroll
    for (int r = 0; r < FD_ROWS; ++r) { l[r] = 0.0; u[r] = INFINITY; }
    double dj = Dt[jc];
    for (int a = 0; a < na; ++a) {
        const double *__restrict__ di = Dt + (size_t)a * nx + i_base;   // uniform: one contiguous scalar read per anchor
        double d[FD_ROWS];
#pragma unroll
        for (int r = 0; r < FD_ROWS; ++r) d[r] = di[r];
        const double djn = Dt[(size_t)min(a + 1, na - 1) * nx + jc];   // the next anchor's column distances, ahead of the arithmetic
#pragma unroll
        for (int r = 0; r < FD_ROWS; ++r) {
            // (the machine's max / min directly: fmax / fmin re-canonicalise the running value -- one more instruction per bound,
            // row and anchor -- for signalling NaNs that distances never are; same results as k_features_tiled's on everything else)
            const double x = d[r] - dj, y = d[r] + dj;
            asm("v_max_f64 %0, %0, |%1|" : "+v"(l[r]) : "v"(x));
            asm("v_min_f64 %0, %0, %1" : "+v"(u[r]) : "v"(y));
        }
        dj = djn;
    }
    const int caj = cA[jc];
    const uint32_t raw_lo = (uint32_t)raw, raw_hi = (uint32_t)(raw >> 32);
    const uint32_t pos_lo = (uint32_t)posb, pos_hi = (uint32_t)((uint64_t)posb >> 32);
#pragma unroll
    for (int r = 0; r < FD_ROWS; ++r) {
        const int64_t i = i_base + r;
        const uint64_t bits = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)raw_hi, r) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)raw_lo, r);
        const bool mine = j > i && j < nx && ((bits >> lane) & 1ull);
        if (!__any(mine)) continue;
        const int64_t pb = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)pos_hi, r) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)pos_lo, r));
        const int cai = __builtin_amdgcn_readlane(cai_r, r);
        if (mine) {
            const int64_t pos = pb + __popcll(bits & ((1ull << lane) - 1ull));
            __builtin_nontemporal_store(l[r], &lb[pos]);
            __builtin_nontemporal_store(u[r], &ub[pos]);
            __builtin_nontemporal_store((Dt[(size_t)caj * nx + i] + Dt[(size_t)cai * nx + j]) / 2, &dad[pos]);
        }
    }
}

// is_anchor / not_computed of the pairs that touch an anchor (annchor.py:286-289): thread per (anchor, point)
__global__ void k_anchor_flags(const int32_t *__restrict__ A, int nA, int64_t nx, const uint64_t *__restrict__ K,
                               const uint32_t *__restrict__ pref, int kw, const int32_t *__restrict__ low,
                               const int64_t *__restrict__ rowstart, uint8_t *__restrict__ anc, uint8_t *__restrict__ ncm)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)nA * nx) return;
    const int64_t a = A[t / nx], o = t % nx;
    if (a == o) return;
    const int64_t i = a < o ? a : o, j = a < o ? o : a;
    const uint64_t bits = K[i * kw + (j >> 6)];
    if (!((bits >> (j & 63)) & 1ull)) return;
    const int64_t pos = rowstart[i] + ((int64_t)pref[i * kw + (j >> 6)] + __popcll(bits & ((1ull << (j & 63)) - 1ull)) - low[i]);
    anc[pos] = 1;
    ncm[pos] = 0;
}

extern "C" int annchor_compute_features(annchor_ctx *c)
{
    if (!c) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->n > 0, ANNCHOR_EINVAL, "locality not built");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->n;
    ANN_TRY(ann_reserve(c, c->lb, 8 * n));
    ANN_TRY(ann_reserve(c, c->ub, 8 * n));
    ANN_TRY(ann_reserve(c, c->dad, 8 * n));
    ANN_TRY(ann_reserve(c, c->RA, 8 * n));
    ANN_TRY(ann_reserve(c, c->prob, 8 * n));
    ANN_TRY(ann_reserve(c, c->anc, n));
    ANN_TRY(ann_reserve(c, c->ncm, n));
    ANN_TRY(ann_reserve(c, c->label, n));
    ANN_CHECK_HIP(c, hipEventRecord(c->call_a, c->stream));
    {
        // algorithmic bytes per pair: 8 (ij) + 3*8 (lb, ub, dad) + 2 (masks)
        ProfScope ps(c, "bounds_dad_features", (double)n * 34.0);
        static const long long tiled_min = getenv("ANNCHOR_FEATURES_TILED_MIN") ? atoll(getenv("ANNCHOR_FEATURES_TILED_MIN")) : (8ll << 20);
        if (c->have_bitmap && c->n >= tiled_min && c->na <= 64) {
            const int kw = (int)((c->nx + 63) / 64);
            const int64_t tasks = ((c->nx + FT_ROWS - 1) / FT_ROWS) * kw;
            const unsigned grid = (unsigned)((tasks + 3) / 4);
#define FT_LAUNCH(NAM) k_features_tiled<NAM><<<grid, 256, 0, c->stream>>>(c->Kbits.as<uint64_t>(), c->Kpref.as<uint32_t>(), kw, \
                c->low.as<int32_t>(), c->rowstart.as<int64_t>(), c->Dt.as<double>(), c->nx, c->na, c->cA.as<int32_t>(), \
                c->anchorRank.as<int32_t>(), c->lb.as<double>(), c->ub.as<double>(), c->dad.as<double>(), c->anc.as<uint8_t>(), \
                c->ncm.as<uint8_t>())
            ANN_CHECK_HIP(c, hipMemsetAsync(c->anc.p, 0, n, c->stream));
            ANN_CHECK_HIP(c, hipMemsetAsync(c->ncm.p, 1, n, c->stream));
            if (c->nA > 0)
                k_anchor_flags<<<ann_blocks((int64_t)c->nA * c->nx, 256), 256, 0, c->stream>>>(
                    c->A.as<int32_t>(), c->nA, c->nx, c->Kbits.as<uint64_t>(), c->Kpref.as<uint32_t>(), kw, c->low.as<int32_t>(),
                    c->rowstart.as<int64_t>(), c->anc.as<uint8_t>(), c->ncm.as<uint8_t>());
            static const char *form = getenv("ANNCHOR_FEATURES_FORM");   // "tiled" / "dense" force a form
            // thinned lists take the anchor-outer form (dense tiles, empty ones skipped); complete lists the row-outer one
            const double density = (double)c->n / (0.5 * (double)c->nx * (double)(c->nx - 1));
            const bool dense_form = c->nx >= FD_ROWS && (form ? strcmp(form, "dense") == 0 : density < 0.5);
            if (dense_form) {
                const int64_t dtasks = ((c->nx + FD_ROWS - 1) / FD_ROWS) * kw;
                k_features_dense<<<(unsigned)((dtasks + 3) / 4), 256, 0, c->stream>>>(c->Kbits.as<uint64_t>(), c->Kpref.as<uint32_t>(), kw,
                    c->low.as<int32_t>(), c->rowstart.as<int64_t>(), c->Dt.as<double>(), c->nx, c->na, c->cA.as<int32_t>(),
                    c->lb.as<double>(), c->ub.as<double>(), c->dad.as<double>());
            } else
            if (c->na <= 8) FT_LAUNCH(8);
            else if (c->na <= 16) FT_LAUNCH(16);
            else if (c->na <= 24) FT_LAUNCH(24);
            else if (c->na <= 32) FT_LAUNCH(32);
            else if (c->na <= 48) FT_LAUNCH(48);
            else FT_LAUNCH(64);
#undef FT_LAUNCH
        } else
        k_features<<<ann_blocks(c->n, 256), 256, 0, c->stream>>>(c->ij.as<int2>(), c->n, c->Dt.as<double>(), c->nx, c->na,
                                                                c->cA.as<int32_t>(), c->anchorRank.as<int32_t>(),
                                                                c->lb.as<double>(), c->ub.as<double>(), c->dad.as<double>(),
                                                                c->anc.as<uint8_t>(), c->ncm.as<uint8_t>());
    }
    ANN_CHECK_HIP(c, hipEventRecord(c->call_b, c->stream));
    c->call_timed = true;
    ANN_CHECK_HIP(c, hipGetLastError());
    c->have_features = true;
    c->have_RA = false; c->sel_prepared = false;
    c->nsamp = 0;
    c->n_unc = c->n_unc_after_features; c->sel_prepared = false;   // (counted by build_locality; -1 on the query path: recount)
    return ANNCHOR_OK;
}

// ------------------------------------------------------------ sampler support
__global__ __launch_bounds__(256) void k_count_flags(const uint8_t *__restrict__ f, int64_t n, unsigned long long *out)
{
    // 16 flags per load (the arrays are 256-byte aligned)
    unsigned long long s = 0;
    const int64_t n16 = n >> 4;
    const uint4 *f16 = reinterpret_cast<const uint4 *>(f);
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n16; t += 4 * step) {
        uint4 v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ann_ldc(f16, t + e * step, n16);
        // flags are 0/1 bytes: the byte sum of a word is its popcount
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (t + e * step < n16) s += __popc(v[e].x) + __popc(v[e].y) + __popc(v[e].z) + __popc(v[e].w);
    }
    if (blockIdx.x == 0)
        for (int64_t t = (n16 << 4) + threadIdx.x; t < n; t += blockDim.x) s += f[t] != 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    // one atomic per workgroup: same-address atomics serialise at ~12.5 ns each (tools/microbench/atomics.hip)
    __shared__ unsigned long long ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0 && (ws[0] | ws[1] | ws[2] | ws[3])) atomicAdd(out, ws[0] + ws[1] + ws[2] + ws[3]);
}

extern "C" int annchor_count_uncomputed(annchor_ctx *c, int64_t *n_unc)
{
    if (!c || !n_unc) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->have_features, ANNCHOR_EINVAL, "features not computed");
    if (c->n_unc >= 0) { *n_unc = c->n_unc; return ANNCHOR_OK; }  // maintained incrementally by the stages below
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    ANN_TRY(ann_reserve(c, c->tmp2, 64));
    ANN_CHECK_HIP(c, hipMemsetAsync(c->tmp2.p, 0, 8, c->stream));
    int blocks = min(ann_blocks(c->n, 256 * 16 * 4), c->prop.multiProcessorCount);
    k_count_flags<<<blocks, 256, 0, c->stream>>>(c->ncm.as<uint8_t>(), c->n, c->tmp2.as<unsigned long long>());
    unsigned long long v = 0;
    ANN_TRY(ann_d2h(c, &v, c->tmp2.p, 8));
    *n_unc = (int64_t)v;
    c->n_unc = (int64_t)v;
    return ANNCHOR_OK;
}

extern "C" int annchor_kth_uncomputed_dad(annchor_ctx *c, const int64_t *ks, int32_t nk, double *out)
{
    if (!c || !ks || !out) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->have_features, ANNCHOR_EINVAL, "features not computed");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    return ann_kth_smallest(c, c->dad.as<double>(), c->ncm.as<uint8_t>(), c->n, ks, nk, out);
}


// sampler bin: lo <= x < hi   (utils.py:547-549); -1 if none
__device__ __forceinline__ int sampler_bin(const BinEdges &b, double x)
{
    for (int k = 0; k < b.nb; ++k)
        if (x >= b.e[k] && x < b.e[k + 1]) return k;
    return -1;
}

#define BC_THREADS 1024
#define BC_ITEMS 8
#define BC_TILE (BC_THREADS * BC_ITEMS)
// Few fat workgroups (their closing atomics on the shared bin counters serialise), eight
// unconditional loads per thread in flight, per-wave LDS counters.
__device__ __forceinline__ void bin_counts_body(const double *__restrict__ dad, const uint8_t *__restrict__ ncm, int64_t n,
                                                const BinEdges &be, unsigned long long *__restrict__ counts)
{
    __shared__ unsigned int lc[BC_THREADS / 64][MAXBINS];
    const int wave = threadIdx.x >> 6;
    for (int t = threadIdx.x; t < (BC_THREADS / 64) * MAXBINS; t += BC_THREADS) (&lc[0][0])[t] = 0;
    __syncthreads();
    const int64_t ntiles = (n + BC_TILE - 1) / BC_TILE;
    uint32_t rc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        double v[BC_ITEMS];
        uint8_t f[BC_ITEMS];
#pragma unroll
        for (int k = 0; k < BC_ITEMS; ++k) {
            const int64_t t = tile * BC_TILE + k * BC_THREADS + threadIdx.x;
            v[k] = ann_ldc(dad, t, n);
            f[k] = ann_ldc(ncm, t, n);
            if (t >= n) f[k] = 0;
        }
        if (be.nb <= 8) {
            // few partitions (the default sampler has 7): membership counted in registers with
            // compare-adds, no LDS atomics in the streaming loop
#pragma unroll
            for (int k = 0; k < BC_ITEMS; ++k) {
                if (!f[k]) continue;
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    rc[b] += (b < be.nb && v[k] >= be.e[b] && v[k] < be.e[b + 1]) ? 1u : 0u;
            }
        } else {
#pragma unroll
            for (int k = 0; k < BC_ITEMS; ++k) {
                const int b = f[k] ? sampler_bin(be, v[k]) : -1;
                if (b >= 0) atomicAdd(&lc[wave][b], 1u);
            }
        }
    }
    if (be.nb <= 8) {
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            uint32_t x = rc[b];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
            if ((threadIdx.x & 63) == 0 && x) lc[wave][b] = x;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < be.nb) {
        unsigned long long s = 0;
        for (int w = 0; w < BC_THREADS / 64; ++w) s += lc[w][threadIdx.x];
        if (s) atomicAdd(&counts[threadIdx.x], s);
    }
}
__global__ __launch_bounds__(BC_THREADS) void k_bin_counts(const double *__restrict__ dad, const uint8_t *__restrict__ ncm,
                                                          int64_t n, BinEdges be, unsigned long long *__restrict__ counts)
{
    bin_counts_body(dad, ncm, n, be, counts);
}

__global__ __launch_bounds__(BC_THREADS) void k_bin_counts_dev(const double *__restrict__ dad, const uint8_t *__restrict__ ncm,
                                                              int64_t n, SamplerStats *__restrict__ st)
{
    __shared__ BinEdges be;
    if (threadIdx.x == 0) be = st->be;
    __syncthreads();
    bin_counts_body(dad, ncm, n, be, st->counts);
}

static int load_bins(annchor_ctx *c, const double *bins, int32_t nbins, BinEdges &be);

// ------------------------------------------------------------ hashed stratified sampling
// DeviceStratifiedSampler (annchor_amd/samplers.py): the stratified draw of Sampler.sample_partition
// (reference annchor/samplers.py:44-73, utils.py:543-578) with an ORDER-FREE random choice.  The
// reference draws `want` members of a partition through a sequential shuffle of its whole population
// (np.random.choice(..., replace=False): an MT19937 stream as long as the pair list, walked by one
// thread).  Here every not-computed pair gets a key = splitmix64(seed_key ^ position) and a partition
// keeps its `want` smallest keys: the same distribution (a uniform random subset), a pure function of
// (seed, position), computable by any number of threads in any order.
//   k_hs_collect  pairs with key <= T_b (T_b = 1.5 * want_b / count_b of the key range + slack) -> per-partition lists
//   k_hs_finish   one workgroup per partition: its `want` smallest (key, position), output by position
__host__ __device__ __forceinline__ unsigned long long ann_splitmix64(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

#define HS_CAP 4096   // list entries per partition (want <= HS_CAP / 2)
struct HsParams {
    unsigned long long seed_key;
    unsigned long long thr[MAXBINS];   // key threshold per partition (~0: take every member)
};

// (The first form looked every pair's partition up -- a search over the edges -- and then loaded that partition's threshold
// from the kernel arguments: a dependent global load per pair, 0.55 ms for 127 M pairs where the column streams in 0.25.  A
// pair whose key exceeds the LARGEST threshold cannot be taken whatever its partition: nearly all of them, and for those the
// hash is all there is to do; edges and thresholds wait in LDS for the rest.)
#define HS_U 8
__global__ __launch_bounds__(256) void k_hs_collect(const double *__restrict__ dad, const uint8_t *__restrict__ ncm, int64_t n,
                                                   BinEdges be, HsParams hp, unsigned long long *__restrict__ lkey,
                                                   int32_t *__restrict__ lpos, uint32_t *__restrict__ lcnt)
{
    __shared__ double se[MAXBINS + 1];
    __shared__ unsigned long long sthr[MAXBINS];
    for (int t = threadIdx.x; t <= be.nb; t += blockDim.x) se[t] = be.e[t];
    for (int t = threadIdx.x; t < be.nb; t += blockDim.x) sthr[t] = hp.thr[t];
    unsigned long long tmax = 0;
    for (int b = 0; b < be.nb; ++b) tmax = hp.thr[b] > tmax ? hp.thr[b] : tmax;   // (uniform)
    __syncthreads();
    const int nb = be.nb;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * HS_U;
    for (int64_t p0 = (int64_t)blockIdx.x * blockDim.x * HS_U + threadIdx.x; p0 < n; p0 += stride) {
        double v[HS_U];
        uint8_t f[HS_U];
#pragma unroll
        for (int e = 0; e < HS_U; ++e) { v[e] = ann_ldc(dad, p0 + (int64_t)e * blockDim.x, n); f[e] = ann_ldc(ncm, p0 + (int64_t)e * blockDim.x, n); }
#pragma unroll
        for (int e = 0; e < HS_U; ++e) {
            const int64_t p = p0 + (int64_t)e * blockDim.x;
            if (p >= n || !f[e]) continue;
            const unsigned long long key = ann_splitmix64(hp.seed_key ^ (unsigned long long)p);
            if (key > tmax) continue;
            int b = -1;
            for (int k = 0; k < nb; ++k)
                if (v[e] >= se[k] && v[e] < se[k + 1]) { b = k; break; }   // (sampler_bin over the LDS copy)
            if (b >= 0 && key <= sthr[b]) {
                const uint32_t o = atomicAdd(&lcnt[b], 1u);
                if (o < HS_CAP) { lkey[(size_t)b * HS_CAP + o] = key; lpos[(size_t)b * HS_CAP + o] = (int32_t)p; }
            }
        }
    }
}

// block b: rank the list of partition b by (key, position); members of rank < want are the sample;
// they are written in ascending position order at out + offset[b]
__global__ __launch_bounds__(1024) void k_hs_finish(const unsigned long long *__restrict__ lkey, const int32_t *__restrict__ lpos,
                                                   const uint32_t *__restrict__ lcnt, const int32_t *__restrict__ want,
                                                   const int32_t *__restrict__ offset, int32_t *__restrict__ out, int32_t *__restrict__ got)
{
    // Two bitonic sorts in LDS: the candidates by (key, position), then the positions of the `want` smallest.  (The first
    // version ranked every candidate against every other: 179 us per call for ~1100 candidates per partition.)
    __shared__ unsigned long long sk[HS_CAP];
    __shared__ int32_t sp[HS_CAP];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int cnt = (int)min(lcnt[b], (uint32_t)HS_CAP);
    const int w = min(want[b], cnt);
    if (tid == 0) got[b] = lcnt[b] > HS_CAP ? -1 : w;
    int P = 64;
    while (P < cnt) P <<= 1;
    for (int t = tid; t < P; t += 1024) {
        sk[t] = t < cnt ? lkey[(size_t)b * HS_CAP + t] : ~0ull;
        sp[t] = t < cnt ? lpos[(size_t)b * HS_CAP + t] : 0x7fffffff;
    }
    __syncthreads();
    for (int k2 = 2; k2 <= P; k2 <<= 1)
        for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
            for (int i = tid; i < P; i += 1024) {
                const int l = i ^ j2;
                if (l > i) {
                    const unsigned long long ka = sk[i], kb = sk[l];
                    const int32_t pa = sp[i], pb = sp[l];
                    const bool gt = ka > kb || (ka == kb && pa > pb);
                    if (gt == ((i & k2) == 0)) { sk[i] = kb; sk[l] = ka; sp[i] = pb; sp[l] = pa; }
                }
            }
            __syncthreads();
        }
    // the chosen positions, ascending inside the partition
    int Q = 64;
    while (Q < w) Q <<= 1;
    for (int t = tid; t < Q; t += 1024) if (t >= w) sp[t] = 0x7fffffff;   // (w <= HS_CAP / 2: the tail of sp is free)
    __syncthreads();
    for (int k2 = 2; k2 <= Q; k2 <<= 1)
        for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
            for (int i = tid; i < Q; i += 1024) {
                const int l = i ^ j2;
                if (l > i) {
                    const int32_t pa = sp[i], pb = sp[l];
                    if ((pa > pb) == ((i & k2) == 0)) { sp[i] = pb; sp[l] = pa; }
                }
            }
            __syncthreads();
        }
    for (int t = tid; t < w; t += 1024) out[offset[b] + t] = sp[t];
}

// the choice itself: sample positions (int32, partition by partition) left on the device at *d_out_p
// got[b] against want[b] on the device: a short or overflowing list raises the sticky sample-step flag (dev_flags[0] = 2)
__global__ void k_hs_check(const int32_t *__restrict__ got, const int32_t *__restrict__ want, int nbins, int32_t *__restrict__ flags)
{
    if (threadIdx.x == 0 && blockIdx.x == 0)
        for (int b = 0; b < nbins; ++b)
            if (got[b] != want[b]) flags[0] = 2;
}

// no_wait: one attempt, its outcome checked on the device (the thresholds leave 1.5 x want + 64 expected keys: a miss is a
// 1e-9 event, answered by the caller's retry the waiting way when it finds the flag)
static int hash_sample_device(annchor_ctx *c, const double *bins, int32_t nbins, const int64_t *counts, const int64_t *want,
                              uint64_t seed_key, int32_t **d_out_p, int64_t *n_out, bool no_wait = false)
{
    BinEdges be;
    ANN_TRY(load_bins(c, bins, nbins, be));
    HsParams hp;
    hp.seed_key = seed_key;
    std::vector<int32_t> h_want(nbins), h_off(nbins);
    int64_t total = 0;
    for (int b = 0; b < nbins; ++b) {
        ANN_REQUIRE(c, want[b] >= 0 && want[b] <= HS_CAP / 2, ANNCHOR_ELIMIT, "at most %d samples per partition", HS_CAP / 2);
        const int64_t w = std::min(want[b], counts[b]);
        h_want[b] = (int32_t)w;
        h_off[b] = (int32_t)total;
        total += w;
        if (counts[b] <= want[b] || counts[b] <= 0) hp.thr[b] = ~0ull;   // the whole partition
        else {
            // expected hits = 1.5 * want + 64 of a uniform key: P(fewer than want) is negligible; a short list is retried below
            const long double frac = std::min<long double>(1.0L, (1.5L * (long double)want[b] + 64.0L) / (long double)counts[b]);
            hp.thr[b] = frac >= 1.0L ? ~0ull : (unsigned long long)(frac * 18446744073709551615.0L);
        }
    }
    ANN_TRY(ann_reserve(c, c->hs_key, sizeof(unsigned long long) * (size_t)nbins * HS_CAP));
    ANN_TRY(ann_reserve(c, c->hs_pos, sizeof(int32_t) * (size_t)nbins * HS_CAP));
    ANN_TRY(ann_reserve(c, c->hs_misc, sizeof(int32_t) * (size_t)(4 * MAXBINS + total + 16)));
    uint32_t *d_cnt = c->hs_misc.as<uint32_t>();
    int32_t *d_want = c->hs_misc.as<int32_t>() + MAXBINS, *d_off = d_want + MAXBINS, *d_got = d_off + MAXBINS, *d_out = d_got + MAXBINS;
    ANN_TRY(ann_h2d(c, d_want, h_want.data(), sizeof(int32_t) * nbins));
    ANN_TRY(ann_h2d(c, d_off, h_off.data(), sizeof(int32_t) * nbins));
    std::vector<int32_t> h_got(nbins);
    for (int attempt = 0; attempt < 8; ++attempt) {
        ANN_CHECK_HIP(c, hipMemsetAsync(d_cnt, 0, sizeof(uint32_t) * MAXBINS, c->stream));
        {
            ProfScope ps(c, "hashed_sample_collect", (double)c->n * 9.0);
            const int blocks = (int)std::min<int64_t>(ann_blocks(c->n, 256 * HS_U), (int64_t)c->prop.multiProcessorCount * 8);
            k_hs_collect<<<blocks, 256, 0, c->stream>>>(c->dad.as<double>(), c->ncm.as<uint8_t>(), c->n, be, hp, c->hs_key.as<unsigned long long>(),
                                                       c->hs_pos.as<int32_t>(), d_cnt);
            k_hs_finish<<<nbins, 1024, 0, c->stream>>>(c->hs_key.as<unsigned long long>(), c->hs_pos.as<int32_t>(), d_cnt, d_want, d_off, d_out,
                                                      d_got);
        }
        ANN_CHECK_HIP(c, hipGetLastError());
        if (no_wait) {
            ANN_TRY(ann_dev_flags(c));
            k_hs_check<<<1, 64, 0, c->stream>>>(d_got, d_want, nbins, c->dev_flags.as<int32_t>());
            break;
        }
        ANN_TRY(ann_d2h(c, h_got.data(), d_got, sizeof(int32_t) * nbins));
        bool redo = false;
        for (int b = 0; b < nbins; ++b) {
            if (h_got[b] < 0) {                       // list overflow: tighten the threshold
                hp.thr[b] = hp.thr[b] / 2; redo = true;
            } else if (h_got[b] < h_want[b]) {        // too few keys under the threshold: widen it
                hp.thr[b] = hp.thr[b] > (~0ull >> 1) ? ~0ull : hp.thr[b] * 2; redo = true;
            }
        }
        if (!redo) break;
        ANN_REQUIRE(c, attempt < 7, ANNCHOR_ESTATE, "hashed sampling did not settle");
    }
    *d_out_p = d_out;
    *n_out = total;
    return ANNCHOR_OK;
}

extern "C" int annchor_hash_sample(annchor_ctx *c, const double *bins, int32_t nbins, const int64_t *counts, const int64_t *want,
                                   uint64_t seed_key, int64_t *positions, int64_t *n_out)
{
    if (!c || !bins || !counts || !want || !positions || !n_out) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->have_features, ANNCHOR_EINVAL, "features not computed");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    int32_t *d_out = nullptr;
    int64_t total = 0;
    ANN_TRY(hash_sample_device(c, bins, nbins, counts, want, seed_key, &d_out, &total));
    std::vector<int32_t> h_out((size_t)total);
    if (total) ANN_TRY(ann_d2h(c, h_out.data(), d_out, sizeof(int32_t) * (size_t)total));
    for (int64_t t = 0; t < total; ++t) positions[t] = h_out[(size_t)t];
    *n_out = total;
    return ANNCHOR_OK;
}

static int load_bins(annchor_ctx *c, const double *bins, int32_t nbins, BinEdges &be)
{
    ANN_REQUIRE(c, nbins >= 1 && nbins <= MAXBINS, ANNCHOR_ELIMIT, "1..%d partitions supported", MAXBINS);
    be.nb = nbins;
    for (int k = 0; k <= nbins; ++k) be.e[k] = bins[k];
    return ANNCHOR_OK;
}

extern "C" int annchor_bin_counts(annchor_ctx *c, const double *bins, int32_t nbins, int64_t *counts)
{
    if (!c || !bins || !counts) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->have_features, ANNCHOR_EINVAL, "features not computed");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    BinEdges be;
    ANN_TRY(load_bins(c, bins, nbins, be));
    ANN_TRY(ann_reserve(c, c->tmp2, 8 * MAXBINS));
    ANN_CHECK_HIP(c, hipMemsetAsync(c->tmp2.p, 0, 8 * MAXBINS, c->stream));
    const int64_t ntiles = (c->n + BC_TILE - 1) / BC_TILE;
    const int blocks = (int)(ntiles <= 256 ? ntiles : std::min<int64_t>(1024, std::max<int64_t>(256, ntiles / 4)));
    {
        ProfScope ps(c, "sampler_bin_counts", (double)c->n * 9.0);
        k_bin_counts<<<blocks, BC_THREADS, 0, c->stream>>>(c->dad.as<double>(), c->ncm.as<uint8_t>(), c->n, be,
                                                   c->tmp2.as<unsigned long long>());
    }
    return ann_d2h(c, counts, c->tmp2.p, 8 * (size_t)nbins);
}

extern "C" int annchor_sampler_stats(annchor_ctx *c, const int64_t *ks, int32_t n_partitions, double *q, double *edges,
                                     int64_t *counts, int32_t *fused)
{
    if (!c || !ks || !q || !edges || !counts || !fused) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->have_features, ANNCHOR_EINVAL, "features not computed");
    ANN_REQUIRE(c, n_partitions >= 2 && n_partitions <= MAXBINS, ANNCHOR_ELIMIT, "2..%d partitions supported", MAXBINS);
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    *fused = 0;
    const unsigned long long *d_prefix = nullptr;
    const int *d_unfinished = nullptr;
    ANN_TRY(ann_reserve(c, c->sstats, sizeof(SamplerStats)));
    SamplerStats *st = c->sstats.as<SamplerStats>();
    {
        // (quantiles -> edges -> zeroed counters: written by the selection's finishing workgroup itself)
        Sel2Epilogue ep;
        memset(&ep, 0, sizeof ep);
        ep.kind = 1; ep.st = st; ep.nparts = n_partitions;
        ANN_TRY(ann_kth_async(c, c->dad.as<double>(), c->ncm.as<uint8_t>(), c->n, ks, 2, &d_prefix, &d_unfinished, &ep));
    }
    if (d_prefix) {
        const int64_t ntiles = (c->n + BC_TILE - 1) / BC_TILE;
        const int blocks = (int)(ntiles <= 256 ? ntiles : std::min<int64_t>(1024, std::max<int64_t>(256, ntiles / 4)));
        {
            ProfScope ps(c, "sampler_bin_counts", (double)c->n * 9.0);
            k_bin_counts_dev<<<blocks, BC_THREADS, 0, c->stream>>>(c->dad.as<double>(), c->ncm.as<uint8_t>(), c->n, st);
        }
        ANN_CHECK_HIP(c, hipGetLastError());
        SamplerStats h;
        if (c->park_refine) {
            // fit(): the refinement launch of the iteration that just chose its candidates is queued BEHIND this download -- the host
            // waits for the statistics only and starts the draw while the GPU refines (queueing it after the wait left the GPU idle
            // for the host's wake-up and the trip through the host language: ~70 us per iteration)
            c->park_refine = false;
            ANN_TRY(ann_d2h_then(c, &h, st, sizeof h, annchor_refine_candidates));
        } else
        ANN_TRY(ann_d2h(c, &h, st, sizeof h));
        const int unfinished = h.unfinished;
        ann_kth_async_done(c);
        if (!unfinished) {
            q[0] = h.q[0]; q[1] = h.q[1];
            for (int b = 0; b <= n_partitions; ++b) edges[b] = h.be.e[b];
            for (int b = 0; b < n_partitions; ++b) counts[b] = (int64_t)h.counts[b];
            *fused = 1;
            return ANNCHOR_OK;
        }
    }
    // long lists (the sampled bracket is checked by the host) and mixed buckets the finishing workgroup could not hold: the
    // quantiles by the general route; the caller computes its edges and asks for the counts
    return ann_kth_smallest(c, c->dad.as<double>(), c->ncm.as<uint8_t>(), c->n, ks, 2, q);
}

// ---- rank-in-bin selection: a blocked scan of per-bin membership counts
#define RB_THREADS 256
#define RB_ITEMS 8
#define RB_TILE (RB_THREADS * RB_ITEMS)

__global__ __launch_bounds__(RB_THREADS) void k_rb_count(const double *__restrict__ dad, const uint8_t *__restrict__ ncm,
                                                        int64_t n, BinEdges be, uint32_t *__restrict__ blkcnt)
{
    __shared__ unsigned int lc[MAXBINS];
    for (int t = threadIdx.x; t < MAXBINS; t += blockDim.x) lc[t] = 0;
    __syncthreads();
    int64_t base = (int64_t)blockIdx.x * RB_TILE;
    double v[RB_ITEMS];
    uint8_t f[RB_ITEMS];
#pragma unroll
    for (int k = 0; k < RB_ITEMS; ++k) {   // all loads in flight, masked afterwards
        const int64_t t = base + (int64_t)k * RB_THREADS + threadIdx.x;
        v[k] = ann_ldc(dad, t, n);
        f[k] = ann_ldc(ncm, t, n);
        if (t >= n) f[k] = 0;
    }
    if (be.nb <= 8) {   // few partitions: register compare-adds, one LDS add per wave and partition
        uint32_t rc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < RB_ITEMS; ++k) {
            if (!f[k]) continue;
#pragma unroll
            for (int b = 0; b < 8; ++b)
                rc[b] += (b < be.nb && v[k] >= be.e[b] && v[k] < be.e[b + 1]) ? 1u : 0u;
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            uint32_t x = rc[b];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
            if ((threadIdx.x & 63) == 0 && x) atomicAdd(&lc[b], x);
        }
    } else {
#pragma unroll
        for (int k = 0; k < RB_ITEMS; ++k) {
            const int b = f[k] ? sampler_bin(be, v[k]) : -1;
            if (b >= 0) atomicAdd(&lc[b], 1u);
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < be.nb; t += blockDim.x) blkcnt[(size_t)blockIdx.x * be.nb + t] = lc[t];
}

__global__ __launch_bounds__(256) void k_rb_scan(uint32_t *__restrict__ blkcnt, int nblocks, int nb)
{
    // block b scans bin b over the tiles (exclusive, in tile order)
    __shared__ uint32_t wsum[4];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (int base = 0; base < nblocks; base += 256) {
        const int k = base + threadIdx.x;
        const uint32_t v = k < nblocks ? blkcnt[(size_t)k * nb + b] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
        }
        __syncthreads();
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t pre = 0, tot = 0;
        for (int w = 0; w < 4; ++w) { if (w < wave) pre += wsum[w]; tot += wsum[w]; }
        if (k < nblocks) blkcnt[(size_t)k * nb + b] = carry + pre + inc - v;
        carry += tot;
    }
}

// slotmap[binbase[b] + rank] = request slot or -1.  A tile is RB_TILE positions; each of the
// four waves owns 512 consecutive ones (eight coalesced 64-wide chunks, all loaded up front),
// counts its bins into LDS so that every wave knows where its ranks start, then ranks its own
// elements with ballots (position order inside a chunk = lane order).
struct RbBase { int64_t v[MAXBINS]; };   // the partitions' offsets in the slot map by value (binbase == nullptr): no upload in front of the launch
__global__ __launch_bounds__(RB_THREADS) void k_rb_emit(const double *__restrict__ dad, const uint8_t *__restrict__ ncm, int64_t n,
                                                       BinEdges be, const uint32_t *__restrict__ blkoff,
                                                       const int64_t *__restrict__ binbase, RbBase rb, const int32_t *__restrict__ slotmap,
                                                       int64_t *__restrict__ positions)
{
    constexpr int CH = RB_TILE / RB_THREADS;   // chunks of 64 per wave
    __shared__ uint32_t wcnt[RB_THREADS / 64][MAXBINS];
    __shared__ int64_t sbase[MAXBINS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = threadIdx.x; t < (RB_THREADS / 64) * MAXBINS; t += RB_THREADS) (&wcnt[0][0])[t] = 0;
    if ((int)threadIdx.x < be.nb) sbase[threadIdx.x] = binbase ? binbase[threadIdx.x] : rb.v[threadIdx.x];
    __syncthreads();
    const int64_t wbase = (int64_t)blockIdx.x * RB_TILE + (int64_t)wave * (CH * 64);
    int b[CH];
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
        const int64_t t = wbase + ch * 64 + lane;
        const double v = ann_ldc(dad, t, n);
        const uint8_t f = ann_ldc(ncm, t, n);
        b[ch] = (t < n && f) ? sampler_bin(be, v) : -1;
    }
#pragma unroll
    for (int ch = 0; ch < CH; ++ch)
        if (b[ch] >= 0) atomicAdd(&wcnt[wave][b[ch]], 1u);
    __syncthreads();
    // lane k carries the running count of bin k (nb <= 64)
    uint32_t myrun = 0;
    if (lane < be.nb) {
        myrun = blkoff[(size_t)blockIdx.x * be.nb + lane];
        for (int w = 0; w < wave; ++w) myrun += wcnt[w][lane];
    }
    uint32_t rank[CH];
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
        rank[ch] = 0;
        for (int k = 0; k < be.nb; ++k) {
            const unsigned long long m = __ballot(b[ch] == k);
            if (!m) continue;
            const uint32_t rk = __shfl(myrun, k);
            if (b[ch] == k) rank[ch] = rk + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (lane == k) myrun += (uint32_t)__popcll(m);
        }
    }
#pragma unroll
    for (int ch = 0; ch < CH; ++ch)
        if (b[ch] >= 0) {
            const int32_t slot = slotmap[sbase[b[ch]] + rank[ch]];
            if (slot >= 0) positions[slot] = wbase + ch * 64 + lane;
        }
}

// also initialises the request's output position to -1 ("not found") and, when given, the caller's
// error flag to 0: two memset launches less per sampling step
__global__ void k_scatter_slots(const int32_t *__restrict__ bin_of, const int64_t *__restrict__ ranks, int64_t nreq,
                                const int64_t *__restrict__ binbase, int32_t *__restrict__ slotmap,
                                int64_t *__restrict__ positions, int32_t *__restrict__ flag)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nreq) {
        slotmap[binbase[bin_of[t]] + ranks[t]] = (int32_t)t;
        positions[t] = -1;
    }
    if (t == 0 && flag) *flag = 0;
}

// positions (int64, device: c->stage_out) of the requested (bin, rank) entries; counts_opt = the
// per-bin totals if the caller already holds them (annchor_bin_counts), else they are recounted
static int select_by_rank_device(annchor_ctx *c, const double *bins, int32_t nbins, const int64_t *counts_opt,
                                 const int32_t *bin_of, const int64_t *ranks, int64_t nreq, int32_t *zero_flag = nullptr)
{
    BinEdges be;
    ANN_TRY(load_bins(c, bins, nbins, be));
    const int64_t n = c->n;
    const int nblocks = ann_blocks(n, RB_TILE);
    // per-bin totals (host needs them to lay out the slot map)
    std::vector<int64_t> counts((size_t)nbins), base((size_t)nbins + 1, 0);
    if (counts_opt) for (int b = 0; b < nbins; ++b) counts[(size_t)b] = counts_opt[b];
    else ANN_TRY(annchor_bin_counts(c, bins, nbins, counts.data()));
    for (int b = 0; b < nbins; ++b) base[(size_t)b + 1] = base[(size_t)b] + counts[(size_t)b];
    for (int64_t t = 0; t < nreq; ++t) {
        ANN_REQUIRE(c, bin_of[t] >= 0 && bin_of[t] < nbins, ANNCHOR_EINVAL, "bin index out of range");
        ANN_REQUIRE(c, ranks[t] >= 0 && ranks[t] < counts[(size_t)bin_of[t]], ANNCHOR_EINVAL, "rank %lld outside bin %d (size %lld)",
                    (long long)ranks[t], bin_of[t], (long long)counts[(size_t)bin_of[t]]);
    }
    const int64_t total = base[(size_t)nbins];
    ANN_TRY(ann_reserve(c, c->blk_cnt, sizeof(uint32_t) * (size_t)nblocks * nbins));
    ANN_TRY(ann_reserve(c, c->tmp0, sizeof(int32_t) * (size_t)(total + 1)));  // slotmap
    ANN_TRY(ann_reserve(c, c->stage_out, sizeof(int64_t) * (size_t)nreq));
    // ranks | bin base offsets | bin ids: one staged upload (three small copies were three launches of ~5 us)
    ANN_TRY(ann_reserve(c, c->stage_in, (sizeof(int32_t) + sizeof(int64_t)) * (size_t)nreq + sizeof(int64_t) * (size_t)(nbins + 1) + 64));
    int64_t *d_ranks = c->stage_in.as<int64_t>();
    int64_t *d_base = d_ranks + nreq;
    int32_t *d_binof = reinterpret_cast<int32_t *>(d_base + nbins + 1);
    {
        std::vector<unsigned char> stage(sizeof(int64_t) * (size_t)(nreq + nbins + 1) + sizeof(int32_t) * (size_t)nreq);
        memcpy(stage.data(), ranks, sizeof(int64_t) * (size_t)nreq);
        memcpy(stage.data() + sizeof(int64_t) * (size_t)nreq, base.data(), sizeof(int64_t) * (size_t)(nbins + 1));
        memcpy(stage.data() + sizeof(int64_t) * (size_t)(nreq + nbins + 1), bin_of, sizeof(int32_t) * (size_t)nreq);
        // (in pieces of one pinned ring slot: a larger pageable copy would make ann_h2d wait for the stream)
        for (size_t o = 0; o < stage.size(); o += annchor_ctx::PIN_SLOT_BYTES)
            ANN_TRY(ann_h2d(c, reinterpret_cast<unsigned char *>(d_ranks) + o, stage.data() + o,
                            std::min(stage.size() - o, (size_t)annchor_ctx::PIN_SLOT_BYTES)));
    }
    ANN_CHECK_HIP(c, hipMemsetAsync(c->tmp0.p, 0xff, sizeof(int32_t) * (size_t)(total + 1), c->stream));
    {
        ProfScope ps(c, "sampler_select_by_rank", (double)n * 18.0);
        k_scatter_slots<<<ann_blocks(nreq, 256), 256, 0, c->stream>>>(d_binof, d_ranks, nreq, d_base,
                                                                     c->tmp0.as<int32_t>(), c->stage_out.as<int64_t>(), zero_flag);
        k_rb_count<<<nblocks, RB_THREADS, 0, c->stream>>>(c->dad.as<double>(), c->ncm.as<uint8_t>(), n, be,
                                                         c->blk_cnt.as<uint32_t>());
        k_rb_scan<<<nbins, 256, 0, c->stream>>>(c->blk_cnt.as<uint32_t>(), nblocks, nbins);
        k_rb_emit<<<nblocks, RB_THREADS, 0, c->stream>>>(c->dad.as<double>(), c->ncm.as<uint8_t>(), n, be, c->blk_cnt.as<uint32_t>(),
                                                d_base, RbBase(), c->tmp0.as<int32_t>(), c->stage_out.as<int64_t>());
    }
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}

extern "C" int annchor_select_by_rank(annchor_ctx *c, const double *bins, int32_t nbins, const int32_t *bin_of,
                                      const int64_t *ranks, int64_t nreq, int64_t *positions)
{
    if (!c || !bins || (nreq > 0 && (!bin_of || !ranks || !positions))) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->have_features, ANNCHOR_EINVAL, "features not computed");
    if (nreq == 0) return ANNCHOR_OK;
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    ANN_TRY(select_by_rank_device(c, bins, nbins, nullptr, bin_of, ranks, nreq));
    return ann_d2h(c, positions, c->stage_out.p, sizeof(int64_t) * (size_t)nreq);
}

// ------------------------------------------------------------ sample plumbing
__global__ void k_gather_features(const int32_t *__restrict__ pos, int64_t m, const double *__restrict__ lb,
                                  const double *__restrict__ ub, const double *__restrict__ dad,
                                  const uint8_t *__restrict__ anc, double *__restrict__ out)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    int32_t p = pos[t];
    out[4 * t + 0] = lb[p];
    out[4 * t + 1] = ub[p];
    out[4 * t + 2] = dad[p];
    out[4 * t + 3] = (double)anc[p];
}

static int upload_positions(annchor_ctx *c, const int64_t *pos, int64_t m, DevBuf &dst)
{
    std::vector<int32_t> p32((size_t)m);
    for (int64_t t = 0; t < m; ++t) {
        ANN_REQUIRE(c, pos[t] >= 0 && pos[t] < c->n, ANNCHOR_EINVAL, "pair position %lld out of range", (long long)pos[t]);
        p32[(size_t)t] = (int32_t)pos[t];
    }
    ANN_TRY(ann_reserve(c, dst, sizeof(int32_t) * (size_t)m));
    return ann_h2d(c, dst.p, p32.data(), sizeof(int32_t) * (size_t)m);
}

extern "C" int annchor_gather_features(annchor_ctx *c, const int64_t *pos, int64_t m, double *feats)
{
    if (!c || (m > 0 && (!pos || !feats))) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->have_features, ANNCHOR_EINVAL, "features not computed");
    if (m == 0) return ANNCHOR_OK;
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    ANN_TRY(upload_positions(c, pos, m, c->tmp3));
    ANN_TRY(ann_reserve(c, c->stage_out, sizeof(double) * 4 * (size_t)m));
    k_gather_features<<<ann_blocks(m, 256), 256, 0, c->stream>>>(c->tmp3.as<int32_t>(), m, c->lb.as<double>(),
                                                                c->ub.as<double>(), c->dad.as<double>(),
                                                                c->anc.as<uint8_t>(), c->stage_out.as<double>());
    ANN_CHECK_HIP(c, hipGetLastError());
    return ann_d2h(c, feats, c->stage_out.p, sizeof(double) * 4 * (size_t)m);
}

__global__ void k_clear_flags(const int32_t *__restrict__ pos, int64_t m, uint8_t *__restrict__ ncm)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < m) ncm[pos[t]] = 0;
}

// the same, plus the sample's distances and the error flag copied next to the staged positions /
// feature rows (annchor_sample_pairs hands everything back in one transfer)
__global__ void k_clear_flags_stage(const int32_t *__restrict__ pos, int64_t m, uint8_t *__restrict__ ncm,
                                    const double *__restrict__ sy, const int32_t *__restrict__ bad, double *__restrict__ st_y,
                                    int64_t *__restrict__ st_bad)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < m) { ncm[pos[t]] = 0; st_y[t] = sy[t]; }
    if (t == 0) *st_bad = *bad;
}

__global__ void k_clear_flags_sticky(const int32_t *__restrict__ pos, int64_t m, uint8_t *__restrict__ ncm,
                                     const int32_t *__restrict__ bad, int32_t *__restrict__ flags)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < m) ncm[pos[t]] = 0;
    if (t == 0 && *bad) flags[0] = 1;
}

extern "C" int annchor_evaluate_samples(annchor_ctx *c, const int64_t *pos, int64_t m, double *sample_y)
{
    if (!c || (m > 0 && (!pos || !sample_y))) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->have_features, ANNCHOR_EINVAL, "features not computed");
    ANN_REQUIRE(c, c->metric != ANNCHOR_METRIC_NONE, ANNCHOR_EINVAL, "no device metric bound to this context");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    ANN_TRY(upload_positions(c, pos, m, c->spos));
    ANN_TRY(ann_reserve(c, c->sy, sizeof(double) * (size_t)m));
    c->nsamp = m;
    c->n_unc = -1; c->sel_prepared = false;   // recount lazily: a custom sampler may hand back already-computed or repeated pairs
    if (m == 0) return ANNCHOR_OK;
    PairSource src;
    src.ij = c->ij.as<int2>();
    src.idx = c->spos.as<int32_t>();
    src.n = m;
    ANN_CHECK_HIP(c, hipEventRecord(c->call_a, c->stream));
    ANN_TRY(ann_metric_launch(c, src, c->sy.as<double>(), nullptr, nullptr));
    ANN_CHECK_HIP(c, hipEventRecord(c->call_b, c->stream));
    c->call_timed = true;
    k_clear_flags<<<ann_blocks(m, 256), 256, 0, c->stream>>>(c->spos.as<int32_t>(), m, c->ncm.as<uint8_t>());
    ANN_CHECK_HIP(c, hipGetLastError());
    return ann_d2h(c, sample_y, c->sy.p, sizeof(double) * (size_t)m);
}

// k_pos_to_i32 + k_gather_features in one launch (the device-resident sampling step)
// CLEAR: the samples also leave the not-computed mask and a missing entry raises the sticky flag at once (what k_clear_flags_sticky
// does behind the metric launch: the metric kernels do not read the mask)
template <bool CLEAR = false>
__global__ void k_pos_gather(const int64_t *__restrict__ pos, int64_t m, int32_t *__restrict__ out, int32_t *__restrict__ bad,
                             const double *__restrict__ lb, const double *__restrict__ ub, const double *__restrict__ dad,
                             const uint8_t *__restrict__ anc, double *__restrict__ feats, uint8_t *__restrict__ ncm = nullptr,
                             int32_t *__restrict__ flags = nullptr)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const int64_t p64 = pos[t];
    if (p64 < 0) { *bad = 1; if (CLEAR) atomicMax(&flags[0], 1); }   // a requested (bin, rank) entry was not found (a trace that gave up -- 3 -- stays visible)
    const int32_t p = (int32_t)(p64 < 0 ? 0 : p64);
    if (CLEAR) ncm[p] = 0;
    out[t] = p;
    feats[4 * t + 0] = lb[p];
    feats[4 * t + 1] = ub[p];
    feats[4 * t + 2] = dad[p];
    feats[4 * t + 3] = (double)anc[p];
}

__global__ void k_pos_to_i32(const int64_t *__restrict__ pos, int64_t m, int32_t *__restrict__ out, int32_t *__restrict__ bad)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    if (pos[t] < 0) *bad = 1;   // a requested (bin, rank) entry was not found
    out[t] = (int32_t)(pos[t] < 0 ? 0 : pos[t]);
}

// The built-in sampling step in one call (get_sample, annchor.py:313-343): (bin, rank) -> pair
// positions, their feature rows, their exact distances, not_computed_mask cleared -- the
// positions never leave the device between the steps and the host waits once.  `counts` are
// the per-bin totals of annchor_bin_counts for the same edges.  The drawn pairs are distinct
// not-computed pairs by construction, so the cached count drops by exactly nreq.
extern "C" int annchor_sample_pairs(annchor_ctx *c, const double *bins, int32_t nbins, const int64_t *counts,
                                    const int32_t *bin_of, const int64_t *ranks, int64_t nreq, int64_t *positions,
                                    double *feats, double *sample_y)
{
    if (!c || !bins || !counts || (nreq > 0 && (!bin_of || !ranks || !positions || !feats || !sample_y))) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->have_features, ANNCHOR_EINVAL, "features not computed");
    ANN_REQUIRE(c, c->metric != ANNCHOR_METRIC_NONE, ANNCHOR_EINVAL, "no device metric bound to this context");
    c->nsamp = nreq;
    if (nreq == 0) return ANNCHOR_OK;
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    // one staging block for everything the host gets back: positions | feature rows | distances | flag
    const size_t stage_bytes = sizeof(double) * (6 * (size_t)nreq + 1);
    ANN_TRY(ann_reserve(c, c->stage_out, stage_bytes));
    ANN_TRY(ann_reserve(c, c->spos, sizeof(int32_t) * (size_t)nreq + 16));
    ANN_TRY(ann_reserve(c, c->sy, sizeof(double) * (size_t)nreq));
    int32_t *bad = c->spos.as<int32_t>() + nreq;
    ANN_TRY(select_by_rank_device(c, bins, nbins, counts, bin_of, ranks, nreq, bad));   // positions -> stage_out[0 .. nreq); *bad = 0
    double *st_feats = c->stage_out.as<double>() + nreq, *st_y = st_feats + 4 * (size_t)nreq;
    int64_t *st_bad = reinterpret_cast<int64_t *>(st_y + nreq);
    k_pos_to_i32<<<ann_blocks(nreq, 256), 256, 0, c->stream>>>(c->stage_out.as<int64_t>(), nreq, c->spos.as<int32_t>(), bad);
    k_gather_features<<<ann_blocks(nreq, 256), 256, 0, c->stream>>>(c->spos.as<int32_t>(), nreq, c->lb.as<double>(),
                                                                   c->ub.as<double>(), c->dad.as<double>(), c->anc.as<uint8_t>(),
                                                                   st_feats);
    PairSource src;
    src.ij = c->ij.as<int2>();
    src.idx = c->spos.as<int32_t>();
    src.n = nreq;
    ANN_CHECK_HIP(c, hipEventRecord(c->call_a, c->stream));
    ANN_TRY(ann_metric_launch(c, src, c->sy.as<double>(), nullptr, nullptr));
    ANN_CHECK_HIP(c, hipEventRecord(c->call_b, c->stream));
    c->call_timed = true;
    k_clear_flags_stage<<<ann_blocks(nreq, 256), 256, 0, c->stream>>>(c->spos.as<int32_t>(), nreq, c->ncm.as<uint8_t>(),
                                                                     c->sy.as<double>(), bad, st_y, st_bad);
    ANN_CHECK_HIP(c, hipGetLastError());
    int64_t h_bad = 0;
    if (c->pin && stage_bytes <= annchor_ctx::PIN_DL_BYTES) {   // one transfer, one wait
        unsigned char *slot = c->pin + (size_t)annchor_ctx::PIN_SLOTS * annchor_ctx::PIN_SLOT_BYTES;
        ANN_CHECK_HIP(c, hipMemcpyAsync(slot, c->stage_out.p, stage_bytes, hipMemcpyDeviceToHost, c->stream));
        ANN_CHECK_HIP(c, ann_sync(c, __func__));
        memcpy(positions, slot, sizeof(int64_t) * (size_t)nreq);
        memcpy(feats, slot + sizeof(double) * (size_t)nreq, sizeof(double) * 4 * (size_t)nreq);
        memcpy(sample_y, slot + sizeof(double) * 5 * (size_t)nreq, sizeof(double) * (size_t)nreq);
        memcpy(&h_bad, slot + sizeof(double) * 6 * (size_t)nreq, sizeof h_bad);
    } else {
        ANN_TRY(ann_d2h(c, positions, c->stage_out.p, sizeof(int64_t) * (size_t)nreq));
        ANN_TRY(ann_d2h(c, feats, st_feats, sizeof(double) * 4 * (size_t)nreq));
        ANN_TRY(ann_d2h(c, sample_y, st_y, sizeof(double) * (size_t)nreq));
        ANN_TRY(ann_d2h(c, &h_bad, st_bad, sizeof h_bad));
    }
    ANN_REQUIRE(c, !h_bad, ANNCHOR_ESTATE, "sample_pairs: a (bin, rank) entry does not exist (stale counts?)");
    if (c->n_unc >= 0) c->n_unc -= nreq;
    c->sel_prepared = false;
    return ANNCHOR_OK;
}

__global__ void k_i32_to_i64_pos(const int32_t *__restrict__ in, int64_t m, int64_t *__restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < m) out[t] = in[t];
}

// The same sampling step with everything left on the device (the device-resident model fit of model.hip follows):
// positions in spos (int32), feature rows in sfeat [m][4], exact distances in sy; a missing (bin, rank) entry raises
// the sticky flag dev_flags[0], read with the selection stage's final state.
extern "C" int annchor_sample_pairs_device(annchor_ctx *c, const double *bins, int32_t nbins, const int64_t *counts,
                                           const int32_t *bin_of, const int64_t *ranks, int64_t nreq)
{
    if (!c || !bins || !counts || (nreq > 0 && (!bin_of || !ranks))) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->have_features, ANNCHOR_EINVAL, "features not computed");
    ANN_REQUIRE(c, c->metric != ANNCHOR_METRIC_NONE, ANNCHOR_EINVAL, "no device metric bound to this context");
    c->nsamp = nreq;
    if (nreq == 0) return ANNCHOR_OK;
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    ANN_TRY(ann_reserve(c, c->stage_out, sizeof(int64_t) * (size_t)nreq));
    ANN_TRY(ann_reserve(c, c->spos, sizeof(int32_t) * (size_t)nreq + 16));
    ANN_TRY(ann_reserve(c, c->sy, sizeof(double) * (size_t)nreq));
    ANN_TRY(ann_reserve(c, c->sfeat, sizeof(double) * 4 * (size_t)nreq));
    ANN_TRY(ann_dev_flags(c));
    int32_t *bad = c->spos.as<int32_t>() + nreq;
    ANN_TRY(select_by_rank_device(c, bins, nbins, counts, bin_of, ranks, nreq, bad));   // positions -> stage_out[0 .. nreq); *bad = 0
    k_pos_gather<false><<<ann_blocks(nreq, 256), 256, 0, c->stream>>>(c->stage_out.as<int64_t>(), nreq, c->spos.as<int32_t>(), bad,
                                                              c->lb.as<double>(), c->ub.as<double>(), c->dad.as<double>(),
                                                              c->anc.as<uint8_t>(), c->sfeat.as<double>());
    PairSource src;
    src.ij = c->ij.as<int2>();
    src.idx = c->spos.as<int32_t>();
    src.n = nreq;
    ANN_CHECK_HIP(c, hipEventRecord(c->call_a, c->stream));
    ANN_TRY(ann_metric_launch(c, src, c->sy.as<double>(), nullptr, nullptr));
    ANN_CHECK_HIP(c, hipEventRecord(c->call_b, c->stream));
    c->call_timed = true;
    k_clear_flags_sticky<<<ann_blocks(nreq, 256), 256, 0, c->stream>>>(c->spos.as<int32_t>(), nreq, c->ncm.as<uint8_t>(), bad,
                                                                      c->dev_flags.as<int32_t>());
    ANN_CHECK_HIP(c, hipGetLastError());
    if (c->n_unc >= 0) c->n_unc -= nreq;
    c->sel_prepared = false;
    return ANNCHOR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The legacy sampler's draw with its backward trace on the DEVICE (round 4).  np.random.choice(ixmask, size, replace=False)
// of the reference (utils.py:543-578) is permutation(c)[:k] of NumPy's legacy stream: a Fisher-Yates shuffle of the whole bin
// (c - 1 rejection-sampled partners) of which only the first k entries are used.  The host library walks the stream (the
// rejection scan is sequential: 0.22 ms at C2) and used to undo the swaps for the k kept positions on helper threads -- the
// two large bins' traces ran until 0.5 ms after the scan started, with the GPU idle behind them.  Here the host only scans:
// every bin's partners J go to pinned memory and are copied to the device on a side stream as soon as the bin is scanned, and
// the trace is three kernels:
//   * undoing the swaps (i, j_i) in reverse execution order i = 1 .. c - 1 moves a kept slot only when i is its position or
//     j_i is; for i >= k the first case cannot occur (position i is untouched before its own step), so a slot at position q
//     moves to the FIRST i >= k, i > q with j_i == q -- and again from there.  `next[v]` = min { i >= k : j_i == v, i != v } is
//     one atomicMin per step (k_tr_steps); a slot's path is a short chase through it (~ln(c / k) hops, k_tr_chase);
//   * the first k - 1 steps involve kept positions only: every slot scans them once (partners in LDS, all slots in lockstep):
//     at i == p the slot moves to j_p, afterwards to i whenever j_i is where it sits.
// The chase's result (the rank of the chosen element inside the bin's population) is written straight into the slot map the
// rank -> pair-position kernels read: no ranks on the host, none uploaded.  Same partners, same trace: the samples and their
// order are those of annchor_legacy_choice_ranks (tests compare the two on random populations).
#define TR_KMAX 8192   // kept entries per bin the chain kernel holds in LDS
struct TraceBins {
    int64_t c[MAXBINS];       // population
    int64_t k[MAXBINS];       // entries kept (min(want, c))
    int64_t joff[MAXBINS];    // offset of the bin's partners in J / of its next[] (-1: not shuffled, the whole bin is taken)
    int64_t base[MAXBINS];    // offset of the bin in the slot map (prefix of the populations)
    int64_t offs[MAXBINS];    // offset of the bin's requests (prefix of k)
    int64_t step0[MAXBINS + 1];   // prefix of the phase-2 steps (c - k per shuffled bin): the scatter's flat index
    int nbins;
};

// One launch, two kinds of workgroups.  Blocks [0, nbins): the first k - 1 steps of a bin (partners in LDS, every kept slot scans
// them once, all slots in lockstep) -> the slots' positions at time k - 1 (q1).  The other blocks: next[joff + v] = min { i in [k, c) :
// j_i == v, i != v } by one atomicMin per step (J holds the partners in stream order: J[t] belongs to i = c - 1 - t), a chunk of
// TR_CHUNK steps at a time.  The two halves are independent; the chase (k_tr_chase) needs both.
// STREAMED form (progress != nullptr): the launch is queued BEFORE the host walks the stream.  J is the host's pinned buffer and
// *progress the number of partner words the host has completed so far (all shuffled bins in order, c - 1 words per bin; release
// store every PUB_EVERY words, hostrng.hip): a workgroup waits until the words of its chunk (of its whole bin for the lockstep
// half) are there, reads them over the fabric and goes on -- the trace finishes a few microseconds after the scan instead of a
// last upload + 45 us of scatter later.  One thread polls (system scope), napping about a third of what the host still needs at
// 8 words / ns; a wait longer than the limit raises flags[0] = 3 and lets the workgroup go (the chase terminates on any next[]:
// its chains only ascend).
#define TR_CHUNK 16384
__device__ __forceinline__ bool tr_wait(const unsigned long long *progress, unsigned long long need, long long timeout_ticks,
                                        int32_t *flags)
{
    const long long t0 = wall_clock64();
    for (;;) {
        // (the POLL is relaxed: an acquire at every poll would invalidate the XCD's L2 again and again under the refinement
        // kernels of a fit's second draw.  The one SYSTEM-scope acquire behind the successful poll is what orders the
        // workgroup's plain loads of J after the host's release store: chunk and bin boundaries are not line-aligned, so a
        // line fetched for a neighbouring chunk before the host finished it must not be served from a device cache -- that
        // the pinned buffer is uncached today is a property of the mapping, not of the memory model.  One fence per chunk /
        // bin; the workgroup barrier behind tr_wait carries it to the other waves of the workgroup.)
        const unsigned long long p = __hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (p >= need) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); return true; }
        if (wall_clock64() - t0 > timeout_ticks) {
            if (flags) atomicMax(&flags[0], 3);
            return false;
        }
        const int naps = (int)min((unsigned long long)((need - p) >> 14), 64ull) + 1;
        for (int q = 0; q < naps; ++q) __builtin_amdgcn_s_sleep(32);
    }
}
// partner words complete before bin b starts (joff counts 33 words of slack and header per shuffled bin on top of its c - 1)
__device__ __forceinline__ int64_t tr_words_before(const TraceBins &tb, int b)
{
    int sb = 0;
    for (int o = 0; o < b; ++o) sb += tb.joff[o] >= 0;
    return tb.joff[b] - 33ll * sb;
}
__global__ __launch_bounds__(1024) void k_tr_steps(TraceBins tb, const uint32_t *__restrict__ J, uint32_t *__restrict__ next,
                                                  uint32_t *__restrict__ q1, int scatter_blocks, int chunk,
                                                  const unsigned long long *__restrict__ progress, long long timeout_ticks,
                                                  int32_t *__restrict__ flags)
{
    extern __shared__ uint32_t jl[];   // jl[i] = j_i for i in [1, k)
    __shared__ int ok_s;
    if ((int)blockIdx.x >= tb.nbins) {
        const int64_t total = tb.step0[tb.nbins];
        for (int64_t g0 = (int64_t)(blockIdx.x - tb.nbins) * chunk; g0 < total; g0 += (int64_t)scatter_blocks * chunk) {
            const int64_t g1 = min(g0 + (int64_t)chunk, total);
            if (progress) {
                if (threadIdx.x == 0) {
                    int b = 0;
                    while (g1 - 1 >= tb.step0[b + 1]) ++b;
                    ok_s = tr_wait(progress, (unsigned long long)(tr_words_before(tb, b) + (g1 - 1 - tb.step0[b]) + 1), timeout_ticks, flags);
                }
                __syncthreads();
                const bool ok = ok_s != 0;
                __syncthreads();
                if (!ok) return;
            }
            for (int64_t g = g0 + threadIdx.x; g < g1; g += blockDim.x) {
                int b = 0;
                while (g >= tb.step0[b + 1]) ++b;
                const int64_t t = g - tb.step0[b];            // t in [0, c - k): i = c - 1 - t >= k
                const uint32_t i = (uint32_t)(tb.c[b] - 1 - t);
                const uint32_t j = J[tb.joff[b] + t];
                if (j != i) atomicMin(&next[tb.joff[b] + j], i);
            }
        }
        return;
    }
    const int b = blockIdx.x;
    const int64_t c = tb.c[b], k = tb.k[b];
    const bool shuffled = tb.joff[b] >= 0;
    if (progress && shuffled) {
        if (threadIdx.x == 0) ok_s = tr_wait(progress, (unsigned long long)(tr_words_before(tb, b) + c - 1), timeout_ticks, flags);
        __syncthreads();
        if (!ok_s) return;
    }
    if (shuffled)
        for (int64_t i = 1 + threadIdx.x; i < k; i += blockDim.x) jl[i] = J[tb.joff[b] + (c - 1 - i)];
    __syncthreads();
    for (int64_t p0 = 0; p0 < k; p0 += blockDim.x) {
        const int64_t p = p0 + threadIdx.x;
        uint32_t q = (uint32_t)p;
        if (shuffled) {
            // nothing happens to a slot before its own step; at i == p it moves to j_p, afterwards to i whenever j_i is where it sits
            int64_t i = p0 < 1 ? 1 : p0;
            for (; i + 8 <= k; i += 8) {
                uint32_t jv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) jv[u] = jl[i + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (i + u == p) q = jv[u];
                    else if (i + u > p && jv[u] == q) q = (uint32_t)(i + u);
                }
            }
            for (; i < k; ++i) {
                const uint32_t j = jl[i];
                if (i == p) q = j;
                else if (i > p && j == q) q = (uint32_t)i;
            }
        }
        if (p < k) q1[tb.offs[b] + p] = q;
    }
}

// one thread per kept slot: from its position at time k - 1 through next[] to the end; the rank goes into the slot map
__global__ __launch_bounds__(256) void k_tr_chase(TraceBins tb, const uint32_t *__restrict__ next, const uint32_t *__restrict__ q1, int64_t nreq,
                                                 int32_t *__restrict__ slotmap, int64_t *__restrict__ positions, int32_t *__restrict__ flag)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0 && flag) *flag = 0;
    if (t >= nreq) return;
    int b = 0;
    while (b + 1 < tb.nbins && t >= tb.offs[b + 1]) ++b;
    uint32_t q = q1[t];
    if (tb.joff[b] >= 0) {
        const uint32_t *nx = next + tb.joff[b];
        for (;;) {
            const uint32_t n2 = nx[q];
            if (n2 == 0xffffffffu) break;
            q = n2;
        }
    }
    slotmap[tb.base[b] + q] = (int32_t)t;
    positions[t] = -1;
}

static hipStream_t g_draw_copy_stream[16] = {};
static hipEvent_t g_draw_copy_event[16] = {};
static uint32_t *g_draw_pin = nullptr;     // [0, TR_PIN_HEAD): the scan's progress word; the partners behind it
static size_t g_draw_pin_words = 0;
static std::mutex g_draw_mu;
#define TR_PIN_HEAD 64

struct DrawUpload {
    annchor_ctx *c;
    hipStream_t copy;
    uint32_t *d_J;
    const int64_t *counts;
    const int64_t *joff;
    int rc;
};
static void draw_after_bin(int b, void *user)
{
    DrawUpload *u = static_cast<DrawUpload *>(user);
    if (hipMemcpyAsync(u->d_J + u->joff[b], g_draw_pin + TR_PIN_HEAD + u->joff[b], sizeof(uint32_t) * (size_t)u->counts[b], hipMemcpyHostToDevice,
                       u->copy) != hipSuccess)
        u->rc = ANNCHOR_EHIP;
}

static void trace_tb_fill(TraceBins &tb, int nbins, const int64_t *counts, const int64_t *want, int64_t *nreq, int64_t *total, int64_t *jwords,
                          int64_t *kmax, bool *ok)
{
    memset(&tb, 0, sizeof tb);
    tb.nbins = nbins;
    *nreq = *total = *jwords = *kmax = 0;
    *ok = false;
    for (int b = 0; b < nbins; ++b) {
        if (counts[b] < 0 || want[b] < 0 || counts[b] >= (1ll << 31)) return;
        tb.c[b] = counts[b];
        tb.k[b] = std::min(counts[b], want[b]);
        tb.base[b] = *total;
        tb.offs[b] = *nreq;
        const bool shuffled = counts[b] >= want[b] && counts[b] >= 2;
        // a shuffled bin of which nothing is kept (k = 0) still consumes its c - 1 stream words on the host, but its phase-2
        // walk would cover t = c - 1, one word past the partners: leave such draws to the host trace (ok stays false)
        if (shuffled && tb.k[b] < 1) return;
        tb.joff[b] = shuffled ? *jwords : -1;
        if (shuffled) *jwords += counts[b] + 32;
        *kmax = std::max(*kmax, tb.k[b]);
        *nreq += tb.k[b];
        *total += counts[b];
    }
    int64_t s = 0;
    for (int b = 0; b < nbins; ++b) { tb.step0[b] = s; if (tb.joff[b] >= 0) s += tb.c[b] - tb.k[b]; }
    tb.step0[nbins] = s;
    *ok = *kmax <= TR_KMAX;
}

// The host's scan and the device's trace of one draw.  `queue_rest` enqueues what follows the trace on the context's stream (the
// rank -> position kernels, the metric launch).  Streamed form (default): the trace kernel goes to the side stream FIRST and reads the
// partners from pinned memory while the host produces them, the chase and `queue_rest` are queued behind it, and the host's scan
// is the last thing the call does -- nothing is left to enqueue when it ends, and during a fit's second draw the trace runs beside
// the refinement kernels instead of behind them.  ANNCHOR_DRAW_STREAM=0 (or more than 32 M partner words): partners uploaded bin by
// bin on the side stream as each bin is scanned, the trace and `queue_rest` queued after the scan.
// The draw buffers (next[], q1, the pinned partners) belong to one draw at a time: a caller holds valid `counts` only after the
// previous draw's mask update has reached the host, i.e. after its trace.
template <typename Rest>
static int draw_scan_and_trace(annchor_ctx *c, const TraceBins &tb, uint32_t seed, const int64_t *counts, const int64_t *want, int64_t jwords,
                               int64_t kmax, int64_t nreq, int32_t *bad, int32_t *flags, Rest queue_rest)
{
    const int nbins = tb.nbins;
    const int64_t steps = tb.step0[nbins];
    const int threads = (int)std::min<int64_t>(1024, std::max<int64_t>(256, (kmax + 63) / 64 * 64));
    const size_t lds = sizeof(uint32_t) * (size_t)(kmax + 1);
    if (lds > 64 * 1024)
        ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)k_tr_steps, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    uint32_t *q1 = c->draw_q1.as<uint32_t>();
    std::lock_guard<std::mutex> lk(g_draw_mu);
    hipStream_t &copy = g_draw_copy_stream[c->device];
    hipEvent_t &ev = g_draw_copy_event[c->device];
    if (!copy) {
        ANN_CHECK_HIP(c, hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
        ANN_CHECK_HIP(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    for (int dv = 0; dv < 16; ++dv)   // (the previous draws' readers -- of any device -- have left the pinned buffer)
        if (g_draw_copy_event[dv]) ANN_CHECK_HIP(c, hipEventSynchronize(g_draw_copy_event[dv]));
    if (g_draw_pin_words < (size_t)jwords + TR_PIN_HEAD) {
        if (g_draw_pin) (void)hipHostFree(g_draw_pin);
        g_draw_pin = nullptr;
        g_draw_pin_words = 0;
        const size_t words = (size_t)jwords + (size_t)jwords / 4 + 4096 + TR_PIN_HEAD;
        ANN_CHECK_HIP(c, hipHostMalloc((void **)&g_draw_pin, sizeof(uint32_t) * words, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent));
        g_draw_pin_words = words;
    }
    unsigned long long *progress = reinterpret_cast<unsigned long long *>(g_draw_pin);
    uint32_t *pinJ = g_draw_pin + TR_PIN_HEAD;
    const char *e_st = getenv("ANNCHOR_DRAW_STREAM");
    void *dev_pin = nullptr;
    bool streamed = !(e_st && atoi(e_st) == 0) && jwords <= (32ll << 20) && jwords > 0;
    if (streamed && hipHostGetDevicePointer(&dev_pin, g_draw_pin, 0) != hipSuccess) { (void)hipGetLastError(); streamed = false; }
    if (streamed) {
        const char *e_to = getenv("ANNCHOR_DRAW_STREAM_TIMEOUT_MS");
        const long long ticks = (e_to ? atoll(e_to) : 5000ll) * 100000ll;   // wall_clock64: 100 MHz
        __atomic_store_n(progress, 0ull, __ATOMIC_RELEASE);
        const int scatter_blocks = steps > 0 ? (int)std::min<int64_t>((steps + TR_CHUNK - 1) / TR_CHUNK, 64) : 0;
        ANN_CHECK_HIP(c, hipMemsetAsync(c->draw_next.p, 0xff, sizeof(uint32_t) * (size_t)jwords, copy));
        k_tr_steps<<<nbins + scatter_blocks, threads, lds, copy>>>(tb, reinterpret_cast<const uint32_t *>(dev_pin) + TR_PIN_HEAD, c->draw_next.as<uint32_t>(), q1,
                                                                  scatter_blocks, TR_CHUNK, reinterpret_cast<const unsigned long long *>(dev_pin), ticks, flags);
        ANN_CHECK_HIP(c, hipGetLastError());
        ANN_CHECK_HIP(c, hipEventRecord(ev, copy));
        // (from here on the trace kernel is waiting for this thread: the scan cannot fail short of running out of memory.  What
        // follows the trace is queued AFTER the scan: while the host scans the GPU has nothing else to do in a fit's first draw,
        // and every launch in front of the scan -- ~5 us of host time each -- would only start the scan later)
        static const bool timing = getenv("ANNCHOR_SYNC_TIMING") != nullptr;
        const long long t_s = timing ? ann_now_ns() : 0;
        const int rc = ann_legacy_scan(seed, counts, want, nbins, pinJ, tb.joff, nullptr, nullptr, progress);
        if (timing) fprintf(stderr, "T scan streamed %lld %lld %lld\n", t_s, t_s, ann_now_ns());
        if (rc != ANNCHOR_OK) { __atomic_store_n(progress, ~0ull, __ATOMIC_RELEASE); return rc; }   // let the kernel go; the call fails
        ANN_CHECK_HIP(c, hipStreamWaitEvent(c->stream, ev, 0));
        {
            // (the profile sees the chase alone: the streamed kernel runs on the side stream and is as long as the host's scan)
            ProfScope ps(c, "sampler_draw_trace", (double)nreq * 16.0);
            k_tr_chase<<<ann_blocks(nreq, 256), 256, 0, c->stream>>>(tb, c->draw_next.as<uint32_t>(), q1, nreq, c->tmp0.as<int32_t>(),
                                                                   c->stage_out.as<int64_t>(), bad);
        }
        ANN_CHECK_HIP(c, hipGetLastError());
        return queue_rest();
    }
    ANN_TRY(ann_reserve(c, c->draw_J, sizeof(uint32_t) * (size_t)std::max<int64_t>(jwords, 1)));
    if (jwords) ANN_CHECK_HIP(c, hipMemsetAsync(c->draw_next.p, 0xff, sizeof(uint32_t) * (size_t)jwords, c->stream));
    DrawUpload up{c, copy, c->draw_J.as<uint32_t>(), counts, tb.joff, ANNCHOR_OK};
    static const bool timing = getenv("ANNCHOR_SYNC_TIMING") != nullptr;
    const long long t_s = timing ? ann_now_ns() : 0;
    ANN_TRY(ann_legacy_scan(seed, counts, want, nbins, pinJ, tb.joff, draw_after_bin, &up));
    if (timing) fprintf(stderr, "T scan uploaded %lld %lld %lld\n", t_s, t_s, ann_now_ns());
    ANN_REQUIRE(c, up.rc == ANNCHOR_OK, ANNCHOR_EHIP, "upload of the draw's partners failed");
    ANN_CHECK_HIP(c, hipEventRecord(ev, copy));
    ANN_CHECK_HIP(c, hipStreamWaitEvent(c->stream, ev, 0));
    const int scatter_blocks = steps > 0 ? (int)std::min<int64_t>(ann_blocks(steps, threads), (int64_t)c->prop.multiProcessorCount * 2) : 0;
    {
        ProfScope ps(c, "sampler_draw_trace", (double)jwords * 8.0);
        k_tr_steps<<<nbins + scatter_blocks, threads, lds, c->stream>>>(tb, c->draw_J.as<uint32_t>(), c->draw_next.as<uint32_t>(), q1, scatter_blocks, threads,
                                                                       nullptr, 0, nullptr);
        k_tr_chase<<<ann_blocks(nreq, 256), 256, 0, c->stream>>>(tb, c->draw_next.as<uint32_t>(), q1, nreq, c->tmp0.as<int32_t>(),
                                                               c->stage_out.as<int64_t>(), bad);
    }
    ANN_CHECK_HIP(c, hipGetLastError());
    return queue_rest();
}

// fills the slot map with -1 (and, for fill2_n > 0, a second array): one launch in front of the rank -> position kernels
__global__ __launch_bounds__(256) void k_fill_i32(int32_t *__restrict__ a, int64_t n, int32_t v)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; t < n; t += stride) {
        if (t + 4 <= n && (reinterpret_cast<uintptr_t>(a + t) & 15) == 0) *reinterpret_cast<int4 *>(a + t) = make_int4(v, v, v, v);
        else for (int64_t u = t; u < n && u < t + 4; ++u) a[u] = v;
    }
}

// The built-in sampling step with the legacy draw (SimpleStratifiedSampler): annchor_legacy_choice_ranks + annchor_sample_pairs_device
// in one call, the draw's trace on the device.  *taken = 0: a bin keeps more than TR_KMAX entries, the seed is outside the legacy
// range ... -- the caller draws on the host as before.  *n_out = number of samples (sum of min(want, counts)).
extern "C" int annchor_sample_pairs_device_draw(annchor_ctx *c, const double *bins, int32_t nbins, const int64_t *counts,
                                                const int64_t *want, uint32_t seed, int64_t *n_out, int32_t *taken)
{
    if (!c || !bins || !counts || !want || !n_out || !taken) return ANNCHOR_EINVAL;
    *taken = 0;
    ANN_REQUIRE(c, c->have_features, ANNCHOR_EINVAL, "features not computed");
    ANN_REQUIRE(c, c->metric != ANNCHOR_METRIC_NONE, ANNCHOR_EINVAL, "no device metric bound to this context");
    ANN_REQUIRE(c, nbins >= 1 && nbins <= MAXBINS, ANNCHOR_ELIMIT, "at most %d partitions", MAXBINS);
    if (c->device < 0 || c->device >= 16 || getenv("ANNCHOR_DRAW_TRACE_HOST")) return ANNCHOR_OK;
    TraceBins tb;
    int64_t nreq = 0, total = 0, jwords = 0, kmax = 0;
    bool ok = false;
    trace_tb_fill(tb, nbins, counts, want, &nreq, &total, &jwords, &kmax, &ok);
    if (!ok) return ANNCHOR_OK;
    c->nsamp = nreq;
    *n_out = nreq;
    *taken = 1;
    if (nreq == 0) return ANNCHOR_OK;
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    BinEdges be;
    ANN_TRY(load_bins(c, bins, nbins, be));
    const int64_t n = c->n;
    const int nblocks = ann_blocks(n, RB_TILE);
    ANN_TRY(ann_reserve(c, c->blk_cnt, sizeof(uint32_t) * (size_t)nblocks * nbins));
    ANN_TRY(ann_reserve(c, c->tmp0, sizeof(int32_t) * (size_t)(total + 1)));  // slotmap
    ANN_TRY(ann_reserve(c, c->stage_out, sizeof(int64_t) * (size_t)nreq));
    ANN_TRY(ann_reserve(c, c->spos, sizeof(int32_t) * (size_t)nreq + 16));
    ANN_TRY(ann_reserve(c, c->sy, sizeof(double) * (size_t)nreq));
    ANN_TRY(ann_reserve(c, c->sfeat, sizeof(double) * 4 * (size_t)nreq));
    ANN_TRY(ann_reserve(c, c->draw_next, sizeof(uint32_t) * (size_t)std::max<int64_t>(jwords, 1)));
    ANN_TRY(ann_reserve(c, c->draw_q1, sizeof(uint32_t) * (size_t)std::max<int64_t>(nreq, 1)));
    ANN_TRY(ann_dev_flags(c));
    int32_t *bad = c->spos.as<int32_t>() + nreq;
    RbBase rbase;
    for (int b = 0; b < nbins; ++b) rbase.v[b] = tb.base[b];
    // queued at once, ahead of the host's scan: the slot map's fill and the half of the rank -> position conversion that does not
    // need the ranks (per-tile counts of every partition and their scan)
    k_fill_i32<<<std::min(ann_blocks(total + 1, 1024), 1024), 256, 0, c->stream>>>(c->tmp0.as<int32_t>(), total + 1, -1);
    {
        ProfScope ps(c, "sampler_select_by_rank", (double)n * 9.0);
        k_rb_count<<<nblocks, RB_THREADS, 0, c->stream>>>(c->dad.as<double>(), c->ncm.as<uint8_t>(), n, be, c->blk_cnt.as<uint32_t>());
        k_rb_scan<<<nbins, 256, 0, c->stream>>>(c->blk_cnt.as<uint32_t>(), nblocks, nbins);
    }
    auto queue_rest = [&]() -> int {
        {
            ProfScope ps(c, "sampler_select_by_rank", (double)n * 9.0);
            k_rb_emit<<<nblocks, RB_THREADS, 0, c->stream>>>(c->dad.as<double>(), c->ncm.as<uint8_t>(), n, be, c->blk_cnt.as<uint32_t>(), nullptr, rbase,
                                                            c->tmp0.as<int32_t>(), c->stage_out.as<int64_t>());
        }
        // (the samples leave the not-computed mask here: the metric kernels do not read it)
        k_pos_gather<true><<<ann_blocks(nreq, 256), 256, 0, c->stream>>>(c->stage_out.as<int64_t>(), nreq, c->spos.as<int32_t>(), bad,
                                                                        c->lb.as<double>(), c->ub.as<double>(), c->dad.as<double>(),
                                                                        c->anc.as<uint8_t>(), c->sfeat.as<double>(), c->ncm.as<uint8_t>(),
                                                                        c->dev_flags.as<int32_t>());
        PairSource src;
        src.ij = c->ij.as<int2>();
        src.idx = c->spos.as<int32_t>();
        src.n = nreq;
        ANN_CHECK_HIP(c, hipEventRecord(c->call_a, c->stream));
        ANN_TRY(ann_metric_launch(c, src, c->sy.as<double>(), nullptr, nullptr));
        ANN_CHECK_HIP(c, hipEventRecord(c->call_b, c->stream));
        c->call_timed = true;
        ANN_CHECK_HIP(c, hipGetLastError());
        return ANNCHOR_OK;
    };
    ANN_TRY(draw_scan_and_trace(c, tb, seed, counts, want, jwords, kmax, nreq, bad, c->dev_flags.as<int32_t>(), queue_rest));
    if (c->n_unc >= 0) c->n_unc -= nreq;
    c->sel_prepared = false;
    return ANNCHOR_OK;
}

// The draw's ranks alone through the device trace (tests: against annchor_legacy_choice_ranks): ranks_out[offs_b + p] for every
// bin, bin by bin, in slot order.  Needs a context (device, stream), no data set.
extern "C" int annchor_legacy_choice_ranks_device(annchor_ctx *c, uint32_t seed, const int64_t *counts, const int64_t *want, int32_t nbins,
                                                  int64_t *ranks_out, int32_t *taken)
{
    if (!c || !counts || !want || !ranks_out || !taken) return ANNCHOR_EINVAL;
    *taken = 0;
    ANN_REQUIRE(c, nbins >= 1 && nbins <= MAXBINS, ANNCHOR_ELIMIT, "at most %d partitions", MAXBINS);
    if (c->device < 0 || c->device >= 16) return ANNCHOR_OK;
    TraceBins tb;
    int64_t nreq = 0, total = 0, jwords = 0, kmax = 0;
    bool ok = false;
    trace_tb_fill(tb, nbins, counts, want, &nreq, &total, &jwords, &kmax, &ok);
    if (!ok) return ANNCHOR_OK;
    *taken = 1;
    if (nreq == 0) return ANNCHOR_OK;
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    ANN_TRY(ann_reserve(c, c->tmp0, sizeof(int32_t) * (size_t)(total + 1)));
    ANN_TRY(ann_reserve(c, c->stage_out, sizeof(int64_t) * (size_t)nreq));
    ANN_TRY(ann_reserve(c, c->draw_next, sizeof(uint32_t) * (size_t)std::max<int64_t>(jwords, 1)));
    ANN_TRY(ann_reserve(c, c->draw_q1, sizeof(uint32_t) * (size_t)std::max<int64_t>(nreq, 1)));
    ANN_TRY(ann_dev_flags(c));
    k_fill_i32<<<std::min(ann_blocks(total + 1, 1024), 1024), 256, 0, c->stream>>>(c->tmp0.as<int32_t>(), total + 1, -1);
    ANN_TRY(draw_scan_and_trace(c, tb, seed, counts, want, jwords, kmax, nreq, nullptr, c->dev_flags.as<int32_t>(), []() -> int { return ANNCHOR_OK; }));
    // ranks from the slot map: slotmap[base_b + rank] = request index
    std::vector<int32_t> sm((size_t)total + 1);
    int32_t fl[16];
    ANN_TRY(ann_d2h2(c, sm.data(), c->tmp0.p, sizeof(int32_t) * (size_t)(total + 1), fl, c->dev_flags.p, sizeof fl));
    if (fl[0] == 3) {
        ANN_CHECK_HIP(c, hipMemsetAsync(c->dev_flags.p, 0, sizeof(int32_t) * 16, c->stream));
        ANN_REQUIRE(c, false, ANNCHOR_EHIP, "draw trace: the host's partner stream did not arrive in time");
    }
    for (int b = 0; b < nbins; ++b)
        for (int64_t r = 0; r < tb.c[b]; ++r) {
            const int32_t t = sm[(size_t)(tb.base[b] + r)];
            if (t >= 0) ranks_out[t] = r;
        }
    return ANNCHOR_OK;
}

// DeviceStratifiedSampler's step with everything left on the device (annchor_hash_sample_pairs without the transfer and
// without a host wait): positions int32 [m] (spos), feature rows (sfeat), exact distances (sy), masks updated; m =
// sum of min(want, counts), returned.
extern "C" int annchor_hash_sample_pairs_device(annchor_ctx *c, const double *bins, int32_t nbins, const int64_t *counts,
                                                const int64_t *want, uint64_t seed_key, int64_t *n_out)
{
    if (!c || !bins || !counts || !want || !n_out) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->have_features, ANNCHOR_EINVAL, "features not computed");
    ANN_REQUIRE(c, c->metric != ANNCHOR_METRIC_NONE, ANNCHOR_EINVAL, "no device metric bound to this context");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    int32_t *d_pos = nullptr;
    int64_t m = 0;
    ANN_TRY(hash_sample_device(c, bins, nbins, counts, want, seed_key, &d_pos, &m, true));
    *n_out = m;
    c->nsamp = m;
    if (m == 0) return ANNCHOR_OK;
    ANN_TRY(ann_reserve(c, c->spos, sizeof(int32_t) * (size_t)m + 16));
    ANN_TRY(ann_reserve(c, c->sy, sizeof(double) * (size_t)m));
    ANN_TRY(ann_reserve(c, c->sfeat, sizeof(double) * 4 * (size_t)m));
    int32_t *bad = c->spos.as<int32_t>() + m;
    ANN_CHECK_HIP(c, hipMemsetAsync(bad, 0, 4, c->stream));
    ANN_CHECK_HIP(c, hipMemcpyAsync(c->spos.p, d_pos, sizeof(int32_t) * (size_t)m, hipMemcpyDeviceToDevice, c->stream));
    k_gather_features<<<ann_blocks(m, 256), 256, 0, c->stream>>>(c->spos.as<int32_t>(), m, c->lb.as<double>(), c->ub.as<double>(),
                                                                c->dad.as<double>(), c->anc.as<uint8_t>(), c->sfeat.as<double>());
    PairSource src;
    src.ij = c->ij.as<int2>();
    src.idx = c->spos.as<int32_t>();
    src.n = m;
    ANN_CHECK_HIP(c, hipEventRecord(c->call_a, c->stream));
    ANN_TRY(ann_metric_launch(c, src, c->sy.as<double>(), nullptr, nullptr));
    ANN_CHECK_HIP(c, hipEventRecord(c->call_b, c->stream));
    c->call_timed = true;
    k_clear_flags_sticky<<<ann_blocks(m, 256), 256, 0, c->stream>>>(c->spos.as<int32_t>(), m, c->ncm.as<uint8_t>(), bad,
                                                                   c->dev_flags.as<int32_t>());
    ANN_CHECK_HIP(c, hipGetLastError());
    if (c->n_unc >= 0) c->n_unc -= m;
    c->sel_prepared = false;
    return ANNCHOR_OK;
}

// The last sampling step's arrays for the host (plugin-visible attributes; not on the fit path): positions int64 [m],
// feature rows float64 [m][4], distances float64 [m], unclipped predictions float64 [m] (NULL: skip).
extern "C" int annchor_download_samples(annchor_ctx *c, int64_t *positions, double *feats, double *sample_y, double *sample_predict)
{
    if (!c) return ANNCHOR_EINVAL;
    const int64_t m = c->nsamp;
    if (m == 0) return ANNCHOR_OK;
    ANN_REQUIRE(c, c->sfeat.p && c->spos.p && c->sy.p, ANNCHOR_ESTATE, "no device-resident sample on this context");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    if (positions) {
        ANN_TRY(ann_reserve(c, c->stage_out, sizeof(int64_t) * (size_t)m));
        k_i32_to_i64_pos<<<ann_blocks(m, 256), 256, 0, c->stream>>>(c->spos.as<int32_t>(), m, c->stage_out.as<int64_t>());
        ANN_CHECK_HIP(c, hipGetLastError());
        ANN_TRY(ann_d2h(c, positions, c->stage_out.p, sizeof(int64_t) * (size_t)m));
    }
    if (feats) ANN_TRY(ann_d2h(c, feats, c->sfeat.p, sizeof(double) * 4 * (size_t)m));
    if (sample_y) ANN_TRY(ann_d2h(c, sample_y, c->sy.p, sizeof(double) * (size_t)m));
    if (sample_predict) {
        ANN_REQUIRE(c, c->spred.p, ANNCHOR_ESTATE, "no predictions for the current sample");
        ANN_TRY(ann_d2h(c, sample_predict, c->spred.p, sizeof(double) * (size_t)m));
    }
    return ANNCHOR_OK;
}


// annchor_hash_sample + annchor_gather_features + annchor_evaluate_samples in one call (device metric):
// positions, feature rows and exact distances come back in one transfer.
extern "C" int annchor_hash_sample_pairs(annchor_ctx *c, const double *bins, int32_t nbins, const int64_t *counts, const int64_t *want,
                                         uint64_t seed_key, int64_t *positions, double *feats, double *sample_y, int64_t *n_out)
{
    if (!c || !bins || !counts || !want || !positions || !feats || !sample_y || !n_out) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->have_features, ANNCHOR_EINVAL, "features not computed");
    ANN_REQUIRE(c, c->metric != ANNCHOR_METRIC_NONE, ANNCHOR_EINVAL, "no device metric bound to this context");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    int32_t *d_pos = nullptr;
    int64_t m = 0;
    ANN_TRY(hash_sample_device(c, bins, nbins, counts, want, seed_key, &d_pos, &m));
    *n_out = m;
    c->nsamp = m;
    if (m == 0) return ANNCHOR_OK;
    const size_t stage_bytes = sizeof(double) * (6 * (size_t)m + 1);
    ANN_TRY(ann_reserve(c, c->stage_out, stage_bytes));
    ANN_TRY(ann_reserve(c, c->spos, sizeof(int32_t) * (size_t)m + 16));
    ANN_TRY(ann_reserve(c, c->sy, sizeof(double) * (size_t)m));
    int32_t *bad = c->spos.as<int32_t>() + m;
    ANN_CHECK_HIP(c, hipMemsetAsync(bad, 0, 4, c->stream));
    ANN_CHECK_HIP(c, hipMemcpyAsync(c->spos.p, d_pos, sizeof(int32_t) * (size_t)m, hipMemcpyDeviceToDevice, c->stream));
    double *st_feats = c->stage_out.as<double>() + m, *st_y = st_feats + 4 * (size_t)m;
    int64_t *st_bad = reinterpret_cast<int64_t *>(st_y + m);
    k_i32_to_i64_pos<<<ann_blocks(m, 256), 256, 0, c->stream>>>(c->spos.as<int32_t>(), m, c->stage_out.as<int64_t>());
    k_gather_features<<<ann_blocks(m, 256), 256, 0, c->stream>>>(c->spos.as<int32_t>(), m, c->lb.as<double>(), c->ub.as<double>(),
                                                                c->dad.as<double>(), c->anc.as<uint8_t>(), st_feats);
    PairSource src;
    src.ij = c->ij.as<int2>();
    src.idx = c->spos.as<int32_t>();
    src.n = m;
    ANN_CHECK_HIP(c, hipEventRecord(c->call_a, c->stream));
    ANN_TRY(ann_metric_launch(c, src, c->sy.as<double>(), nullptr, nullptr));
    ANN_CHECK_HIP(c, hipEventRecord(c->call_b, c->stream));
    c->call_timed = true;
    k_clear_flags_stage<<<ann_blocks(m, 256), 256, 0, c->stream>>>(c->spos.as<int32_t>(), m, c->ncm.as<uint8_t>(), c->sy.as<double>(), bad,
                                                                  st_y, st_bad);
    ANN_CHECK_HIP(c, hipGetLastError());
    if (c->pin && stage_bytes <= annchor_ctx::PIN_DL_BYTES) {   // one transfer, one wait
        unsigned char *slot = c->pin + (size_t)annchor_ctx::PIN_SLOTS * annchor_ctx::PIN_SLOT_BYTES;
        ANN_CHECK_HIP(c, hipMemcpyAsync(slot, c->stage_out.p, stage_bytes, hipMemcpyDeviceToHost, c->stream));
        ANN_CHECK_HIP(c, ann_sync(c, __func__));
        memcpy(positions, slot, sizeof(int64_t) * (size_t)m);
        memcpy(feats, slot + sizeof(double) * (size_t)m, sizeof(double) * 4 * (size_t)m);
        memcpy(sample_y, slot + sizeof(double) * 5 * (size_t)m, sizeof(double) * (size_t)m);
    } else {
        ANN_TRY(ann_d2h(c, positions, c->stage_out.p, sizeof(int64_t) * (size_t)m));
        ANN_TRY(ann_d2h(c, feats, st_feats, sizeof(double) * 4 * (size_t)m));
        ANN_TRY(ann_d2h(c, sample_y, st_y, sizeof(double) * (size_t)m));
    }
    if (c->n_unc >= 0) c->n_unc -= m;
    c->sel_prepared = false;
    return ANNCHOR_OK;
}

extern "C" int annchor_set_samples(annchor_ctx *c, const int64_t *pos, int64_t m, const double *sample_y)
{
    if (!c || (m > 0 && (!pos || !sample_y))) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->have_features, ANNCHOR_EINVAL, "features not computed");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    ANN_TRY(upload_positions(c, pos, m, c->spos));
    ANN_TRY(ann_reserve(c, c->sy, sizeof(double) * (size_t)m));
    ANN_TRY(ann_h2d(c, c->sy.p, sample_y, sizeof(double) * (size_t)m));
    c->nsamp = m;
    c->n_unc = -1; c->sel_prepared = false;
    if (m > 0) k_clear_flags<<<ann_blocks(m, 256), 256, 0, c->stream>>>(c->spos.as<int32_t>(), m, c->ncm.as<uint8_t>());
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}

// ------------------------------------------------- predict / clip / label / merge
__device__ __forceinline__ double reg_predict(const RegModel &m, double l, double u, double d)
{
    // regression bin: lo < F <= hi (regressors.py:84-87); pairs outside every bin keep 0
    int b = -1;
    for (int k = 0; k < m.nb; ++k)
        if (d > m.e[k] && d <= m.e[k + 1]) b = k;
    if (b < 0) return 0.0;
    return ((m.w[b][0] * l + m.w[b][1] * u) + m.w[b][2] * d) + m.c[b];
}

__device__ __forceinline__ int err_label(const RegModel &m, double d)
{
    // error bins: lo <= F <= hi, later bins overwrite (error_predictors.py:61-66)
    int b = -1;
    for (int k = 0; k < m.nb; ++k)
        if (d >= m.e[k] && d <= m.e[k + 1]) b = k;
    return b;
}

#define PM_U 4
__global__ __launch_bounds__(256) void k_predict_merge(int64_t n, const RegModel *__restrict__ mp, int first, int is_metric,
                                                      const int2 *__restrict__ ij, const double *__restrict__ Dt,
                                                      int64_t nx, const int32_t *__restrict__ anchorRank,
                                                      const double *__restrict__ lb, const double *__restrict__ ub,
                                                      const double *__restrict__ dad, const uint8_t *__restrict__ anc,
                                                      const uint8_t *__restrict__ ncm, double *__restrict__ RA,
                                                      uint8_t *__restrict__ label, int stream)
{
    // four pairs per thread: their twelve column reads are in flight together (one pair per thread and 500 000 workgroups of
    // 256 pairs: 1.04 ms per call at 127 M pairs, 0.52 of the HBM rate)
    const RegModel &m = *mp;   // uniform address: scalar loads
    const int64_t p0 = (int64_t)blockIdx.x * (PM_U * 256) + threadIdx.x;
    double l[PM_U], u[PM_U], d[PM_U];
    uint8_t an[PM_U], nc[PM_U];
#pragma unroll
    for (int e = 0; e < PM_U; ++e) {
        const int64_t p = p0 + (int64_t)e * 256;
        const int64_t pc = p < n ? p : n - 1;
        l[e] = ann_load(lb + pc, stream); u[e] = ann_load(ub + pc, stream); d[e] = ann_load(dad + pc, stream);
        an[e] = is_metric ? (uint8_t)0 : anc[pc];
        nc[e] = first ? (uint8_t)1 : ncm[pc];
    }
#pragma unroll
    for (int e = 0; e < PM_U; ++e) {
        const int64_t p = p0 + (int64_t)e * 256;
        if (p >= n) continue;
        double pr = reg_predict(m, l[e], u[e], d[e]);
        pr = fmin(fmax(pr, l[e]), u[e]);  // np.clip(pred, lb, ub)
        if (an[e]) {
            // annchor.py:368-372: anchor pairs take their exact value from D; a later
            // anchor in A overrides an earlier one
            const int2 q = ij[p];
            const int ri = anchorRank[q.x], rj = anchorRank[q.y];
            pr = (ri > rj) ? Dt[(size_t)ri * nx + q.y] : Dt[(size_t)rj * nx + q.x];
        }
        if (nc[e]) ann_store(RA + p, pr, stream);
        const int lbl = err_label(m, d[e]);
        ann_store(label + p, (uint8_t)(lbl < 0 ? 255 : lbl), stream);
    }
}

__global__ void k_sample_predict_scatter(const int32_t *__restrict__ pos, const double *__restrict__ sy, int64_t ms,
                                         const RegModel *__restrict__ mp, const double *__restrict__ lb, const double *__restrict__ ub,
                                         const double *__restrict__ dad, double *__restrict__ RA,
                                         double *__restrict__ spred)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ms) return;
    const RegModel &m = *mp;
    const int32_t p = pos[t];
    spred[t] = reg_predict(m, lb[p], ub[p], dad[p]);  // unclipped (annchor.py:357)
    RA[p] = sy[t];                                    // annchor.py:380
}

int ann_predict_merge_device(annchor_ctx *c, const RegModel *d_model, int first_iteration, int is_metric)
{
    ANN_CHECK_HIP(c, hipEventRecord(c->call_a, c->stream));
    {
        // algorithmic bytes per pair: 3*8 read (lb, ub, dad) + 1 (mask) + 8 (RA) + 1 (label)
        ProfScope ps(c, "predict_clip_label_merge", (double)c->n * 34.0);
        k_predict_merge<<<ann_blocks(c->n, 256 * PM_U), 256, 0, c->stream>>>(
            c->n, d_model, first_iteration, is_metric, c->ij.as<int2>(), c->Dt.as<double>(), c->nx, c->anchorRank.as<int32_t>(),
            c->lb.as<double>(), c->ub.as<double>(), c->dad.as<double>(), c->anc.as<uint8_t>(), c->ncm.as<uint8_t>(),
            c->RA.as<double>(), c->label.as<uint8_t>(), c->n >= ANN_STREAM_MIN_PAIRS);
    }
    ANN_CHECK_HIP(c, hipEventRecord(c->call_b, c->stream));
    c->call_timed = true;
    c->have_RA = true; c->sel_prepared = false;
    if (c->nsamp > 0) {
        ANN_TRY(ann_reserve(c, c->spred, sizeof(double) * (size_t)c->nsamp));
        k_sample_predict_scatter<<<ann_blocks(c->nsamp, 256), 256, 0, c->stream>>>(
            c->spos.as<int32_t>(), c->sy.as<double>(), c->nsamp, d_model, c->lb.as<double>(), c->ub.as<double>(),
            c->dad.as<double>(), c->RA.as<double>(), c->spred.as<double>());
    }
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}

extern "C" int annchor_predict_merge(annchor_ctx *c, const double *bins, int32_t nb, const double *W, const double *cc,
                                     int32_t first_iteration, int32_t is_metric, double *sample_predict)
{
    if (!c || !bins || !W || !cc) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->have_features, ANNCHOR_EINVAL, "features not computed");
    ANN_REQUIRE(c, nb >= 1 && nb <= MAXBINS, ANNCHOR_ELIMIT, "1..%d partitions supported", MAXBINS);
    ANN_REQUIRE(c, first_iteration || c->have_RA, ANNCHOR_EINVAL, "RefineApprox not initialised");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    RegModel m;
    memset(&m, 0, sizeof m);
    m.nb = nb;
    for (int k = 0; k <= nb; ++k) m.e[k] = bins[k];
    for (int k = 0; k < nb; ++k) {
        m.w[k][0] = W[3 * k]; m.w[k][1] = W[3 * k + 1]; m.w[k][2] = W[3 * k + 2];
        m.c[k] = cc[k];
    }
    ANN_TRY(ann_reserve(c, c->model, sizeof(DeviceModel)));
    ANN_TRY(ann_h2d(c, c->model.p, &m, sizeof m));   // (DeviceModel begins with its RegModel)
    c->model_fitted = false; c->errs_on_device = false; c->model_cache_valid = false;
    ANN_TRY(ann_predict_merge_device(c, c->model.as<RegModel>(), first_iteration, is_metric));
    if (c->nsamp > 0 && sample_predict) ANN_TRY(ann_d2h(c, sample_predict, c->spred.p, sizeof(double) * (size_t)c->nsamp));
    return ANNCHOR_OK;
}

__global__ __launch_bounds__(256) void k_merge_host_pred(int64_t n, const double *__restrict__ pred, int first,
                                                        int is_metric, const int2 *__restrict__ ij,
                                                        const double *__restrict__ Dt, int64_t nx,
                                                        const int32_t *__restrict__ anchorRank,
                                                        const double *__restrict__ lb, const double *__restrict__ ub,
                                                        const uint8_t *__restrict__ anc, const uint8_t *__restrict__ ncm,
                                                        double *__restrict__ RA)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    double pr = fmin(fmax(pred[p], lb[p]), ub[p]);
    if (!is_metric && anc[p]) {
        const int2 q = ij[p];
        const int ri = anchorRank[q.x], rj = anchorRank[q.y];
        pr = (ri > rj) ? Dt[(size_t)ri * nx + q.y] : Dt[(size_t)rj * nx + q.x];
    }
    if (first || ncm[p]) RA[p] = pr;
}

__global__ void k_scatter_f64(const int32_t *__restrict__ pos, const double *__restrict__ v, int64_t m, double *__restrict__ dst)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < m) dst[pos[t]] = v[t];
}

extern "C" int annchor_merge_host_prediction(annchor_ctx *c, const double *pred, int32_t first_iteration, int32_t is_metric)
{
    if (!c || !pred) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->have_features, ANNCHOR_EINVAL, "features not computed");
    ANN_REQUIRE(c, first_iteration || c->have_RA, ANNCHOR_EINVAL, "RefineApprox not initialised");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    ANN_TRY(ann_reserve(c, c->stage_in, sizeof(double) * (size_t)c->n));
    ANN_TRY(ann_h2d(c, c->stage_in.p, pred, sizeof(double) * (size_t)c->n));
    k_merge_host_pred<<<ann_blocks(c->n, 256), 256, 0, c->stream>>>(
        c->n, c->stage_in.as<double>(), first_iteration, is_metric, c->ij.as<int2>(), c->Dt.as<double>(), c->nx,
        c->anchorRank.as<int32_t>(), c->lb.as<double>(), c->ub.as<double>(), c->anc.as<uint8_t>(), c->ncm.as<uint8_t>(),
        c->RA.as<double>());
    if (c->nsamp > 0)
        k_scatter_f64<<<ann_blocks(c->nsamp, 256), 256, 0, c->stream>>>(c->spos.as<int32_t>(), c->sy.as<double>(), c->nsamp,
                                                                       c->RA.as<double>());
    ANN_CHECK_HIP(c, hipGetLastError());
    c->have_RA = true; c->sel_prepared = false;
    return ANNCHOR_OK;
}

__global__ void k_labels_from_i64(const int64_t *__restrict__ in, int64_t n, uint8_t *__restrict__ out)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = (in[t] >= 0 && in[t] < 255) ? (uint8_t)in[t] : (uint8_t)255;
}

extern "C" int annchor_set_labels(annchor_ctx *c, const int64_t *labels)
{
    if (!c || !labels) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->have_features, ANNCHOR_EINVAL, "features not computed");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    ANN_TRY(ann_reserve(c, c->stage_in, sizeof(int64_t) * (size_t)c->n));
    ANN_TRY(ann_h2d(c, c->stage_in.p, labels, sizeof(int64_t) * (size_t)c->n));
    k_labels_from_i64<<<ann_blocks(c->n, 256), 256, 0, c->stream>>>(c->stage_in.as<int64_t>(), c->n, c->label.as<uint8_t>());
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}
