// select.hip -- candidate selection for refinement.
//
// Replaces Annchor.select_refine_candidate_pairs (reference annchor/annchor.py:395-473)
// with guarantee_nmin / argpartition (annchor/utils.py:600-621) and get_probs
// (annchor/utils.py:581-589):
//   thresh[i]  = value at sorted position n_neighbors of RefineApprox[I[i]]
//   guarantee_nmin (first iteration): rows with too few computed pairs force their
//                best not-computed pairs to RefineApprox = -1
//   p          = max(thresh[i], thresh[j]) - RefineApprox      on not-computed pairs
//   prob       = searchsorted(errs[label], p, 'left') / len(errs[label])
//   candidates = top n_refine by prob, next = the following n_refine*(lookahead-1)
// Tie rule (the reference's np.argpartition is arbitrary inside tie groups):
// (prob descending, pair position ascending).
//
// guarantee_nmin is sequential over rows in the reference (row i sees the -1 marks
// left by rows < i).  Here the expensive part -- each row's nmin+1 smallest
// not-computed entries -- is computed for all rows in parallel; the sequential
// sweep that remains touches nmin+1 entries per row and runs in a single wavefront.
#include "common.h"
#include "rowsel.h"
#include "selstate.h"

// ------------------------------------------------------------------- thresholds
__global__ __launch_bounds__(ROW_THREADS) void k_row_thresh(const int64_t *__restrict__ Iptr, RowSrc src, uint32_t k,
                                                           double *__restrict__ thresh, int cap)
{
    __shared__ RowSelShared sh;
    __shared__ RowCand rc;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_rt[];
    uint64_t *keys = reinterpret_cast<uint64_t *>(dyn_rt);   // [cap] (fallback path only)
    const int64_t i = row_of_block(gridDim.x);
    const int64_t b = Iptr[i];
    const int len = (int)(Iptr[i + 1] - b);
    if (len <= 0) { if (threadIdx.x == 0) thresh[i] = -INFINITY; return; }  // empty row (query form): never the max
    const RowView rv = row_view(src, i, b);
    const uint32_t kk = k >= (uint32_t)len ? (uint32_t)len - 1 : k;   // clamped like row_kth_key
    {
        // fast path: the k+1 smallest are among the entries below a sampled threshold
        const int cnt = row_candidates(rc, len, (int)kk + 1, [&](int s) { return ann_key_asc(rv.val(s)); }, [](int) { return true; },
                                       [](int, uint64_t, bool) {});
        const int cnts = cnt > (int)kk ? row_cand_shrink(rc, cnt, (int)kk + 1, src.shrink_min) : -1;
        if (cnts > (int)kk) {
            for (int e = threadIdx.x; e < cnts; e += ROW_THREADS) {
                const uint64_t ke = rc.key[e];
                uint32_t less = 0, leq = 0;
                for (int o = 0; o < cnts; ++o) { const uint64_t ko = rc.key[o]; less += ko < ke; leq += ko <= ke; }
                if (less <= kk && kk < leq) thresh[i] = ann_key_asc_inv(ke);   // every writer holds the same value
            }
            return;
        }
    }
    uint64_t res;
    if (len <= cap) {
        for (int s = threadIdx.x; s < len; s += ROW_THREADS) keys[s] = ann_key_asc(rv.val(s));
        __syncthreads();
        res = row_kth_key(sh, len, k, [&](int s) { return keys[s]; });
    } else {
        res = row_kth_key(sh, len, k, [&](int s) { return ann_key_asc(rv.val(s)); });
    }
    if (threadIdx.x == 0) thresh[i] = ann_key_asc_inv(res);
}

// -------------------------------------------------------------- guarantee_nmin
// per row: number of computed entries and the L smallest not-computed entries
// sorted by (value, slot)
__global__ __launch_bounds__(ROW_THREADS) void k_gn_lists(const int64_t *__restrict__ Iptr, RowSrc src,
                                                         const int2 *__restrict__ ij, int L, double *__restrict__ gl_val,
                                                         int32_t *__restrict__ gl_pos, int32_t *__restrict__ gl_oth,
                                                         int32_t *__restrict__ gl_cnt, int32_t *__restrict__ gl_ncomp, int cap)
{
    __shared__ RowSelShared sh;
    __shared__ RowCand rc;
    __shared__ uint32_t cnt_lt, n_unc_s;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    uint64_t *keys = reinterpret_cast<uint64_t *>(dyn);       // [cap] (fallback path only)
    uint64_t *lkey = keys + cap;                              // [L]
    int32_t *lslot = reinterpret_cast<int32_t *>(lkey + L);   // [L]
    const int64_t i = row_of_block(gridDim.x);
    const int64_t b = Iptr[i];
    const int len = (int)(Iptr[i + 1] - b);
    const bool in_lds = len <= cap;
    const uint64_t KINF = ~0ull;  // computed entries sort last
    const RowView rv = row_view(src, i, b);
    const int32_t *Iidx = src.Iidx;
    auto key_of = [&](int s) -> uint64_t {
        const double v = rv.val(s);      // both loads issued, then the select
        const bool u = rv.unc(s);
        return u ? ann_key_asc(v) : KINF;
    };
    if (threadIdx.x == 0) { cnt_lt = 0; n_unc_s = 0; }
    __syncthreads();
    uint32_t my_unc = 0;
    // one streaming pass: count the not-computed entries, keep those below a sampled threshold
    const int fast = row_candidates(rc, len, L, [&](int s) { return ann_key_asc(rv.val(s)); }, [&](int s) { return rv.unc(s); },
                                    [&](int s, uint64_t kk, bool un) {
        if (in_lds) keys[s] = un ? kk : KINF;
        my_unc += un;
    });
    if (my_unc) atomicAdd(&n_unc_s, my_unc);
    __syncthreads();
    const int n_unc = (int)n_unc_s;
    const int want = min(L, n_unc);
    if (threadIdx.x == 0) { gl_cnt[i] = want; gl_ncomp[i] = len - n_unc; }
    if (want == 0) return;
    const int fasts = fast >= want ? row_cand_shrink(rc, fast, want, src.shrink_min) : -1;
    if (fasts >= want) {
        // the `want` smallest by (key, slot) are the candidates of rank < want
        for (int e = threadIdx.x; e < fasts; e += ROW_THREADS) {
            const uint64_t ke = rc.key[e];
            const int32_t se = rc.slot[e];
            int r = 0;
            for (int o = 0; o < fasts; ++o) { const uint64_t ko = rc.key[o]; r += (ko < ke) || (ko == ke && rc.slot[o] < se); }
            if (r < want) {
                const int32_t p = Iidx[b + se];
                const int2 q = ij[p];
                gl_val[i * L + r] = ann_key_asc_inv(ke);
                gl_pos[i * L + r] = p;
                gl_oth[i * L + r] = q.x == (int)i ? q.y : q.x;
            }
        }
        return;
    }
    auto kf = [&](int s) -> uint64_t { return in_lds ? keys[s] : key_of(s); };
    const uint64_t t = row_kth_key(sh, len, (uint32_t)(want - 1), kf);
    // strictly smaller entries (at most want-1 of them), any order
    for (int s = threadIdx.x; s < len; s += ROW_THREADS) {
        const uint64_t kk = kf(s);
        if (kk < t) {
            const uint32_t o = atomicAdd(&cnt_lt, 1u);
            lkey[o] = kk; lslot[o] = s;
        }
    }
    __syncthreads();
    const uint32_t nlt = cnt_lt;
    // entries equal to t, in slot order, until the list is full
    uint32_t run = nlt;
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
    // rank sort by (key, slot)
    for (int e = threadIdx.x; e < want; e += ROW_THREADS) {
        const uint64_t ke = lkey[e];
        const int32_t se = lslot[e];
        int r = 0;
        for (int o = 0; o < want; ++o) r += (lkey[o] < ke) || (lkey[o] == ke && lslot[o] < se);
        const int32_t p = Iidx[b + se];
        const int2 q = ij[p];
        gl_val[i * L + r] = ann_key_asc_inv(ke);
        gl_pos[i * L + r] = p;
        gl_oth[i * L + r] = q.x == (int)i ? q.y : q.x;
    }
}

// twin[i][e] = slot of the same pair in the other endpoint's list, or -1
__global__ void k_gn_twin(int64_t nx, int L, const int32_t *__restrict__ gl_pos, const int32_t *__restrict__ gl_oth,
                          const int32_t *__restrict__ gl_cnt, int32_t *__restrict__ gl_twin)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nx * L) return;
    const int64_t i = t / L;
    const int e = (int)(t - i * L);
    int tw = -1;
    if (e < gl_cnt[i]) {
        const int32_t p = gl_pos[t], j = gl_oth[t];
        const int cj = gl_cnt[j];
        for (int o = 0; o < cj; ++o)
            if (gl_pos[(int64_t)j * L + o] == p) { tw = o; break; }
    }
    gl_twin[t] = tw;
}

// The sweep as a fixed-point iteration.  utils.py:611-619 walks the rows in order; what row i does depends
// only on the marks EARLIER rows put on its pairs: m of them, then i marks its unmarked not-computed
// entries below the (n_todo - m)-th unmarked value.  That is a triangular system -- marks(i) =
// F_i(marks(0..i-1)) -- whose unique solution is reached by iterating all rows at once from "no marks":
// after round r every row whose chain of deciding earlier rows is <= r long is final, and a round that
// changes nothing proves the fixed point.  The chains are short (strings fixture, 1600 rows: 6 rounds
// + 1), so a handful of fully parallel rounds replace one wave's walk over all rows (0.6 ms of the 5.8 ms
// fit there, 4.8 ms at 16 000 rows).
// One wave per row, lane = list entry.  masks: bit e of row i = "row i marks its entry e" (uint32
// [nx][Lw], two buffers); mout[b][i] = marks of earlier rows on pairs of row i that are outside i's list
// (three rotating buffers: read, accumulate for the next round, clear for the one after).
__global__ __launch_bounds__(256) void k_gn_round(int64_t nx, int nmin, int L, int Lw, const double *__restrict__ gl_val,
                                                 const int32_t *__restrict__ gl_oth, const int32_t *__restrict__ gl_twin,
                                                 const int32_t *__restrict__ gl_cnt, const int32_t *__restrict__ gl_ncomp,
                                                 const uint32_t *__restrict__ cur, uint32_t *__restrict__ nxt,
                                                 const int32_t *__restrict__ mout_cur, int32_t *__restrict__ mout_nxt,
                                                 int32_t *__restrict__ mout_clr, int32_t *__restrict__ changed, int32_t *__restrict__ err)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= nx) return;
    if (lane == 0) mout_clr[i] = 0;
    const int cnt = gl_cnt[i], ncomp = gl_ncomp[i];
    const int ntodo = nmin - ncomp;
    const int chunks = (L + 63) / 64;
    bool active = !(ntodo <= 0 || (cnt == 0 && ncomp == 0));
    if (active && cnt <= ntodo && cnt < L) { if (lane == 0) *err = 1; active = false; }
    // marks of earlier rows on this row's entries
    int m = active ? mout_cur[i] : 0;
    if (active)
        for (int ch = 0; ch < chunks; ++ch) {
            const int e = ch * 64 + lane;
            bool earlier = false;
            if (e < cnt) {
                const int32_t o = gl_oth[i * L + e], tw = gl_twin[i * L + e];
                earlier = o < i && tw >= 0 && ((cur[(int64_t)o * Lw + (tw >> 5)] >> (tw & 31)) & 1u);
            }
            m += __popcll(__ballot(earlier));
        }
    const int need = ntodo + 1 - m;
    double t = 0;
    bool found = false;
    if (active && need > 0) {
        int cum = 0;
        for (int ch = 0; ch < chunks && !found; ++ch) {
            const int e = ch * 64 + lane;
            bool um = false;
            double v = 0.0;
            if (e < cnt) {
                const int32_t o = gl_oth[i * L + e], tw = gl_twin[i * L + e];
                v = gl_val[i * L + e];
                um = !(o < i && tw >= 0 && ((cur[(int64_t)o * Lw + (tw >> 5)] >> (tw & 31)) & 1u));
            }
            const unsigned long long mb = __ballot(um);
            const int cm = __popcll(mb);
            if (cum + cm >= need) {
                const int want = need - cum - 1;
                const int myrank = __popcll(mb & ((1ull << lane) - 1ull));
                const unsigned long long hit = __ballot(um && myrank == want);
                t = __shfl(v, __ffsll((unsigned long long)hit) - 1);
                found = true;
            }
            cum += cm;
        }
        if (!found && lane == 0) *err = 2;
    }
    bool diff = false;
    for (int ch = 0; ch < chunks; ++ch) {
        const int e = ch * 64 + lane;
        bool mark = false;
        int32_t o = 0, tw = -1;
        if (found && e < cnt) {
            o = gl_oth[i * L + e]; tw = gl_twin[i * L + e];
            const bool earlier = o < i && tw >= 0 && ((cur[(int64_t)o * Lw + (tw >> 5)] >> (tw & 31)) & 1u);
            mark = !earlier && gl_val[i * L + e] < t;
        }
        const unsigned long long mb = __ballot(mark);
        const int w0 = 2 * ch, w1 = 2 * ch + 1;
        if (lane == 0) {
            if (w0 < Lw) { const uint32_t v0 = (uint32_t)mb; diff |= cur[i * Lw + w0] != v0; nxt[i * Lw + w0] = v0; }
            if (w1 < Lw) { const uint32_t v1 = (uint32_t)(mb >> 32); diff |= cur[i * Lw + w1] != v1; nxt[i * Lw + w1] = v1; }
        }
        // a mark on a pair that the later row o does not list still counts among o's marked entries
        if (mark && tw < 0 && o > i) atomicAdd(&mout_nxt[o], 1);
    }
    if (lane == 0 && diff) atomicOr(changed, 1);
}

// the marks of the fixed point go into the pair list (RA = -1)
__global__ void k_gn_apply(int64_t nx, int L, int Lw, const int32_t *__restrict__ gl_pos, const int32_t *__restrict__ gl_cnt,
                           const uint32_t *__restrict__ masks, double *__restrict__ RA)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nx * L) return;
    const int64_t i = t / L;
    const int e = (int)(t - i * L);
    if (e < gl_cnt[i] && ((masks[i * Lw + (e >> 5)] >> (e & 31)) & 1u)) RA[gl_pos[t]] = -1.0;
}

// LDS form of the sequential sweep (utils.py:611-619): all state that one row hands to
// the next -- "this entry of your list is already -1" flags and the per-row count of -1
// entries -- lives in LDS; the per-row lists are read-only and prefetched one row ahead.
__global__ __launch_bounds__(64) void k_gn_sweep_lds(int64_t nx, int nmin, int L, const double *__restrict__ gl_val,
                                                    const int32_t *__restrict__ gl_pos, const int32_t *__restrict__ gl_oth,
                                                    const int32_t *__restrict__ gl_twin, const int32_t *__restrict__ gl_cnt,
                                                    const int32_t *__restrict__ gl_ncomp, double *__restrict__ RA,
                                                    int32_t *__restrict__ err)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    const int lane = threadIdx.x;
    const int Lw = (L + 31) / 32;
    uint32_t *mflag = reinterpret_cast<uint32_t *>(dyn);   // [nx][Lw] bit e: entry e of the row is marked
    int32_t *mcount = reinterpret_cast<int32_t *>(mflag + (size_t)nx * Lw);  // [nx]
    for (int64_t t = lane; t < nx * Lw; t += 64) mflag[t] = 0;
    for (int64_t t = lane; t < nx; t += 64) mcount[t] = 0;
    __syncthreads();
    const int chunks = (L + 63) / 64;
    // prefetch registers for chunk 0 of the next row
    double nv = 0; int32_t np_ = 0, no = 0, nt = -1; int ncnt = 0, nncomp = 0;
    auto fetch = [&](int64_t r) {
        if (r < nx) {
            ncnt = gl_cnt[r]; nncomp = gl_ncomp[r];
            if (lane < L) { nv = gl_val[r * L + lane]; np_ = gl_pos[r * L + lane]; no = gl_oth[r * L + lane]; nt = gl_twin[r * L + lane]; }
        }
    };
    fetch(0);
    for (int64_t i = 0; i < nx; ++i) {
        const double v0 = nv; const int32_t p0 = np_, o0 = no, t0 = nt; const int cnt = ncnt, ncomp = nncomp;
        fetch(i + 1);
        const int ntodo = nmin - ncomp;
        if (ntodo <= 0 || (cnt == 0 && ncomp == 0)) continue;   // enough computed, or an empty row
        if (cnt <= ntodo && cnt < L) { if (lane == 0) *err = 1; continue; }
        const int need = ntodo + 1 - mcount[i];
        if (need <= 0) continue;
        double t = 0;
        bool found = false;
        int cum = 0;
        for (int ch = 0; ch < chunks && !found; ++ch) {
            const int e = ch * 64 + lane;
            const bool valid = e < cnt;
            const double v = ch == 0 ? v0 : (valid ? gl_val[i * L + e] : 0.0);
            const bool um = valid && !((mflag[i * Lw + (e >> 5)] >> (e & 31)) & 1u);
            const unsigned long long m = __ballot(um);
            const int cm = __popcll(m);
            if (cum + cm >= need) {
                const int want = need - cum - 1;
                const int myrank = __popcll(m & ((1ull << lane) - 1ull));
                const unsigned long long hit = __ballot(um && myrank == want);
                t = __shfl(v, __ffsll((unsigned long long)hit) - 1);
                found = true;
            }
            cum += cm;
        }
        if (!found) { if (lane == 0) *err = 2; continue; }
        for (int ch = 0; ch < chunks; ++ch) {
            const int e = ch * 64 + lane;
            if (e < cnt) {
                const double v = ch == 0 ? v0 : gl_val[i * L + e];
                const int32_t p = ch == 0 ? p0 : gl_pos[i * L + e];
                const int32_t o = ch == 0 ? o0 : gl_oth[i * L + e];
                const int32_t tw = ch == 0 ? t0 : gl_twin[i * L + e];
                if (v < t && !((mflag[i * Lw + (e >> 5)] >> (e & 31)) & 1u)) {
                    RA[p] = -1.0;
                    if (tw >= 0) atomicOr(&mflag[(int64_t)o * Lw + (tw >> 5)], 1u << (tw & 31));
                    atomicAdd(&mcount[o], 1);
                }
            }
        }
        __syncthreads();  // single wave: orders this row's LDS updates before the next row's reads
    }
}


// Same sweep, with the read-only per-row lists streamed through a double-buffered LDS ring
// by all four waves of the workgroup while wave 0 alone performs the (sequential) sweep:
// the global-memory latency of the lists no longer sits between consecutive rows.
#define GN_B 32          // rows per batch
#define GN_LMAX 64       // list length handled by this form
__global__ __launch_bounds__(256) void k_gn_sweep_ring(int64_t nx, int nmin, int L, const double *__restrict__ gl_val,
                                                      const int32_t *__restrict__ gl_pos, const int32_t *__restrict__ gl_oth,
                                                      const int32_t *__restrict__ gl_twin, const int32_t *__restrict__ gl_cnt,
                                                      const int32_t *__restrict__ gl_ncomp, double *__restrict__ RA,
                                                      int32_t *__restrict__ err)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Lw = (L + 31) / 32;
    const int BL = GN_B * L;                                   // entries per batch
    double *rval = reinterpret_cast<double *>(dyn);                      // [2][BL] + 64 slack
    int32_t *rpos = reinterpret_cast<int32_t *>(rval + 2 * BL + 64);     // [2][BL] + 64 slack
    int32_t *roth = rpos + 2 * BL + 64, *rtwin = roth + 2 * BL + 64;     // [2][BL] + 64 slack each
    int32_t *rcnt = rtwin + 2 * BL + 64, *rncomp = rcnt + 2 * GN_B;      // [2][GN_B] each
    uint32_t *mflag = reinterpret_cast<uint32_t *>(rncomp + 2 * GN_B);   // [nx][Lw] + 2 slack
    int32_t *mcount = reinterpret_cast<int32_t *>(mflag + (size_t)nx * Lw + 2);  // [nx]
    for (int64_t t = threadIdx.x; t < nx * Lw; t += 256) mflag[t] = 0;
    for (int64_t t = threadIdx.x; t < nx; t += 256) mcount[t] = 0;
    constexpr int PER = (GN_B * GN_LMAX + 255) / 256;          // 8 entries per thread at most
    double tv[PER]; int32_t tp[PER], to[PER], tt[PER];
    int32_t tc = 0, tn = 0;
    const int nbatch = (int)((nx + GN_B - 1) / GN_B);
    auto load = [&](int b) {
        const int64_t base = (int64_t)b * BL, lim = nx * L;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int e = q * 256 + threadIdx.x;
            const bool ok = e < BL && base + e < lim;
            tv[q] = ok ? gl_val[base + e] : 0.0;
            tp[q] = ok ? gl_pos[base + e] : 0;
            to[q] = ok ? gl_oth[base + e] : 0;
            tt[q] = ok ? gl_twin[base + e] : -1;
        }
        if (threadIdx.x < GN_B) {
            const int64_t r = (int64_t)b * GN_B + threadIdx.x;
            tc = r < nx ? gl_cnt[r] : 0;
            tn = r < nx ? gl_ncomp[r] : nmin;
        }
    };
    auto stash = [&](int slot) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int e = q * 256 + threadIdx.x;
            if (e < BL) { rval[slot * BL + e] = tv[q]; rpos[slot * BL + e] = tp[q]; roth[slot * BL + e] = to[q]; rtwin[slot * BL + e] = tt[q]; }
        }
        if (threadIdx.x < GN_B) { rcnt[slot * GN_B + threadIdx.x] = tc; rncomp[slot * GN_B + threadIdx.x] = tn; }
    };
    load(0);
    stash(0);
    __syncthreads();
    for (int b = 0; b < nbatch; ++b) {
        const int slot = b & 1;
        if (b + 1 < nbatch) load(b + 1);
        if (wave == 0) {
            // Software pipeline over the rows of the batch: the nine LDS operands of row rr+1
            // are requested before row rr is decided, so their latency hides behind row rr's
            // register work.  They cannot contain row rr's own marks; at most one of those can
            // concern row rr+1 (the pair (i, i+1) is unique) and it is applied to the prefetched
            // registers directly.  Marks of earlier rows are in LDS already (a wave's LDS
            // operations execute in order).
            const int e = lane;   // L <= 64: one entry per lane
            struct RowOps { int cnt, ncomp, mc; uint32_t fl; double v; int32_t p, o, tw; };
            auto fetch = [&](int rr, int64_t i) {
                RowOps r;
                r.cnt = rcnt[slot * GN_B + rr]; r.ncomp = rncomp[slot * GN_B + rr];
                r.mc = mcount[i];
                // UNCONDITIONAL loads (lanes >= L read the next row's entries / the slack behind
                // the ring: never used, `um` masks them with e < cnt <= L): a predicated load
                // becomes an exec-mask branch per operand and splits the block into several
                // LDS round trips
                r.fl = mflag[i * Lw + (e >> 5)];
                r.v = rval[slot * BL + rr * L + e];
                r.p = rpos[slot * BL + rr * L + e];
                r.o = roth[slot * BL + rr * L + e];
                r.tw = rtwin[slot * BL + rr * L + e];
                return r;
            };
            const int64_t i0 = (int64_t)b * GN_B;
            const int nrows = (int)min((int64_t)GN_B, nx - i0);
            RowOps cur = fetch(0, i0);
            for (int rr = 0; rr < nrows; ++rr) {
                const int64_t i = i0 + rr;
                const bool more = rr + 1 < nrows;
                RowOps nxt = cur;
                if (more) nxt = fetch(rr + 1, i + 1);
                asm volatile("" : "+v"(nxt.fl), "+v"(nxt.p), "+v"(nxt.o), "+v"(nxt.tw));   // keep the prefetch here
                const int cnt = cur.cnt, ncomp = cur.ncomp, mc = cur.mc;
                const uint32_t fl = cur.fl;
                const double v = cur.v;
                const int32_t p = cur.p, o = cur.o, tw = cur.tw;
                const int ntodo = nmin - ncomp;
                const int need = ntodo + 1 - mc;
                const bool empty = cnt == 0 && ncomp == 0;
                const bool um = e < cnt && !((fl >> (e & 31)) & 1u);
                const unsigned long long m = __ballot(um);
                bool mark = false;
                if (!(ntodo <= 0 || empty || need <= 0)) {
                    if ((cnt <= ntodo && cnt < L) || (int)__popcll(m) < need) { if (lane == 0) *err = 1; }
                    else {
                        const int myrank = __popcll(m & ((1ull << lane) - 1ull));
                        const unsigned long long hit = __ballot(um && myrank == need - 1);
                        const int src = __ffsll((unsigned long long)hit) - 1;
                        const double t = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src),
                                                          __builtin_amdgcn_readlane(__double2loint(v), src));
                        mark = um && v < t;
                        if (mark) {
                            RA[p] = -1.0;
                            if (tw >= 0) atomicOr(&mflag[(int64_t)o * Lw + (tw >> 5)], 1u << (tw & 31));
                            atomicAdd(&mcount[o], 1);
                        }
                    }
                }
                if (more) {   // this row's mark on the pair (i, i+1), if any, reaches the prefetched registers
                    const unsigned long long hn = __ballot(mark && o == (int32_t)(i + 1));
                    if (hn) {
                        const int srcn = __ffsll((unsigned long long)hn) - 1;
                        const int twn = __builtin_amdgcn_readlane(tw, srcn);
                        nxt.mc += 1;
                        if (twn >= 0 && (e >> 5) == (twn >> 5)) nxt.fl |= 1u << (twn & 31);
                    }
                }
                cur = nxt;
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (b + 1 < nbatch) stash(slot ^ 1);
        __syncthreads();
    }
}

__device__ __forceinline__ uint8_t ld_u8_agent(const uint8_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int32_t ld_i32_agent(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// single wavefront, sequential over rows (utils.py:611-619)
__global__ __launch_bounds__(64) void k_gn_sequential(int64_t nx, int nmin, int L, const double *__restrict__ gl_val,
                                                     const int32_t *__restrict__ gl_pos, const int32_t *__restrict__ gl_cnt,
                                                     const int32_t *__restrict__ gl_ncomp, const int2 *__restrict__ ij,
                                                     double *__restrict__ RA, uint8_t *__restrict__ marked,
                                                     int32_t *__restrict__ markcount, int32_t *__restrict__ err)
{
    const int lane = threadIdx.x;
    for (int64_t i = 0; i < nx; ++i) {
        const int ntodo = nmin - gl_ncomp[i];
        if (ntodo <= 0) continue;
        const int cnt = gl_cnt[i];
        if (cnt == 0 && gl_ncomp[i] == 0) continue;   // empty row
        // the reference's np.partition(a, n_todo) needs n_todo < len(a)
        if (cnt <= ntodo && cnt < L) { if (lane == 0) *err = 1; continue; }
        const int need = ntodo + 1 - ld_i32_agent(&markcount[i]);
        if (need <= 0) continue;
        // value of the need-th smallest not-yet-marked entry
        double t = 0;
        bool found = false;
        int cum = 0;
        for (int base = 0; base < cnt && !found; base += 64) {
            const int e = base + lane;
            const bool valid = e < cnt;
            const double v = valid ? gl_val[i * L + e] : 0.0;
            const int32_t p = valid ? gl_pos[i * L + e] : 0;
            const bool um = valid && !ld_u8_agent(&marked[p]);
            const unsigned long long m = __ballot(um);
            const int cmask = __popcll(m);
            if (cum + cmask >= need) {
                const int want = need - cum - 1;  // 0-based rank inside this chunk
                const int myrank = __popcll(m & ((1ull << lane) - 1ull));
                const unsigned long long hit = __ballot(um && myrank == want);
                const int src = __ffsll((unsigned long long)hit) - 1;
                t = __shfl(v, src);
                found = true;
            }
            cum += cmask;
        }
        if (!found) { if (lane == 0) *err = 2; continue; }
        for (int base = 0; base < cnt; base += 64) {
            const int e = base + lane;
            if (e < cnt) {
                const double v = gl_val[i * L + e];
                const int32_t p = gl_pos[i * L + e];
                if (v < t && !ld_u8_agent(&marked[p])) {
                    RA[p] = -1.0;
                    __hip_atomic_store(&marked[p], (uint8_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int2 q = ij[p];
                    const int other = q.x == (int)i ? q.y : q.x;
                    atomicAdd(&markcount[other], 1);
                }
            }
        }
        // the marks of this row must be visible (in L2) before the next row reads them
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// Bucket index over a label's sorted errors: nb = 2 * len equal-width buckets over [first, last].
// bucket(x) is a monotone non-decreasing function of x (float subtraction, multiplication by a non-negative
// constant, floor and clamping all are), so with T[k] = #{entries with bucket < k} every entry in a bucket
// below bucket(x) is < x and every entry in a bucket above it is > x: searchsorted(E, x, 'left') lies in
// [T[k], T[k + 1]], k = bucket(x) -- one table read and a search over the one or two entries of a bucket
// instead of ~13 probes.  Exact for any input (ties, infinities: they clamp; NaN: bucket 0, as comparisons
// with NaN are false in the search as well).
struct EcdfIndex {
    double emin, scale;
    int32_t nb, toff;   // buckets of this label, offset of its nb + 1 table entries
};
__device__ __forceinline__ int ecdf_bucket(const EcdfIndex &ix, double x)
{
    const double t = (x - ix.emin) * ix.scale;
    if (!(t >= 0.0)) return 0;
    if (t >= (double)ix.nb) return ix.nb - 1;
    return (int)t;
}
// one block per label
__global__ __launch_bounds__(256) void k_ecdf_index(const double *__restrict__ errs, const int64_t *__restrict__ errptr,
                                                   EcdfIndex *__restrict__ index, uint32_t *__restrict__ table)
{
    const int b = blockIdx.x;
    const int64_t lo = errptr[b], hi = errptr[b + 1];
    const int len = (int)(hi - lo);
    EcdfIndex ix;
    ix.nb = len > 0 ? 2 * len : 1;
    ix.toff = (int32_t)(2 * lo + b);
    ix.emin = len > 0 ? errs[lo] : 0.0;
    const double span = len > 0 ? errs[hi - 1] - ix.emin : 0.0;
    ix.scale = span > 0.0 ? (double)ix.nb / span : 0.0;
    if (!(ix.scale < INFINITY)) ix.scale = 0.0;
    uint32_t *T = table + ix.toff;
    for (int k = threadIdx.x; k <= ix.nb; k += blockDim.x) T[k] = 0u;
    __syncthreads();
    // T[k + 1] counts the entries of bucket k; the entries are sorted, so bucket numbers are non-decreasing
    // along the list and T[k] = index of the first entry whose bucket is >= k: written by the first entry of
    // every bucket run, gaps filled by a running maximum
    for (int e = threadIdx.x; e < len; e += blockDim.x) atomicAdd(&T[ecdf_bucket(ix, errs[lo + e]) + 1], 1u);
    __syncthreads();
    // inclusive prefix over the nb + 1 counts: a contiguous chunk per thread, chunk totals scanned in LDS
    __shared__ uint32_t csum[256];
    const int per = (ix.nb + 1 + 255) / 256;
    const int k0 = threadIdx.x * per, k1 = min(k0 + per, ix.nb + 1);
    uint32_t mine = 0;
    for (int k = k0; k < k1; ++k) mine += T[k];
    csum[threadIdx.x] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int q = 0; q < 256; ++q) { const uint32_t v = csum[q]; csum[q] = run; run += v; }
        index[b] = ix;
    }
    __syncthreads();
    uint32_t run = csum[threadIdx.x];
    for (int k = k0; k < k1; ++k) { run += T[k]; T[k] = run; }
}

// -------------------------------------------------------------------- ECDF prob
__global__ __launch_bounds__(256) void k_prob(int64_t n, const int2 *__restrict__ ij, const double *__restrict__ thresh,
                                             const double *__restrict__ RA, const uint8_t *__restrict__ ncm,
                                             const uint8_t *__restrict__ label, const double *__restrict__ errs,
                                             const int64_t *__restrict__ errptr, int nlabels, int lds_cap_entries,
                                             double *__restrict__ prob, int stream, const EcdfIndex *__restrict__ index,
                                             const uint32_t *__restrict__ table)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    double *le = reinterpret_cast<double *>(dyn);
    __shared__ int64_t lptr[257];
    __shared__ EcdfIndex lix[256];   // (the labels' bucket indices: three gathers per pair less for the vector memory pipeline)
    for (int t = threadIdx.x; t <= nlabels; t += blockDim.x) lptr[t] = errptr[t];
    if (index)
        for (int t = threadIdx.x; t < nlabels; t += blockDim.x) lix[t] = index[t];
    __syncthreads();
    // (the lists' total length is only known on the device when they were fitted there: annchor_fit_errors_device)
    const bool errs_in_lds = lptr[nlabels] <= (int64_t)lds_cap_entries;
    if (errs_in_lds) {
        const int64_t tot = lptr[nlabels];
        for (int64_t t = threadIdx.x; t < tot; t += blockDim.x) le[t] = errs[t];
        __syncthreads();
    }
    const double *E = errs_in_lds ? le : errs;
    // Eight pairs per thread per step: their streams (mask, pair, RA, label) are loaded together,
    // then the two threshold gathers, then the eight binary searches advance in lock step -- one
    // pair at a time the chain mask -> pair -> thresholds -> ~13 dependent LDS probes ran at
    // memory latency (2.0 ms for 127 M pairs, 20 % of the HBM rate).
    constexpr int PI = 8;   // (int32 search indices: the error lists hold at most the sample count)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * PI;
    for (int64_t p0 = (int64_t)blockIdx.x * blockDim.x * PI + threadIdx.x; p0 < n; p0 += stride) {
        uint8_t m[PI], lbv[PI];
        int2 q[PI];
        double ra[PI];
#pragma unroll
        for (int e = 0; e < PI; ++e) {
            const int64_t p = p0 + (int64_t)e * blockDim.x;
            const bool in = p < n;
            const int64_t pc = in ? p : 0;
            m[e] = ann_load(ncm + pc, stream);
            const long long qq = ann_load(reinterpret_cast<const long long *>(ij) + pc, stream);
            q[e] = make_int2((int)(qq & 0xffffffffll), (int)(qq >> 32));
            ra[e] = ann_load(RA + pc, stream);
            lbv[e] = ann_load(label + pc, stream);
            if (!in) { m[e] = 0; q[e] = make_int2(0, 0); lbv[e] = 0; }
        }
        double pv[PI];
#pragma unroll
        for (int e = 0; e < PI; ++e) pv[e] = fmax(thresh[q[e].x], thresh[q[e].y]) - ra[e];
        int32_t b[PI], lo[PI], hi[PI];
        bool busy = false;
#pragma unroll
        for (int e = 0; e < PI; ++e) {
            const bool search = m[e] && (int)lbv[e] < nlabels;
            b[e] = search ? (int32_t)lptr[lbv[e]] : 0;
            lo[e] = b[e];
            hi[e] = search ? (int32_t)lptr[lbv[e] + 1] : 0;
            if (index && lo[e] < hi[e]) {   // bucket index: the answer lies inside one bucket
                const EcdfIndex ix = lix[lbv[e]];
                const int k = ecdf_bucket(ix, pv[e]);
                const uint32_t t0 = table[ix.toff + k], t1 = table[ix.toff + k + 1];
                hi[e] = b[e] + (int32_t)t1;
                lo[e] = b[e] + (int32_t)t0;
            }
            busy |= lo[e] < hi[e];
        }
        // searchsorted(side='left'): number of entries < pv
        while (busy) {
            busy = false;
#pragma unroll
            for (int e = 0; e < PI; ++e)
                if (lo[e] < hi[e]) {
                    const int32_t mid = (lo[e] + hi[e]) >> 1;
                    if (E[mid] < pv[e]) lo[e] = mid + 1; else hi[e] = mid;
                    busy |= lo[e] < hi[e];
                }
        }
#pragma unroll
        for (int e = 0; e < PI; ++e) {
            const int64_t p = p0 + (int64_t)e * blockDim.x;
            if (p >= n) continue;
            double pr = -1.0;
            if (m[e]) {
                if ((int)lbv[e] < nlabels) {
                    const int32_t len = (int32_t)lptr[lbv[e] + 1] - b[e];
                    pr = (double)(lo[e] - b[e]) / (double)len;
                } else {
                    pr = 0.0;
                }
            }
            ann_store(prob + p, pr, stream);
        }
    }
}

// ----------------------------------------------------- top-K split + compaction
#define CP_THREADS 256
#define CP_ITEMS 8
#define CP_SUBTILE (CP_THREADS * CP_ITEMS)
// sub-tiles per workgroup (a tile = nsub x 2048 pairs): 4 on long lists -- the single-workgroup scan of the tile counters
// (k_cut_scan) walked 62 K tiles of 2048 at 127 M pairs in 175 us --, 1 on short ones (C2: 625 workgroups of one sub-tile)
static inline int cp_nsub(int64_t n) { return n >= (16ll << 20) ? 4 : 1; }


#define TIE_CAP 65536

// (position * 0x9E3779B97F4A7C15 mod 2^64) >> 11
__host__ __device__ __forceinline__ unsigned long long ann_tie_scramble(int64_t p)
{
    return ((unsigned long long)p * 0x9E3779B97F4A7C15ull) >> 11;
}

// ---- the scrambled-position cut inside the group on each probability cut, device only:
//   k_tie_hist     group sizes, pairs above each cut, histogram of the members' top TIE_HB key bits
//   k_tie_pick     the histogram bin holding the wanted rank, and the rank inside it
//   k_tie_collect  the members of that bin (a few dozen: the keys are uniform) into a short list
//   k_tie_select   the wanted rank of the list (MSB-first byte radix in one workgroup)
// Groups of any size take this route (the lookahead cut of the strings workload sits in a group of
// ~5 * 10^5 pairs); only a list longer than TIE_CAP -- keys piling up in one of 4096 bins -- falls back
// to the host-driven general selection.
#define TIE_HB 12
#define TIE_BINS (1 << TIE_HB)
#define TIE_SHIFT (53 - TIE_HB)
#define TIE_U 8   // probabilities in flight per thread in the two streaming passes (with three workgroups per CU: 12 MB chip-wide)
__global__ __launch_bounds__(256) void k_tie_hist(const double *__restrict__ prob, int64_t n, CutState *__restrict__ cs,
                                                 uint32_t *__restrict__ ghist /*[2][TIE_BINS]*/)
{
    __shared__ uint32_t h[2][TIE_BINS];
    __shared__ unsigned long long acc[4];
    for (int t = threadIdx.x; t < 2 * TIE_BINS; t += 256) (&h[0][0])[t] = 0;
    if (threadIdx.x < 4) acc[threadIdx.x] = 0;
    __syncthreads();
    const double t1 = cs->t1, t5 = cs->t5;
    const bool need1 = !cs->all1, need5 = !cs->all5;
    unsigned long long g1 = 0, g5 = 0, n1 = 0, n5 = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * TIE_U;
    for (int64_t p0 = (int64_t)blockIdx.x * blockDim.x * TIE_U + threadIdx.x; p0 < n; p0 += stride) {
        double v[TIE_U];
#pragma unroll
        for (int e = 0; e < TIE_U; ++e) v[e] = ann_ldc(prob, p0 + (int64_t)e * blockDim.x, n);
#pragma unroll
        for (int e = 0; e < TIE_U; ++e) {
            const int64_t p = p0 + (int64_t)e * blockDim.x;
            if (p >= n || !(v[e] >= 0.0)) continue;
            g1 += v[e] > t1;
            g5 += v[e] > t5;
            const bool m1 = need1 && v[e] == t1, m5 = need5 && v[e] == t5;
            if (m1 || m5) {
                const uint32_t bin = (uint32_t)(ann_tie_scramble(p) >> TIE_SHIFT);
                if (m1) { atomicAdd(&h[0][bin], 1u); ++n1; }
                if (m5) { atomicAdd(&h[1][bin], 1u); ++n5; }
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        g1 += __shfl_xor(g1, off); g5 += __shfl_xor(g5, off); n1 += __shfl_xor(n1, off); n5 += __shfl_xor(n5, off);
    }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&acc[0], g1); atomicAdd(&acc[1], g5); atomicAdd(&acc[2], n1); atomicAdd(&acc[3], n5); }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * TIE_BINS; t += 256) {
        const uint32_t c = (&h[0][0])[t];
        if (c) atomicAdd(&ghist[t], c);
    }
    if (threadIdx.x == 0) {
        if (acc[0]) atomicAdd((unsigned long long *)&cs->tie_gt1, acc[0]);
        if (acc[1]) atomicAdd((unsigned long long *)&cs->tie_gt5, acc[1]);
        if (acc[2]) atomicAdd((unsigned long long *)&cs->tie_n1, acc[2]);
        if (acc[3]) atomicAdd((unsigned long long *)&cs->tie_n5, acc[3]);
    }
}

// one workgroup: for each cut the bin that holds the e-th smallest key of the group (e = K - above) and
// the rank inside that bin; rk is settled here when the whole group (or none of it) is taken.
// Leaves the histogram zeroed for the next call.
__global__ __launch_bounds__(1024) void k_tie_pick(CutState *__restrict__ cs, uint32_t *__restrict__ ghist)
{
    __shared__ unsigned long long wsum[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int q = 0; q < 2; ++q) {
        const bool all = q == 0 ? cs->all1 : cs->all5;
        const long long cnt = q == 0 ? cs->tie_n1 : cs->tie_n5;
        const long long e = q == 0 ? cs->K1 - cs->tie_gt1 : cs->K5 - cs->tie_gt5;
        uint32_t *h = ghist + q * TIE_BINS;
        constexpr int PER = TIE_BINS / 1024;
        unsigned long long c[PER], mine = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) { c[j] = h[threadIdx.x * PER + j]; h[threadIdx.x * PER + j] = 0; mine += c[j]; }
        unsigned long long inc = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const unsigned long long up = __shfl_up(inc, off); if (lane >= off) inc += up; }
        __syncthreads();
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        unsigned long long before = inc - mine;
        for (int w2 = 0; w2 < wave; ++w2) before += wsum[w2];
        if (threadIdx.x == 0) {
            unsigned long long rk = ~0ull;       // whole group taken
            long long bin = -1;
            if (!all && e < cnt) { if (e <= 0) rk = 0ull; else bin = -2; }   // -2: to be located below
            if (q == 0) { cs->rk1 = rk; cs->tie_bin1 = bin; } else { cs->rk5 = rk; cs->tie_bin5 = bin; }
        }
        __syncthreads();
        if (!all && e < cnt && e > 0 && (long long)before < e && (long long)(before + mine) >= e) {
            long long accb = (long long)before;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                if (accb < e && accb + (long long)c[j] >= e) {
                    if (q == 0) { cs->tie_bin1 = threadIdx.x * PER + j; cs->tie_rem1 = e - accb; cs->tie_len1 = (long long)c[j]; }
                    else { cs->tie_bin5 = threadIdx.x * PER + j; cs->tie_rem5 = e - accb; cs->tie_len5 = (long long)c[j]; }
                }
                accb += (long long)c[j];
            }
        }
        __syncthreads();
    }
}

// members of the picked bins -> short lists (append order is irrelevant: k_tie_select ranks the keys).  A workgroup keeps its
// finds in LDS and reserves their slots with ONE atomic per list at the end: an atomic on one global address costs ~12.5 ns,
// serialised over the whole chip (tools/microbench/atomics.hip) -- with a group of 10^8 pairs on the cut the picked bin holds
// 3 x 10^4 of them, and appending them one by one made this pass 0.58 ms where the histogram pass over the same column takes 0.19.
#define TIE_LCAP 1024
__global__ __launch_bounds__(256) void k_tie_collect(const double *__restrict__ prob, int64_t n, CutState *__restrict__ cs,
                                                    unsigned long long *__restrict__ list1, unsigned long long *__restrict__ list5,
                                                    long long cap)
{
    __shared__ unsigned long long lbuf[2][TIE_LCAP];
    __shared__ uint32_t lcnt[2];
    __shared__ unsigned long long lbase[2];
    const long long b1 = cs->tie_bin1, b5 = cs->tie_bin5;
    if (b1 < 0 && b5 < 0) return;
    if (threadIdx.x < 2) lcnt[threadIdx.x] = 0;
    __syncthreads();
    const double t1 = cs->t1, t5 = cs->t5;
    auto put = [&](int q, unsigned long long kk) {
        const uint32_t o = atomicAdd(&lcnt[q], 1u);
        if (o < TIE_LCAP) lbuf[q][o] = kk;
        else {   // (more finds than the staging holds: straight to the list)
            const unsigned long long g = atomicAdd((unsigned long long *)(q == 0 ? &cs->tie_got1 : &cs->tie_got5), 1ull);
            if ((long long)g < cap) (q == 0 ? list1 : list5)[g] = kk;
        }
    };
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * TIE_U;
    for (int64_t p0 = (int64_t)blockIdx.x * blockDim.x * TIE_U + threadIdx.x; p0 < n; p0 += stride) {
        double v[TIE_U];
#pragma unroll
        for (int e = 0; e < TIE_U; ++e) v[e] = ann_ldc(prob, p0 + (int64_t)e * blockDim.x, n);
#pragma unroll
        for (int e = 0; e < TIE_U; ++e) {
            const int64_t p = p0 + (int64_t)e * blockDim.x;
            if (p >= n || !(v[e] >= 0.0)) continue;
            const bool m1 = b1 >= 0 && v[e] == t1, m5 = b5 >= 0 && v[e] == t5;
            if (m1 || m5) {
                const unsigned long long kk = ann_tie_scramble(p);
                const long long bin = (long long)(kk >> TIE_SHIFT);
                if (m1 && bin == b1) put(0, kk);
                if (m5 && bin == b5) put(1, kk);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const uint32_t m = min(lcnt[threadIdx.x], (uint32_t)TIE_LCAP);
        lbase[threadIdx.x] = m ? atomicAdd((unsigned long long *)(threadIdx.x == 0 ? &cs->tie_got1 : &cs->tie_got5), (unsigned long long)m) : 0ull;
    }
    __syncthreads();
    for (int q = 0; q < 2; ++q) {
        const uint32_t m = min(lcnt[q], (uint32_t)TIE_LCAP);
        unsigned long long *list = q == 0 ? list1 : list5;
        for (uint32_t t = threadIdx.x; t < m; t += blockDim.x)
            if ((long long)(lbase[q] + t) < cap) list[lbase[q] + t] = lbuf[q][t];
    }
}

// one workgroup: the tie_rem-th smallest key of each short list (MSB-first byte radix)
__global__ __launch_bounds__(1024) void k_tie_select(CutState *__restrict__ cs, const unsigned long long *__restrict__ list1,
                                                    const unsigned long long *__restrict__ list5, long long cap)
{
    __shared__ uint32_t hist[256];
    __shared__ unsigned long long prefix_s;
    __shared__ long long krem_s;
    for (int q = 0; q < 2; ++q) {
        const long long bin = q == 0 ? cs->tie_bin1 : cs->tie_bin5;
        if (bin < 0) continue;     // rk already final
        const long long cnt = q == 0 ? cs->tie_len1 : cs->tie_len5;
        const long long e = q == 0 ? cs->tie_rem1 : cs->tie_rem5;
        const unsigned long long *list = q == 0 ? list1 : list5;
        unsigned long long rk = 0ull;
        if (cnt > cap) { if (threadIdx.x == 0) cs->tie_overflow = 1; }   // rk = 0: take none for now (the lists hold at most K entries): the host resolves it
        else {
            if (threadIdx.x == 0) { prefix_s = 0; krem_s = e - 1; }
            __syncthreads();
            for (int pass = 1; pass < 8; ++pass) {   // keys are < 2^53: the top byte is zero
                const int shift = 56 - 8 * pass;
                const unsigned long long himask = ~0ull << (shift + 8);
                if (threadIdx.x < 256) hist[threadIdx.x] = 0;
                __syncthreads();
                const unsigned long long pre = prefix_s;
                for (long long t = threadIdx.x; t < cnt; t += 1024) {
                    const unsigned long long kk = list[t];
                    if ((kk & himask) == pre) atomicAdd(&hist[(uint32_t)(kk >> shift) & 0xffu], 1u);
                }
                __syncthreads();
                if (threadIdx.x < 64) {   // one wave locates the digit: 4 bins per lane + a wave scan
                    uint32_t c4[4], s4 = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { c4[j] = hist[threadIdx.x * 4 + j]; s4 += c4[j]; }
                    uint32_t inc = s4;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) { const uint32_t up = __shfl_up(inc, off); if ((int)threadIdx.x >= off) inc += up; }
                    long long k = krem_s;
                    long long acc4 = (long long)(inc - s4);
                    if (acc4 <= k && k < acc4 + (long long)s4) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (acc4 <= k && k < acc4 + (long long)c4[j]) {
                                prefix_s = pre | ((unsigned long long)(threadIdx.x * 4 + j) << shift);
                                krem_s = k - acc4;
                            }
                            acc4 += c4[j];
                        }
                    }
                }
                __syncthreads();
            }
            rk = prefix_s;
        }
        if (threadIdx.x == 0) { if (q == 0) cs->rk1 = rk; else cs->rk5 = rk; }
        __syncthreads();
    }
}

__global__ void k_tie_flags(const double *__restrict__ prob, int64_t n, double t, uint8_t *__restrict__ flag, double *__restrict__ scr)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) {
        flag[p] = prob[p] == t ? 1 : 0;   // prob >= 0 only on not-computed pairs, and t >= 0 here
        scr[p] = (double)ann_tie_scramble(p);
    }
}

// class of a pair against one cut: 2 = above (taken), 1 = on the cut (taken in position order), 0 = below
__device__ __forceinline__ int cut_class(double v, double t, unsigned long long rk, const double *RA, int64_t p)
{
    if (v > t) return 2;
    if (v != t) return 0;
    if (rk == ~0ull) return 2;          // the whole group is taken
    const unsigned long long kk = ann_tie_scramble(p);
    return kk < rk ? 2 : (kk == rk ? 1 : 0);
}

__global__ __launch_bounds__(CP_THREADS) void k_cut_count(const double *__restrict__ prob, const double *__restrict__ RA, int64_t n,
                                                         const CutState *__restrict__ cs, uint32_t *__restrict__ blk, int nsub)
{
    __shared__ uint32_t acc[4];
    if (threadIdx.x < 4) acc[threadIdx.x] = 0;
    __syncthreads();
    const double t1 = cs->t1, t5 = cs->t5;
    const unsigned long long rk1 = cs->rk1, rk5 = cs->rk5;
    uint32_t g1 = 0, q1 = 0, g5 = 0, q5 = 0;
    for (int sub = 0; sub < nsub; ++sub) {
        const int64_t base = ((int64_t)blockIdx.x * nsub + sub) * CP_SUBTILE;
        if (base >= n) break;
        double v[CP_ITEMS];
#pragma unroll
        for (int k = 0; k < CP_ITEMS; ++k) v[k] = ann_ldc(prob, base + (int64_t)k * CP_THREADS + threadIdx.x, n);   // one batch in flight
#pragma unroll
        for (int k = 0; k < CP_ITEMS; ++k) {
            const int64_t p = base + (int64_t)k * CP_THREADS + threadIdx.x;
            if (p < n && v[k] >= 0.0) {
                const int c1 = cut_class(v[k], t1, rk1, RA, p), c5 = cut_class(v[k], t5, rk5, RA, p);
                g1 += c1 == 2; q1 += c1 == 1; g5 += c5 == 2; q5 += c5 == 1;
            }
        }
    }
    // four per-lane counters packed into one wave reduction (each <= 64 * CP_ITEMS * nsub = 2048 < 2^16)
    unsigned long long pk = (unsigned long long)g1 | ((unsigned long long)q1 << 16) | ((unsigned long long)g5 << 32) |
                            ((unsigned long long)q5 << 48);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pk += __shfl_xor(pk, off);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&acc[0], (uint32_t)(pk & 0xffff)); atomicAdd(&acc[1], (uint32_t)((pk >> 16) & 0xffff));
        atomicAdd(&acc[2], (uint32_t)((pk >> 32) & 0xffff)); atomicAdd(&acc[3], (uint32_t)(pk >> 48));
    }
    __syncthreads();
    if (threadIdx.x < 4) blk[(size_t)blockIdx.x * 4 + threadIdx.x] = acc[threadIdx.x];
}

// single block: totals, then per-tile offsets {eq1 prefix, eq5 prefix, cand offset, next offset}.
// 1024 threads own four consecutive tiles each per step and scan two 32-bit counters packed in
// one 64-bit word (every total is < 2^31: pair positions are int32), so 62 K tiles (127 M pairs)
// take 16 steps of two workgroup scans instead of 243 steps of four.
#define CS_THREADS 1024
#define CS_ITEMS 4
__global__ __launch_bounds__(CS_THREADS) void k_cut_scan(const uint32_t *__restrict__ blk, int nb, CutState *__restrict__ cs,
                                                        int64_t *__restrict__ off)
{
    __shared__ unsigned long long wsum[CS_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto block_scan = [&](unsigned long long v, unsigned long long *total) -> unsigned long long {
        unsigned long long inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { unsigned long long x = __shfl_up(inc, o); if (lane >= o) inc += x; }
        __syncthreads();
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        unsigned long long base = 0, tot = 0;
        for (int w = 0; w < CS_THREADS / 64; ++w) { const unsigned long long x = wsum[w]; if (w < wave) base += x; tot += x; }
        *total = tot;
        return base + inc - v;
    };
    const uint4 *blk4 = reinterpret_cast<const uint4 *>(blk);   // {gt1, eq1, gt5, eq5} per tile
    // totals of gt1 / gt5
    unsigned long long g = 0;
    for (int t0 = threadIdx.x; t0 < nb; t0 += CS_THREADS * CS_ITEMS) {
        uint4 b[CS_ITEMS];
#pragma unroll
        for (int e = 0; e < CS_ITEMS; ++e) b[e] = ann_ldc(blk4, t0 + e * CS_THREADS, nb);
#pragma unroll
        for (int e = 0; e < CS_ITEMS; ++e)
            if (t0 + e * CS_THREADS < nb) g += (unsigned long long)b[e].x | ((unsigned long long)b[e].z << 32);
    }
    unsigned long long T;
    block_scan(g, &T);
    const int64_t T1 = (int64_t)(T & 0xffffffffull), T5 = (int64_t)(T >> 32);
    const int64_t e1 = cs->all1 ? (1ll << 62) : cs->K1 - T1;
    const int64_t e5 = cs->all5 ? (1ll << 62) : cs->K5 - T5;
    const bool both_all = cs->all1 && cs->all5;
    unsigned long long c_eq = 0, c_out = 0;   // running {eq1 | eq5 << 32}, {cand | next << 32}
    for (int base = 0; base < nb; base += CS_THREADS * CS_ITEMS) {
        const int t0 = base + threadIdx.x * CS_ITEMS;   // thread-contiguous tiles: scan order = tile order
        uint4 b[CS_ITEMS];
#pragma unroll
        for (int e = 0; e < CS_ITEMS; ++e) {
            b[e] = ann_ldc(blk4, t0 + e, nb);
            if (t0 + e >= nb) b[e] = make_uint4(0, 0, 0, 0);
        }
        unsigned long long eqs = 0;
#pragma unroll
        for (int e = 0; e < CS_ITEMS; ++e) eqs += (unsigned long long)b[e].y | ((unsigned long long)b[e].w << 32);
        unsigned long long tot;
        unsigned long long ex = c_eq + block_scan(eqs, &tot);
        c_eq += tot;
        int64_t p1[CS_ITEMS], p5[CS_ITEMS];
        unsigned long long outv[CS_ITEMS], outs = 0;
#pragma unroll
        for (int e = 0; e < CS_ITEMS; ++e) {
            p1[e] = (int64_t)(ex & 0xffffffffull);
            p5[e] = (int64_t)(ex >> 32);
            const int64_t eq1 = b[e].y, eq5 = b[e].w;
            const int64_t take1 = min(eq1, max((int64_t)0, e1 - p1[e]));
            const int64_t take5 = min(eq5, max((int64_t)0, e5 - p5[e]));
            const int64_t ncand = (int64_t)b[e].x + take1;
            const int64_t nbig = (int64_t)b[e].z + take5;
            // when everything is taken for both lists, `next` is the full list again (annchor.py:444-446)
            const int64_t nnext = both_all ? nbig : nbig - ncand;
            outv[e] = (unsigned long long)ncand | ((unsigned long long)nnext << 32);
            outs += outv[e];
            ex += (unsigned long long)eq1 | ((unsigned long long)eq5 << 32);
        }
        unsigned long long exo = c_out + block_scan(outs, &tot);
        c_out += tot;
#pragma unroll
        for (int e = 0; e < CS_ITEMS; ++e) {
            if (t0 + e < nb) {
                int64_t *o = off + (size_t)(t0 + e) * 4;
                o[0] = p1[e]; o[1] = p5[e]; o[2] = (int64_t)(exo & 0xffffffffull); o[3] = (int64_t)(exo >> 32);
            }
            exo += outv[e];
        }
    }
    if (threadIdx.x == 0) { cs->e1 = e1; cs->e5 = e5; cs->ncand = (int64_t)(c_out & 0xffffffffull); cs->nnext = (int64_t)(c_out >> 32); }
}

__global__ __launch_bounds__(CP_THREADS) void k_cut_emit(const double *__restrict__ prob, const double *__restrict__ RA, int64_t n,
                                                        const CutState *__restrict__ cs, const int64_t *__restrict__ off,
                                                        int32_t *__restrict__ cand, int32_t *__restrict__ next, int nsub)
{
    __shared__ uint32_t wsum[CP_THREADS / 64];
    const double t1 = cs->t1, t5 = cs->t5;
    const unsigned long long rk1 = cs->rk1, rk5 = cs->rk5;
    const int64_t e1 = cs->e1, e5 = cs->e5;
    const bool both_all = cs->all1 && cs->all5;
    auto scan2 = [&](uint32_t a, uint32_t b, uint32_t *ea, uint32_t *eb, uint32_t *ta, uint32_t *tb) {
        // packs two counters (each < 2^16 per sub-tile) into one 32-bit scan
        uint32_t tot;
        const uint32_t packed = row_block_scan((a << 16) | b, wsum, &tot);
        __syncthreads();
        *ea = packed >> 16;
        *eb = packed & 0xffffu;
        *ta = tot >> 16;
        *tb = tot & 0xffffu;
    };
    // running offsets of the tile: class-1 ranks of the two cuts, write cursors of the two lists
    int64_t R1 = off[(size_t)blockIdx.x * 4], R5 = off[(size_t)blockIdx.x * 4 + 1];
    int64_t WC = off[(size_t)blockIdx.x * 4 + 2], WN = off[(size_t)blockIdx.x * 4 + 3];
    for (int sub = 0; sub < nsub; ++sub) {
        const int64_t sbase = ((int64_t)blockIdx.x * nsub + sub) * CP_SUBTILE;
        if (sbase >= n) break;   // (uniform)
        const int64_t base = sbase + (int64_t)threadIdx.x * CP_ITEMS;  // thread-contiguous: keeps position order
        double v[CP_ITEMS];
        int8_t k1[CP_ITEMS], k5[CP_ITEMS];
        uint32_t q1 = 0, q5 = 0;
#pragma unroll
        for (int k = 0; k < CP_ITEMS; ++k) {
            v[k] = (base + k < n) ? prob[base + k] : -1.0;
            k1[k] = k5[k] = 0;
            if (v[k] >= 0.0) {
                k1[k] = (int8_t)cut_class(v[k], t1, rk1, RA, base + k);
                k5[k] = (int8_t)cut_class(v[k], t5, rk5, RA, base + k);
                q1 += k1[k] == 1; q5 += k5[k] == 1;
            }
        }
        uint32_t x1, x5, tq1, tq5;
        scan2(q1, q5, &x1, &x5, &tq1, &tq5);
        int64_t r1 = R1 + x1, r5 = R5 + x5;
        uint8_t fc[CP_ITEMS], fn[CP_ITEMS];
        uint32_t nc = 0, nn = 0;
#pragma unroll
        for (int k = 0; k < CP_ITEMS; ++k) {
            bool c1 = false, c5 = false;
            if (v[k] >= 0.0) {
                c1 = k1[k] == 2 || (k1[k] == 1 && r1 < e1);
                c5 = k5[k] == 2 || (k5[k] == 1 && r5 < e5);
                r1 += k1[k] == 1;
                r5 += k5[k] == 1;
            }
            fc[k] = c1;
            fn[k] = both_all ? c5 : (c5 && !c1);
            nc += fc[k];
            nn += fn[k];
        }
        uint32_t oc, on, tc, tn;
        scan2(nc, nn, &oc, &on, &tc, &tn);
        int64_t wc = WC + oc, wn = WN + on;
#pragma unroll
        for (int k = 0; k < CP_ITEMS; ++k) {
            if (fc[k]) cand[wc++] = (int32_t)(base + k);
            if (fn[k]) next[wn++] = (int32_t)(base + k);
        }
        R1 += tq1; R5 += tq5; WC += tc; WN += tn;
    }
}

// the selection's device state: CutState, then (at SEL_ERR_OFF) the int32 error flag of the guarantee_nmin sweep
#define SEL_ERR_OFF 256
#define SEL_STATE_BYTES (SEL_ERR_OFF + 256)
static_assert(sizeof(CutState) <= SEL_ERR_OFF, "the error flag sits behind the cut state");
static inline int32_t *gn_err_ptr(annchor_ctx *c) { return reinterpret_cast<int32_t *>(c->sel_state.as<char>() + SEL_ERR_OFF); }
size_t ann_sel_state_bytes() { return SEL_STATE_BYTES; }
#define GN_BATCH 8   // guarantee_nmin rounds between two looks at the "changed" flags

// Stage A of the selection: row thresholds and guarantee_nmin (lists, first batch of rounds) -- functions of
// RefineApprox and the mask only, launched without a host wait.
static int select_stage_a(annchor_ctx *c, int32_t n_neighbors, int32_t nmin)
{
    const int64_t n = c->n, nx = c->nx;
    c->gn_pending = false;
    ANN_TRY(ann_reserve(c, c->thresh, sizeof(double) * (size_t)nx));
    ANN_CHECK_HIP(c, hipEventRecord(c->call_a, c->stream));
    RowSrc rsrc;
    ANN_TRY(ann_transpose_columns(c, &rsrc));   // large lists: column-ordered copy of the column-like halves (shared by the two row kernels below)
    {
        // algorithmic bytes: every pair value is read from both of its rows: 2n * (8 + 4)
        ProfScope ps(c, "row_kth_threshold", (double)n * 24.0);
        int cap = 2;   // streamed rows: no LDS copy (a rare fallback row re-reads global memory)
        if (!rsrc.T) ANN_TRY(row_pick_cap(c, k_row_thresh, nx, nx, 0, &cap));
        ANN_TRY(row_lds_prepare(c, k_row_thresh, (size_t)cap * 8));
        k_row_thresh<<<(int)nx, ROW_THREADS, (size_t)cap * 8, c->stream>>>(c->Iptr.as<int64_t>(), rsrc, (uint32_t)n_neighbors,
                                                                         c->thresh.as<double>(), cap);
    }
    if (nmin > 0) {
        const int L = nmin + 1;
        ANN_REQUIRE(c, L <= 1024, ANNCHOR_ELIMIT, "nmin=%d too large", nmin);
        ANN_TRY(ann_reserve(c, c->gl_val, sizeof(double) * (size_t)nx * L));
        ANN_TRY(ann_reserve(c, c->gl_pos, sizeof(int32_t) * (size_t)nx * L * 3));  // pos | other endpoint | twin
        ANN_TRY(ann_reserve(c, c->gl_cnt, sizeof(int32_t) * (size_t)nx));
        ANN_TRY(ann_reserve(c, c->gl_ncomp, sizeof(int32_t) * (size_t)nx));
        ANN_TRY(ann_reserve(c, c->sel_state, SEL_STATE_BYTES));   // (the sweep's error flag sits behind the cut state: one download brings both)
        if (!c->gn_err_clean) {   // sweep error flag (it stays zero unless a sweep fails: cleared again only then)
            ANN_CHECK_HIP(c, hipMemsetAsync(gn_err_ptr(c), 0, 4, c->stream));
        }
        c->gn_err_clean = false;   // (until the flag has been read back as zero)
        {
            ProfScope ps(c, "guarantee_nmin_lists", (double)n * 26.0);
            const size_t tail = (((size_t)L * 12) + 15) & ~(size_t)15;
            int cap = 2;
            if (!rsrc.T) ANN_TRY(row_pick_cap(c, k_gn_lists, nx, nx, tail, &cap));
            ANN_TRY(row_lds_prepare(c, k_gn_lists, (size_t)cap * 8 + tail));
            k_gn_lists<<<(int)nx, ROW_THREADS, (size_t)cap * 8 + tail, c->stream>>>(
                c->Iptr.as<int64_t>(), rsrc, c->ij.as<int2>(), L,
                c->gl_val.as<double>(), c->gl_pos.as<int32_t>(), c->gl_pos.as<int32_t>() + (size_t)nx * L,
                c->gl_cnt.as<int32_t>(), c->gl_ncomp.as<int32_t>(), cap);
        }
        const size_t sweep_lds = (size_t)nx * (((size_t)L + 31) / 32 * 4 + 4);
        const size_t ring_lds = sweep_lds + 2 * (size_t)GN_B * L * 20 + 64 * 20 + 4 * GN_B * 4 + 64 + 16;
        // ANNCHOR_GN_SWEEP = rounds (default) | sequential: the one-wave walks below (tests compare the two)
        const char *gn_env = getenv("ANNCHOR_GN_SWEEP");
        if (!gn_env || strcmp(gn_env, "sequential") != 0) {
            ProfScope ps(c, "guarantee_nmin_sweep", (double)nx * L * 21.0);
            int32_t *oth = c->gl_pos.as<int32_t>() + (size_t)nx * L, *twin = oth + (size_t)nx * L;
            k_gn_twin<<<ann_blocks(nx * L, 256), 256, 0, c->stream>>>(nx, L, c->gl_pos.as<int32_t>(), oth,
                                                                     c->gl_cnt.as<int32_t>(), twin);
            const int Lw = (L + 31) / 32;
            const size_t mask_bytes = sizeof(uint32_t) * (size_t)nx * Lw, mout_bytes = sizeof(int32_t) * (size_t)nx;

            ANN_TRY(ann_reserve(c, c->gn_state, 2 * mask_bytes + 3 * mout_bytes + sizeof(int32_t) * GN_BATCH));
            uint32_t *masks[2] = {c->gn_state.as<uint32_t>(), c->gn_state.as<uint32_t>() + (size_t)nx * Lw};
            int32_t *mout[3];
            for (int q = 0; q < 3; ++q) mout[q] = reinterpret_cast<int32_t *>(c->gn_state.as<char>() + 2 * mask_bytes + q * mout_bytes);
            int32_t *changed = reinterpret_cast<int32_t *>(c->gn_state.as<char>() + 2 * mask_bytes + 3 * mout_bytes);
            // (masks, counters and the round flags behind them: one memset)
            ANN_CHECK_HIP(c, hipMemsetAsync(c->gn_state.p, 0, 2 * mask_bytes + 3 * mout_bytes + sizeof(int32_t) * GN_BATCH, c->stream));
            // first batch of rounds now; whether they settled is looked at in select_stage_finish (the host may do other
            // work in between: annchor_select_prepare)
            int round = 0;
            for (int q = 0; q < GN_BATCH; ++q, ++round)
                k_gn_round<<<ann_blocks(nx * 64, 256), 256, 0, c->stream>>>(
                    nx, nmin, L, Lw, c->gl_val.as<double>(), oth, twin, c->gl_cnt.as<int32_t>(), c->gl_ncomp.as<int32_t>(),
                    masks[round & 1], masks[(round + 1) & 1], mout[round % 3], mout[(round + 1) % 3], mout[(round + 2) % 3],
                    changed + q, gn_err_ptr(c));
            c->gn_pending = true; c->gn_round = round; c->gn_L = L;
        } else if (L <= GN_LMAX && ring_lds <= 156 * 1024) {
            ProfScope ps(c, "guarantee_nmin_sweep", (double)nx * L * 21.0);
            int32_t *oth = c->gl_pos.as<int32_t>() + (size_t)nx * L, *twin = oth + (size_t)nx * L;
            k_gn_twin<<<ann_blocks(nx * L, 256), 256, 0, c->stream>>>(nx, L, c->gl_pos.as<int32_t>(), oth,
                                                                     c->gl_cnt.as<int32_t>(), twin);
            ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)k_gn_sweep_ring, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)ring_lds));
            k_gn_sweep_ring<<<1, 256, ring_lds, c->stream>>>(nx, nmin, L, c->gl_val.as<double>(), c->gl_pos.as<int32_t>(), oth, twin,
                                                            c->gl_cnt.as<int32_t>(), c->gl_ncomp.as<int32_t>(),
                                                            c->RA.as<double>(), gn_err_ptr(c));
        } else if (sweep_lds <= 150 * 1024) {
            ProfScope ps(c, "guarantee_nmin_sweep", (double)nx * L * 21.0);
            int32_t *oth = c->gl_pos.as<int32_t>() + (size_t)nx * L, *twin = oth + (size_t)nx * L;
            k_gn_twin<<<ann_blocks(nx * L, 256), 256, 0, c->stream>>>(nx, L, c->gl_pos.as<int32_t>(), oth,
                                                                     c->gl_cnt.as<int32_t>(), twin);
            if (sweep_lds > 64 * 1024)
                ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)k_gn_sweep_lds, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     (int)sweep_lds));
            k_gn_sweep_lds<<<1, 64, sweep_lds, c->stream>>>(nx, nmin, L, c->gl_val.as<double>(), c->gl_pos.as<int32_t>(), oth,
                                                           twin, c->gl_cnt.as<int32_t>(), c->gl_ncomp.as<int32_t>(),
                                                           c->RA.as<double>(), gn_err_ptr(c));
        } else {
            // (the row walk's own marks: only this route needs them -- two memsets, one of them a byte per pair, used to run
            // in front of every selection)
            ANN_TRY(ann_reserve(c, c->marked, (size_t)n));
            ANN_TRY(ann_reserve(c, c->markcount, sizeof(int32_t) * (size_t)nx));
            ANN_CHECK_HIP(c, hipMemsetAsync(c->marked.p, 0, (size_t)n, c->stream));
            ANN_CHECK_HIP(c, hipMemsetAsync(c->markcount.p, 0, sizeof(int32_t) * (size_t)nx, c->stream));
            ProfScope ps(c, "guarantee_nmin_sweep", (double)nx * L * 13.0);
            k_gn_sequential<<<1, 64, 0, c->stream>>>(nx, nmin, L, c->gl_val.as<double>(), c->gl_pos.as<int32_t>(),
                                                    c->gl_cnt.as<int32_t>(), c->gl_ncomp.as<int32_t>(), c->ij.as<int2>(),
                                                    c->RA.as<double>(), c->marked.as<uint8_t>(), c->markcount.as<int32_t>(),
                                                    gn_err_ptr(c));
        }
    }
    return ANNCHOR_OK;
}

// The rounds launched by stage A: read their flags, run further batches until one settles, apply the marks.
static int select_stage_finish(annchor_ctx *c)
{
    if (!c->gn_pending) return ANNCHOR_OK;
    const int64_t nx = c->nx;
    const int L = c->gn_L, Lw = (L + 31) / 32, nmin = L - 1;
    const size_t mask_bytes = sizeof(uint32_t) * (size_t)nx * Lw, mout_bytes = sizeof(int32_t) * (size_t)nx;
    uint32_t *masks[2] = {c->gn_state.as<uint32_t>(), c->gn_state.as<uint32_t>() + (size_t)nx * Lw};
    int32_t *mout[3];
    for (int q = 0; q < 3; ++q) mout[q] = reinterpret_cast<int32_t *>(c->gn_state.as<char>() + 2 * mask_bytes + q * mout_bytes);
    int32_t *changed = reinterpret_cast<int32_t *>(c->gn_state.as<char>() + 2 * mask_bytes + 3 * mout_bytes);
    int32_t *oth = c->gl_pos.as<int32_t>() + (size_t)nx * L, *twin = oth + (size_t)nx * L;
    int round = c->gn_round;
    for (;;) {
        int32_t h_changed[GN_BATCH];
        ANN_TRY(ann_d2h(c, h_changed, changed, sizeof h_changed));
        // a round without a change: its input (and output) is the fixed point, and so is everything after it
        if (h_changed[GN_BATCH - 1] == 0) break;
        ANN_REQUIRE(c, round <= (int)nx + GN_BATCH, ANNCHOR_EHIP, "guarantee_nmin rounds did not settle");
        ANN_CHECK_HIP(c, hipMemsetAsync(changed, 0, sizeof(int32_t) * GN_BATCH, c->stream));
        for (int q = 0; q < GN_BATCH; ++q, ++round)
            k_gn_round<<<ann_blocks(nx * 64, 256), 256, 0, c->stream>>>(
                nx, nmin, L, Lw, c->gl_val.as<double>(), oth, twin, c->gl_cnt.as<int32_t>(), c->gl_ncomp.as<int32_t>(),
                masks[round & 1], masks[(round + 1) & 1], mout[round % 3], mout[(round + 1) % 3], mout[(round + 2) % 3],
                changed + q, gn_err_ptr(c));
    }
    k_gn_apply<<<ann_blocks(nx * L, 256), 256, 0, c->stream>>>(nx, L, Lw, c->gl_pos.as<int32_t>(), c->gl_cnt.as<int32_t>(),
                                                              masks[round & 1], c->RA.as<double>());
    ANN_CHECK_HIP(c, hipGetLastError());
    c->gn_pending = false;
    return ANNCHOR_OK;
}

// Thresholds and guarantee_nmin ahead of annchor_select_candidates (same n_neighbors / nmin): the caller fits its
// error model on the host while they run.  Anything that changes RefineApprox or the mask in between voids it.
size_t ann_tie_hist_bytes() { return sizeof(uint32_t) * 2 * TIE_BINS; }

extern "C" int annchor_select_prepare(annchor_ctx *c, int32_t n_neighbors, int32_t nmin)
{
    if (!c) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->have_RA, ANNCHOR_EINVAL, "RefineApprox not initialised");
    ANN_REQUIRE(c, n_neighbors >= 1 && nmin >= 0, ANNCHOR_EINVAL, "bad selection parameters");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    ANN_TRY(select_stage_a(c, n_neighbors, nmin));
    c->sel_prepared = true; c->sel_k = n_neighbors; c->sel_nmin = nmin;
    return ANNCHOR_OK;
}

extern "C" int annchor_select_candidates(annchor_ctx *c, int32_t n_neighbors, int32_t nmin, const double *errs,
                                         const int64_t *err_ptr, int32_t nlabels, int64_t n_refine, int32_t lookahead,
                                         int64_t *n_cand, int64_t *n_next)
{
    if (!c || (errs && !err_ptr) || !n_cand || !n_next) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->have_RA, ANNCHOR_EINVAL, "RefineApprox not initialised");
    // errs == NULL: the residual lists annchor_fit_errors_device left in device memory (nlabels = its partitions)
    const bool dev_errs = errs == nullptr;
    ANN_REQUIRE(c, !dev_errs || (c->errs_on_device && nlabels == c->model_nb), ANNCHOR_ESTATE,
                "no device-resident residual lists for %d labels (annchor_fit_errors_device)", nlabels);
    ANN_REQUIRE(c, nlabels >= 1 && nlabels <= 255, ANNCHOR_ELIMIT, "1..255 error labels supported");
    ANN_REQUIRE(c, n_neighbors >= 1 && lookahead >= 1 && n_refine >= 0, ANNCHOR_EINVAL, "bad selection parameters");
    if (!dev_errs)
        for (int b = 0; b < nlabels; ++b)
            ANN_REQUIRE(c, err_ptr[b + 1] > err_ptr[b], ANNCHOR_ESTATE, "error bin %d has no samples", b);
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    const int64_t n = c->n, nx = c->nx;
    if (!(c->sel_prepared && c->sel_k == n_neighbors && c->sel_nmin == nmin)) ANN_TRY(select_stage_a(c, n_neighbors, nmin));
    c->sel_prepared = false;
    ANN_TRY(select_stage_finish(c));
    (void)nx;
    // ---- probabilities
    // device-resident lists: their exact total is on the device; a sample on an inner edge counts twice, so <= 2 m
    const int64_t nerr = dev_errs ? 2 * c->nsamp + 64 : err_ptr[nlabels];   // (upper bound when the lists live on the device)
    if (!dev_errs) {
        ANN_TRY(ann_reserve(c, c->errs, sizeof(double) * (size_t)nerr));
        ANN_TRY(ann_reserve(c, c->errptr, sizeof(int64_t) * (size_t)(MAXBINS > nlabels ? MAXBINS + 1 : nlabels + 1)));
        ANN_TRY(ann_h2d(c, c->errs.p, errs, sizeof(double) * (size_t)nerr));
        ANN_TRY(ann_h2d(c, c->errptr.p, err_ptr, sizeof(int64_t) * (size_t)(nlabels + 1)));
        c->errs_on_device = false; c->model_cache_valid = false;
    }
    {
        // LDS copy of the lists: exact size known (host lists), or room for the usual case of the device lists (every
        // sample once + a few on shared edges); the kernel checks the real total against the capacity
        const int64_t lds_entries = dev_errs ? c->nsamp + 64 : nerr;
        const int in_lds = lds_entries * 8 <= 60 * 1024;
        const size_t dyn = in_lds ? (size_t)lds_entries * 8 : 0;
        int blocks = min(ann_blocks(n, 256 * 8), c->prop.multiProcessorCount * 8);
        // algorithmic bytes per pair: 8 (ij) + 8 (RA) + 2 (mask, label) + 8 (prob)
        ProfScope ps(c, "ecdf_probability", (double)n * 26.0);
        // long lists: a bucket index over every label's errors (one small launch) replaces most of the search
        static const long long index_min = getenv("ANNCHOR_ECDF_INDEX_MIN") ? atoll(getenv("ANNCHOR_ECDF_INDEX_MIN")) : (4ll << 20);
        EcdfIndex *index = nullptr;
        uint32_t *table = nullptr;
        if (n >= index_min && nlabels > 0 && nerr > 0 && nerr < (1ll << 29)) {
            const size_t idx_bytes = (sizeof(EcdfIndex) * (size_t)nlabels + 15) & ~(size_t)15;
            ANN_TRY(ann_reserve(c, c->ecdf_index, idx_bytes + sizeof(uint32_t) * (size_t)(2 * nerr + nlabels + 1)));
            index = c->ecdf_index.as<EcdfIndex>();
            table = reinterpret_cast<uint32_t *>(c->ecdf_index.as<char>() + idx_bytes);
            k_ecdf_index<<<nlabels, 256, 0, c->stream>>>(c->errs.as<double>(), c->errptr.as<int64_t>(), index, table);
        }
        k_prob<<<blocks, 256, dyn, c->stream>>>(n, c->ij.as<int2>(), c->thresh.as<double>(), c->RA.as<double>(),
                                               c->ncm.as<uint8_t>(), c->label.as<uint8_t>(), c->errs.as<double>(),
                                               c->errptr.as<int64_t>(), nlabels, in_lds ? (int)lds_entries : 0, c->prob.as<double>(), n >= ANN_STREAM_MIN_PAIRS,
                                               index, table);
    }
    ANN_CHECK_HIP(c, hipGetLastError());
    // (the sweep's error flag is read with the final state below: one host wait less)
    // ---- cut values
    int64_t n_unc = 0;
    ANN_TRY(annchor_count_uncomputed(c, &n_unc));
    CutState cs;
    memset(&cs, 0, sizeof cs);
    cs.K1 = n_refine;
    cs.K5 = n_refine * (int64_t)lookahead;
    cs.all1 = n_refine >= n_unc;
    cs.all5 = cs.all1 || cs.K5 >= n_unc;
    cs.t1 = cs.t5 = INFINITY;
    // The cut values stay on the device when the selection can be enqueued without a host wait (lists short of the sampled
    // bracket): the cut kernels read them from the state, the "could not finish" flag comes back with the final state and the
    // cut is redone the waiting way if it is set (degenerate keys; never at the reference's configurations).
    const unsigned long long *d_cut_prefix = nullptr;
    const int *d_cut_unfinished = nullptr;
    int64_t cut_ks[2];
    int cut_nk = 0;
    auto cut_values_waiting = [&]() -> int {
        double tv[2];
        // flag = ncm: prob >= 0 exactly on not-computed pairs
        ANN_TRY(ann_kth_smallest(c, c->prob.as<double>(), c->ncm.as<uint8_t>(), n, cut_ks, cut_nk, tv));
        int q = 0;
        if (!cs.all1) cs.t1 = tv[q++];
        if (!cs.all5) cs.t5 = tv[q++];
        return ANNCHOR_OK;
    };
    if (cs.all1) cs.t1 = -1.0;
    if (cs.all5) cs.t5 = -1.0;
    if (n_refine == 0) { cs.all1 = cs.all5 = 0; cs.t1 = cs.t5 = INFINITY; cs.K1 = cs.K5 = 0; }
    const int nsub = cp_nsub(n);
    const int nb = ann_blocks(n, (int64_t)CP_SUBTILE * nsub);
    ANN_TRY(ann_reserve(c, c->sel_state, SEL_STATE_BYTES));
    ANN_TRY(ann_reserve(c, c->blk_cnt, sizeof(uint32_t) * 4 * (size_t)nb));
    ANN_TRY(ann_reserve(c, c->blk_off, sizeof(int64_t) * 4 * (size_t)nb));
    const int64_t maxc = cs.all1 ? n_unc : cs.K1, maxn = cs.all5 ? n_unc : cs.K5;
    ANN_TRY(ann_reserve(c, c->cand, sizeof(int32_t) * (size_t)(maxc + 1)));
    ANN_TRY(ann_reserve(c, c->next, sizeof(int32_t) * (size_t)(maxn + 1)));
    cs.rk1 = cs.rk5 = ~0ull;
    if (n_refine > 0 && !(cs.all1 && cs.all5)) {
        if (!cs.all1) cut_ks[cut_nk++] = n_unc - cs.K1;
        if (!cs.all5) cut_ks[cut_nk++] = n_unc - cs.K5;
        if (!getenv("ANNCHOR_CUT_WAIT")) {
            // (the selection's finishing workgroup writes the state the cut kernels read: the host's copy of it rides as a kernel argument)
            Sel2Epilogue ep;
            memset(&ep, 0, sizeof ep);
            ep.kind = 2; ep.cs = c->sel_state.as<CutState>(); ep.cs_init = cs; ep.need1 = !cs.all1; ep.need5 = !cs.all5;
            ep.force_redo = getenv("ANNCHOR_CUT_FORCE_REDO") ? 1 : 0;
            ANN_TRY(ann_kth_async(c, c->prob.as<double>(), c->ncm.as<uint8_t>(), n, cut_ks, cut_nk, &d_cut_prefix, &d_cut_unfinished, &ep));
        }
        if (!d_cut_prefix) ANN_TRY(cut_values_waiting());
    }
    const CutState cs_in = cs;   // (for a second attempt)
    if (!d_cut_prefix) ANN_TRY(ann_h2d(c, c->sel_state.p, &cs, sizeof cs));
    const bool ties = n_refine > 0 && !(cs.all1 && cs.all5);
    ANN_TRY(ann_reserve(c, c->tie_lists, sizeof(unsigned long long) * 2 * TIE_CAP));
    unsigned long long *tl1 = c->tie_lists.as<unsigned long long>(), *tl5 = tl1 + TIE_CAP;
    auto split = [&]() {
        ProfScope ps(c, "topk_split_compact", (double)n * 16.0 + (double)(maxc + maxn) * 4.0);
        k_cut_count<<<nb, CP_THREADS, 0, c->stream>>>(c->prob.as<double>(), c->RA.as<double>(), n, c->sel_state.as<CutState>(),
                                                     c->blk_cnt.as<uint32_t>(), nsub);
        k_cut_scan<<<1, CS_THREADS, 0, c->stream>>>(c->blk_cnt.as<uint32_t>(), nb, c->sel_state.as<CutState>(), c->blk_off.as<int64_t>());
        k_cut_emit<<<nb, CP_THREADS, 0, c->stream>>>(c->prob.as<double>(), c->RA.as<double>(), n, c->sel_state.as<CutState>(),
                                                    c->blk_off.as<int64_t>(), c->cand.as<int32_t>(), c->next.as<int32_t>(), nsub);
    };
    auto tie_groups = [&]() -> int {
        // the groups on the two cuts and their RefineApprox cuts (device only: no host wait)
        ProfScope ps(c, "topk_tie_groups", (double)n * 8.0);
        // (the two streaming passes: 256 workgroups, one per CU with 8 KB in flight each, read a 1 GB column at 1.8 TB/s)
        // three per CU: every workgroup ends with up to 8192 global atomics (its histogram), so more of them read faster and
        // flush longer -- 1 / 2 / 3 / 5 per CU: 1.06 / 0.91 / 0.88 / 0.92 ms per call at 127 M pairs
        const int tb = (int)std::min<int64_t>(ann_blocks(n, 256 * TIE_U), (int64_t)c->prop.multiProcessorCount * 3);
        const char *cap_env = getenv("ANNCHOR_TIE_CAP");   // tests force the large-group route on small inputs
        const long long cap = cap_env ? std::min<long long>(TIE_CAP, std::max<long long>(1, atoll(cap_env))) : TIE_CAP;
        ANN_TRY(ann_reserve(c, c->tie_hist, sizeof(uint32_t) * 2 * TIE_BINS));
        if (c->tie_hist_clean != c->tie_hist.p) {   // (k_tie_pick leaves it zeroed for the next call)
            ANN_CHECK_HIP(c, hipMemsetAsync(c->tie_hist.p, 0, sizeof(uint32_t) * 2 * TIE_BINS, c->stream));
            c->tie_hist_clean = c->tie_hist.p;
        }
        k_tie_hist<<<tb, 256, 0, c->stream>>>(c->prob.as<double>(), n, c->sel_state.as<CutState>(), c->tie_hist.as<uint32_t>());
        k_tie_pick<<<1, 1024, 0, c->stream>>>(c->sel_state.as<CutState>(), c->tie_hist.as<uint32_t>());
        k_tie_collect<<<tb, 256, 0, c->stream>>>(c->prob.as<double>(), n, c->sel_state.as<CutState>(), tl1, tl5, cap);
        k_tie_select<<<1, 1024, 0, c->stream>>>(c->sel_state.as<CutState>(), tl1, tl5, cap);
        return ANNCHOR_OK;
    };
    if (ties) ANN_TRY(tie_groups());
    split();
    ANN_CHECK_HIP(c, hipEventRecord(c->call_b, c->stream));
    c->call_timed = true;
    ANN_CHECK_HIP(c, hipGetLastError());
    if (nmin > 0) {
        unsigned char both[SEL_ERR_OFF + 4];
        ANN_TRY(ann_d2h(c, both, c->sel_state.p, sizeof both));
        int32_t e = 0;
        memcpy(&cs, both, sizeof cs);
        memcpy(&e, both + SEL_ERR_OFF, 4);
        c->gn_err_clean = e == 0;
        ANN_REQUIRE(c, e == 0, ANNCHOR_ESTATE, "guarantee_nmin: a row has fewer not-computed candidates than it must refine");
    } else {
        ANN_TRY(ann_d2h(c, &cs, c->sel_state.p, sizeof cs));
    }
    if (d_cut_prefix) {
        ann_kth_async_done(c);
        if (cs.sel_unfinished) {   // the selection needs its byte passes: cut values the waiting way, then everything above again
            cs = cs_in;
            ANN_TRY(cut_values_waiting());
            ANN_TRY(ann_h2d(c, c->sel_state.p, &cs, sizeof cs));
            if (ties) ANN_TRY(tie_groups());
            split();
            ANN_CHECK_HIP(c, hipGetLastError());
            ANN_TRY(ann_d2h(c, &cs, c->sel_state.p, sizeof cs));
        }
    }
    if (cs.tie_overflow) {
        // a group on a cut larger than TIE_CAP (e.g. a third of the pool at probability 0): its RefineApprox
        // cut comes from the general selection over all pairs, then the split is redone
        ANN_TRY(ann_reserve(c, c->marked, (size_t)n));
        for (int q = 0; q < 2; ++q) {
            const bool all = q == 0 ? cs.all1 : cs.all5;
            const int64_t cnt = q == 0 ? cs.tie_n1 : cs.tie_n5, e = q == 0 ? cs.K1 - cs.tie_gt1 : cs.K5 - cs.tie_gt5;
            unsigned long long rk = ~0ull;
            if (!all && e < cnt) {
                if (e <= 0) rk = 0ull;
                else {
                    ANN_TRY(ann_reserve(c, c->colT, sizeof(double) * (size_t)n));   // scratch: the scrambled positions as doubles (53 bits: exact)
                    k_tie_flags<<<ann_blocks(n, 256), 256, 0, c->stream>>>(c->prob.as<double>(), n, q == 0 ? cs.t1 : cs.t5, c->marked.as<uint8_t>(),
                                                                          c->colT.as<double>());
                    int64_t kq = e - 1;
                    double rv = 0;
                    ANN_TRY(ann_kth_smallest(c, c->colT.as<double>(), c->marked.as<uint8_t>(), n, &kq, 1, &rv));
                    rk = (unsigned long long)rv;
                }
            }
            if (q == 0) cs.rk1 = rk; else cs.rk5 = rk;
        }
        cs.tie_overflow = 0; cs.e1 = cs.e5 = 0; cs.ncand = cs.nnext = 0;
        ANN_TRY(ann_h2d(c, c->sel_state.p, &cs, sizeof cs));
        split();
        ANN_CHECK_HIP(c, hipGetLastError());
        ANN_TRY(ann_d2h(c, &cs, c->sel_state.p, sizeof cs));
    }
    c->ncand = cs.ncand;
    c->n_unc = n_unc;  // candidates are distinct not-computed pairs: refinement lowers the count by ncand
    c->cand_marked = false;
    c->nnext = cs.nnext;
    *n_cand = cs.ncand;
    *n_next = cs.nnext;
    return ANNCHOR_OK;
}

// ------------------------------------------------------------------- refinement
__global__ void k_mark_candidates(const int32_t *__restrict__ pos, int64_t m, uint8_t *__restrict__ ncm)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < m) ncm[pos[t]] = 0;
}

extern "C" int annchor_mark_candidates(annchor_ctx *c)
{
    if (!c) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->have_RA, ANNCHOR_EINVAL, "RefineApprox not initialised");
    if (c->ncand == 0 || c->cand_marked) return ANNCHOR_OK;
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    k_mark_candidates<<<ann_blocks(c->ncand, 256), 256, 0, c->stream>>>(c->cand.as<int32_t>(), c->ncand, c->ncm.as<uint8_t>());
    ANN_CHECK_HIP(c, hipGetLastError());
    if (c->n_unc >= 0) c->n_unc -= c->ncand;
    c->cand_marked = true; c->sel_prepared = false;
    return ANNCHOR_OK;
}

// action 1: park the refinement launch -- the next annchor_sampler_stats queues it behind its download; action 2: launch it now
// if it is still parked (the statistics took another route)
extern "C" int annchor_park_refine(annchor_ctx *c, int32_t action)
{
    if (!c || (action != 1 && action != 2)) return ANNCHOR_EINVAL;
    if (action == 1) { c->park_refine = true; return ANNCHOR_OK; }
    if (!c->park_refine) return ANNCHOR_OK;
    c->park_refine = false;
    return annchor_refine_candidates(c);
}

extern "C" int annchor_refine_candidates(annchor_ctx *c)
{
    if (!c) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->have_RA, ANNCHOR_EINVAL, "RefineApprox not initialised");
    ANN_REQUIRE(c, c->metric != ANNCHOR_METRIC_NONE, ANNCHOR_EINVAL, "no device metric bound to this context");
    if (c->ncand == 0) return ANNCHOR_OK;
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    PairSource src;
    src.ij = c->ij.as<int2>();
    src.idx = c->cand.as<int32_t>();
    src.n = c->ncand;
    ANN_CHECK_HIP(c, hipEventRecord(c->call_a, c->stream));
    ANN_TRY(ann_metric_launch(c, src, nullptr, c->RA.as<double>(), c->ncm.as<uint8_t>()));
    ANN_CHECK_HIP(c, hipEventRecord(c->call_b, c->stream));
    c->call_timed = true;
    ANN_CHECK_HIP(c, hipGetLastError());
    if (c->n_unc >= 0 && !c->cand_marked) c->n_unc -= c->ncand;
    c->cand_marked = true; c->sel_prepared = false;
    return ANNCHOR_OK;
}

__global__ void k_write_refined(const int32_t *__restrict__ pos, const double *__restrict__ v, int64_t m,
                                double *__restrict__ RA, uint8_t *__restrict__ ncm)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < m) { RA[pos[t]] = v[t]; ncm[pos[t]] = 0; }
}

extern "C" int annchor_set_refined(annchor_ctx *c, const double *exact, int64_t n_cand)
{
    if (!c || (n_cand > 0 && !exact)) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->have_RA, ANNCHOR_EINVAL, "RefineApprox not initialised");
    ANN_REQUIRE(c, n_cand == c->ncand, ANNCHOR_EINVAL, "expected %lld refined values, got %lld", (long long)c->ncand,
                (long long)n_cand);
    if (n_cand == 0) return ANNCHOR_OK;
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    ANN_TRY(ann_reserve(c, c->stage_in, sizeof(double) * (size_t)n_cand));
    ANN_TRY(ann_h2d(c, c->stage_in.p, exact, sizeof(double) * (size_t)n_cand));
    k_write_refined<<<ann_blocks(n_cand, 256), 256, 0, c->stream>>>(c->cand.as<int32_t>(), c->stage_in.as<double>(), n_cand,
                                                                   c->RA.as<double>(), c->ncm.as<uint8_t>());
    ANN_CHECK_HIP(c, hipGetLastError());
    if (c->n_unc >= 0 && !c->cand_marked) c->n_unc -= n_cand;
    c->cand_marked = true; c->sel_prepared = false;
    return ANNCHOR_OK;
}
