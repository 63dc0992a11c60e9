// emd.hip -- exact optimal transport (Wasserstein / Kantorovich) between histograms.
//
// Replaces, for f = wasserstein (reference annchor/utils.py:75-86 ->
// pynndescent.distances.kantorovich(x, y, cost=M): restrict x, y to their supports,
// normalise each to unit mass, return the optimum of the transportation LP), the
// evaluator get_exact(f, X, IJ) of annchor/utils.py:110-177.
//
// Sinkhorn is NOT used for the returned value: SURVEY.md section 7 (hard part 2)
// measured entropic OT 30-100x outside the tolerance that parity needs (and inexact
// anchor distances void the triangle bounds).  The kernel is an exact primal-dual
// solver shaped for a wavefront:
//   * one wavefront per pair; lane j owns sink j (its demand, potential v_j, tentative
//     distance, predecessor) and lane i owns source i (supply, u_i);
//   * Dijkstra on reduced costs over the dense bipartite graph: "pop the nearest
//     unscanned sink" is a DPP min-reduction, "relax a source row" is one LDS row read
//     of the cost matrix executed by all lanes;
//   * the flow matrix F (n x m float64, <= 33 KB) lives in LDS, one slab per wave; the
//     ground-cost matrix (<= 32 KB) is staged in LDS once per workgroup;
//   * control flow is wave-uniform (scalar branches); no atomics, no global scratch.
// Latency/branch bound by nature: reported as pairs/s and
// microseconds per pair, not as an HBM or MFMA fraction.
#include "common.h"

#define EMD_MAXB 64

struct EmdArgs {
    const double *hist;
    const double *cost;
    int nb;
    int S;          // padded row stride of the flow slab (odd); 0: per solve, (number of sinks) | 1
    int waves;      // waves per block
    int slab_bytes; // LDS bytes per wave (flow slab + support index lists)
    const int2 *ij;
    const int32_t *idx;
    const int32_t *anchor;
    int64_t n;
    double *out;
    double *RA;
    uint8_t *ncm;
    int32_t *fail;  // set if the iteration guard trips (sticky)
    int32_t *work;  // the launch's work counter (next unclaimed solve); *work_next: the next launch's, zeroed by this one
    int32_t *work_next;
    int reduce;     // metric ground cost: solve on the differences of the two (scaled) histograms
    long long *dbg; // -DEMD_PROFILE
    int dantzig_cap; // k_emd_ns: pivots under Dantzig's rule before Bland's takes over (-1: 16 (n + m) + 64; tests force 0)
    // max-min picking fused into a one-to-all launch (PairSource::pick_*): the launch derives its own anchor from the previous round's row
    const double *pick_row;
    double *pick_runmin;
    int32_t *pick_out;
    int pick_reset, pick_nx;
    const int32_t *hs_bin;   // k_emd_ns<.., true>: [nx][32] bins of the non-zero entries (ascending), their masses, their number
    const double *hs_val;
    const int32_t *hs_cnt;
    double eps;     // k_emd_ns: an arc enters the basis when its reduced cost is below -eps (2^-43 x the largest ground cost)
};

// ---- wave-level min over lanes (double), result broadcast; DPP, no LDS traffic
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_f64(double v, double identity)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    int ilo = __double2loint(identity), ihi = __double2hiint(identity);
    lo = __builtin_amdgcn_update_dpp(ilo, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(ihi, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_min_f64(double v)
{
    const double inf = INFINITY;
    v = fmin(v, dpp_f64<0xb1, 0xf>(v, inf));   // quad_perm [1,0,3,2]
    v = fmin(v, dpp_f64<0x4e, 0xf>(v, inf));   // quad_perm [2,3,0,1]
    v = fmin(v, dpp_f64<0x141, 0xf>(v, inf));  // row_half_mirror
    v = fmin(v, dpp_f64<0x140, 0xf>(v, inf));  // row_mirror
    v = fmin(v, dpp_f64<0x142, 0xa>(v, inf));  // row_bcast:15 -> rows 1,3
    v = fmin(v, dpp_f64<0x143, 0xc>(v, inf));  // row_bcast:31 -> rows 2,3
    // lane 63 now holds the minimum of all 64 lanes
    int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// Minimum of NON-NEGATIVE doubles (+inf allowed): they order like their bit patterns, so the high words are
// reduced first and the low words among the lanes that hold the minimal high word -- twelve 32-bit DPP minimum
// steps instead of six 64-bit ones made of two moves and a v_min_f64 each (the kernel is issue bound and this
// reduction runs once per Dijkstra pop).
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0xb1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x4e, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x141, 0xf, 0xf, false));  // row_half_mirror
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x140, 0xf, 0xf, false));  // row_mirror
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x142, 0xa, 0xf, false));  // row_bcast:15 -> rows 1,3
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x143, 0xc, 0xf, false));  // row_bcast:31 -> rows 2,3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ double wave_min_nonneg_f64(double v)
{
    const uint32_t hi = (uint32_t)__double2hiint(v), lo = (uint32_t)__double2loint(v);
    const uint32_t mh = wave_min_u32(hi);
    const uint32_t ml = wave_min_u32(hi == mh ? lo : 0xffffffffu);
    return __hiloint2double((int)mh, (int)ml);
}

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// Flow / mass type: double in general; int32 when both histograms are integer valued (then
// supplies x_i * sum(y) and demands y_j * sum(x) are exact integers and the flow slab is half
// the size, which doubles the waves a CU can hold).
__device__ __forceinline__ double rl(double v, int lane) { return readlane_f64(v, lane); }
__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ double tmin(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ int tmin(int a, int b) { return a < b ? a : b; }

// -DEMD_PROFILE: per-solve cycle counts by phase (small launches only), printed for the slowest solve of a launch
#ifdef EMD_PROFILE
#define EP_DECL long long ep_t = (long long)__builtin_readcyclecounter(), ep_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int ep_cnt[4] = {0, 0, 0, 0}
#define EP(i) do { const long long ep_n = (long long)__builtin_readcyclecounter(); ep_acc[i] += ep_n - ep_t; ep_t = ep_n; } while (0)
#define EP_CNT(i) (++ep_cnt[i])
#else
#define EP_DECL do { } while (0)
#define EP(i) do { } while (0)
#define EP_CNT(i) do { } while (0)
#endif

// T: masses and flows in registers; FT: the flow slab's storage type (int16 when every flow fits: half the LDS per wave again)
template <typename T, typename FT> __global__ __launch_bounds__(1024) void k_emd(EmdArgs a)
{
    constexpr bool INTEGRAL = sizeof(T) == 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nb = a.nb;
    double *costL = reinterpret_cast<double *>(smem);                       // [nb][nb]
    unsigned char *slab = reinterpret_cast<unsigned char *>(costL + nb * nb) + (size_t)wave * a.slab_bytes;
    FT *F = reinterpret_cast<FT *>(slab);                                   // [<=64][S] flow slab
    int *rowsL = reinterpret_cast<int *>(slab + a.slab_bytes - 2 * EMD_MAXB * sizeof(int));  // [64] support of x
    int *colsL = rowsL + EMD_MAXB;                                          // [64] support of y
    for (int t = threadIdx.x; t < nb * nb; t += blockDim.x) costL[t] = a.cost[t];
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.work_next = 0;   // (two counters used in turn: no memset between launches)
    __syncthreads();

    // Solves are claimed one at a time from a counter: their durations differ by an order of magnitude (near pairs need a few
    // searches, far pairs scan the whole graph every time), and a fixed share per wave left most of the chip waiting for the
    // unluckiest wave of a launch.
    const int64_t wave_global = (int64_t)blockIdx.x * a.waves + wave;
    const int64_t wave_total = (int64_t)gridDim.x * a.waves;
    for (int64_t t = wave_global;;) {
        if (t >= a.n) break;
        EP_DECL;
        int pi, pj;
        int64_t opos = t;
        if (a.anchor) { pi = *a.anchor; pj = (int)t; }
        else {
            int64_t q = a.idx ? a.idx[t] : t;
            int2 p = a.ij[q];
            pi = p.x; pj = p.y;
            if (a.idx) opos = q;
        }
        pi = __builtin_amdgcn_readfirstlane(pi);
        pj = __builtin_amdgcn_readfirstlane(pj);
        const double *hx = a.hist + (size_t)pi * nb, *hy = a.hist + (size_t)pj * nb;
        // masses and their sums in index order (sequential index order: exact parity
        // of the normalisation for non-integer inputs)
        const double xk = lane < nb ? hx[lane] : 0.0, yk = lane < nb ? hy[lane] : 0.0;
        double sa = 0, sb = 0;
        for (int k = 0; k < nb; ++k) { sa += readlane_f64(xk, k); sb += readlane_f64(yk, k); }
        // masses of bin `lane` in the solver's units: 1 / (sa * sb) (exact integers) or unit total mass
        T xm, ym;
        if (INTEGRAL) { xm = (T)(xk * sb); ym = (T)(yk * sa); }
        else { xm = (T)(xk / sa); ym = (T)(yk / sb); }
        if (a.reduce) {
            // metric ground cost (c_kk = 0, triangle inequality): an optimal plan leaves min(x_k, y_k) on bin k, so only the
            // differences travel -- sources and sinks become disjoint and fewer (two digits share most of their pixels), and
            // the saturated (k, k) arcs whose backward edges the shortest-path searches would otherwise walk are gone
            const T d = xm - ym;
            xm = d > (T)0 ? d : (T)0;
            ym = d < (T)0 ? -d : (T)0;
        }
        const unsigned long long mx = __ballot(xm != (T)0), my = __ballot(ym != (T)0);
        const int n = __popcll(mx), m = __popcll(my);
        const int S = a.S ? a.S : (m | 1);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (xm != (T)0) rowsL[__popcll(mx & below)] = lane;
        if (ym != (T)0) colsL[__popcll(my & below)] = lane;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int myrow = lane < n ? rowsL[lane] : 0;   // source `lane` is histogram bin myrow
        const int mycol = lane < m ? colsL[lane] : 0;   // sink `lane` is histogram bin mycol
        // supply of source `lane`, demand of sink `lane`: the masses of their bins
        T a_rem = __shfl(xm, myrow), b_rem = __shfl(ym, mycol);
        if (lane >= n) a_rem = (T)0;
        if (lane >= m) b_rem = (T)0;
        double u = 0.0;                                  // potential of source `lane`
        // v_j = min_i C[i][j] (first minimal source on ties): dual feasible with u = 0, and
        // arc (amin_j, j) is tight for every sink
        double v = INFINITY;
        int amin = 0;
        for (int i = 0; i < n; ++i) {
            const int r = __builtin_amdgcn_readlane(myrow, i);
            if (lane < m) {
                const double cc = costL[r * nb + mycol];
                if (cc < v) { v = cc; amin = i; }
            }
        }
        // zero the flow slab
        for (int i = 0; i < n; ++i)
            if (lane < m) F[i * S + lane] = (FT)0;
        __builtin_amdgcn_wave_barrier();
        // greedy start on the tight arcs (complementary slackness holds: flow only where the
        // reduced cost is zero).  For near-by histograms most mass sits on identical bins
        // (cost 0) and is routed here, before any shortest-path search.
        for (int j = 0; j < m; ++j) {
            const int i = __builtin_amdgcn_readlane(amin, j);
            const T f = tmin(rl(a_rem, i), rl(b_rem, j));
            if (f > (T)0) {
                if (lane == 0) F[i * S + j] = (FT)f;
                if (lane == i) a_rem -= f;
                if (lane == j) b_rem -= f;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

        EP(0);
        int guard = 64 * (n + m) + 1024;
        bool failed = false, dust = false;
        for (int s = 0; s < n && !dust && !failed; ++s) {
            const int rs = __builtin_amdgcn_readlane(myrow, s);
            for (;;) {
                const T as = rl(a_rem, s);
                if (!(as > (T)0)) break;
                if (--guard < 0) { failed = true; break; }
                // ---- Dijkstra from source s on reduced costs
                const double us = readlane_f64(u, s);
                // (reduced costs are >= 0 up to rounding: clamped, so that the tentative distances are non-negative
                // doubles -- the invariant the wave minimum below relies on)
                double dist = lane < m ? fmax(costL[rs * nb + mycol] - us - v, 0.0) : INFINITY;
                int pred = s;
                unsigned long long sinkdone = 0, srcdone = 1ull << s;
                double srcdist = 0.0;   // valid on lanes whose srcdone bit is set
                int srcfrom = -1;
                double mu = 0.0;
                // The search does not stop at the first sink with demand left: it goes on until the demand it has met covers
                // what source s still has to place (or every sink is scanned).  One shortest-path tree then carries several
                // augmentations -- a source whose mass splits over two or three sinks used to start its search over for each of
                // them and re-scan the same near sinks.
                int prank = -1;         // order in which THIS lane's sink was reached, among the sinks with demand left
                int nhit = 0;
                T need = as;
                EP(1); EP_CNT(0);
                for (;;) {
                    const bool open = lane < m && !((sinkdone >> lane) & 1ull);
                    // nearest open sink: non-negative doubles order like their bit patterns -- minimum of the high words first;
                    // only when several open sinks share it (equal distances: tight arcs at 0, symmetric costs) do the low words
                    // need their own reduction
                    const uint32_t dhi = open ? (uint32_t)__double2hiint(dist) : 0x7ff00000u, dlo = (uint32_t)__double2loint(dist);
                    const uint32_t mh = wave_min_u32(dhi);
                    unsigned long long hit = __ballot(open && dhi == mh);
                    double best;
                    if (hit & (hit - 1)) {
                        const uint32_t ml = wave_min_u32(open && dhi == mh ? dlo : 0xffffffffu);
                        hit = __ballot(open && dhi == mh && dlo == ml);
                        best = __hiloint2double((int)mh, (int)ml);
                    } else {
                        best = hit ? readlane_f64(dist, __ffsll((unsigned long long)hit) - 1) : INFINITY;
                    }
                    EP(2); EP_CNT(1);
                    if (!hit) break;                       // every sink scanned
                    const int js = __ffsll((unsigned long long)hit) - 1;  // first index on ties
                    sinkdone |= 1ull << js;
                    mu = best;
                    const T bj = rl(b_rem, js);
                    if (bj > (T)0) {
                        if (lane == js) prank = nhit;
                        ++nhit;
                        need -= tmin(need, bj);
                        if (!(need > (T)0)) break;
                    }
                    // sources with flow into js that are not scanned yet
                    const T fcol = lane < n ? (T)F[lane * S + js] : (T)0;
                    unsigned long long todo = __ballot(lane < n && fcol > (T)0 && !((srcdone >> lane) & 1ull));
                    EP(3);
                    while (todo) {
                        EP_CNT(2);
                        const int i = __ffsll((unsigned long long)todo) - 1;
                        todo &= todo - 1;
                        srcdone |= 1ull << i;
                        if (lane == i) { srcdist = mu; srcfrom = js; }
                        const int ri = __builtin_amdgcn_readlane(myrow, i);
                        const double ui = readlane_f64(u, i);
                        if (lane < m && !((sinkdone >> lane) & 1ull)) {
                            const double nd = fmax(mu + (costL[ri * nb + mycol] - ui - v), 0.0);
                            if (nd < dist) { dist = nd; pred = i; }
                        }
                    }
                    EP(4);
                }
                if (nhit == 0) { dust = true; break; }     // only rounding dust left
                // ---- potentials: every scanned node against the last distance popped -- the tree's arcs are tight afterwards
                if ((srcdone >> lane) & 1ull) u += mu - srcdist;
                if ((sinkdone >> lane) & 1ull) v -= mu - dist;
                EP(5);
                // ---- augment along the tree to every sink reached with demand left, nearest first; a path's bottleneck is taken
                // from the flows as the earlier augmentations of this search left them
                for (int k = 0; k < nhit; ++k) {
                    const int jend = __ffsll((unsigned long long)__ballot(prank == k)) - 1;
                    T delta = tmin(rl(a_rem, s), rl(b_rem, jend));
                    for (int j = jend; delta > (T)0;) {
                        const int i = __builtin_amdgcn_readlane(pred, j);
                        if (i == s) break;
                        const int jj = __builtin_amdgcn_readlane(srcfrom, i);
                        delta = tmin(delta, (T)F[i * S + jj]);   // uniform address: LDS broadcast
                        j = jj;
                    }
                    if (!(delta > (T)0)) continue;
                    for (int j = jend;;) {
                        const int i = __builtin_amdgcn_readlane(pred, j);
                        if (lane == 0) F[i * S + j] = (FT)((T)F[i * S + j] + delta);
                        if (i == s) break;
                        const int jj = __builtin_amdgcn_readlane(srcfrom, i);
                        if (lane == 0) F[i * S + jj] = (FT)((T)F[i * S + jj] - delta);
                        j = jj;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    if (lane == s) a_rem -= delta;
                    if (lane == jend) b_rem -= delta;
                    EP_CNT(3);
                }
                EP(6);
            }
        }
        // ---- objective
        double tot = 0.0;
        for (int i = 0; i < n; ++i) {
            const int r = __builtin_amdgcn_readlane(myrow, i);
            if (lane < m) tot += (double)F[i * S + lane] * costL[r * nb + mycol];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
        if (INTEGRAL) tot /= sa * sb;
        if (failed) { tot = NAN; if (lane == 0) *a.fail = 1; }
        if (lane == 0) {
            if (a.out) a.out[t] = tot;
            if (a.RA) { a.RA[opos] = tot; a.ncm[opos] = 0; }
        }
        __builtin_amdgcn_wave_barrier();
#ifdef EMD_PROFILE
        EP(7);
        if (a.dbg && t < 4096 && lane == 0) {
            for (int q = 0; q < 8; ++q) a.dbg[t * 16 + q] = ep_acc[q];
            for (int q = 0; q < 4; ++q) a.dbg[t * 16 + 8 + q] = ep_cnt[q];
            a.dbg[t * 16 + 12] = n; a.dbg[t * 16 + 13] = m;
        }
#endif
        {   // the next solve: the first wave_total ones were handed out by position
            int nxt = 0;
            if (lane == 0) nxt = atomicAdd(a.work, 1);
            t = wave_total + (int64_t)__builtin_amdgcn_readfirstlane(nxt);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_emd_ns: the same LP by the transportation simplex on a spanning-tree basis, for metric ground costs (the common mass of
// the two histograms cancels: sources and sinks are disjoint bins, n + m <= 64 -- ONE LANE PER NODE).
//
// Why: the shortest-path solver above does ~200-350 Dijkstra pops per solve on the digits (up to 1700 on far pairs: ~100
// searches of ~17 pops), each a dependent chain of a DPP minimum, an LDS column read and LDS row reads; an anchor round lasts as
// long as its slowest solve.  The simplex needs 20-30 pivots (at most ~75; tools/sim/emd_simplex_sim.cpp counts them on the same
// pairs) and keeps the whole basis in registers:
//   * lane k < n is source k, lane n + j sink j; per lane: histogram bin, parent in the basis tree, flow on the arc to the
//     parent (always source -> sink), potential, and the SET OF NODES BELOW IT as a 64-bit mask (`sub`);
//   * pricing (Dantzig): every sink lane scans the sources (one LDS cost read each, independent), one wave minimum;
//   * the cycle of the entering arc (x, y) needs no tree walk: the ancestors of x are the lanes whose `sub` holds bit x (one
//     ballot), likewise y; the cycle's arcs are the lanes in exactly one of the two sets; which of them lose flow follows from
//     the node type and the side; theta and the leaving arc are one wave minimum, the flow update one masked add;
//   * the part of the tree cut off by the leaving arc is `sub` of its lower node (one readlane): potentials shift by the
//     entering arc's reduced cost there, the `sub` masks of the old / new ancestors lose / gain it, and the parent pointers
//     along the short path from the entering arc's end to the leaving arc are reversed;
//   * start: row-minimum rule (each step closes exactly one line: a spanning tree with n + m - 1 arcs, degenerate ones
//     included), potentials in reverse closing order.
// No flow slab: LDS holds the ground costs and 512 B per wave.  Degenerate pivots are rare (< 1 %); should Dantzig's rule not
// finish within its cap the loop continues under Bland's rule (lowest index in, lowest index out), which cannot cycle.
#define EMD_NS_WAVE_BYTES (2 * EMD_MAXB * (int)sizeof(int) + 64 * (int)sizeof(double))

template <typename T> __device__ __forceinline__ T wave_min_nonneg(T v);
template <> __device__ __forceinline__ int wave_min_nonneg<int>(int v) { return (int)wave_min_u32((uint32_t)v); }
template <> __device__ __forceinline__ double wave_min_nonneg<double>(double v) { return wave_min_nonneg_f64(v); }
__device__ __forceinline__ unsigned long long rl64(unsigned long long v, int lane)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return ((unsigned long long)hi << 32) | lo;
}

// SPARSE: histograms of more than 64 bins with at most 32 non-zero entries each (annchor_set_histograms keeps them as (bin, mass)
// lists): lanes 0..31 take x's entries, lanes 32..63 y's; the ground costs stay in global memory (read at the start of a solve, in
// the start rule and for the objective -- never per pivot)
template <typename T, bool SPARSE> __global__ __launch_bounds__(1024) void k_emd_ns(EmdArgs a)
{
    constexpr bool INTEGRAL = sizeof(T) == 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nb = a.nb;
    double *costL = reinterpret_cast<double *>(smem);                       // [nb][nb] (not SPARSE)
    // per wave: [64] support of x, [64] support of y (ints), [64] potentials by node (doubles)
    unsigned char *wv = reinterpret_cast<unsigned char *>(costL + (SPARSE ? 0 : nb * nb)) + (size_t)wave * EMD_NS_WAVE_BYTES;
    auto COST = [&](int ra, int cb) -> double { return SPARSE ? a.cost[(size_t)ra * nb + cb] : costL[ra * nb + cb]; };
    int *rowsL = reinterpret_cast<int *>(wv);
    int *colsL = rowsL + EMD_MAXB;
    double *potL = reinterpret_cast<double *>(wv + 2 * EMD_MAXB * sizeof(int));
    if (!SPARSE)
        for (int t = threadIdx.x; t < nb * nb; t += blockDim.x) costL[t] = a.cost[t];
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.work_next = 0;   // (two counters used in turn: no memset between launches)
    __syncthreads();
    const double eps = a.eps;
    const unsigned long long lanebit = 1ull << lane;
    // ---- the anchor of a max-min round (pickers.py:47-50): every wave derives it for itself from the previous round's row --
    // running minimum (idempotent: workgroup 0 also stores it), first arg-max -- so the picker needs no launch of its own
    int picked = -1;
    if (a.pick_row) {
        double bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int j = lane; j < a.pick_nx; j += 64) {
            const double d = a.pick_row[j];
            const double v = a.pick_reset ? d : fmin(a.pick_runmin[j], d);
            if (blockIdx.x == 0 && wave == 0) a.pick_runmin[j] = v;
            argmax_combine(bv, bi, v, j);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(bv, off);
            const int oi = __shfl_xor(bi, off);
            argmax_combine(bv, bi, ov, oi);
        }
        picked = bi;
        if (blockIdx.x == 0 && threadIdx.x == 0) *a.pick_out = bi;
    }

    const int64_t wave_global = (int64_t)blockIdx.x * a.waves + wave;
    const int64_t wave_total = (int64_t)gridDim.x * a.waves;
    for (int64_t t = wave_global;;) {
        if (t >= a.n) break;
        EP_DECL;
        int pi, pj;
        int64_t opos = t;
        if (a.anchor) { pi = picked >= 0 ? picked : *a.anchor; pj = (int)t; }
        else {
            int64_t q = a.idx ? a.idx[t] : t;
            int2 p = a.ij[q];
            pi = p.x; pj = p.y;
            if (a.idx) opos = q;
        }
        pi = __builtin_amdgcn_readfirstlane(pi);
        pj = __builtin_amdgcn_readfirstlane(pj);
        double sa = 0, sb = 0;
        T xm, ym;
        int ebin = lane;   // histogram bin of the entry this lane holds (dense: bin = lane)
        if constexpr (SPARSE) {
            const int side = lane >> 5, e = lane & 31;
            const int cx = a.hs_cnt[pi], cy = a.hs_cnt[pj];
            const int pidx = side ? pj : pi, cnt = side ? cy : cx;
            ebin = e < cnt ? a.hs_bin[(size_t)pidx * 32 + e] : -1 - lane;   // (empty slots: distinct negative numbers, matching nothing)
            const double ev = e < cnt ? a.hs_val[(size_t)pidx * 32 + e] : 0.0;
            for (int k = 0; k < cx; ++k) sa += readlane_f64(ev, k);          // entries are in bin order: the dense loop's sums
            for (int k = 0; k < cy; ++k) sb += readlane_f64(ev, 32 + k);
            T mine;
            if (INTEGRAL) mine = (T)(ev * (side ? sa : sb));
            else mine = (T)(ev / (side ? sb : sa));
            // the other histogram's mass on the same bin
            T other = (T)0;
            for (int k = 0; k < cy; ++k) {
                const int b = __builtin_amdgcn_readlane(ebin, 32 + k);
                const T v = rl(mine, 32 + k);
                if (side == 0 && ebin == b) other = v;
            }
            for (int k = 0; k < cx; ++k) {
                const int b = __builtin_amdgcn_readlane(ebin, k);
                const T v = rl(mine, k);
                if (side == 1 && ebin == b) other = v;
            }
            const T d = mine - other;     // the common mass stays where it is (metric cost)
            const T pos = d > (T)0 ? d : (T)0;
            xm = side == 0 ? pos : (T)0;
            ym = side == 1 ? pos : (T)0;
        } else {
            const double *hx = a.hist + (size_t)pi * nb, *hy = a.hist + (size_t)pj * nb;
            const double xk = lane < nb ? hx[lane] : 0.0, yk = lane < nb ? hy[lane] : 0.0;
            for (int k = 0; k < nb; ++k) { sa += readlane_f64(xk, k); sb += readlane_f64(yk, k); }
            if (INTEGRAL) { xm = (T)(xk * sb); ym = (T)(yk * sa); }
            else { xm = (T)(xk / sa); ym = (T)(yk / sb); }
            // the common mass stays where it is (metric cost): only the differences travel
            const T d = xm - ym;
            xm = d > (T)0 ? d : (T)0;
            ym = d < (T)0 ? -d : (T)0;
        }
        const unsigned long long mx = __ballot(xm != (T)0), my = __ballot(ym != (T)0);
        const int n = __popcll(mx), m = __popcll(my), N = n + m;
        double tot = 0.0;
        bool failed = false;
        if (n > 0 && m > 0) {
            const unsigned long long below = lanebit - 1ull;
            if (xm != (T)0) rowsL[__popcll(mx & below)] = lane;
            if (ym != (T)0) colsL[__popcll(my & below)] = lane;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            // ---- nodes: lane k < n = source k, lane n + j = sink j
            const bool is_src = lane < n, is_snk = lane >= n && lane < N;
            const int holder = is_src ? rowsL[lane] : (is_snk ? colsL[lane - n] : 0);   // the lane that holds this node's entry
            const int ebin_of = __shfl(ebin, holder);   // (every lane takes part: a holder lane may lie beyond the N node lanes)
            const int bin = lane < N ? ebin_of : 0;
            if constexpr (SPARSE) {   // the lists now carry the nodes' bins (dense: they already do)
                __builtin_amdgcn_wave_barrier();
                if (is_src) rowsL[lane] = bin;
                if (is_snk) colsL[lane - n] = bin;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            // ---- the arcs, dealt over the lanes: arc a = slot * 64 + lane is (source a / m, sink a % m); its ground cost and the
            // LDS offsets of its ends' potentials stay in registers for the whole solve (<= 16 slots: n m <= 1024)
            const int nm = n * m;
            double ac[16];
            int aij[16];
            {
                const float inv_m = 1.0f / (float)m;
#pragma unroll
                for (int sl = 0; sl < 16; ++sl) {
                    ac[sl] = 0.0; aij[sl] = -1;
                    if (sl * 64 < nm) {
                        const int aidx = sl * 64 + lane;
                        const int ac_ = min(aidx, nm - 1);
                        const int ai = (int)(((float)ac_ + 0.5f) * inv_m);
                        const int aj = ac_ - ai * m;
                        ac[sl] = COST(rowsL[ai], colsL[aj]);
                        if (aidx < nm) aij[sl] = (ai << 16) | (n + aj);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            const T mass_x = __shfl(xm, holder), mass_y = __shfl(ym, holder);
            T rem = is_src ? mass_x : (is_snk ? mass_y : (T)0);       // supply / demand still to place (start rule)
            const unsigned long long allmask = N == 64 ? ~0ull : ((1ull << N) - 1ull);
            const unsigned long long srcmask = (1ull << n) - 1ull, snkmask = allmask & ~srcmask;
            int parent = -1;
            T pflow = (T)0;
            double pot = 0.0, pcost = 0.0;
            unsigned long long sub = lane < N ? lanebit : 0ull;
            int ord = -1;
            // ---- start: row-minimum rule.  The first open source takes its cheapest open sink; the assignment exhausts one of
            // the two, which closes and hangs below the other (never the last source while several sinks are open: with the totals
            // balanced their demands are zero then, and they close one by one on degenerate arcs).
            unsigned long long open = allmask;
            int step = 0;
            EP(0);
            while (open & (open - 1)) {
                const unsigned long long osrc = open & srcmask;
                const int i = __ffsll(osrc) - 1;                       // (an open source always exists while two lines are open)
                const int bi = __builtin_amdgcn_readlane(bin, i);
                const bool cand = is_snk && ((open >> lane) & 1ull);
                const double c = cand ? COST(bi, bin) : INFINITY;
                // (minimum of the high words first; the low words only when several candidates share it)
                const uint32_t chi = (uint32_t)__double2hiint(c);
                const uint32_t mh = wave_min_u32(chi);
                unsigned long long hitc = __ballot(cand && chi == mh);
                if (hitc & (hitc - 1)) {
                    const uint32_t ml = wave_min_u32(cand && chi == mh ? (uint32_t)__double2loint(c) : 0xffffffffu);
                    hitc = __ballot(cand && chi == mh && (uint32_t)__double2loint(c) == ml);
                }
                const int j = __ffsll(hitc) - 1;
                const double cmin = readlane_f64(c, j);
                const T ai = rl(rem, i), bj = rl(rem, j);
                const T f = tmin(ai, bj);
                const unsigned long long osnk = open & snkmask;
                const bool last_src = (osrc & (osrc - 1)) == 0, last_snk = (osnk & (osnk - 1)) == 0;
                // (the exhausted line closes; never the last line of its kind while several of the other kind are open -- with
                // balanced totals what those still hold is zero, or rounding dust for non-integer masses)
                const bool close_src = (ai - f > (T)0) ? (last_snk && !last_src) : !(last_src && !last_snk);
                const int cn = close_src ? i : j, on = close_src ? j : i;
                if (lane == i || lane == j) rem -= f;
                const unsigned long long csub = rl64(sub, cn);
                if (lane == cn) { parent = on; pflow = f; pcost = cmin; ord = step; }
                if (lane == on) sub |= csub;
                open &= ~(1ull << cn);
                ++step;
            }
            EP(1);
            // potentials: the root (the line left open) at 0, the others in reverse closing order -- u_i + v_j = c_ij on tree arcs
            for (int s = step - 1; s >= 0; --s) {
                const int cn = __ffsll((unsigned long long)__ballot(ord == s)) - 1;
                const int pn = __builtin_amdgcn_readlane(parent, cn);
                const double pp = readlane_f64(pot, pn);
                if (lane == cn) pot = pcost - pp;
            }
            EP(2);
            // ---- pivots
            const int dantzig_cap = a.dantzig_cap >= 0 ? a.dantzig_cap : 16 * N + 64, total_cap = dantzig_cap + 4096;
            int piv = 0;
            for (;; ++piv) {
                if (piv >= total_cap) { failed = true; break; }
                const bool bland = piv >= dantzig_cap;
                // pricing: every lane its <= 16 arcs; the ends' potentials through LDS (independent reads, one wait)
                potL[lane] = pot;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                double best = 0.0;
                int bestij = -1;
                // (four slots per uniform branch: their eight LDS reads are in flight together)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (g * 256 < nm) {
                        double pu[4], pv[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int e = aij[4 * g + q] < 0 ? 0 : aij[4 * g + q];
                            pu[q] = potL[e >> 16];
                            pv[q] = potL[e & 0xffff];
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const double rc = (ac[4 * g + q] - pu[q]) - pv[q];
                            const bool take = aij[4 * g + q] >= 0 && (bland ? (bestij < 0 && rc < -eps) : (rc < best));
                            if (take) { best = rc; bestij = aij[4 * g + q]; }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                EP(3); EP_CNT(0);
                int x, y;
                double rcin;
                {
                    int from;
                    if (!bland) {
                        rcin = wave_min_f64(best);
                        if (!(rcin < -eps)) break;                     // optimal
                        from = __ffsll((unsigned long long)__ballot(bestij >= 0 && best == rcin)) - 1;
                    } else {
                        // lowest arc index: a lane's first hit is its lowest slot; arcs are numbered slot * 64 + lane
                        const unsigned long long cands = __ballot(bestij >= 0);
                        if (!cands) break;
                        const uint32_t key = bestij >= 0 ? (uint32_t)((bestij >> 16) * 64 + (bestij & 0xffff)) : 0xffffffffu;
                        const uint32_t kmin = wave_min_u32(key);
                        from = __ffsll((unsigned long long)__ballot(key == kmin)) - 1;
                        rcin = readlane_f64(best, from);
                    }
                    const int e = __builtin_amdgcn_readlane(bestij, from);
                    x = e >> 16; y = e & 0xffff;
                }
                // ---- the cycle: tree path x ~> y plus the entering arc.  Ancestors (incl. the node itself) by `sub`.
                const unsigned long long AX = __ballot((sub >> x) & 1ull), AY = __ballot((sub >> y) & 1ull);
                const unsigned long long xs = AX & ~AY, ys = AY & ~AX;
                // theta enters on x -> y and travels y ~> top ~> x: upwards on y's side (a sink below a source loses), downwards on
                // x's side (a source below a sink loses)
                const unsigned long long dec = (xs & srcmask) | (ys & snkmask);
                const bool in_dec = (dec >> lane) & 1ull;
                T theta;
                int leave;
                if (INTEGRAL) {
                    theta = wave_min_nonneg<T>(in_dec ? pflow : (T)0x7fffffff);
                } else {
                    theta = wave_min_nonneg<T>(in_dec ? pflow : (T)INFINITY);
                }
                {
                    const unsigned long long ties = __ballot(in_dec && pflow == theta);
                    leave = __ffsll(ties) - 1;
                    if (bland && (ties & (ties - 1))) {
                        // lowest arc index (sink-major) among the ties
                        const int asrc = is_src ? lane : parent, asnk = is_src ? parent : lane;
                        const uint32_t aidx = ((ties >> lane) & 1ull) ? (uint32_t)(asnk * 64 + asrc) : 0xffffffffu;
                        const uint32_t amin = wave_min_u32(aidx);
                        leave = __ffsll((unsigned long long)__ballot(aidx == amin)) - 1;
                    }
                }
                if (((xs | ys) >> lane) & 1ull) pflow += in_dec ? -theta : theta;
                EP(4);
                // ---- the tree: T2 = what hangs below the leaving arc; it re-attaches through the entering arc
                const unsigned long long T2 = rl64(sub, leave);
                const bool on_x = (xs >> leave) & 1ull;
                const int q = on_x ? x : y, p = on_x ? y : x;
                if ((T2 >> lane) & 1ull) pot += (is_src == (q < n)) ? rcin : -rcin;
                else if ((sub >> leave) & 1ull) sub &= ~T2;              // old ancestors of the cut part
                if (!((T2 >> lane) & 1ull) && ((sub >> p) & 1ull)) sub |= T2;   // its new ancestors: p and everything above p
                int w = q, cpar = p;
                T cflow = theta;
                unsigned long long prevsub = 0ull;
                for (;;) {
                    const int opar = __builtin_amdgcn_readlane(parent, w);
                    const T oflow = rl(pflow, w);
                    const unsigned long long osub = rl64(sub, w);
                    if (lane == w) { parent = cpar; pflow = cflow; sub = T2 & ~prevsub; }
                    if (w == leave) break;
                    cpar = w; cflow = oflow; prevsub = osub; w = opar;
                    EP_CNT(1);
                }
                EP(5);
            }
            // ---- objective: flow x cost over the tree arcs
            {
                const int pb = __shfl(bin, parent < 0 ? lane : parent);
                const double cst = (lane < N && parent >= 0) ? COST(is_src ? bin : pb, is_src ? pb : bin) : 0.0;
                tot = (lane < N && parent >= 0) ? (double)pflow * cst : 0.0;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
            if (INTEGRAL) tot /= sa * sb;
        }
        if (failed) { tot = NAN; if (lane == 0) *a.fail = 1; }
        if (lane == 0) {
            if (a.out) a.out[t] = tot;
            if (a.RA) { a.RA[opos] = tot; a.ncm[opos] = 0; }
        }
        __builtin_amdgcn_wave_barrier();
#ifdef EMD_PROFILE
        EP(6);
        if (a.dbg && t < 4096 && lane == 0) {
            for (int q = 0; q < 8; ++q) a.dbg[t * 16 + q] = ep_acc[q];
            for (int q = 0; q < 4; ++q) a.dbg[t * 16 + 8 + q] = ep_cnt[q];
            a.dbg[t * 16 + 12] = n; a.dbg[t * 16 + 13] = m;
        }
#endif
        {
            int nxt = 0;
            if (lane == 0) nxt = atomicAdd(a.work, 1);
            t = wave_total + (int64_t)__builtin_amdgcn_readfirstlane(nxt);
        }
    }
}

int ann_emd_launch(annchor_ctx *c, const PairSource &src, double *d_out, double *d_RA, uint8_t *d_ncm)
{
    if (src.n == 0) return ANNCHOR_OK;
    EmdArgs a;
    a.hist = c->hist.as<double>();
    a.cost = c->cost.as<double>();
    a.nb = c->nbins;
    a.ij = src.ij; a.idx = src.idx; a.anchor = src.anchor; a.n = src.n;
    a.out = d_out; a.RA = d_RA; a.ncm = d_ncm;
    a.pick_row = nullptr; a.pick_runmin = nullptr; a.pick_out = nullptr; a.pick_reset = 0; a.pick_nx = 0;
    if (!c->supp.p) {
        ANN_TRY(ann_reserve(c, c->supp, 64));
        ANN_CHECK_HIP(c, hipMemsetAsync(c->supp.p, 0, 64, c->stream));
        c->emd_epoch = 0;
    }
    a.fail = c->supp.as<int32_t>();
    a.work = a.fail + 4 + (c->emd_epoch & 1);
    a.work_next = a.fail + 4 + ((c->emd_epoch + 1) & 1);
    ++c->emd_epoch;
    a.reduce = c->cost_is_metric && !getenv("ANNCHOR_EMD_NO_REDUCE");
    const size_t cost_bytes = sizeof(double) * (size_t)a.nb * a.nb;
    const bool integral = c->hist_integral;
    {
        // metric ground cost: the transportation simplex with one lane per node (ANNCHOR_EMD_SOLVER=ssp keeps the shortest-path solver)
        const char *e = getenv("ANNCHOR_EMD_SOLVER");
        if (a.reduce && !(e && !strcmp(e, "ssp"))) {
            a.eps = c->cost_max * 1.1368683772161603e-13;   // 2^-43
            a.S = 0; a.slab_bytes = 0; a.dbg = nullptr;
            if (src.anchor && src.pick_fused && c->nx <= 65536) {
                *src.pick_fused = true;
                if (src.pick_row) {
                    a.pick_row = src.pick_row; a.pick_runmin = src.pick_runmin; a.pick_out = src.pick_out;
                    a.pick_reset = src.pick_reset; a.pick_nx = (int)c->nx;
                }
            }
            a.dantzig_cap = -1;
            if (const char *dc = getenv("ANNCHOR_EMD_DANTZIG_CAP")) a.dantzig_cap = atoi(dc);
            int waves = 16;
            const int64_t spread = (src.n + 2 * (int64_t)c->prop.multiProcessorCount - 1) / (2 * (int64_t)c->prop.multiProcessorCount);
            if (spread < waves) waves = (int)std::max<int64_t>(spread, 1);
            if (const char *w = getenv("ANNCHOR_EMD_WAVES")) { const int ww = atoi(w); if (ww >= 1 && ww < waves) waves = ww; }
            a.waves = waves;
            const bool sparse = a.nb > EMD_MAXB;
            a.hs_bin = c->hs_bin.as<int32_t>(); a.hs_val = c->hs_val.as<double>(); a.hs_cnt = c->hs_cnt.as<int32_t>();
            const size_t lds = (sparse ? 0 : cost_bytes) + (size_t)waves * EMD_NS_WAVE_BYTES;
            const void *fn = sparse ? (integral ? (const void *)k_emd_ns<int, true> : (const void *)k_emd_ns<double, true>)
                                    : (integral ? (const void *)k_emd_ns<int, false> : (const void *)k_emd_ns<double, false>);
            ANN_CHECK_HIP(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            int64_t blocks = (src.n + waves - 1) / waves;
            int per_cu = 1;   // resident workgroups only: the waves claim their solves from a counter
            ANN_CHECK_HIP(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, waves * 64, lds));
            const int64_t resident = (int64_t)c->prop.multiProcessorCount * std::max(per_cu, 1);
            if (blocks > resident) blocks = resident;
#ifdef EMD_PROFILE
            static long long *dbg = nullptr;
            if (!dbg) (void)hipMalloc(&dbg, sizeof(long long) * 4096 * 16);
            if (src.n <= 4096) { a.dbg = dbg; (void)hipMemsetAsync(dbg, 0, sizeof(long long) * 4096 * 16, c->stream); }
#endif
            ProfScope ps(c, "wasserstein_pairs", (double)src.n * (2.0 * std::min(a.nb, 64) * 8 + 8));
            if (sparse) {
                if (integral) k_emd_ns<int, true><<<(int)blocks, waves * 64, lds, c->stream>>>(a);
                else k_emd_ns<double, true><<<(int)blocks, waves * 64, lds, c->stream>>>(a);
            } else {
                if (integral) k_emd_ns<int, false><<<(int)blocks, waves * 64, lds, c->stream>>>(a);
                else k_emd_ns<double, false><<<(int)blocks, waves * 64, lds, c->stream>>>(a);
            }
            ANN_CHECK_HIP(c, hipGetLastError());
#ifdef EMD_PROFILE
            if (a.dbg) {
                static int calls = 0;
                if (++calls % 20 == 7) {
                    std::vector<long long> h((size_t)4096 * 16);
                    (void)hipStreamSynchronize(c->stream);
                    (void)hipMemcpy(h.data(), dbg, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
                    int64_t worst = 0; long long wt = 0; double tot[8] = {0};
                    for (int64_t t = 0; t < src.n; ++t) { long long sum = 0; for (int q = 0; q < 8; ++q) { sum += h[t * 16 + q]; tot[q] += h[t * 16 + q]; } if (sum > wt) { wt = sum; worst = t; } }
                    const char *nm[7] = {"setup", "start basis", "potentials", "pricing", "cycle+theta", "tree update", "objective"};
                    fprintf(stderr, "k_emd_ns launch of %lld solves, %d waves per block: slowest solve %lld: %lld cycles, n %lld m %lld, pivots %lld path steps %lld\n",
                            (long long)src.n, waves, (long long)worst, wt, h[worst * 16 + 12], h[worst * 16 + 13], h[worst * 16 + 8], h[worst * 16 + 9]);
                    for (int q = 0; q < 7; ++q) fprintf(stderr, "   %-14s slowest %8lld   mean %8.0f\n", nm[q], h[worst * 16 + q], tot[q] / (double)src.n);
                }
            }
#endif
            return ANNCHOR_OK;
        }
    }
    ANN_REQUIRE(c, a.nb <= EMD_MAXB, ANNCHOR_ELIMIT, "histograms of %d bins need a metric ground cost (the shortest-path solver takes up to %d bins)", a.nb, EMD_MAXB);
    a.eps = 0.0;
    const bool narrow = integral && c->hist_fits_i16 && !getenv("ANNCHOR_EMD_WIDE_FLOWS");
    // Flow slab: n sources x (m | 1) sinks.  The supports of the raw histograms bound n and m by max_support; with the common
    // mass cancelled the two supports are disjoint (n + m <= bins), which halves the worst case again.
    size_t entries = 0;
    if (a.reduce) {
        a.S = 0;   // stride per solve
        for (int n = 1; n <= c->max_support; ++n) {
            const int m = std::min(c->max_support, a.nb - n);
            if (m >= 1) entries = std::max(entries, (size_t)n * (size_t)(m | 1));
        }
        if (!entries) entries = 1;
    } else {
        a.S = c->max_support | 1;  // odd stride: conflict-free column reads
        entries = (size_t)c->max_support * a.S;
    }
    const size_t esz = narrow ? 2 : integral ? 4 : 8;
    const size_t slab = ((esz * entries + 2 * EMD_MAXB * sizeof(int)) + 15) & ~(size_t)15;
    a.slab_bytes = (int)slab;
    // The solver is latency bound (one DPP minimum per Dijkstra pop on the critical path): as many waves per SIMD as the LDS
    // takes.  A workgroup holds at most 16 waves, so two workgroups per CU, each with its own copy of the ground costs.
    int per_cu = 2;
    int waves = (int)(((160 * 1024) / per_cu - cost_bytes) / slab);
    if (waves < 8) { per_cu = 1; waves = (int)((160 * 1024 - cost_bytes) / slab); }
    if (waves > 16) waves = 16;
    // a small launch (an anchor round: one solve per point) is as slow as its slowest solve: spread it over every SIMD of the
    // chip instead of filling a few CUs
    {
        const int64_t spread = (src.n + 2 * (int64_t)c->prop.multiProcessorCount - 1) / (2 * (int64_t)c->prop.multiProcessorCount);
        if (spread < waves) waves = (int)std::max<int64_t>(spread, 1);
    }
    if (const char *w = getenv("ANNCHOR_EMD_WAVES")) { const int ww = atoi(w); if (ww >= 1 && ww < waves) waves = ww; }   // tuning / occupancy experiments
    ANN_REQUIRE(c, waves >= 1, ANNCHOR_ELIMIT, "histogram support %d needs more LDS than a CU has", c->max_support);
    a.waves = waves;
    const size_t lds = cost_bytes + slab * waves;
    const void *fn = narrow ? (const void *)k_emd<int, int16_t> : integral ? (const void *)k_emd<int, int> : (const void *)k_emd<double, double>;
    ANN_CHECK_HIP(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t blocks = (src.n + waves - 1) / waves;
    if (blocks > (int64_t)per_cu * c->prop.multiProcessorCount) blocks = (int64_t)per_cu * c->prop.multiProcessorCount;  // resident blocks only (LDS bound)
    a.dbg = nullptr;
#ifdef EMD_PROFILE
    static long long *dbg = nullptr;
    if (!dbg) (void)hipMalloc(&dbg, sizeof(long long) * 4096 * 16);
    if (src.n <= 4096) { a.dbg = dbg; (void)hipMemsetAsync(dbg, 0, sizeof(long long) * 4096 * 16, c->stream); }
#endif
    ProfScope ps(c, "wasserstein_pairs", (double)src.n * (2.0 * a.nb * 8 + 8));
    if (narrow) k_emd<int, int16_t><<<(int)blocks, waves * 64, lds, c->stream>>>(a);
    else if (integral) k_emd<int, int><<<(int)blocks, waves * 64, lds, c->stream>>>(a);
    else k_emd<double, double><<<(int)blocks, waves * 64, lds, c->stream>>>(a);
#ifdef EMD_PROFILE
    if (a.dbg) {
        static int calls = 0;
        if (++calls % 20 == 7) {
            std::vector<long long> h((size_t)4096 * 16);
            (void)hipStreamSynchronize(c->stream);
            (void)hipMemcpy(h.data(), dbg, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
            int64_t worst = 0; long long wt = 0; double tot[8] = {0};
            for (int64_t t = 0; t < src.n; ++t) { long long sum = 0; for (int q = 0; q < 8; ++q) { sum += h[t * 16 + q]; tot[q] += h[t * 16 + q]; } if (sum > wt) { wt = sum; worst = t; } }
            const char *nm[8] = {"setup", "search init", "pop min", "column+ballot", "relax", "potentials", "augment", "objective"};
            fprintf(stderr, "k_emd launch of %lld solves, %d waves per block: slowest solve %lld: %lld cycles, n %lld m %lld, searches %lld pops %lld relaxed %lld augmentations %lld\n",
                    (long long)src.n, waves, (long long)worst, wt, h[worst * 16 + 12], h[worst * 16 + 13], h[worst * 16 + 8], h[worst * 16 + 9], h[worst * 16 + 10], h[worst * 16 + 11]);
            for (int q = 0; q < 8; ++q) fprintf(stderr, "   %-14s slowest %8lld   mean %8.0f\n", nm[q], h[worst * 16 + q], tot[q] / (double)src.n);
        }
    }
#endif
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}
