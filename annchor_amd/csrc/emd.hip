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
    int S;          // padded row stride of the flow slab (odd)
    int waves;      // waves per block
    int slab_bytes; // LDS bytes per wave (flow slab + support index lists)
    const int2 *ij;
    const int32_t *idx;
    const int32_t *anchor;
    int64_t n;
    double *out;
    double *RA;
    uint8_t *ncm;
    int32_t *fail;  // set if the iteration guard trips
    int reduce;     // metric ground cost: solve on the differences of the two (scaled) histograms
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

template <typename T> __global__ __launch_bounds__(1024) void k_emd(EmdArgs a)
{
    constexpr bool INTEGRAL = sizeof(T) == 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nb = a.nb, S = a.S;
    double *costL = reinterpret_cast<double *>(smem);                       // [nb][nb]
    unsigned char *slab = reinterpret_cast<unsigned char *>(costL + nb * nb) + (size_t)wave * a.slab_bytes;
    T *F = reinterpret_cast<T *>(slab);                                     // [<=64][S] flow slab
    int *rowsL = reinterpret_cast<int *>(slab + a.slab_bytes - 2 * EMD_MAXB * sizeof(int));  // [64] support of x
    int *colsL = rowsL + EMD_MAXB;                                          // [64] support of y
    for (int t = threadIdx.x; t < nb * nb; t += blockDim.x) costL[t] = a.cost[t];
    __syncthreads();

    const int64_t wave_global = (int64_t)blockIdx.x * a.waves + wave;
    const int64_t wave_stride = (int64_t)gridDim.x * a.waves;
    for (int64_t t = wave_global; t < a.n; t += wave_stride) {
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
            if (lane < m) F[i * S + lane] = (T)0;
        __builtin_amdgcn_wave_barrier();
        // greedy start on the tight arcs (complementary slackness holds: flow only where the
        // reduced cost is zero).  For near-by histograms most mass sits on identical bins
        // (cost 0) and is routed here, before any shortest-path search.
        for (int j = 0; j < m; ++j) {
            const int i = __builtin_amdgcn_readlane(amin, j);
            const T f = tmin(rl(a_rem, i), rl(b_rem, j));
            if (f > (T)0) {
                if (lane == 0) F[i * S + j] = f;
                if (lane == i) a_rem -= f;
                if (lane == j) b_rem -= f;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

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
                for (;;) {
                    const bool open = lane < m && !((sinkdone >> lane) & 1ull);
                    const double best = wave_min_nonneg_f64(open ? dist : INFINITY);
                    const unsigned long long hit = __ballot(open && dist == best);
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
                    const T fcol = lane < n ? F[lane * S + js] : (T)0;
                    unsigned long long todo = __ballot(lane < n && fcol > (T)0 && !((srcdone >> lane) & 1ull));
                    while (todo) {
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
                }
                if (nhit == 0) { dust = true; break; }     // only rounding dust left
                // ---- potentials: every scanned node against the last distance popped -- the tree's arcs are tight afterwards
                if ((srcdone >> lane) & 1ull) u += mu - srcdist;
                if ((sinkdone >> lane) & 1ull) v -= mu - dist;
                // ---- augment along the tree to every sink reached with demand left, nearest first; a path's bottleneck is taken
                // from the flows as the earlier augmentations of this search left them
                for (int k = 0; k < nhit; ++k) {
                    const int jend = __ffsll((unsigned long long)__ballot(prank == k)) - 1;
                    T delta = tmin(rl(a_rem, s), rl(b_rem, jend));
                    for (int j = jend; delta > (T)0;) {
                        const int i = __builtin_amdgcn_readlane(pred, j);
                        if (i == s) break;
                        const int jj = __builtin_amdgcn_readlane(srcfrom, i);
                        delta = tmin(delta, F[i * S + jj]);   // uniform address: LDS broadcast
                        j = jj;
                    }
                    if (!(delta > (T)0)) continue;
                    for (int j = jend;;) {
                        const int i = __builtin_amdgcn_readlane(pred, j);
                        if (lane == 0) F[i * S + j] += delta;
                        if (i == s) break;
                        const int jj = __builtin_amdgcn_readlane(srcfrom, i);
                        if (lane == 0) F[i * S + jj] -= delta;
                        j = jj;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    if (lane == s) a_rem -= delta;
                    if (lane == jend) b_rem -= delta;
                }
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
    }
}

int ann_emd_launch(annchor_ctx *c, const PairSource &src, double *d_out, double *d_RA, uint8_t *d_ncm)
{
    if (src.n == 0) return ANNCHOR_OK;
    EmdArgs a;
    a.hist = c->hist.as<double>();
    a.cost = c->cost.as<double>();
    a.nb = c->nbins;
    int S = c->max_support | 1;  // odd stride: conflict-free column reads
    a.S = S;
    a.ij = src.ij; a.idx = src.idx; a.anchor = src.anchor; a.n = src.n;
    a.out = d_out; a.RA = d_RA; a.ncm = d_ncm;
    ANN_TRY(ann_reserve(c, c->supp, 64));
    a.fail = c->supp.as<int32_t>();
    a.reduce = c->cost_is_metric && !getenv("ANNCHOR_EMD_NO_REDUCE");
    ANN_CHECK_HIP(c, hipMemsetAsync(a.fail, 0, 4, c->stream));
    const size_t cost_bytes = sizeof(double) * (size_t)a.nb * a.nb;
    const bool integral = c->hist_integral;
    const size_t slab = (((integral ? 4 : 8) * (size_t)c->max_support * S + 2 * EMD_MAXB * sizeof(int)) + 15) & ~(size_t)15;
    a.slab_bytes = (int)slab;
    int waves = (int)((160 * 1024 - cost_bytes) / slab);
    if (waves > 16) waves = 16;
    if (const char *w = getenv("ANNCHOR_EMD_WAVES")) { const int ww = atoi(w); if (ww >= 1 && ww < waves) waves = ww; }   // tuning / occupancy experiments
    ANN_REQUIRE(c, waves >= 1, ANNCHOR_ELIMIT, "histogram support %d needs more LDS than a CU has", c->max_support);
    a.waves = waves;
    const size_t lds = cost_bytes + slab * waves;
    const void *fn = integral ? (const void *)k_emd<int> : (const void *)k_emd<double>;
    ANN_CHECK_HIP(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t blocks = (src.n + waves - 1) / waves;
    if (blocks > c->prop.multiProcessorCount) blocks = c->prop.multiProcessorCount;  // one resident block per CU (LDS bound)
    ProfScope ps(c, "wasserstein_pairs", (double)src.n * (2.0 * a.nb * 8 + 8));
    if (integral) k_emd<int><<<(int)blocks, waves * 64, lds, c->stream>>>(a);
    else k_emd<double><<<(int)blocks, waves * 64, lds, c->stream>>>(a);
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}
