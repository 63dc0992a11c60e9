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
__global__ __launch_bounds__(ROW_THREADS) void k_comp_count(const int64_t *__restrict__ Iptr, const int32_t *__restrict__ Iidx,
                                                           const uint8_t *__restrict__ ncm, int32_t *__restrict__ cnt)
{
    __shared__ uint32_t acc;
    if (threadIdx.x == 0) acc = 0;
    __syncthreads();
    const int64_t i = row_of_block(gridDim.x), b = Iptr[i];
    const int len = (int)(Iptr[i + 1] - b);
    uint32_t s = 0;
    for (int k = threadIdx.x; k < len; k += ROW_THREADS) s += !ncm[Iidx[b + k]];
    if (s) atomicAdd(&acc, s);
    __syncthreads();
    if (threadIdx.x == 0) cnt[i] = (int32_t)acc;
}

__global__ __launch_bounds__(ROW_THREADS) void k_comp_fill(const int64_t *__restrict__ Iptr, const int32_t *__restrict__ Iidx,
                                                          const uint8_t *__restrict__ ncm, const int2 *__restrict__ ij,
                                                          const double *__restrict__ RA, const int64_t *__restrict__ cptr,
                                                          int32_t *__restrict__ cidx, double *__restrict__ cval)
{
    __shared__ uint32_t wsum[ROW_THREADS / 64];
    const int64_t i = row_of_block(gridDim.x), b = Iptr[i];
    const int len = (int)(Iptr[i + 1] - b);
    int64_t w = cptr[i];
    for (int base = 0; base < len; base += ROW_THREADS) {
        const int k = base + threadIdx.x;
        int32_t p = 0;
        uint32_t f = 0;
        if (k < len) { p = Iidx[b + k]; f = !ncm[p]; }
        uint32_t tot;
        const uint32_t ex = row_block_scan(f, wsum, &tot);
        if (f) {
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
    {
        ProfScope ps(c, "computed_neighbour_csr", (double)c->n * 2 * 5.0);
        k_comp_count<<<(int)nx, ROW_THREADS, 0, c->stream>>>(c->Iptr.as<int64_t>(), c->Iidx.as<int32_t>(), c->ncm.as<uint8_t>(),
                                                            c->tmp1.as<int32_t>());
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
        k_comp_fill<<<(int)nx, ROW_THREADS, 0, c->stream>>>(c->Iptr.as<int64_t>(), c->Iidx.as<int32_t>(), c->ncm.as<uint8_t>(),
                                                           c->ij.as<int2>(), c->RA.as<double>(), c->cptr.as<int64_t>(),
                                                           c->cidx.as<int32_t>(), c->cval.as<double>());
    }
    {
        // algorithmic bytes per lookahead pair: both computed lists once, 12 B per entry
        const int64_t counted = c->n_unc >= 0 ? 2 * (c->n - c->n_unc) : total;   // computed entries, both directions
        const double avg = nx > 0 ? (double)counted / (double)nx : 0.0;
        ProfScope ps(c, "update_bounds_intersect", (double)c->nnext * (2.0 * avg * 12.0 + 36.0));
        k_update_bounds<<<ann_blocks(c->nnext * 64, 256), 256, 0, c->stream>>>(
            c->next.as<int32_t>(), c->nnext, c->ij.as<int2>(), c->cptr.as<int64_t>(), c->cidx.as<int32_t>(),
            c->cval.as<double>(), c->lb.as<double>(), c->ub.as<double>());
    }
    ANN_CHECK_HIP(c, hipEventRecord(c->call_b, c->stream));
    c->call_timed = true;
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}

// -------------------------------------------------------------------- get_nn
// block per row: keys d' = RA (+ row max on not-computed entries); the nn-1 smallest
// by (d', slot); output values are the un-shifted RA (utils.py:417-428)
__global__ __launch_bounds__(ROW_THREADS) void k_get_nn(const int64_t *__restrict__ Iptr, const int32_t *__restrict__ Iidx,
                                                       const double *__restrict__ RA, const uint8_t *__restrict__ ncm,
                                                       const int2 *__restrict__ ij, int nn, int64_t *__restrict__ ngi,
                                                       double *__restrict__ ngd, int cap)
{
    __shared__ RowSelShared sh;
    __shared__ double wmax[ROW_THREADS / 64];
    __shared__ uint32_t cnt_lt;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    const int L = nn - 1;
    uint64_t *keys = reinterpret_cast<uint64_t *>(dyn);      // [cap]
    uint64_t *lkey = keys + cap;                             // [L]
    int32_t *lslot = reinterpret_cast<int32_t *>(lkey + L);  // [L]
    const int64_t i = row_of_block(gridDim.x), b = Iptr[i];
    const int len = (int)(Iptr[i + 1] - b);
    if (threadIdx.x == 0) { ngi[i * nn] = i; ngd[i * nn] = 0.0; cnt_lt = 0; }
    // row maximum of RA (utils.py:418)
    double mx = -INFINITY;
    for (int s = threadIdx.x; s < len; s += ROW_THREADS) mx = fmax(mx, RA[Iidx[b + s]]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3]));
    const bool in_lds = len <= cap;
    auto key_of = [&](int s) -> uint64_t {
        const int32_t p = Iidx[b + s];
        double d = RA[p];
        if (ncm[p]) d += mx;
        return ann_key_asc(d);
    };
    if (in_lds) {
        for (int s = threadIdx.x; s < len; s += ROW_THREADS) keys[s] = key_of(s);
        __syncthreads();
    }
    auto kf = [&](int s) -> uint64_t { return in_lds ? keys[s] : key_of(s); };
    const int want = min(L, len);
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
    {
        ProfScope ps(c, "row_topk_graph", (double)c->n * 2 * 13.0 + (double)cells * 16.0);
        const size_t tail = (((size_t)(nn - 1) * 12) + 15) & ~(size_t)15;
        const int cap = row_lds_cap(c->nx, tail);
        ANN_TRY(row_lds_prepare(c, k_get_nn, (size_t)cap * 8 + tail));
        k_get_nn<<<(int)c->nx, ROW_THREADS, (size_t)cap * 8 + tail, c->stream>>>(
            c->Iptr.as<int64_t>(), c->Iidx.as<int32_t>(), c->RA.as<double>(), c->ncm.as<uint8_t>(), c->ij.as<int2>(), nn, d_i,
            d_d, cap);
    }
    ANN_CHECK_HIP(c, hipEventRecord(c->call_b, c->stream));
    c->call_timed = true;
    ANN_CHECK_HIP(c, hipGetLastError());
    if (direct) {
        ANN_CHECK_HIP(c, hipStreamSynchronize(c->stream));
        memcpy(ng_idx, slot, cells * 8);
        memcpy(ng_dist, slot + cells * 8, cells * 8);
        return ANNCHOR_OK;
    }
    if (c->pin && cells * 16 <= annchor_ctx::PIN_DL_BYTES) {
        // indices and distances sit back to back: one transfer into the pinned download region
        ANN_CHECK_HIP(c, hipMemcpyAsync(slot, d_i, cells * 16, hipMemcpyDeviceToHost, c->stream));
        ANN_CHECK_HIP(c, hipStreamSynchronize(c->stream));
        memcpy(ng_idx, slot, cells * 8);
        memcpy(ng_dist, slot + cells * 8, cells * 8);
        return ANNCHOR_OK;
    }
    ANN_TRY(ann_d2h(c, ng_idx, d_i, cells * 8));
    return ann_d2h(c, ng_dist, d_d, cells * 8);
}
