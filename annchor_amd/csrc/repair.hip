// repair.hip -- exact repair of the rows the split-fp16 tile kernels flag (k_st_repair).
//
// The split-fp16 kernels (knnbf.hip, knnbk.hip) SELECT by |x|^2 + |y|^2 - 2 x.y with products good to ~2^-22 |x||y| and re-rank
// what they keep exactly; their guard flags a row when its K-th exact distance comes within the measured error of its list's last
// approximate entry -- a neighbour may then have been left outside the list (tight clusters far from the centre: |x|^2 >> d^2).
// No product form in float32 resolves such rows (the exact-f32 MFMA kernel evaluates the same expanded form), and beyond 256
// dimensions there was no second kernel at all.  The reference computes np.linalg.norm(x - y) on the float32 rows
// (annchor/distances.py:8-13): DIFFERENCES.  So the flagged ROWS are done again that way: one workgroup per row tile that holds
// any (the kernels leave a 128-bit mask per row tile), row by row, over the column tiles that row tile evaluated (its bitmap row;
// without a recorded bitmap -- a full-budget build -- every column tile whose interval bound stays below the row's K-th exact
// distance so far): two threads per column sum (x - y)^2 in float32, the columns that beat the row's K-th entry are collected and
// inserted in (d^2, column) order.  ~1 ms per flagged row at N = 10^6 (512 tiles of 128 columns x 128 dimensions); well-conditioned
// data flags none (C3: 0 rows), N = 8 x 10^6 a handful.  (First form of the round: the whole row tile again, 128 rows against every
// evaluated tile on one CU -- 50 ms for ONE flagged row tile at N = 8 x 10^6, whatever the number of ranks.)
// Any padded dimension, any list length of the split kernels (K <= 62), graph builds and queries.
#include "streamed.h"

#define RP_THREADS 256
#define RP_MAXDIM 1024

// the flagged rows as a list: one thread per mask word, entries (row tile << 7 | row) appended in any order
__global__ void k_st_repair_list(const uint32_t *__restrict__ guard_tiles, int64_t nwords, uint32_t *__restrict__ list, uint32_t *__restrict__ count,
                                 uint32_t cap)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwords) return;
    uint32_t bits = guard_tiles[w];
    while (bits) {
        const int b = __builtin_ctz(bits);
        bits &= bits - 1;
        const uint32_t slot = atomicAdd(count, 1u);
        if (slot < cap) list[slot] = (uint32_t)((w >> 2) << 7) | (uint32_t)(((w & 3) << 5) + b);
    }
}

__global__ __launch_bounds__(RP_THREADS) void k_st_repair(KnnArgs a, int dimp, const uint32_t *__restrict__ list, const uint32_t *__restrict__ count)
{
    if (blockIdx.x >= *count) return;
    const uint32_t ent = list[blockIdx.x];
    const int bt = (int)(ent >> 7);
    uint32_t gw[4] = {0, 0, 0, 0};
    gw[(ent & 127) >> 5] = 1u << (ent & 31);   // (one row per workgroup: the rows of a tile run side by side)
    __shared__ float xrow[RP_MAXDIM];
    __shared__ float ld[ST_KMAX_BIG];        // the row's list: exact d^2 ascending by (d^2, column)
    __shared__ int32_t lc[ST_KMAX_BIG];
    __shared__ float cd[ST_T];               // candidates of one column tile: what beats the K-th entry
    __shared__ int32_t cc_[ST_T];
    __shared__ int ncand;
    const int K = a.K;
    const int I = a.tile_begin + bt;
    const int64_t grow0 = (int64_t)I * ST_T;
    const int tid = threadIdx.x, cl = tid >> 1, hf = tid & 1;   // two threads per column: halves of the dimensions
    const uint32_t *eb = a.eval_bits ? a.eval_bits + (size_t)bt * a.eval_halves * a.eval_words : nullptr;
    const float *slb = a.scr_lb + (size_t)bt * a.nt_all;      // valid interval bounds of (this row tile, every column tile)
    const int hd = dimp >> 1;                                // (dimp is a multiple of 32)
    for (int row = 0; row < ST_T; ++row) {
        if (!((gw[row >> 5] >> (row & 31)) & 1u)) continue;   // (uniform)
        __syncthreads();
        for (int k = tid; k < dimp; k += RP_THREADS) xrow[k] = a.Rs[(size_t)(grow0 + row) * dimp + k];
        if (tid < K) { ld[tid] = INFINITY; lc[tid] = 0x7fffffff; }
        if (tid == 0) ncand = 0;
        __syncthreads();
        // (word by word: a bitmap read per column tile -- 62 500 of them at N = 8 x 10^6 for ~660 evaluated -- was the kernel's time)
        for (int w = 0; w < (a.nt_all + 31) / 32; ++w) {
        uint32_t wbits = eb ? eb[w] : 0xffffffffu;
        while (wbits) {   // (uniform)
            const int J = 32 * w + __builtin_ctz(wbits);
            wbits &= wbits - 1;
            if (J >= a.nt_all) break;
            const float thr = ld[K - 1];
            const int32_t thc = lc[K - 1];
            if (!eb) {   // (with a bitmap: the tile phase's own set of column tiles, its budget)
                const float lb = slb[J];
                if (!(lb * lb < thr) && !(J == I && !a.query)) continue;
            }
            const int64_t col = (int64_t)J * ST_T + cl;
            const float4 *y = reinterpret_cast<const float4 *>(a.Xs + (size_t)col * dimp + hf * hd);
            const float4 *x = reinterpret_cast<const float4 *>(xrow + hf * hd);
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 8   // (eight loads in flight per thread: one at a time, a tile cost sixteen global round trips)
            for (int t = 0; t < hd / 4; ++t) {
                const float4 u = x[t], v = y[t];
                const float dx = u.x - v.x, dy = u.y - v.y, dz = u.z - v.z, dw = u.w - v.w;
                s0 += dx * dx; s1 += dy * dy; s2 += dz * dz; s3 += dw * dw;
            }
            float d = (s0 + s1) + (s2 + s3);
            d += __shfl_xor(d, 1);
            const bool real = a.rs[col] < INFINITY && !(!a.query && col == grow0 + row);   // (padding columns carry +inf; a point is not its own neighbour)
            if (hf == 0 && real && (d < thr || (d == thr && (int32_t)col < thc))) {
                const int slot = atomicAdd(&ncand, 1);
                cd[slot] = d; cc_[slot] = (int32_t)col;
            }
            __syncthreads();
            if (ncand) {   // (uniform)
                if (tid == 0) {
                    for (int q = 0; q < ncand; ++q) {
                        const float dq = cd[q];
                        const int32_t cq = cc_[q];
                        if (!(dq < ld[K - 1] || (dq == ld[K - 1] && cq < lc[K - 1]))) continue;
                        int p = K - 1;
                        while (p > 0 && (dq < ld[p - 1] || (dq == ld[p - 1] && cq < lc[p - 1]))) { ld[p] = ld[p - 1]; lc[p] = lc[p - 1]; --p; }
                        ld[p] = dq; lc[p] = cq;
                    }
                    ncand = 0;
                }
                __syncthreads();
            }
        }
        }
        __syncthreads();
        if (tid < K) {
            a.out_d2[((size_t)bt * ST_T + row) * K + tid] = ld[tid];
            a.out_col[((size_t)bt * ST_T + row) * K + tid] = ld[tid] < INFINITY ? lc[tid] : 0x7fffffff;
        }
    }
}

// the flagged rows of the tile phase just run (guard_tiles [tile_count][4]: a bitmask of flagged rows per row tile; `flagged`: how
// many, as the host read it), exactly: a list of them, then one workgroup per row
int ann_stream_repair_flagged(annchor_ctx *c, StreamState *s, const KnnArgs &a, int dim_padded, const uint32_t *guard_tiles, int64_t flagged)
{
    ANN_REQUIRE(c, dim_padded <= RP_MAXDIM && a.K <= ST_KMAX_BIG, ANNCHOR_ELIMIT, "exact repair: padded dim %d, %d-entry lists", dim_padded, a.K);
    if (flagged <= 0) return ANNCHOR_OK;
    const int64_t cap = std::min<int64_t>(flagged, (int64_t)a.tile_count * ST_T);
    ANN_TRY(ann_stream_reserve(c, s->guard_list, sizeof(uint32_t) * (size_t)(cap + 1)));
    uint32_t *cnt = s->guard_list.as<uint32_t>(), *list = cnt + 1;
    ANN_CHECK_HIP(c, hipMemsetAsync(cnt, 0, sizeof(uint32_t), c->stream));
    const int64_t nwords = (int64_t)a.tile_count * 4;
    k_st_repair_list<<<ann_blocks(nwords, 256), 256, 0, c->stream>>>(guard_tiles, nwords, list, cnt, (uint32_t)cap);
    k_st_repair<<<(unsigned)cap, RP_THREADS, 0, c->stream>>>(a, dim_padded, list, cnt);
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}
