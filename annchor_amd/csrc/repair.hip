// repair.hip -- exact repair of the row tiles the split-fp16 tile kernels flag (k_st_repair).
//
// The split-fp16 kernels (knnbf.hip, knnbk.hip) SELECT by |x|^2 + |y|^2 - 2 x.y with products good to ~2^-22 |x||y| and re-rank
// what they keep exactly; their guard flags a row when its K-th exact distance comes within the measured error of its list's last
// approximate entry -- a neighbour may then have been left outside the list (tight clusters far from the centre: |x|^2 >> d^2).
// No product form in float32 resolves such rows (the exact-f32 MFMA kernel evaluates the same expanded form), and beyond 256
// dimensions there was no second kernel at all.  The reference computes np.linalg.norm(x - y) on the float32 rows
// (annchor/distances.py:8-13): DIFFERENCES.  So the flagged row tiles are done again that way: one workgroup per flagged row tile,
// over the column tiles that row tile evaluated (its bitmap row; without a recorded bitmap -- a full-budget build -- every column
// tile whose interval bound stays below the row tile's worst exact K-th distance), sum (x - y)^2 in float32 on the vector ALUs,
// sorted insertion per row by (d^2, column).  A slow path by construction (~20 us per tile pair at 300 dimensions, against 9 us
// for the split kernel's 128): it runs for the flagged row tiles only, and well-conditioned data flags none (C3: 0 rows).
// Any padded dimension, any list length of the split kernels (K <= 62), graph builds and queries.
#include "streamed.h"

#define RP_THREADS 256
#define RP_KC 32   // dimensions per staged chunk
#define RP_CH 64   // columns per pass (half a column tile)

__global__ __launch_bounds__(RP_THREADS) void k_st_repair(KnnArgs a, int dimp, const uint32_t *__restrict__ guard_tiles)
{
    const int bt = blockIdx.x;
    if (!guard_tiles[bt]) return;   // (uniform)
    extern __shared__ __attribute__((aligned(16))) unsigned char rsm[];
    const int K = a.K;
    float *ld = reinterpret_cast<float *>(rsm);                       // [ST_T][K] exact d^2, ascending by (d^2, column)
    int32_t *lc = reinterpret_cast<int32_t *>(ld + ST_T * K);         // [ST_T][K]
    float *d2m = reinterpret_cast<float *>(lc + ST_T * K);            // [ST_T][RP_CH + 1]
    float *xs = d2m + ST_T * (RP_CH + 1);                             // [ST_T][RP_KC + 1]
    float *ys = xs + ST_T * (RP_KC + 1);                              // [RP_CH][RP_KC + 1]
    __shared__ float thr_w[RP_THREADS / 64];
    const int I = a.tile_begin + bt;
    const int64_t grow0 = (int64_t)I * ST_T;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;        // thread (ty, tx): rows 8 ty .. + 7, columns 4 tx .. + 3 of a pass
    for (int q = tid; q < ST_T * K; q += RP_THREADS) { ld[q] = INFINITY; lc[q] = 0x7fffffff; }
    const uint32_t *eb = a.eval_bits ? a.eval_bits + (size_t)bt * a.eval_halves * a.eval_words : nullptr;
    const float *slb = a.scr_lb + (size_t)bt * a.nt_all;             // valid interval bounds of (this row tile, every column tile)
    float thrmax = INFINITY;   // (uniform) worst K-th exact d^2 over the tile's real rows
    __syncthreads();
    for (int J = 0; J < a.nt_all; ++J) {
        if (eb) {
            if (!((eb[J >> 5] >> (J & 31)) & 1u)) continue;            // the tile phase's own set of column tiles (its budget)
        } else {
            const float lb = slb[J];
            if (!(lb * lb < thrmax) && !(J == I && !a.query)) continue;
        }
        for (int h = 0; h < ST_T / RP_CH; ++h) {
            const int64_t c0 = (int64_t)J * ST_T + h * RP_CH;
            float acc[8][4];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
            for (int k0 = 0; k0 < dimp; k0 += RP_KC) {
                __syncthreads();
                for (int u = tid; u < ST_T * RP_KC / 4; u += RP_THREADS) {
                    const int row = u / (RP_KC / 4), q4 = u % (RP_KC / 4);
                    const float4 v = *reinterpret_cast<const float4 *>(a.Rs + (size_t)(grow0 + row) * dimp + k0 + 4 * q4);
                    float *d = xs + row * (RP_KC + 1) + 4 * q4;
                    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                }
                for (int u = tid; u < RP_CH * RP_KC / 4; u += RP_THREADS) {
                    const int cl = u / (RP_KC / 4), q4 = u % (RP_KC / 4);
                    const float4 v = *reinterpret_cast<const float4 *>(a.Xs + (size_t)(c0 + cl) * dimp + k0 + 4 * q4);
                    float *d = ys + cl * (RP_KC + 1) + 4 * q4;
                    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                }
                __syncthreads();
#pragma unroll 4
                for (int kk = 0; kk < RP_KC; ++kk) {
                    float xv[8], yv[4];
#pragma unroll
                    for (int i = 0; i < 8; ++i) xv[i] = xs[(8 * ty + i) * (RP_KC + 1) + kk];
#pragma unroll
                    for (int j = 0; j < 4; ++j) yv[j] = ys[(4 * tx + j) * (RP_KC + 1) + kk];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float df = xv[i] - yv[j];
                            acc[i][j] += df * df;
                        }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t cc = c0 + 4 * tx + j;
                const bool colreal = a.rs[cc] < INFINITY;   // (padding columns carry +inf)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = 8 * ty + i;
                    const bool self = !a.query && cc == grow0 + row;   // a point is not its own neighbour
                    d2m[row * (RP_CH + 1) + 4 * tx + j] = (colreal && !self) ? acc[i][j] : INFINITY;
                }
            }
            __syncthreads();
            if (tid < ST_T && a.rr[grow0 + tid] < INFINITY) {   // one thread per (real) row: sorted insertion of what beats the row's K-th entry
                const int row = tid;
                float *rd = ld + row * K;
                int32_t *rc = lc + row * K;
                for (int cix = 0; cix < RP_CH; ++cix) {
                    const float d = d2m[row * (RP_CH + 1) + cix];
                    const int32_t cc = (int32_t)(c0 + cix);
                    if (!(d < rd[K - 1] || (d == rd[K - 1] && cc < rc[K - 1]))) continue;
                    int p = K - 1;
                    while (p > 0 && (d < rd[p - 1] || (d == rd[p - 1] && cc < rc[p - 1]))) { rd[p] = rd[p - 1]; rc[p] = rc[p - 1]; --p; }
                    rd[p] = d; rc[p] = cc;
                }
            }
            __syncthreads();
        }
        if (!eb) {   // the pruning threshold: the worst K-th exact d^2 of the tile's real rows
            float t = (tid < ST_T && a.rr[grow0 + tid] < INFINITY) ? ld[tid * K + K - 1] : -1.f;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) t = fmaxf(t, __shfl_xor(t, off));
            if ((tid & 63) == 0) thr_w[tid >> 6] = t;
            __syncthreads();
            thrmax = fmaxf(fmaxf(thr_w[0], thr_w[1]), fmaxf(thr_w[2], thr_w[3]));
            __syncthreads();
        }
    }
    __syncthreads();
    for (int q = tid; q < ST_T * K; q += RP_THREADS) {
        const int row = q / K;
        if (!(a.rr[grow0 + row] < INFINITY)) continue;   // (padding rows keep what the tile kernel wrote)
        a.out_d2[(size_t)bt * ST_T * K + q] = ld[q];
        a.out_col[(size_t)bt * ST_T * K + q] = ld[q] < INFINITY ? lc[q] : 0x7fffffff;
    }
}

// the flagged row tiles of the tile phase just run (guard_tiles[tile_count]: rows flagged per row tile), exactly
int ann_stream_repair_flagged(annchor_ctx *c, const KnnArgs &a, int dim_padded, const uint32_t *guard_tiles)
{
    const size_t lds = sizeof(float) * ((size_t)2 * ST_T * a.K + (size_t)ST_T * (RP_CH + 1) + (size_t)ST_T * (RP_KC + 1) + (size_t)RP_CH * (RP_KC + 1));
    ANN_REQUIRE(c, lds <= 150 * 1024, ANNCHOR_ELIMIT, "exact repair: %d-entry lists need %zu B of LDS", a.K, lds);
    ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)k_st_repair, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    k_st_repair<<<a.tile_count, RP_THREADS, lds, c->stream>>>(a, dim_padded, guard_tiles);
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}
