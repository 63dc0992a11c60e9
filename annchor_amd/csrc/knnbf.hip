// knnbf.hip -- the tile phase of the streamed k-NN build on the 16-bit matrix cores (k_st_knnbf): split-fp16 tile GEMMs,
// operands by LDS-DMA, exact re-ranking of what is kept.
//
// Same algorithm as k_st_knn (streamed.hip): a workgroup owns a 128-row tile, ranks the column tiles, evaluates them as
// tile GEMMs and keeps the best columns per row in LDS.  What is different, and why (tools/microbench/shadow.hip,
// pingpong.hip, f16_split.hip, measured on MI355X):
//   * v_mfma_f32_32x32x2_f32 runs at the f32 VECTOR rate and, it turns out, in the vector ALUs' issue slot: nothing
//     hides in its shadow (64 cycles per MFMA bare; 84 with two v_fma behind each, 110 with eight), so an f32 tile
//     kernel pays for every threshold test, LDS write and address computation in full -- k_st_knn's 60 % of the f32
//     peak is that.
//   * v_mfma_f32_32x32x16_f16 / _bf16 is a real matrix pipe: 32 cycles per MFMA with up to four VALU instructions behind
//     each for free, and sixteen times the f32 rate.  With every float split into two fp16 (x = hi + lo: 22 bits of
//     mantissa) a dot product is hi.hi + hi.lo + lo.hi -- three MFMAs per 16 dimensions, 24 per 32 x 32 x 128 block =
//     768 matrix-pipe cycles against 4096 for the exact f32 stream -- with a measured error of 2^-22 |x||y| (rms 2^-24.7):
//     the accuracy of the f32 MFMA stream itself (2^-21.5, rms 2^-23.9; split bf16: 2^-19).  fp16's range is handled once
//     per data set: the rows are centred (c = mean of the anchors) and scaled by the power of two that puts the largest
//     |coordinate| in (2^12, 2^13] -- exact operations; what underflows below 2^-24 is 37 octaves under the largest value.
//   * What error is left only matters at the boundary of a row's list.  The lists hold K + ST_BF_MARGIN entries chosen by
//     the split distance, and the kernel's epilogue recomputes the EXACT float32 distance sum (x - y)^2 of everything kept
//     (from the original rows) and hands the best K on; it also counts the rows whose K-th exact distance lies within twice
//     the measured error of the list's last approximate entry -- if more than 1 row in 200 is flagged (neighbours closer
//     together than float32 products of |x|^2 resolve) the host repeats the tile phase on k_st_knn.  The reported
//     distances are exact float32 as before; the exactness tests (rtol 1e-5 against float64 brute force with the full
//     budget) hold unchanged.
//   * While a wave streams MFMAs back to back, the SIMD's other wave issues NOTHING (pingpong.hip: a partner's VALU or
//     LDS work beside an MFMA chain takes exactly chain + its own time, whatever s_setprio says).  A producer /
//     consumer split inside a SIMD therefore serialises; what overlaps is one wave's LATENCY (the LDS round trips of a
//     list merge, a barrier wait) with another wave's issue.  Hence two independent 4-wave workgroups per CU, one wave of
//     each on every SIMD, each wave doing everything for its 32 rows: stream, test, insert, merge.
//
// Per slab (32 columns) and wave: request the NEXT slab (LDS-DMA `global_load_lds_dwordx4`: no staging registers, no
// ds_write pass; two ring slots), 16 ds_read_b128 of operands, 24 MFMAs with the previous slab's threshold test in their
// shadow, survivor inserts, a cooperative list merge, wait for the request, one workgroup barrier.  The LDS image of a slab
// is lane-linear, so the bank-conflict-free layout is made on the SOURCE side: 16-byte unit kq of column c (units
// 0..DIM/8-1 the hi halves, then the lo halves) sits at unit kq ^ f(c) of the column's run and the operand reads apply the
// same XOR.
//
// Decisions that steer the stream (skip a ranked tile whose bound has fallen behind the thresholds, early stop, budget)
// are taken by every wave from barrier-separated LDS state, for the tile AFTER the one in the stream and from the
// thresholds as merged through the tile BEFORE it -- a fixed lag, so runs are reproducible.
#include "streamed.h"

#define STB_THREADS 256
#define ST_BF_MARGIN 2   // list entries beyond K kept by the split-fp16 distance (re-ranked exactly at the end)
#define ST3_CS 8         // three-slot form: candidate slots per row and slab
#define ST3_DIRECT 6     // three-slot form: up to this many survivors per wave and slab go straight into their lists

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ST_PROFILE builds: per-wave cycle sums by segment (a.prof[0..7], printed by knn_tile_phase):
//   0 MFMA stream (operand reads + MFMA issue)   1 barrier after the stream   2 choice of the next tile
//   3 threshold test + survivor inserts           4 LDS-DMA requests           5 barrier after the requests
//   6 merge (+ publish, run prologue / tail)      7 ranking, selection rounds, the rest
#ifdef ST_PROFILE
// (ordered: nothing may be scheduled across the time stamp, and it waits for the wave's outstanding LDS / scalar traffic)
__device__ __forceinline__ long long st8_now()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return (long long)t;
}
#define P8(i) { const long long pf_n = st8_now(); pf[i] += pf_n - pf_t; pf_t = pf_n; }
// sub-segment stamp: time since the last P8 / PS goes to slot i (8..15) without resetting the P8 clock
#define PS(i) { const long long pf_n = st8_now(); pf[i] += pf_n - pf_s; pf_s = pf_n; }
#define PS0 { pf_s = st8_now(); }
#else
#define P8(i)
#define PS(i)
#define PS0
#endif

// R3 (round 6): a ring of THREE slots, slabs requested two ahead, the slab barrier replaced by per-wave progress flags (the
// waves of a workgroup may drift a slab apart), the columns' norms by LDS-DMA beside the operands, eight candidate slots per
// row and slab (a row with more survivors -- the first tiles of a row tile -- merges in rounds).  See `run3` below.
template <int DIM, int KMAX, bool R3 = false> struct KnnSharedB {
    static constexpr int SLABF = ST_SLAB * DIM;                                 // floats per operand slab
    static constexpr int NSLOT = R3 ? 3 : 2;
    static constexpr int CS = R3 ? ST3_CS : ST_SLAB;                            // candidate slots per row and slab
    static constexpr int RINGF = NSLOT * SLABF * 4 >= 12288 ? NSLOT * SLABF : 12288 / 4;   // (>= sizeof(SelBuf))
    float ring[RINGF];   // FIRST (LDS-DMA destinations stay below 64 KB); slot = slab parity (R3: slab sequence number mod 3).
                         // Between runs the selection's sort buffers (SelBuf) live here (R3: and its histogram; at the end
                         // the exact distances)
    float norms[R3 ? 3 * 64 : 4];      // R3: squared norms of the columns of the slab in each slot (LDS-DMA destination)
    float cand_d[ST_T][CS + 1];        // (two-slot form -- between runs: the selection's 4096-bin histogram; at the end: exact distances)
    uint8_t cand_c[ST_T][R3 ? CS : ST_SLAB + 4];
    float list_d[ST_T][KMAX + 1];
    int32_t list_c[ST_T][KMAX + 1];
    float thr[ST_T];
    float hb[ST_T];      // (|x_row|^2 - thr[row]) / 2: column c passes the row's test iff x_row . x_c > hb[row] + |x_c|^2 / 2
    float rrow[ST_T];    // |x_row|^2
    int cnt[ST_T];
    float loI[64], hiI[64], midI[64];
    uint32_t slab_id[4][ST_SLAB];   // join passes: ordered column index of each column of a slab (slot = slab & 3)
    float run_vb[ST_KEEP];     // the current round's tiles in rank order: valid bound, tile
    int32_t run_j[ST_KEEP];
    float wave_thr[2][4];   // worst k-th squared distance per 32-row group, published at the end of tile n into [n & 1]
    int wave_ins[2][4];     // list insertions per wave (cumulative), likewise
    uint32_t run_ev[ST_KEEP / 32];            // R3: entries of the current run's tile list that were evaluated (thread 0 writes)
    alignas(16) int f_loaded[4];              // R3: per wave, the last slab (sequence number) whose pieces it requested have landed
    alignas(16) int f_done[4];                // R3: per wave, the last slab whose operands it has read
    int nsurv;
    int sel_bin;
    uint32_t sel_before;
};
struct SelBuf {   // candidate tiles of a selection round (aliases the operand ring, idle between runs)
    float surv_lb[ST_SURV];
    float surv_vb[ST_SURV];
    int32_t surv_j[ST_SURV];
};

// swizzle of a column's 16-byte units (see the header comment); UPC = units per column
template <int UPC> __device__ __forceinline__ int unit_swz(int col) { return UPC >= 16 ? (col & 15) : ((col >> 1) & (UPC - 1)); }

// End of a slab: the wave's LDS-DMA pieces of the next slab have landed and its LDS traffic is done; the barrier hands the
// next slab to every wave and this slab's ring slot back to the requests.  (The requests are invisible to the compiler's
// wait bookkeeping -- inline asm; sched_barrier: the memory clobber alone does not keep register-only instructions on
// their side.)
__device__ __forceinline__ void slab_end()
{
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// JOIN: a join pass (streamed.hip: k_st_join_cands has collected the row tile's candidate columns) -- the "tiles" are runs of
// 128 gathered columns of the candidate list, the lists start from the previous phase's, nothing is ranked or pruned.
template <int DIM, int KMAX, bool JOIN = false, bool R3 = false> __global__ __launch_bounds__(STB_THREADS, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_st_knnbf(KnnArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smemb[];
    using Sh = KnnSharedB<DIM, KMAX, R3>;
    Sh &sh = *reinterpret_cast<Sh *>(smemb);
    static_assert(!(JOIN && R3), "the join passes run the two-slot form");
    constexpr int UPC = DIM / 4;            // 16-byte units per column
    constexpr int NV = UPC / 2;             // operand reads (ds_read_b128) per slab and lane: G hi + G lo
    constexpr int NPIECE = UPC * ST_SLAB / 64;   // 1 KB pieces per slab
    constexpr int NI = NPIECE / 4;          // pieces per loading wave
    static_assert(NI == 1 || NI == 2 || NI == 4, "pieces per loading wave");
    static_assert(sizeof(sh.ring) <= 65536, "LDS-DMA destinations must stay below 64 KB");
    static_assert(sizeof(SelBuf) <= sizeof(sh.ring) && sizeof(SelBuf) == 12288, "selection buffers alias the ring");
    static_assert(R3 || KMAX <= ST_SLAB + 1, "the exact re-ranking reuses cand_d with row stride KMAX");
    static_assert(!R3 || (sizeof(sh.ring) >= 12288 + 16384 && sizeof(sh.ring) >= sizeof(float) * ST_T * KMAX), "R3: histogram and exact distances alias the ring");
    static_assert(!R3 || sizeof(sh.ring) + sizeof(sh.norms) <= 65536, "LDS-DMA destinations must stay below 64 KB");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rg = wave;   // this wave's 32-row group
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smemb;
    int bt;
    {   // XCD-banded row-tile assignment (block b runs on XCD b % 8): neighbours in the k-d order share an L2
        const int nb_ = gridDim.x, q = nb_ >> 3, r = nb_ & 7, x = blockIdx.x & 7, y = blockIdx.x >> 3;
        bt = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    }
    const int I = a.tile_begin + bt;
    const int64_t grow0 = (int64_t)I * ST_T;
    const int K = a.K;
    const int KL = min(KMAX, K + ST_BF_MARGIN);   // list entries kept by the split-fp16 distance
    const int col = lane & 31, half = lane >> 5;
    const int rowbase = rg * 32;
    const int rowq = rowbase + 4 * half;   // C layout: row = rowq + (r & 3) + 8 (r >> 2), col = lane & 31
    // ---- row operand in registers, split: lane holds row (lane & 31), dimensions 16 g + 8 half .. + 7 of k-step g as
    // eight fp16 hi parts and eight lo parts of the centred, scaled values
    constexpr int G = DIM / 16;
    f16x8 ah[G], al[G];
    const float scale = a.cvec[DIM];              // power of two: the centred data's largest |coordinate| becomes <= 2^13
    const float inv_scale2 = 1.f / (scale * scale);
    float rr_c;   // this lane's half of |scale (x_row - c)|^2 (summed with the other half below)
    {
        const float *xr = a.Rs + (size_t)(grow0 + rowbase + col) * DIM + 8 * half;
        const float *cv = a.cvec + 8 * half;
        float acc2 = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float4 t0 = *reinterpret_cast<const float4 *>(xr + 16 * g), t1 = *reinterpret_cast<const float4 *>(xr + 16 * g + 4);
            const float4 c0 = *reinterpret_cast<const float4 *>(cv + 16 * g), c1 = *reinterpret_cast<const float4 *>(cv + 16 * g + 4);
            const float xu[8] = {t0.x - c0.x, t0.y - c0.y, t0.z - c0.z, t0.w - c0.w, t1.x - c1.x, t1.y - c1.y, t1.z - c1.z, t1.w - c1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // (query rows may lie outside the data's range: clamped into fp16's -- their selection is then approximate,
                // the distances of what is selected stay exact)
                const float x = fminf(fmaxf(xu[j] * scale, -60000.f), 60000.f);
                const _Float16 h = (_Float16)x;
                ah[g][j] = h;
                al[g][j] = (_Float16)(x - (float)h);
                acc2 += x * x;
            }
        }
        rr_c = acc2 + __shfl_xor(acc2, 32);
    }
    if (threadIdx.x < ST_T) {
        const int row = threadIdx.x;
        const bool real = a.rr[grow0 + row] < INFINITY;
        sh.cnt[row] = 0;
        if constexpr (JOIN) {
            // the lists as the previous phase left them (exact d^2, original units -> scaled); the KL - K spare entries start
            // as copies of the K-th value without a column, so that the row's threshold is its current K-th distance
            // (all of a row's entries requested before the first is looked at: one at a time, 2 x K dependent global round trips opened
            // every workgroup of a pass)
            float dv[KMAX];
            int32_t cv[KMAX];
            const float *pd = a.out_d2 + ((size_t)bt * ST_T + row) * K;
            const int32_t *pc = a.lists_all + ((size_t)grow0 + row) * K;
#pragma unroll
            for (int q = 0; q < KMAX; ++q) { dv[q] = pd[min(q, K - 1)]; cv[q] = pc[min(q, K - 1)]; }
            float last = INFINITY;
#pragma unroll
            for (int q = 0; q < KMAX; ++q) {
                const bool have = q < K;
                const float d = have ? dv[q] * (scale * scale) : last;
                sh.list_d[row][q] = q < KL ? d : INFINITY;
                sh.list_c[row][q] = have ? cv[q] : 0x7fffffff;
                if (have) last = d;
            }
            sh.thr[row] = real ? last : -1.f;
        } else {
            sh.thr[row] = real ? INFINITY : -1.f;   // padding rows never accept candidates
            for (int q = 0; q < KMAX; ++q) { sh.list_d[row][q] = INFINITY; sh.list_c[row][q] = 0x7fffffff; }
        }
    }
    if (lane < 32) {
        const bool real = a.rr[grow0 + rowbase + lane] < INFINITY;
        sh.rrow[rowbase + lane] = real ? rr_c : INFINITY;   // |scale (x_row - c)|^2
        // (hb = (rrow - thr) / 2; thr of this row was written by thread rowbase + lane above: another wave -> after the barrier)
    }
    if ((int)threadIdx.x < a.na) {
        sh.loI[threadIdx.x] = a.rlo[(size_t)threadIdx.x * a.nt_r + I];
        sh.hiI[threadIdx.x] = a.rhi[(size_t)threadIdx.x * a.nt_r + I];
        sh.midI[threadIdx.x] = a.rmid[(size_t)threadIdx.x * a.nt_r + I];
    }
    if (threadIdx.x < 8) { sh.wave_ins[threadIdx.x >> 2][threadIdx.x & 3] = 0; sh.wave_thr[threadIdx.x >> 2][threadIdx.x & 3] = INFINITY; }
    if (threadIdx.x == 0) sh.nsurv = 0;
    if (R3 && threadIdx.x < 8) (&sh.f_loaded[0])[threadIdx.x] = -1;   // (f_loaded[4], f_done[4]: nothing requested, nothing read)
    int ins = 0;         // list insertions counted by this lane (the first lane of a merge group)
    int processed = 0;   // column tiles scheduled so far (uniform)
    int tdone = 0;       // column tiles completed and published (uniform); tile n publishes into slot n & 1
    int win_start = 0, win_ins = 0;
    bool dried = false;
    uint32_t *ebits = (!JOIN && a.eval_bits) ? a.eval_bits + (size_t)bt * a.eval_words : nullptr;   // (a join pass only reads them)
#ifdef ST_PROFILE
    long long pf[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long pf_t = st8_now();
    long long pf_s = pf_t;
#endif
    __syncthreads();
    if (threadIdx.x < ST_T) {
        const int row = threadIdx.x;
        const float t = sh.thr[row];
        sh.hb[row] = t < 0.f ? INFINITY : (t < INFINITY ? 0.5f * (sh.rrow[row] - t) : -INFINITY);
    }
    __syncthreads();

    // ---------------------------------------------------------------- the pieces of a phase
    // operand slab `slab` of column tile J -> ring slot `slab`: this wave's NI pieces
    // (the lane -> (column, unit) map of a piece never changes: byte offsets inside a slab's 32 rows, once)
    uint32_t loff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int u = (rg * NI + i) * 64 + lane;
        const int c = u / UPC, x = u % UPC;
        loff[i] = (uint32_t)(c * DIM * 4 + ((x ^ unit_swz<UPC>(c)) << 4));
    }
    const char *xb = reinterpret_cast<const char *>(a.Xb);   // [n_all][2][DIM] fp16: hi parts, then lo parts of every ordered row (centred, scaled)
    const uint32_t *ulist = JOIN ? a.ucand + (size_t)bt * a.ucap : nullptr;   // join: sorted candidate columns, 0xffffffff padded to 128
    uint32_t js_id = 0;       // join passes: the lane's column id in the slab after the one being streamed, and that slab's first column
    int64_t js_c0 = -1;
    uint32_t jn_id[NI];       // join passes: the column ids behind the wave's pieces of the slab after the one last requested
    int64_t jn_c0 = -1;       // ... and that slab's first column (-1: none requested)
    auto issue_slab = [&](int J, int slab) {
        if constexpr (JOIN) {
            // gathered columns: every lane's source is its own column's row (per-lane 64-bit addresses).  The columns' ids were
            // requested while the slab before this one was requested (the slabs of a pass come in a fixed order): read here, they
            // were an L2 round trip in front of every slab's requests
            const uint32_t dst = lds0 + (uint32_t)((slab & 1) * ST_SLAB * DIM * 4 + rg * NI * 1024);
            const int64_t c0 = (int64_t)J * ST_T + slab * ST_SLAB;
            const char *srcs[NI];
            const bool have_ids = jn_c0 == c0;   // (uniform)
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int u = (rg * NI + i) * 64 + lane;
                const uint32_t id = have_ids ? jn_id[i] : ulist[c0 + u / UPC];
                srcs[i] = xb + (size_t)(id == 0xffffffffu ? 0u : id) * (DIM * 4) + (loff[i] - (uint32_t)((u / UPC) * DIM * 4));
            }
            unsigned keep;
#pragma unroll
            for (int i = 0; i < NI; ++i)
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(srcs[i]), "s"(dst + (uint32_t)(i * 1024)) : "memory");
            // the ids of the slab that follows in the pass's order (the next slab of the tile, or slab 0 of the next run of columns)
            const int64_t n0 = c0 + ST_SLAB;
            jn_c0 = -1;
            if (n0 + ST_SLAB <= (int64_t)a.ucap) {
                jn_c0 = n0;
#pragma unroll
                for (int i = 0; i < NI; ++i) jn_id[i] = ulist[n0 + ((rg * NI + i) * 64 + lane) / UPC];
            }
            return;
        }
        const char *src = xb + ((size_t)J * ST_T + slab * ST_SLAB) * (DIM * 4);              // wave-uniform: an SGPR pair
        const uint32_t dst = lds0 + (uint32_t)((slab & 1) * ST_SLAB * DIM * 4 + rg * NI * 1024);   // this wave's pieces of the slot
        unsigned keep;
        if constexpr (NI == 4)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
                         "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
                         "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
                         "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(loff[0]), "v"(loff[1]), "v"(loff[2]), "v"(loff[3]), "s"(src), "s"(dst) : "memory", "scc");
        else if constexpr (NI == 2)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                         "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(loff[0]), "v"(loff[1]), "s"(src), "s"(dst) : "memory", "scc");
        else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(loff[0]), "s"(src), "s"(dst) : "memory");
    };
    int cur_slot = 0;           // R3: ring slot of the slab being streamed (uniform)
    float hqr[16];              // R3: hb of the lane's 16 rows, kept across slabs
#pragma unroll
    for (int r = 0; r < 16; ++r) hqr[r] = 0.f;
    bool hq_stale = true;       // R3: a merge of this wave has changed hb since hqr was read (uniform)
    unsigned long long pany = 0;   // R3: OR over the rows of the pending slab's test masks (lanes = columns with a survivor)
    int seq = 0;                // R3: sequence number of the slab being streamed (uniform; monotonic over the kernel)
    const uint32_t flag_addr = lds0 + (uint32_t)((const unsigned char *)&sh.f_loaded[0] - smemb);
    auto set_flag = [&](int which, int v) {   // R3 -- which: 0 = f_loaded[wave], 1 = f_done[wave]
        if (lane == 0) asm volatile("ds_write_b32 %0, %1" : : "v"(flag_addr + (uint32_t)(which * 16 + wave * 4)), "v"(v) : "memory");
    };
    f32x16 acc0, acc1;          // slabs 0, 2 / slabs 1, 3 of a tile: one is streamed into while the other one is tested
    float rj_c = 0.f;           // squared norm of the lane's column in the slab being streamed (requested at its start)
    // the slab whose accumulators wait for their test (the one streamed before the current one)
    bool pend = false;
    int pJ = 0, pslab = 0;
    float prj = 0.f;
    uint32_t ppass = 0;
    // 3 DIM / 16 MFMAs of this wave's 32 rows against the 32 columns in ring slot `slab` (columns of tile J) into accC, with
    // the threshold test of the PREVIOUS slab's accumulators accP in their shadow (the 16-bit matrix pipe takes an MFMA
    // every 32 cycles and up to four vector instructions behind each one for free, tools/microbench/shadow.hip): row r of
    // the lane's column passes iff x_r . x_c > hb[r] + |x_c|^2 / 2 (hb: one LDS word per row, kept by the merge).  The
    // columns' squared norms are requested at the start of their slab and used one slab later: a global round trip under
    // load is thousands of cycles.
    auto stream_slab = [&](int J, int slab, f32x16 &accC, const f32x16 &accP) __attribute__((always_inline)) {
        PS0
        if constexpr (JOIN) {
            // (the lane's column id: requested while the slab before was streamed -- id -> norm is a chain of two global reads, and
            // the LDS write of the id waited out the first of them at the head of every slab: 27 % of the pass's wave cycles)
            const int64_t s0 = (int64_t)J * ST_T + slab * ST_SLAB;
            const uint32_t id = js_c0 == s0 ? js_id : ulist[s0 + col];
            rj_c = id == 0xffffffffu ? INFINITY : a.rsb[id];   // (padding never passes: x.y > hb + inf is false)
            if (wave == 0 && lane < ST_SLAB) sh.slab_id[slab & 3][lane] = id;
            js_c0 = -1;
            if (s0 + 2 * ST_SLAB <= (int64_t)a.ucap) { js_c0 = s0 + ST_SLAB; js_id = ulist[js_c0 + col]; }
        } else if constexpr (R3) {
            rj_c = sh.norms[cur_slot * 64 + col];   // (came with the operands)
        } else {
            rj_c = a.rsb[(int64_t)J * ST_T + slab * ST_SLAB + col];
        }
        const float4 *base = reinterpret_cast<const float4 *>(&sh.ring[(R3 ? cur_slot : (slab & 1)) * (ST_SLAB * DIM)]) + col * UPC;
        const int gsw = half ^ unit_swz<UPC>(col);
        float hq[16];   // (first: LDS data returns in order, and the test must not wait for the operands behind it)
        if constexpr (R3) {
            // the rows' thresholds stay in registers from slab to slab; read again after a merge of this wave changed them
            if (hq_stale) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 h4 = *reinterpret_cast<const float4 *>(&sh.hb[rowq + 8 * q]);
                    hqr[4 * q] = h4.x; hqr[4 * q + 1] = h4.y; hqr[4 * q + 2] = h4.z; hqr[4 * q + 3] = h4.w;
                }
                hq_stale = false;
            }
        } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 h4 = *reinterpret_cast<const float4 *>(&sh.hb[rowq + 8 * q]);
            hq[4 * q] = h4.x; hq[4 * q + 1] = h4.y; hq[4 * q + 2] = h4.z; hq[4 * q + 3] = h4.w;
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        }
        float4 b[NV];   // b[g]: hi parts of k-step g; b[G + g]: lo parts
#pragma unroll
        for (int v = 0; v < NV; ++v) b[v] = base[(2 * v) ^ gsw];
        // all LDS reads first, then MFMAs with the test's vector instructions between them (left alone the scheduler sinks
        // each read to its use and the stream waits out an LDS round trip every few MFMAs)
        __builtin_amdgcn_sched_group_barrier(0x100, NV, 0);
        PS(8)    // operand reads landed (the stamp waits for them)
        constexpr int NM = 3 * G;                       // MFMAs
        constexpr int TPM = (16 + NM - 1) / NM;         // row tests per MFMA
        if constexpr (R3) {
            // R3: the accumulators start from -|x_c|^2 / 2 (the column's norm came with the operands), so the row test is ONE
            // compare per row, x_r . x_c - |x_c|^2 / 2 > hb[r], written to a wave mask (v_cmp into an SGPR pair) of which only
            // the OR over the rows is kept: 16 vector instructions per slab in the MFMAs' shadow instead of 64 (add, compare,
            // select, or -- the two waves of a SIMD take turns at the issue port, and beside an MFMA stream its own wave's
            // vector instructions are the only ones that issue: instruction count is what the tile phase's time is made of).
            // The rare slab with a survivor rebuilds the per-lane row masks in insert_merge.
            const float nh = -0.5f * rj_c;
#pragma unroll
            for (int r = 0; r < 16; ++r) accC[r] = nh;
            unsigned long long any = 0;
#pragma unroll
            for (int g = 0; g < G; ++g) {   // small terms first
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const int m = 3 * g + t;
                    if (t == 0) accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[g], __builtin_bit_cast(f16x8, b[g]), accC, 0, 0, 0);
                    if (t == 1) accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[g], __builtin_bit_cast(f16x8, b[G + g]), accC, 0, 0, 0);
                    if (t == 2) accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[g], __builtin_bit_cast(f16x8, b[g]), accC, 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < TPM; ++u) {
                        const int r = m * TPM + u;
                        if (r < 16) any |= __ballot(accP[r] > hqr[r]);
                    }
                    asm volatile("" : "+v"(accC));
                    // every operand read of the slab has been issued by now (the write below is a memory barrier to the compiler,
                    // and LDS serves a wave's instructions in order): the slot may be requested again once every wave says so
                    if (m == 2) set_flag(1, seq);
                }
            }
            pany = pend ? any : 0ull;
            return;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) accC[r] = 0.f;
        const float hrj = 0.5f * prj;
        uint32_t pass = 0;
#pragma unroll
        for (int g = 0; g < G; ++g) {   // small terms first
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int m = 3 * g + t;
                if (t == 0) accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[g], __builtin_bit_cast(f16x8, b[g]), accC, 0, 0, 0);
                if (t == 1) accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[g], __builtin_bit_cast(f16x8, b[G + g]), accC, 0, 0, 0);
                if (t == 2) accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[g], __builtin_bit_cast(f16x8, b[g]), accC, 0, 0, 0);
#pragma unroll
                for (int u = 0; u < TPM; ++u) {
                    const int r = m * TPM + u;
                    if (r < 16) pass |= (accP[r] > hq[r] + hrj ? 1u : 0u) << r;
                }
                // (pinned by data flow: the tests are pure arithmetic and every scheduling hint -- sched_group_barrier,
                // sched_barrier -- left them all behind the last MFMA; an empty asm that "uses" the accumulator and the
                // mask keeps MFMA m and test m on this side of it)
                asm volatile("" : "+v"(accC), "+v"(pass));
            }
        }
        ppass = pend ? pass : 0u;
        PS(9)    // MFMAs issued
    };
    // what the shadow test let through (slab pslab of tile pJ, accumulators accP, column norms prj): survivors into the rows'
    // candidate slots, then the merge into the sorted lists
    auto insert_merge = [&](const f32x16 &accP) __attribute__((always_inline)) {
        const int J = pJ, slab = pslab;
        const float rj = prj;
        uint32_t pass = ppass;
        if constexpr (R3) {
            if (!pany) return;   // (uniform) no row of this wave let a column of the slab through: nothing to insert, nothing to merge
            hq_stale = true;
            // A warm row tile lets one or two columns per wave and slab through (a row's 16-entry list takes ~100 insertions over
            // ~66 000 streamed columns): the candidate slots, their counters and the batched merge below are machinery for the
            // first tiles, when every column passes.  Few survivors go straight into their lists, one at a time and wave-uniform:
            // the survivor's row test mask names lane and register, its accumulator comes by v_readlane, lanes e = 0 .. 15 hold the
            // row's list, the position is a popcount of the comparison ballot, the tail moves up by a DPP row shift.
            // (how many: the columns with a survivor -- a column that passes for several rows counts once)
            const bool self_t = !a.query && (int64_t)J * ST_T == grow0;
            if (!self_t && __popcll(pany) <= ST3_DIRECT) {
                static_assert(!R3 || KMAX == 16, "the direct insert keeps a list in one 16-lane row");
                const int32_t col0 = (int32_t)(J * ST_T + slab * ST_SLAB);
                const int e = lane & 15;
                uint32_t lp = 0;   // the lane's rows with a survivor
#pragma unroll
                for (int r = 0; r < 16; ++r) lp |= (accP[r] > hqr[r] ? 1u : 0u) << r;
                unsigned long long lanes = __ballot(lp != 0);
                while (lanes) {   // (uniform) lanes = columns with a survivor
                    const int l = (int)__builtin_ctzll(lanes);
                    lanes &= lanes - 1;
                    uint32_t pl = (uint32_t)__builtin_amdgcn_readlane((int)lp, l);
                    while (pl) {   // (uniform) that column's rows
                        const int r = __builtin_ctz(pl);
                        pl &= pl - 1;
                        const float av = accP[r];   // (r is uniform: a relative register read, not a select chain)
                        const float dv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, av), l));
                        const int rowl = rowbase + 4 * (l >> 5) + (r & 3) + 8 * (r >> 2);
                        const int32_t cc = col0 + (l & 31);
                        const float rr = sh.rrow[rowl];
                        const float ld = e < KL ? sh.list_d[rowl][e] : INFINITY;
                        const int32_t lc = e < KL ? sh.list_c[rowl][e] : 0x7fffffff;
                        const float d2 = fmaxf(rr - 2.f * dv, 0.f);
                        const bool before = e < KL && (ld < d2 || (ld == d2 && lc < cc));
                        const int pos = __popcll(__ballot(before) & 0xffffull);
                        if (pos < KL) {   // (uniform)
                            const float pd = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ld), 0x111, 0xf, 0xf, false));   // row_shr:1
                            const int32_t pc = __builtin_amdgcn_update_dpp(0, lc, 0x111, 0xf, 0xf, false);
                            const float nd = e > pos ? pd : (e == pos ? d2 : ld);
                            const int32_t nc = e > pos ? pc : (e == pos ? cc : lc);
                            if (lane < 16 && e >= pos && e < KL) { sh.list_d[rowl][e] = nd; sh.list_c[rowl][e] = nc; }
                            if (lane == KL - 1) { sh.thr[rowl] = nd; sh.hb[rowl] = 0.5f * (rr - nd); }
                            ins += (lane == 0 && pos < K) ? 1 : 0;
                        }
                    }
                }
                wave_fence_lds();
                return;
            }
            pass = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) pass |= (accP[r] > hqr[r] ? 1u : 0u) << r;
        }
        PS0
        const bool self_tile = !JOIN && !a.query && (int64_t)J * ST_T == grow0;
        PS(10)   // thresholds read, accumulators there, 16 tests
        if (pass) {
            if (self_tile) {   // a point is not its own neighbour
                const int dcol = slab * ST_SLAB + col - rowq;
                if (dcol >= 0 && dcol < 32 && (dcol & 4) == 0) pass &= ~(1u << ((dcol & 3) + 4 * (dcol >> 3)));
            }
        }
        // (R3: Sh::CS = 8 candidate slots per row; a survivor that finds them taken waits for the next round of {insert, merge} --
        // the first tiles of a row tile, when every column still passes)
        uint32_t defer;
        do {
        defer = 0;
        {
            while (pass) {
                const int g = __builtin_ctz(pass);
                pass &= pass - 1;
                const int rowl = rowq + (g & 3) + 8 * (g >> 2);
                float ag = accP[0];
#pragma unroll
                for (int t = 1; t < 16; ++t) ag = g == t ? accP[t] : ag;
                const float d2 = R3 ? fmaxf(sh.rrow[rowl] - 2.f * ag, 0.f)    // (R3: ag = x_r . x_c - |x_c|^2 / 2)
                                    : fmaxf(sh.rrow[rowl] + rj - 2.f * ag, 0.f);
                const int slot = atomicAdd(&sh.cnt[rowl], 1);
                if (!R3 || slot < Sh::CS) {
                    sh.cand_d[rowl][slot] = d2;
                    sh.cand_c[rowl][slot] = (uint8_t)col;
                } else {
                    defer |= 1u << g;
                }
            }
        }
        PS(11)   // survivor inserts
        wave_fence_lds();
        P8(3)
        // ---- merge, cooperatively: a group of GL lanes (GL = KMAX: 16 or 32) holds one row's sorted list in registers,
        // one entry per lane; a candidate's position is a popcount over the group's comparison ballot and the entries behind
        // it move up by one lane (a DPP row shift for 16-lane groups).  64 / GL rows at a time -- the lane-per-row form
        // walked every list through dependent LDS round trips (a fifth of the kernel).
        {
            constexpr int GL = KMAX;                  // lanes per row
            constexpr int NG = 64 / GL;               // rows per batch
            const int grpi = lane / GL, e = lane % GL;
            const int mycnt = lane < 32 ? sh.cnt[rowbase + lane] : 0;
            unsigned long long todo = __ballot(mycnt > 0);
            const int32_t col0 = (int32_t)(J * ST_T + slab * ST_SLAB);
            while (todo) {
                int rsel = -1;   // this group's row (relative to rowbase)
#pragma unroll
                for (int k = 0; k < NG; ++k) {
                    const int r = todo ? (int)__builtin_ctzll(todo) : -1;
                    if (todo) todo &= todo - 1;
                    rsel = grpi == k ? r : rsel;
                }
                const int row = rowbase + max(rsel, 0);
                const bool live = rsel >= 0;
                const int nc = live ? (R3 ? min(sh.cnt[row], Sh::CS) : sh.cnt[row]) : 0;
                float ld = (live && e < KL) ? sh.list_d[row][e] : INFINITY;
                int32_t lc = (live && e < KL) ? sh.list_c[row][e] : 0x7fffffff;
                int q = 0;
                while (__ballot(q < nc)) {
                    const bool on = q < nc;
                    const float d = on ? sh.cand_d[row][q] : INFINITY;
                    int32_t cc;
                    if constexpr (JOIN) cc = on ? (int32_t)sh.slab_id[slab & 3][sh.cand_c[row][q]] : 0x7fffffff;
                    else cc = on ? col0 + sh.cand_c[row][q] : 0x7fffffff;
                    const bool before = e < KL && (ld < d || (ld == d && lc < cc));   // entries that stay ahead of the candidate
                    const unsigned long long bb = __ballot(before);
                    const unsigned long long gmask = (GL == 64) ? ~0ull : ((1ull << GL) - 1);
                    const int pos = __popcll((bb >> (grpi * GL)) & gmask);
                    bool skip = false;
                    if constexpr (JOIN) {
                        // gathered columns: the row itself and columns the row already lists may come by
                        const unsigned long long dup = __ballot(e < KL && lc == cc);
                        skip = ((dup >> (grpi * GL)) & gmask) != 0 || (int64_t)cc == grow0 + row;
                    }
                    float pd;
                    int32_t pc;
                    if constexpr (GL == 16) {
                        pd = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ld), 0x111, 0xf, 0xf, false));   // row_shr:1
                        pc = __builtin_amdgcn_update_dpp(0, lc, 0x111, 0xf, 0xf, false);
                    } else {
                        pd = __shfl_up(ld, 1, GL);
                        pc = __shfl_up(lc, 1, GL);
                    }
                    if (on && !skip && pos < KL) {
                        ld = e > pos ? pd : (e == pos ? d : ld);
                        lc = e > pos ? pc : (e == pos ? cc : lc);
                        ins += (e == 0 && pos < K) ? 1 : 0;   // the yield that stops the tile phase counts what reaches the K entries handed on
                    }
                    ++q;
                }
                if (live) {
                    if (e < KL) { sh.list_d[row][e] = ld; sh.list_c[row][e] = lc; }
                    if (e == KL - 1) { sh.thr[row] = ld; sh.hb[row] = 0.5f * (sh.rrow[row] - ld); }
                    if (e == 0) sh.cnt[row] = 0;
                }
            }
        }
        wave_fence_lds();
        pass = defer;
        } while (R3 && __ballot(defer != 0));
        P8(6)
    };
    // the pending slab's test without a stream to hide it in (end of a run)
    auto test_only = [&](const f32x16 &accP) __attribute__((always_inline)) {
        uint32_t pass = 0;
        const float hrj = 0.5f * prj;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 h4 = *reinterpret_cast<const float4 *>(&sh.hb[rowq + 8 * q]);
            const float hv[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) pass |= (accP[4 * q + e] > hv[e] + (R3 ? 0.f : hrj) ? 1u : 0u) << (4 * q + e);
            if constexpr (R3) { hqr[4 * q] = hv[0]; hqr[4 * q + 1] = hv[1]; hqr[4 * q + 2] = hv[2]; hqr[4 * q + 3] = hv[3]; }
        }
        ppass = pend ? pass : 0u;
        if constexpr (R3) pany = pend ? __ballot(pass != 0) : 0ull;   // (insert_merge rebuilds the same row masks from hqr)
    };
    // the wave's insertion count and its rows' worst k-th distance, at the end of a tile
    auto publish = [&]() __attribute__((always_inline)) {
        int wins = ins;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) wins += __shfl_xor(wins, off);
        float t = lane < 32 ? sh.thr[rowbase + lane] : -1.f;   // padding rows: -1
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) t = fmaxf(t, __shfl_xor(t, off));
        if (lane == 0) { sh.wave_ins[(tdone + 1) & 1][wave] = wins; sh.wave_thr[(tdone + 1) & 1][rg] = t * inv_scale2; }   // (the lists are in scaled units, the tile bounds are not)
    };
    // (as of the last tile whose publication a barrier separates from the reader: tile `tdone`)
    auto thrmax_now = [&]() {
        const float *w = sh.wave_thr[tdone & 1];
        return fmaxf(fmaxf(w[0], w[1]), fmaxf(w[2], w[3]));
    };

    // One run of the stream over a list of tiles in rank order: list entry q is (tile `jl(q)`, valid bound `vb(q)`);
    // entries whose bound has fallen behind the thresholds are skipped.  Uniform: every wave takes the same path.
    auto run = [&](int ns, auto jl, auto vb) {
        int q = 0;
        auto next_tile = [&](int in_stream) -> int {
            if (!JOIN && a.early_window > 0 && !dried) {
                const int done = processed - in_stream;   // tiles completed (the one in the stream is not)
                if (done - win_start >= a.early_window) {
                    int cur = 0;
#pragma unroll
                    for (int w = 0; w < 4; ++w) cur += sh.wave_ins[tdone & 1][w];
                    if (cur - win_ins < a.early_tau) dried = true;
                    else { win_start = done; win_ins = cur; }
                }
            }
            if (dried) return -1;
            const float tm = thrmax_now();
            while (q < ns && (JOIN || processed < a.max_tiles)) {
                const int J = jl(q);
                const float lb = vb(q);
                ++q;
                if (lb * lb < tm) {
                    ++processed;
                    if (ebits && threadIdx.x == 0) atomicOr(&ebits[J >> 5], 1u << (J & 31));   // (no return value: nothing to wait for)
                    return J;
                }
            }
            return -1;
        };
        P8(7)
        int J = next_tile(0);
        if (J < 0) return;
        // fill: slab 0 of the first tile
        issue_slab(J, 0);
        slab_end();
        P8(6)
        pend = false;
        // one slab: request the next one, stream (the previous slab's test in the shadow), insert / merge the previous slab
        // (the column norms requested at the start of a slab have landed by its end -- slab_end waits for everything -- but the
        // compiler does not know: without this it parks its own wait, and the whole shadow test behind it, after the MFMAs)
#define RJ_LANDED asm volatile("" : "+v"(prj));
        auto step = [&](int Jc, int sl, f32x16 &accC, const f32x16 &accP) {
            stream_slab(Jc, sl, accC, accP);
            P8(0)
            if (pend) insert_merge(accP);
            pend = true; pJ = Jc; pslab = sl; prj = rj_c;
        };
        for (;;) {
            int Jn = -1;
            issue_slab(J, 1);
            P8(4)
            step(J, 0, acc0, acc1);
#ifdef ST_PROFILE
            PS0
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            PS(12)   // own requests landed
            slab_end();
            PS(13)   // the other waves arrived
#else
            slab_end();
#endif
            RJ_LANDED
            P8(1)
            issue_slab(J, 2);
            P8(4)
            step(J, 1, acc1, acc0);
            slab_end();
            RJ_LANDED
            P8(1)
            issue_slab(J, 3);
            P8(4)
            step(J, 2, acc0, acc1);
            slab_end();
            RJ_LANDED
            P8(1)
            // ---- slab 3: the next tile is chosen (thresholds / insertion counts as of the tile before J: published before the
            // last barrier of that tile and untouched since -- the same choice in every wave) and its slab 0 requested
            Jn = next_tile(1);
            P8(2)
            if (Jn >= 0) issue_slab(Jn, 0);
            P8(4)
            step(J, 3, acc1, acc0);
            publish();     // (the lists as merged through slab 2 of this tile; slab 3's survivors go in during the next slab)
            ++tdone;
            slab_end();
            RJ_LANDED
            P8(1)
            if (Jn < 0) break;
            J = Jn;
        }
        // ---- tail: the last slab's test and merge, and the thresholds the selection of the next round reads
#undef RJ_LANDED
        test_only(acc1);
        insert_merge(acc1);
        pend = false;
        publish();
        ++tdone;
        slab_end();
        P8(6)
    };

    // ---------------------------------------------------------------- R3: the stream without workgroup barriers
    // Slabs carry a sequence number `seq` (monotonic over the kernel); slab s lives in ring slot s % 3 and is requested two
    // slabs ahead.  Per wave and slab s:   its own pieces of slab s + 1 (requested a whole slab ago) have landed: f_loaded[wave]
    // = s + 1  ->  wait until every wave has READ slab s - 1 (its slot is the one slab s + 2 goes to) and LANDED its pieces
    // of slab s  ->  request slab s + 2  ->  operand reads + MFMAs of slab s (slab s - 1's test in their shadow; f_done[wave]
    // = s as soon as the reads are issued)  ->  inserts / merge of slab s - 1.  What a wave waits for is that the others have
    // issued their reads of the slab BEFORE: the MFMAs, the test and the merge of a slab -- two thirds of it -- are slack, where
    // the two-slot form met at a barrier after every slab (27 % of its wave cycles).  The only vector-memory operations inside a run are the
    // LDS-DMA requests (the columns' norms are one more request per slab; the evaluated-tile bits are kept in LDS
    // and flushed at the end of the run), so vmcnt counts exactly them.  The tile after J is chosen at J's slab 2 (its slab 0
    // is requested there) from the state published at the end of the tile before J, as in the two-slot form: f_done >= s - 1
    // implies every wave has finished that tile's last slab, publication included (LDS serves a wave's instructions in order).
    constexpr int OPS = NI + 1;
    const uint32_t norm_addr = lds0 + (uint32_t)((const unsigned char *)&sh.norms[0] - smemb);
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    auto issue3 = [&](int Jv, int slab, int sq) __attribute__((always_inline)) {
        // (uniform values; under register pressure the compiler keeps some of them in vector registers, and the requests want their
        // source in a scalar register pair)
        auto scalar_ptr = [](const void *ptr) -> const char * {
            const uint64_t v = (uint64_t)(uintptr_t)ptr;
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
            return reinterpret_cast<const char *>((uintptr_t)(((uint64_t)hi << 32) | lo));
        };
        const int J = __builtin_amdgcn_readfirstlane(Jv);
        const int slot = __builtin_amdgcn_readfirstlane(sq % 3);
        const char *src = scalar_ptr(xb + ((size_t)J * ST_T + slab * ST_SLAB) * (DIM * 4));
        const uint32_t dst = lds0 + (uint32_t)(slot * ST_SLAB * DIM * 4 + rg * NI * 1024);      // this wave's pieces of the slot
        unsigned keep;
        if constexpr (NI == 4)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
                         "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
                         "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
                         "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(loff[0]), "v"(loff[1]), "v"(loff[2]), "v"(loff[3]), "s"(src), "s"(dst) : "memory", "scc");
        else if constexpr (NI == 2)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                         "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(loff[0]), "v"(loff[1]), "s"(src), "s"(dst) : "memory", "scc");
        else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(loff[0]), "s"(src), "s"(dst) : "memory");
        // the 32 columns' squared norms: every wave requests them (the same 128 bytes to the same place: one request more per
        // wave and slab keeps vmcnt the same for all of them; lanes 32 .. 63 repeat lanes 0 .. 31)
        const char *nsrc = scalar_ptr(a.rsb + (size_t)J * ST_T + slab * ST_SLAB);
        const uint32_t noff = (uint32_t)((lane & 31) * 4);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(noff), "s"(nsrc), "s"(norm_addr + (uint32_t)(slot * 256)) : "memory");
    };
    auto wait_flags = [&](int need) __attribute__((always_inline)) {   // every wave has read slab need - 1 and its pieces of slab need have landed
        for (;;) {
            i32x4 L, D;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(L), "=&v"(D) : "v"(flag_addr) : "memory");
            const int ml = min(min(L[0], L[1]), min(L[2], L[3])), md = min(min(D[0], D[1]), min(D[2], D[3]));
            if (__builtin_amdgcn_readfirstlane((ml >= need && md >= need - 1) ? 1 : 0)) break;
            __builtin_amdgcn_s_sleep(1);
        }
    };
    auto run3 = [&](int ns, auto jl, auto vb) __attribute__((always_inline)) {
        int q = 0;
        if (threadIdx.x == 0)
            for (int t = 0; t < ST_KEEP / 32; ++t) sh.run_ev[t] = 0;
        auto next_tile = [&](int in_stream) -> int {
            if (a.early_window > 0 && !dried) {
                const int done = processed - in_stream;   // tiles completed (the one in the stream is not)
                if (done - win_start >= a.early_window) {
                    int cur = 0;
#pragma unroll
                    for (int w = 0; w < 4; ++w) cur += sh.wave_ins[tdone & 1][w];
                    if (cur - win_ins < a.early_tau) dried = true;
                    else { win_start = done; win_ins = cur; }
                }
            }
            if (dried) return -1;
            const float tm = thrmax_now();
            while (q < ns && processed < a.max_tiles) {
                const int J = jl(q);
                const float lb = vb(q);
                ++q;
                if (lb * lb < tm) {
                    ++processed;
                    if (ebits && threadIdx.x == 0) sh.run_ev[(q - 1) >> 5] |= 1u << ((q - 1) & 31);   // (flushed at the end of the run)
                    return __builtin_amdgcn_readfirstlane(J);   // (uniform, and the requests want it in a scalar register)
                }
            }
            return -1;
        };
        int J = next_tile(0);
        if (J < 0) return;
        // fill: slabs 0 and 1 of the first tile (the ring is idle: a workgroup barrier precedes every run)
        issue3(J, 0, seq);
        issue3(J, 1, seq + 1);
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"(OPS) : "memory");
        set_flag(0, seq);
        pend = false;
        // one slab of the stream: (Ji, si) = the slab to request (two ahead; Ji < 0: the run ends before it), have_next: slab
        // seq + 1 exists (it was requested one slab ago)
        auto top3 = [&](bool have_next) __attribute__((always_inline)) {
            // the wave's own pieces of slab seq + 1 were requested a whole slab ago: landed (nothing else is outstanding)
            asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
            if (have_next) set_flag(0, seq + 1);
            wait_flags(seq);
        };
        auto slab3 = [&](int Jc, int sl, f32x16 &accC, const f32x16 &accP, int Ji, int si) __attribute__((always_inline)) {
            if (Ji >= 0) issue3(Ji, si, seq + 2);
            cur_slot = seq % 3;
            stream_slab(Jc, sl, accC, accP);   // (f_done[wave] = seq behind its third MFMA)
            if (pend) insert_merge(accP);
            pend = true; pJ = Jc; pslab = sl; prj = rj_c;
            if (sl == 3) { publish(); ++tdone; }   // (the lists as merged through slab 2 of this tile)
            ++seq;
        };
        for (;;) {
            top3(true);
            slab3(J, 0, acc0, acc1, J, 2);
            top3(true);
            slab3(J, 1, acc1, acc0, J, 3);
            top3(true);
            const int Jn = next_tile(1);
            slab3(J, 2, acc0, acc1, Jn, 0);
            top3(Jn >= 0);
            slab3(J, 3, acc1, acc0, Jn, 1);
            if (Jn < 0) break;
            J = Jn;
        }
        // ---- tail: the last slab's test and merge, and the thresholds the selection of the next round reads
        test_only(acc1);
        insert_merge(acc1);
        pend = false;
        publish();
        ++tdone;
        __syncthreads();
        if (ebits)
            for (int t = threadIdx.x; t < ns; t += STB_THREADS)
                if ((sh.run_ev[t >> 5] >> (t & 31)) & 1u) {
                    const int Jt = jl(t);
                    atomicOr(&ebits[Jt >> 5], 1u << (Jt & 31));
                }
    };
    auto run_tiles = [&](int ns, auto jl, auto vb) {
        if constexpr (R3) run3(ns, jl, vb);
        else run(ns, jl, vb);
    };

    // ---- phase A: the row tile against itself (gives every row K finite candidates); query rows are not part of the
    // data set and start from the ranked tiles directly
    if constexpr (JOIN) {
        const int nu = a.ucount[bt];
        const int nchunks = (nu + ST_T - 1) / ST_T;
        run(nchunks, [&](int q) { return q; }, [&](int) { return 0.f; });
        processed = nchunks;
    } else if (!a.query) {
        run_tiles(1, [&](int) { return I; }, [&](int) { return 0.f; });
    }

    if constexpr (!JOIN) {
    // ---- phase B: all other column tiles, exactly as k_st_knn ranks and selects them (streamed.hip): rank key and
    // valid bound of every column tile into a scratch row, then rounds of {3-level radix selection of the next ST_KEEP
    // tiles in (key, tile) order, collect, sort, stream}
    float *skey = a.scr_key + (size_t)bt * a.nt_all;
    float *slb = a.scr_lb + (size_t)bt * a.nt_all;
    if (!a.pre_ranked)   // (k_st_rank_pairs has filled the scratch rows: streamed.hip)
    for (int J = threadIdx.x; J < a.nt_all; J += STB_THREADS) {
        float lb = 0.f, lbc = 0.f;
        for (int an = 0; an < a.na; ++an) {
            const float lj = a.lo[(size_t)an * a.nt_all + J], hj = a.hi[(size_t)an * a.nt_all + J];
            const float gap = fmaxf(sh.loI[an] - hj, lj - sh.hiI[an]);
            // slack for the float32 rounding of D (bounds must stay valid lower bounds)
            lb = fmaxf(lb, gap - 4e-6f * (fabsf(hj) + fabsf(sh.hiI[an])));
            const float dm = a.mid[(size_t)an * a.nt_all + J] - sh.midI[an];
            lbc += dm * dm;   // rank key: squared L2 distance between the tiles' mean anchor vectors
        }
        skey[J] = ((J == I && !a.query) || !(lbc < INFINITY)) ? INFINITY : lbc;   // +inf: never a candidate
        slb[J] = lb;
    }
    __syncthreads();   // block-scope visibility of the scratch row (same CU)
    uint32_t *hist = R3 ? reinterpret_cast<uint32_t *>(&sh.ring[12288 / 4])      // R3: behind the sort buffers in the idle ring
                        : reinterpret_cast<uint32_t *>(&sh.cand_d[0][0]);        // 4096 bins; cand_d is idle between runs
    SelBuf &sb = *reinterpret_cast<SelBuf *>(&sh.ring[0]);           // the ring is idle between runs too
    static_assert(R3 || sizeof(sh.cand_d) >= 4096 * sizeof(uint32_t), "histogram does not fit");
    uint32_t done_bits = 0;   // (done_bits, done_j): key bits / index of the last tile already considered
    int done_j = -1;
    // SHORT LIST (round 5).  A selection round sweeps the scratch row four times (three histogram levels + the collection): at
    // 62 500 column tiles that is 2 MB per round and row tile, a fifth of the kernel's wave cycles once the ranking had left it.
    // Instead: ONE histogram sweep of the key's top 12 bits finds the key bound below which ~4 ST_KEEP eligible tiles lie, ONE
    // more sweep copies those (key, bound, tile) to a short list, and the rounds select from the list -- the same tiles in the
    // same order: every eligible tile below the key bound is in the list, and the list is rebuilt behind the cursor when a round
    // finds fewer than ST_KEEP eligible tiles in it while tiles beyond its bound remain.
    uint32_t *clk = a.scr_cl ? a.scr_cl + (size_t)bt * 3 * ST_CL_CAP : nullptr;   // [3][ST_CL_CAP]: key bits, bound bits, tile
    int cl_n = 0;                               // (uniform)
    bool cl_valid = false, cl_complete = false;
    // entries (key bits, valid bound, tile) of the short list or of the whole scratch row, thread-strided
    auto sweep = [&](bool from_list, auto f) {
        // (eight entries' loads in flight per thread: one entry at a time, a sweep of the 62 500 tiles of N = 8 x 10^6 was 244 dependent
        // global round trips per thread -- the selection rounds were 15 % of the two-stage kernel there, 7 % at N = 10^6)
        constexpr int SW = 8;
        if (from_list) {
            int q = threadIdx.x;
            for (; q + (SW - 1) * STB_THREADS < cl_n; q += SW * STB_THREADS) {
                uint32_t kb[SW], lbb[SW], jj[SW];
#pragma unroll
                for (int u = 0; u < SW; ++u) { kb[u] = clk[q + u * STB_THREADS]; lbb[u] = clk[ST_CL_CAP + q + u * STB_THREADS]; jj[u] = clk[2 * ST_CL_CAP + q + u * STB_THREADS]; }
#pragma unroll
                for (int u = 0; u < SW; ++u) f(kb[u], __uint_as_float(lbb[u]), (int)jj[u]);
            }
            for (; q < cl_n; q += STB_THREADS) f(clk[q], __uint_as_float(clk[ST_CL_CAP + q]), (int)clk[2 * ST_CL_CAP + q]);
        } else {
            int J = threadIdx.x;
            for (; J + (SW - 1) * STB_THREADS < a.nt_all; J += SW * STB_THREADS) {
                uint32_t kb[SW];
                float lbv[SW];
#pragma unroll
                for (int u = 0; u < SW; ++u) { kb[u] = __float_as_uint(skey[J + u * STB_THREADS]); lbv[u] = slb[J + u * STB_THREADS]; }
#pragma unroll
                for (int u = 0; u < SW; ++u) f(kb[u], lbv[u], J + u * STB_THREADS);
            }
            for (; J < a.nt_all; J += STB_THREADS) f(__float_as_uint(skey[J]), slb[J], J);
        }
    };
    // first bin whose cumulative count reaches `want` among nbins bins of `hist`: thread t owns bins [per t, per (t+1)); the
    // bin in sh.sel_bin (-1: the total stays below `want`), the count before it in sh.sel_before
    auto find_bin = [&](int nbins, uint32_t want) {
        const int per = nbins >= STB_THREADS ? nbins / STB_THREADS : 1;
        const bool owner = (int)threadIdx.x * per < nbins;
        uint32_t mine = 0;
        if (owner)
            for (int q = 0; q < per; ++q) mine += hist[threadIdx.x * per + q];
        uint32_t incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        uint32_t *wtot = reinterpret_cast<uint32_t *>(&sb.surv_lb[0]);   // 4 wave totals (surv_lb is idle here)
        if (threadIdx.x == 0) sh.sel_bin = -1;
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        uint32_t before = incl - mine;
        for (int w2 = 0; w2 < wave; ++w2) before += wtot[w2];
        if (owner && before < want && before + mine >= want) {
            uint32_t ac = before;
            int q = threadIdx.x * per;
            for (;; ++q) { if (ac + hist[q] >= want) break; ac += hist[q]; }
            sh.sel_bin = q;
            sh.sel_before = ac;
        }
        __syncthreads();
    };
    for (;;) {
        const float thrmax = thrmax_now();
        // (a round selects what the budget can still use, twice over for the entries the bounds will drop: the warm-up of the two-stage
        // kernel -- 33 tiles -- selected, collected and sorted 512 like everyone else; the tiles and their order are the same)
        const uint32_t keep = (uint32_t)min(ST_KEEP, max(64, 2 * (a.max_tiles - processed)));
        uint32_t prefix = 0;
        uint32_t want = keep;
        bool all = false, use_list = false;
        for (int attempt = 0; attempt < 2; ++attempt) {
            if (clk && !cl_valid) {
                // ---- (re)build the short list behind the cursor
                for (int q = threadIdx.x; q < 4096; q += STB_THREADS) hist[q] = 0;
                __syncthreads();
                sweep(false, [&](uint32_t kb, float lb, int J) {
                    const bool after_done = kb > done_bits || (kb == done_bits && J > done_j);
                    if (kb < 0x7f800000u && after_done && lb * lb < thrmax) atomicAdd(&hist[kb >> 20], 1u);
                });
                __syncthreads();
                find_bin(4096, ST_CL_TARGET);
                const int bb = sh.sel_bin;
                cl_complete = bb < 0;                                  // fewer than the target in all: the list holds every eligible tile
                const uint32_t through = cl_complete ? 0u : sh.sel_before + hist[bb];
                __syncthreads();
                if (cl_complete || through <= ST_CL_CAP) {   // (else: a bin of equal leading key bits larger than the list -- the row is swept this round)
                    const uint32_t bound_bits = cl_complete ? 0x7f800000u : (uint32_t)(bb + 1) << 20;
                    if (threadIdx.x == 0) sh.nsurv = 0;
                    __syncthreads();
                    sweep(false, [&](uint32_t kb, float lb, int J) {
                        const bool after_done = kb > done_bits || (kb == done_bits && J > done_j);
                        if (kb < bound_bits && after_done && lb * lb < thrmax) {
                            const int slot = atomicAdd(&sh.nsurv, 1);
                            if (slot < ST_CL_CAP) { clk[slot] = kb; clk[ST_CL_CAP + slot] = __float_as_uint(lb); clk[2 * ST_CL_CAP + slot] = (uint32_t)J; }
                        }
                    });
                    __syncthreads();
                    cl_n = min(sh.nsurv, ST_CL_CAP);
                    cl_valid = true;
                    __syncthreads();
                }
            }
            use_list = clk && cl_valid;
            prefix = 0; want = keep; all = false;
            for (int level = 0; level < 3 && !all; ++level) {
                const int shift = level == 0 ? 20 : level == 1 ? 8 : 0;
                const int nbins = level == 2 ? 256 : 4096;
                const uint32_t pmask = level == 0 ? 0u : level == 1 ? 0xfff00000u : 0xffffff00u;
                for (int q = threadIdx.x; q < nbins; q += STB_THREADS) hist[q] = 0;
                __syncthreads();
                sweep(use_list, [&](uint32_t kb, float lb, int J) {
                    const bool after_done = kb > done_bits || (kb == done_bits && J > done_j);
                    if (kb < 0x7f800000u && after_done && lb * lb < thrmax && (kb & pmask) == prefix)
                        atomicAdd(&hist[(kb >> shift) & (nbins - 1)], 1u);
                });
                __syncthreads();
                find_bin(nbins, want);
                if (sh.sel_bin < 0) all = true;
                else { prefix |= (uint32_t)sh.sel_bin << shift; want -= sh.sel_before; }
                __syncthreads();
            }
            if (!use_list || !all || cl_complete) break;
            cl_valid = false;   // fewer than ST_KEEP eligible tiles left in the list, and tiles beyond its bound remain: rebuild, select again
        }
        const uint32_t cut_bits = all ? 0x7f7fffffu : prefix;   // take keys <= cut (ties resolved by the sort below)
        if (threadIdx.x == 0) sh.nsurv = 0;
        __syncthreads();
        sweep(use_list, [&](uint32_t kb, float lb, int J) {
            const bool after_done = kb > done_bits || (kb == done_bits && J > done_j);
            if (kb < 0x7f800000u && after_done && lb * lb < thrmax && kb <= cut_bits) {
                const int slot = atomicAdd(&sh.nsurv, 1);
                if (slot < ST_SURV) { sb.surv_lb[slot] = __uint_as_float(kb); sb.surv_vb[slot] = lb; sb.surv_j[slot] = J; }
            }
        });
        __syncthreads();
        int ns = min(sh.nsurv, ST_SURV);
        if (ns == 0) break;
        {   // sort by (rank key, J): bitonic over the next power of two >= ns slots
            int sortn = 2;
            while (sortn < ns) sortn <<= 1;
            for (int q = threadIdx.x; q < sortn; q += STB_THREADS)
                if (q >= ns) { sb.surv_lb[q] = INFINITY; sb.surv_j[q] = 0x7fffffff; }
            __syncthreads();
            for (int k2 = 2; k2 <= sortn; k2 <<= 1)
                for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
                    for (int q = threadIdx.x; q < sortn; q += STB_THREADS) {
                        const int p2 = q ^ j2;
                        if (p2 > q) {
                            const bool up = (q & k2) == 0;
                            const float lq = sb.surv_lb[q], lp = sb.surv_lb[p2];
                            const int jq = sb.surv_j[q], jp = sb.surv_j[p2];
                            const bool gt = lq > lp || (lq == lp && jq > jp);
                            if (gt == up) {
                                sb.surv_lb[q] = lp; sb.surv_lb[p2] = lq; sb.surv_j[q] = jp; sb.surv_j[p2] = jq;
                                const float t = sb.surv_vb[q]; sb.surv_vb[q] = sb.surv_vb[p2]; sb.surv_vb[p2] = t;
                            }
                        }
                    }
                    __syncthreads();
                }
        }
        const bool more = !all;          // the selection was cut at ST_KEEP: later tiles remain
        if (ns > (int)keep && more) ns = (int)keep;
        const uint32_t round_last_bits = __float_as_uint(sb.surv_lb[ns - 1]);
        const int round_last_j = sb.surv_j[ns - 1];
        // the round's tiles leave the ring before the stream takes it back
        for (int q = threadIdx.x; q < ns; q += STB_THREADS) { sh.run_j[q] = sb.surv_j[q]; sh.run_vb[q] = sb.surv_vb[q]; }
        __syncthreads();   // hist (cand_d) and the sort buffers (ring) are idle again: the stream may run
        run_tiles(ns, [&](int q) { return sh.run_j[q]; }, [&](int q) { return sh.run_vb[q]; });
        done_bits = round_last_bits;
        done_j = round_last_j;
        __syncthreads();
        if (dried) break;
        if (processed >= a.max_tiles) break;
        if (!more) break;   // the selection saw every eligible tile
    }
    }   // (!JOIN)
    __syncthreads();
    // ---- exact re-ranking: the lists hold KL >= K columns chosen by the split-fp16 distance (error ~1e-5 relative on a
    // neighbour's d^2); their exact float32 distances sum (x - y)^2 decide which K are handed on, and in which order
    if (threadIdx.x == 0) sh.nsurv = 0;
    if (threadIdx.x < 4) sh.wave_ins[0][threadIdx.x] = 0;   // (idle now: the guard's bitmask of flagged rows)
    {
        float *ex = R3 ? &sh.ring[0] : &sh.cand_d[0][0];   // [ST_T][KMAX] exact d^2 (cand_d -- R3: the ring -- is idle now; row stride KMAX <= ST_SLAB + 1)
        for (int q = threadIdx.x; q < ST_T * KL; q += STB_THREADS) {
            const int row = q / KL, e = q - row * KL;
            const int32_t cc = sh.list_c[row][e];
            float d2 = INFINITY;
            bool known = false;
            if constexpr (JOIN) {
                // a column the row already listed before this pass: its exact d^2 is in the previous lists (a pass replaces a few
                // per cent of the entries; re-reading two 512-byte rows for every kept entry was a quarter of the pass)
                if (cc != 0x7fffffff) {
                    const int32_t *ol = a.lists_all + ((size_t)grow0 + row) * K;
                    const float *od = a.out_d2 + ((size_t)bt * ST_T + row) * K;
                    for (int t = 0; t < K; ++t)
                        if (ol[t] == cc) { known = true; d2 = od[t]; }
                }
            }
            if (!known && cc != 0x7fffffff && sh.list_d[row][e] < INFINITY) {
                const float4 *x = reinterpret_cast<const float4 *>(a.Rs + (size_t)(grow0 + row) * DIM);
                const float4 *y = reinterpret_cast<const float4 *>(a.Xs + (size_t)cc * DIM);
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 8
                for (int t = 0; t < DIM / 4; ++t) {
                    const float4 u = x[t], v = y[t];
                    const float dx = u.x - v.x, dy = u.y - v.y, dz = u.z - v.z, dw = u.w - v.w;
                    s0 += dx * dx; s1 += dy * dy; s2 += dz * dz; s3 += dw * dw;
                }
                d2 = (s0 + s1) + (s2 + s3);
            }
            ex[row * KMAX + e] = d2;
        }
        __syncthreads();
        if (threadIdx.x < ST_T) {   // one thread per row: insertion sort of <= 32 entries by (exact d^2, column)
            const int row = threadIdx.x;
            // guard, part 1: the largest error the split products made on this row's kept entries
            float eps = 0.f;
            int nfin = 0;
            for (int e = 0; e < KL; ++e) {
                const float ap = sh.list_d[row][e] * inv_scale2, exv = ex[row * KMAX + e];   // (the lists are in scaled units)
                if (exv < INFINITY) { eps = fmaxf(eps, fabsf(ap - exv)); ++nfin; }
            }
            for (int e = 1; e < KL; ++e) {
                const float d = ex[row * KMAX + e];
                const int32_t cc = sh.list_c[row][e];
                int p = e;
                while (p > 0 && (d < ex[row * KMAX + p - 1] || (d == ex[row * KMAX + p - 1] && cc < sh.list_c[row][p - 1]))) {
                    ex[row * KMAX + p] = ex[row * KMAX + p - 1];
                    sh.list_c[row][p] = sh.list_c[row][p - 1];
                    --p;
                }
                ex[row * KMAX + p] = d;
                sh.list_c[row][p] = cc;
            }
            // guard, part 2: a column left outside the list has an approximate d^2 >= the list's last approximate entry; it can
            // only belong among the K nearest if its exact d^2 is below the K-th exact one, i.e. if the products were off by more
            // than the room between the two -- flagged when that room is within twice the measured error
            if (!JOIN && nfin > K && ex[row * KMAX + K - 1] + 2.f * eps > sh.list_d[row][KL - 1] * inv_scale2) {
                atomicAdd(&sh.nsurv, 1);   // (nsurv is idle here)
                atomicOr(reinterpret_cast<uint32_t *>(&sh.wave_ins[0][0]) + (row >> 5), 1u << (row & 31));   // (so are the waves' counters: the flagged rows)
            }
        }
        __syncthreads();
        for (int q = threadIdx.x; q < ST_T * K; q += STB_THREADS) {
            const int row = q / K, e = q - row * K;
            const float d2 = ex[row * KMAX + e];
            float *od = JOIN ? a.out_d2_new : a.out_d2;
            int32_t *oc = JOIN ? a.out_col_new : a.out_col;
            od[((size_t)bt * ST_T + row) * K + e] = d2;
            oc[((size_t)bt * ST_T + row) * K + e] = d2 < INFINITY ? sh.list_c[row][e] : 0x7fffffff;
        }
    }
    if constexpr (JOIN) {
        if (threadIdx.x == 0) atomicAdd(a.evals + 2, (unsigned long long)processed);   // slot 2: join chunks (0: tile phase, 1: pass yield)
        int wins = ins;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) wins += __shfl_xor(wins, off);
        if (lane == 0 && wins) atomicAdd(a.updates, (unsigned long long)wins);
    } else if (threadIdx.x == 0) {
        atomicAdd(a.evals, (unsigned long long)processed);
        if (sh.nsurv) atomicAdd(a.evals + 3, (unsigned long long)sh.nsurv);   // slot 3: rows flagged by the guard
        if (a.guard_tiles)                                                        // ... and which rows: those are done again exactly (repair.hip)
            for (int w = 0; w < 4; ++w) a.guard_tiles[(size_t)bt * 4 + w] = (uint32_t)sh.wave_ins[0][w];
    }
#ifdef ST_PROFILE
    P8(7)
    if (lane == 0 && a.prof)
        for (int i = 0; i < 16; ++i) atomicAdd(a.prof + i, (unsigned long long)pf[i]);
#endif
}

// the three-slot form (R3): 128 dimensions, 16-entry lists, tile phase and queries (not the join passes)
template <int DIM, int KMAX> static int launchb3(annchor_ctx *c, const KnnArgs &a)
{
    const size_t lds = sizeof(KnnSharedB<DIM, KMAX, true>);
    ANN_REQUIRE(c, lds <= 80 * 1024, ANNCHOR_ELIMIT, "streamed k-NN (three-slot form) needs %zu B of LDS", lds);
    ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)k_st_knnbf<DIM, KMAX, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    k_st_knnbf<DIM, KMAX, false, true><<<a.tile_count, STB_THREADS, lds, c->stream>>>(a);
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}

template <int DIM, int KMAX> static int launchb(annchor_ctx *c, const KnnArgs &a, bool join)
{
    const size_t lds = sizeof(KnnSharedB<DIM, KMAX>);
    ANN_REQUIRE(c, lds <= 160 * 1024, ANNCHOR_ELIMIT, "streamed k-NN (split-fp16 form) needs %zu B of LDS", lds);
    if (join) {
        ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)k_st_knnbf<DIM, KMAX, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_st_knnbf<DIM, KMAX, true><<<a.tile_count, STB_THREADS, lds, c->stream>>>(a);
    } else {
        ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)k_st_knnbf<DIM, KMAX, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_st_knnbf<DIM, KMAX, false><<<a.tile_count, STB_THREADS, lds, c->stream>>>(a);
    }
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}

// The tile phase through the split-fp16 kernel when the shape fits it (padded dim <= 128: a slab of 32 columns is 16 KB there;
// K + ST_BF_MARGIN <= 32 list entries; the split copy of the columns exists); *handled = false sends the caller to the
// exact-f32 kernel k_st_knn.
int ann_stream_launch_knnbf(annchor_ctx *c, const KnnArgs &a, int dim_padded, bool *handled, bool join)
{
    *handled = true;
    if (a.K + ST_BF_MARGIN > ST_KMAX || !a.Xb || !a.rsb || !a.cvec) { *handled = false; return ANNCHOR_OK; }
    const bool k16 = a.K + ST_BF_MARGIN <= 16;
    static const char *kern = getenv("ANNCHOR_ST_KERNEL");
    if (kern && !strcmp(kern, "bf3") && !join && k16 && dim_padded == 128) return launchb3<128, 16>(c, a);
    switch (dim_padded) {
    case 32: return k16 ? launchb<32, 16>(c, a, join) : launchb<32, ST_KMAX>(c, a, join);
    case 64: return k16 ? launchb<64, 16>(c, a, join) : launchb<64, ST_KMAX>(c, a, join);
    case 128: return k16 ? launchb<128, 16>(c, a, join) : launchb<128, ST_KMAX>(c, a, join);
    default: *handled = false; return ANNCHOR_OK;
    }
}

// ------------------------------------------------------------------ the split copy of the ordered rows
// The expanded form |x|^2 + |y|^2 - 2 x.y loses what |x|^2 exceeds d^2 by, so the rows are centred first: c = mean of
// the anchors' coordinates (the max-min anchors span the data; every rank knows all of them: no collective).  Distances do not
// change; the exact re-ranking and the final distances use the original rows.
__global__ void k_st_centre(const float *__restrict__ avecs, int na, int dim, int dimp, float *__restrict__ cvec)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) { cvec[dimp] = 0.f; cvec[dimp + 1] = 0.f; }   // [dimp]: largest |x - c| (then the scale), as float bits for atomicMax
    if (k >= dimp) return;
    float s = 0.f;
    if (avecs && k < dim)
        for (int r = 0; r < na; ++r) s += avecs[(size_t)r * dim + k];
    cvec[k] = (avecs && k < dim && na > 0) ? s / (float)na : 0.f;
}

// largest |coordinate| of the centred rows (non-negative floats order like their bit patterns)
__global__ __launch_bounds__(256) void k_st_absmax(const float *__restrict__ Xs, const float *__restrict__ rs, int64_t n4, int dimp,
                                                   float *__restrict__ cvec)
{
    const int per = dimp / 4;
    float m = 0.f;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = t / per;
        if (!(rs[row] < INFINITY)) continue;
        const float4 v = reinterpret_cast<const float4 *>(Xs)[t], cc = reinterpret_cast<const float4 *>(cvec)[t - row * per];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x - cc.x), fabsf(v.y - cc.y))), fmaxf(fabsf(v.z - cc.z), fabsf(v.w - cc.w)));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(reinterpret_cast<unsigned int *>(cvec + dimp + 1), __float_as_uint(m));
}

// the power of two that brings the largest |coordinate| to (2^12, 2^13]: fp16 holds 65504, a row's squared norm stays far
// inside float32, and scaling by a power of two is exact
__global__ void k_st_scale(float *__restrict__ cvec, int dimp)
{
    const float m = cvec[dimp + 1];
    int e = 0;
    if (m > 0.f && m < INFINITY) { (void)frexpf(m, &e); e = 13 - e; }   // m = f 2^e', f in [0.5, 1): m 2^(13 - e') in [2^12, 2^13)
    cvec[dimp] = ldexpf(1.f, max(-100, min(100, e)));
}

// Xb[row] = {fp16 hi parts of the row's dimp centred, scaled floats, then their lo parts}: hi = fp16(x) (round to nearest
// even), lo = fp16(x - hi) -- the same bytes per row as the float32 copy; three MFMAs (hi.hi + hi.lo + lo.hi) then
// reproduce x.y to ~2^-22 |x||y|, the accuracy of the exact f32 MFMA stream (tools/microbench/f16_split.hip).
// rsb[row] = |scale (x - c)|^2 (+inf on padding rows).  dimp / 4 threads per row, one float4 each.
__global__ __launch_bounds__(256) void k_st_split_f16(const float *__restrict__ Xs, const float *__restrict__ rs, const float *__restrict__ cvec,
                                                      int64_t n_pad, int dimp, uint16_t *__restrict__ Xb, float *__restrict__ rsb)
{
    // min(32, dimp / 4) threads per row, a float4 each per 128 dimensions (dimp 32 / 64: 8 / 16 threads, one float4 each)
    const int per = dimp >= 128 ? 32 : dimp / 4;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row = t / per;
    const int q0 = (int)(t - row * per);
    const float scale = cvec[dimp];
    float acc = 0.f;
    if (row < n_pad) {
        const bool real = rs[row] < INFINITY;
        for (int q = q0; q < dimp / 4; q += per) {
            const float4 v = reinterpret_cast<const float4 *>(Xs + (size_t)row * dimp)[q], cc = reinterpret_cast<const float4 *>(cvec)[q];
            const float x[4] = {real ? (v.x - cc.x) * scale : 0.f, real ? (v.y - cc.y) * scale : 0.f, real ? (v.z - cc.z) * scale : 0.f,
                                real ? (v.w - cc.w) * scale : 0.f};
            uint16_t h[4], l[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const _Float16 hb = (_Float16)x[j];
                const _Float16 lb = (_Float16)(x[j] - (float)hb);
                h[j] = __builtin_bit_cast(uint16_t, hb);
                l[j] = __builtin_bit_cast(uint16_t, lb);
                acc += x[j] * x[j];
            }
            uint16_t *dst = Xb + (size_t)row * dimp * 2 + 4 * q;
            *reinterpret_cast<uint2 *>(dst) = uint2{(uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16)};
            *reinterpret_cast<uint2 *>(dst + dimp) = uint2{(uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16)};
        }
    }
    for (int off = per >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (row < n_pad && q0 == 0) rsb[row] = rs[row] < INFINITY ? acc : INFINITY;
}

int ann_stream_split_rows(annchor_ctx *c, StreamState *s)
{
    const int64_t n4 = s->n_pad * (int64_t)(s->dimp / 4);
    ANN_TRY(ann_stream_reserve(c, s->Xb, sizeof(uint16_t) * 2 * (size_t)s->n_pad * s->dimp));
    ANN_TRY(ann_stream_reserve(c, s->rsb, sizeof(float) * (size_t)s->n_pad));
    ANN_TRY(ann_stream_reserve(c, s->cvec, sizeof(float) * (size_t)(s->dimp + 2)));
    const bool have = s->avecs.p != nullptr && s->na > 0 && s->avecs.cap >= sizeof(float) * (size_t)s->na * s->dim;
    k_st_centre<<<ann_blocks(s->dimp, 128), 128, 0, c->stream>>>(have ? s->avecs.as<float>() : nullptr, s->na, s->dim, s->dimp, s->cvec.as<float>());
    k_st_absmax<<<(int)std::min<int64_t>(ann_blocks(n4, 256), 4096), 256, 0, c->stream>>>(s->Xs.as<float>(), s->rs.as<float>(), n4, s->dimp,
                                                                                        s->cvec.as<float>());
    k_st_scale<<<1, 1, 0, c->stream>>>(s->cvec.as<float>(), s->dimp);
    k_st_split_f16<<<ann_blocks(s->n_pad * (s->dimp >= 128 ? 32 : s->dimp / 4), 256), 256, 0, c->stream>>>(s->Xs.as<float>(), s->rs.as<float>(), s->cvec.as<float>(), s->n_pad, s->dimp,
                                                              s->Xb.as<uint16_t>(), s->rsb.as<float>());
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}
