// knnbk.hip -- k_st_knnbk: the split-fp16 tile kernel of knnbf.hip for rows of MORE than 128 dimensions (padded dim 256 .. 1024,
// a multiple of 128), k-blocked.
//
// k_st_knnbk keeps a wave's 32 rows x 128 dimensions as fp16 hi / lo operands in registers for the whole kernel and finishes a
// 32 x 32 block of dot products per column slab.  Beyond 128 dimensions the operands do not fit, so the dimensions are cut into
// NKB = dim / 128 blocks and a column TILE (128 columns = 4 slabs) is finished block by block: four accumulators (one per slab)
// stay in registers across the blocks; per block the wave converts its rows' 128 dimensions of that block to hi / lo operands
// (from the float32 rows, centred and scaled as in knnbf.hip) and streams the block's four column slabs -- the same 16 KB
// LDS-DMA slabs, the same ring, the same operand layout (a slab of block kb = the block's 128 hi halves and 128 lo halves of 32
// columns: two 256-byte runs of every column's row in the split copy).  After the last block the four slabs are tested against
// the rows' thresholds and merged into the lists exactly as knnbf.hip does it slab by slab.  Rows are read once per column tile
// (NKB x 64 KB per workgroup) beside the columns' NKB x 64 KB: twice the bytes per flop of the 128-dimension kernel.
// Everything else -- ranking of the column tiles, selection rounds, pruning, early stop, budget, the join passes' gathered
// columns, the exact float32 re-ranking of the K + 2 kept columns and the guard count -- is knnbf.hip's code, with the padded
// dimension a run-time number.  (The exact-f32 kernel k_st_knn stops at padded dim 256: beyond it there is no fallback for
// data the guard flags as ill-conditioned -- the caller is told through annchor_stream_last_kernel.)
#include "streamed.h"

#define STBK_THREADS 256
#define ST_BF_MARGIN 2   // list entries beyond K kept by the split-fp16 distance (re-ranked exactly at the end)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ST_PROFILE builds: per-wave cycle sums by segment (a.prof[0..7], printed by knn_tile_phase):
//   0 MFMA stream (operand reads + MFMA issue)   1 barrier after the stream   2 choice of the next tile
//   3 threshold test + survivor inserts           4 LDS-DMA requests           5 barrier after the requests
//   6 merge (+ publish, run prologue / tail)      7 ranking, selection rounds, the rest
#ifdef ST_PROFILE
// (ordered: nothing may be scheduled across the time stamp, and it waits for the wave's outstanding LDS / scalar traffic)
__device__ __forceinline__ long long stk_now()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return (long long)t;
}
#define P8(i) { const long long pf_n = stk_now(); pf[i] += pf_n - pf_t; pf_t = pf_n; }
// sub-segment stamp: time since the last P8 / PS goes to slot i (8..15) without resetting the P8 clock
#define PS(i) { const long long pf_n = stk_now(); pf[i] += pf_n - pf_s; pf_s = pf_n; }
#define PS0 { pf_s = stk_now(); }
#else
#define P8(i)
#define PS(i)
#define PS0
#endif

template <int DIM, int KMAX> struct KnnSharedK {
    static constexpr int SLABF = ST_SLAB * DIM;                                 // floats per operand slab
    static constexpr int RINGF = 2 * SLABF * 4 >= 12288 ? 2 * SLABF : 12288 / 4;   // (>= sizeof(SelBufK))
    float ring[RINGF];   // FIRST (LDS-DMA destinations stay below 64 KB); two slots, slot = slab parity.  Between runs the
                         // selection's sort buffers (SelBuf) live here
    float cand_d[ST_T][ST_SLAB + 1];   // (between runs: the selection's 4096-bin histogram; at the end: exact distances)
    uint8_t cand_c[ST_T][ST_SLAB + 4];
    float list_d[ST_T][KMAX + 1];
    int32_t list_c[ST_T][KMAX + 1];
    float thr[ST_T];
    float hb[ST_T];      // (|x_row|^2 - thr[row]) / 2: column c passes the row's test iff x_row . x_c > hb[row] + |x_c|^2 / 2
    float rrow[ST_T];    // |x_row|^2
    int cnt[ST_T];
    float loI[64], hiI[64], midI[64];
    uint32_t slab_id[8][ST_SLAB];   // join passes: ordered column index of each column of the tile's four slabs, by tile parity
    float run_vb[ST_KEEP];     // the current round's tiles in rank order: valid bound, tile
    int32_t run_j[ST_KEEP];
    float wave_thr[2][4];   // worst k-th squared distance per 32-row group, published at the end of tile n into [n & 1]
    int wave_ins[2][4];     // list insertions per wave (cumulative), likewise
    int nsurv;
    int sel_bin;
    uint32_t sel_before;
};
struct SelBufK {   // candidate tiles of a selection round (aliases the operand ring, idle between runs)
    float surv_lb[ST_SURV];
    float surv_vb[ST_SURV];
    int32_t surv_j[ST_SURV];
};

// swizzle of a column's 16-byte units (see the header comment); UPC = units per column
template <int UPC> __device__ __forceinline__ int unit_swzk(int col) { return UPC >= 16 ? (col & 15) : ((col >> 1) & (UPC - 1)); }

// End of a slab: the wave's LDS-DMA pieces of the next slab have landed and its LDS traffic is done; the barrier hands the
// next slab to every wave and this slab's ring slot back to the requests.  (The requests are invisible to the compiler's
// wait bookkeeping -- inline asm; sched_barrier: the memory clobber alone does not keep register-only instructions on
// their side.)
__device__ __forceinline__ void slab_endk()
{
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// JOIN: a join pass (streamed.hip: k_st_join_cands has collected the row tile's candidate columns) -- the "tiles" are runs of
// 128 gathered columns of the candidate list, the lists start from the previous phase's, nothing is ranked or pruned.
template <int KMAX, bool JOIN = false> __global__ __launch_bounds__(STBK_THREADS, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_st_knnbk(KnnArgs a)
{
    constexpr int DIM = 128;                // dimensions per block; a.dimr (a multiple of DIM) = the rows' padded dimension
    const int dimr = __builtin_amdgcn_readfirstlane(a.dimr), nkb = dimr / DIM;
    extern __shared__ __attribute__((aligned(16))) unsigned char smemb[];
    KnnSharedK<DIM, KMAX> &sh = *reinterpret_cast<KnnSharedK<DIM, KMAX> *>(smemb);
    constexpr int UPC = DIM / 4;            // 16-byte units per column
    constexpr int NV = UPC / 2;             // operand reads (ds_read_b128) per slab and lane: G hi + G lo
    constexpr int NPIECE = UPC * ST_SLAB / 64;   // 1 KB pieces per slab
    constexpr int NI = NPIECE / 4;          // pieces per loading wave
    static_assert(NI == 1 || NI == 2 || NI == 4, "pieces per loading wave");
    static_assert(sizeof(sh.ring) <= 65536, "LDS-DMA destinations must stay below 64 KB");
    static_assert(sizeof(SelBufK) <= sizeof(sh.ring) && sizeof(SelBufK) == 12288, "selection buffers alias the ring");
    static_assert(KMAX <= ST_SLAB + 1 || sizeof(sh.ring) >= sizeof(float) * ST_T * KMAX, "the exact re-ranking's scratch: cand_d, or the ring for 64-entry lists");
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
    const float scale = a.cvec[dimr];              // power of two: the centred data's largest |coordinate| becomes <= 2^13
    const float inv_scale2 = 1.f / (scale * scale);
    const float *xrow = a.Rs + (size_t)(grow0 + rowbase + col) * dimr + 8 * half;
    // block kb of the wave's rows -> operand registers (hi / lo halves of the centred, scaled values); returns the lane's share
    // of the block's squared norm
    // Data rows are columns too: their split operands already sit in the column copy (a.Xb: [row][hi: dimr halves | lo: dimr
    // halves]) -- two 16-byte loads per k-step instead of eight floats, a subtraction, a scaling and two conversions per value, for
    // every block of every tile (the conversions were a third of the kernel at d = 768).  Query rows (another context's, maybe
    // outside the data's range) are converted here as before.
    const bool rows_split = !a.query && a.Rs == a.Xs;   // (uniform)
    const char *xbrow = reinterpret_cast<const char *>(a.Xb) + (size_t)(grow0 + rowbase + col) * dimr * 4 + 16 * half;
    auto load_rows = [&](int kb) __attribute__((always_inline)) -> float {
        if (rows_split) {
            const char *hp = xbrow + (size_t)kb * (DIM * 2);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                ah[g] = *reinterpret_cast<const f16x8 *>(hp + 32 * g);
                al[g] = *reinterpret_cast<const f16x8 *>(hp + (size_t)dimr * 2 + 32 * g);
            }
            return 0.f;
        }
        const float *xr = xrow + kb * DIM;
        const float *cv = a.cvec + kb * DIM + 8 * half;
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
        return acc2;
    };
    float rr_c;   // |scale (x_row - c)|^2 over all blocks
    {
        float acc2 = 0.f;
        for (int kb = 0; kb < nkb; ++kb) acc2 += load_rows(kb);
        rr_c = acc2 + __shfl_xor(acc2, 32);
        if (rows_split) rr_c = a.rsb[grow0 + rowbase + col];   // (the split pass's own sum: a row's norm as a row = its norm as a column)
    }
    if (threadIdx.x < ST_T) {
        const int row = threadIdx.x;
        const bool real = a.rr[grow0 + row] < INFINITY;
        sh.cnt[row] = 0;
        if constexpr (JOIN) {
            // the lists as the previous phase left them (exact d^2, original units -> scaled); the KL - K spare entries start
            // as copies of the K-th value without a column, so that the row's threshold is its current K-th distance
            float last = INFINITY;
            for (int q = 0; q < KMAX; ++q) {
                const bool have = q < K;
                const float d = have ? a.out_d2[((size_t)bt * ST_T + row) * K + q] * (scale * scale) : last;
                sh.list_d[row][q] = q < KL ? d : INFINITY;
                sh.list_c[row][q] = have ? a.lists_all[((size_t)grow0 + row) * K + q] : 0x7fffffff;
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
    int ins = 0;         // list insertions counted by this lane (the first lane of a merge group)
    int processed = 0;   // column tiles scheduled so far (uniform)
    int tdone = 0;       // column tiles completed and published (uniform); tile n publishes into slot n & 1
    int win_start = 0, win_ins = 0;
    bool dried = false;
    uint32_t *ebits = (!JOIN && a.eval_bits) ? a.eval_bits + (size_t)bt * a.eval_words : nullptr;   // (a join pass only reads them)
#ifdef ST_PROFILE
    long long pf[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long pf_t = stk_now();
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
    uint32_t loff[NI], lrow[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int u = (rg * NI + i) * 64 + lane;
        const int c = u / UPC, x = u % UPC;
        const int xs = x ^ unit_swzk<UPC>(c);   // the source unit that lands at unit x: 0..15 hi halves, 16..31 lo halves of the block
        lrow[i] = (uint32_t)(c * dimr * 4);
        loff[i] = lrow[i] + (uint32_t)(xs < 16 ? xs * 16 : dimr * 2 + (xs - 16) * 16);   // (+ 256 kb: block kb of the row)
    }
    const char *xb = reinterpret_cast<const char *>(a.Xb);   // [n_all][2][dimr] fp16: hi parts, then lo parts of every ordered row (centred, scaled)
    const uint32_t *ulist = JOIN ? a.ucand + (size_t)bt * a.ucap : nullptr;   // join: sorted candidate columns, 0xffffffff padded to 128
    // operand slab `slab` of column tile J, dimension block kb -> ring slot `slot`: this wave's NI pieces
    uint32_t jn_id[NI];       // join passes: the column ids behind the wave's pieces of the slab that follows the one last requested
    int64_t jn_c0 = -1;
    auto issue_slab = [&](int J, int slab, int kb, int slot) __attribute__((always_inline)) {
        // (uniform by construction; the loops they live in end on values read from LDS, which the compiler cannot know to be)
        J = __builtin_amdgcn_readfirstlane(J); kb = __builtin_amdgcn_readfirstlane(kb); slot = __builtin_amdgcn_readfirstlane(slot);
        const uint32_t dst = lds0 + (uint32_t)(slot * ST_SLAB * DIM * 4 + rg * NI * 1024);
        if constexpr (JOIN) {
            // gathered columns: every lane's source is its own column's row (per-lane 64-bit addresses)
            // (the ids: requested with the slab before this one, as in knnbf.hip -- the order of a pass is fixed: the tile's next
            // slab, the next block's first, the next run's first)
            const int64_t c0 = (int64_t)J * ST_T + slab * ST_SLAB;
            const char *srcs[NI];
            const bool have_ids = jn_c0 == c0;   // (uniform)
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int u = (rg * NI + i) * 64 + lane;
                const uint32_t id = have_ids ? jn_id[i] : ulist[c0 + u / UPC];
                srcs[i] = xb + (size_t)(id == 0xffffffffu ? 0u : id) * ((size_t)dimr * 4) + (loff[i] - lrow[i]) + (size_t)kb * 256;
            }
            unsigned keep;
#pragma unroll
            for (int i = 0; i < NI; ++i)
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(srcs[i]), "s"(dst + (uint32_t)(i * 1024)) : "memory");
            const int64_t n0 = slab < 3 ? c0 + ST_SLAB : (kb + 1 < nkb ? (int64_t)J * ST_T : (int64_t)(J + 1) * ST_T);
            jn_c0 = -1;
            if (n0 + ST_SLAB <= (int64_t)a.ucap) {
                jn_c0 = n0;
#pragma unroll
                for (int i = 0; i < NI; ++i) jn_id[i] = ulist[n0 + ((rg * NI + i) * 64 + lane) / UPC];
            }
            return;
        }
        const unsigned long long sa64 = (unsigned long long)(uintptr_t)(xb + ((size_t)J * ST_T + slab * ST_SLAB) * ((size_t)dimr * 4) + (size_t)kb * 256);
        const char *src = reinterpret_cast<const char *>(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(sa64 >> 32)) << 32) |
                                                         (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sa64));   // wave-uniform: an SGPR pair
        unsigned keep;
        static_assert(NI == 4, "128-dimension blocks: four pieces per wave");
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(loff[0]), "v"(loff[1]), "v"(loff[2]), "v"(loff[3]), "s"(src), "s"(dst) : "memory", "scc");
    };
    f32x16 acc[4];              // the four slabs of the column tile in the stream: summed over the dimension blocks
    float rj4[4] = {0.f, 0.f, 0.f, 0.f};   // squared norms of the lane's column in each of them (requested at the tile's first block)
    // the slab whose accumulators are being tested / merged
    bool pend = false;
    int pJ = 0, pslab = 0, ppar = 0, tpar = 0;   // (tpar: parity of the tile in the stream -- the join passes' column ids are double-buffered by it)
    float prj = 0.f;
    uint32_t ppass = 0;
    // 3 DIM / 16 MFMAs of this wave's 32 rows (block operands ah / al) against the 32 columns in ring slot `slot` into accC
    auto mfma_slab = [&](int slot, f32x16 &accC) __attribute__((always_inline)) {
        const float4 *base = reinterpret_cast<const float4 *>(&sh.ring[slot * (ST_SLAB * DIM)]) + col * UPC;
        const int gsw = half ^ unit_swzk<UPC>(col);
        float4 b[NV];   // b[g]: hi parts of k-step g; b[G + g]: lo parts
#pragma unroll
        for (int v = 0; v < NV; ++v) b[v] = base[(2 * v) ^ gsw];
        __builtin_amdgcn_sched_group_barrier(0x100, NV, 0);
#pragma unroll
        for (int g = 0; g < G; ++g) {   // small terms first
            accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[g], __builtin_bit_cast(f16x8, b[g]), accC, 0, 0, 0);
            accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[g], __builtin_bit_cast(f16x8, b[G + g]), accC, 0, 0, 0);
            accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[g], __builtin_bit_cast(f16x8, b[g]), accC, 0, 0, 0);
        }
    };
    // the norms (and, join passes, the ids) of the four slabs' columns of tile J
    auto tile_norms = [&](int J) __attribute__((always_inline)) {
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            if constexpr (JOIN) {
                const uint32_t id = ulist[(int64_t)J * ST_T + sl * ST_SLAB + col];
                rj4[sl] = id == 0xffffffffu ? INFINITY : a.rsb[id];   // (padding never passes: x.y > hb + inf is false)
                if (wave == 0 && lane < ST_SLAB) sh.slab_id[(tpar << 2) + sl][lane] = id;
            } else {
                rj4[sl] = a.rsb[(int64_t)J * ST_T + sl * ST_SLAB + col];
            }
        }
    };
    // what the shadow test let through (slab pslab of tile pJ, accumulators accP, column norms prj): survivors into the rows'
    // candidate slots, then the merge into the sorted lists
    auto insert_merge = [&](const f32x16 &accP) __attribute__((always_inline)) {
        const int J = pJ, slab = pslab;
        const float rj = prj;
        uint32_t pass = ppass;
        PS0
        const bool self_tile = !JOIN && !a.query && (int64_t)J * ST_T == grow0;
        PS(10)   // thresholds read, accumulators there, 16 tests
        if (pass) {
            if (self_tile) {   // a point is not its own neighbour
                const int dcol = slab * ST_SLAB + col - rowq;
                if (dcol >= 0 && dcol < 32 && (dcol & 4) == 0) pass &= ~(1u << ((dcol & 3) + 4 * (dcol >> 3)));
            }
            while (pass) {
                const int g = __builtin_ctz(pass);
                pass &= pass - 1;
                const int rowl = rowq + (g & 3) + 8 * (g >> 2);
                float ag = accP[0];
#pragma unroll
                for (int t = 1; t < 16; ++t) ag = g == t ? accP[t] : ag;
                const float d2 = fmaxf(sh.rrow[rowl] + rj - 2.f * ag, 0.f);
                const int slot = atomicAdd(&sh.cnt[rowl], 1);
                sh.cand_d[rowl][slot] = d2;
                sh.cand_c[rowl][slot] = (uint8_t)col;
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
                const int nc = live ? sh.cnt[row] : 0;
                float ld = (live && e < KL) ? sh.list_d[row][e] : INFINITY;
                int32_t lc = (live && e < KL) ? sh.list_c[row][e] : 0x7fffffff;
                int q = 0;
                while (__ballot(q < nc)) {
                    const bool on = q < nc;
                    const float d = on ? sh.cand_d[row][q] : INFINITY;
                    int32_t cc;
                    if constexpr (JOIN) cc = on ? (int32_t)sh.slab_id[(ppar << 2) + slab][sh.cand_c[row][q]] : 0x7fffffff;
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
            for (int e = 0; e < 4; ++e) pass |= (accP[4 * q + e] > hv[e] + hrj ? 1u : 0u) << (4 * q + e);
        }
        ppass = pend ? pass : 0u;
    };
    // the wave's insertion count and its rows' worst k-th distance, at the end of a tile
    auto publish = [&]() {
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
    auto run = [&](int ns, auto jl, auto vb) __attribute__((always_inline)) {
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
        // fill: block 0 of slab 0 of the first tile
        int sidx = 0;   // slab counter of the run: ring slot = sidx & 1
        issue_slab(J, 0, 0, 0);
        (void)load_rows(0);
        slab_endk();
        P8(6)
        for (;;) {
            tile_norms(J);
            int Jn = -1;
#pragma unroll
            for (int sl = 0; sl < 4; ++sl)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[sl][r] = 0.f;
            for (int kb = 0; kb < nkb; ++kb) {
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) {
                    // the next slab of the sequence: the block's next slab, the next block's first, or the next tile's
                    if (sl < 3) issue_slab(J, sl + 1, kb, (sidx + 1) & 1);
                    else if (kb + 1 < nkb) issue_slab(J, 0, kb + 1, (sidx + 1) & 1);
                    else {
                        // the next tile is chosen from the thresholds / insertion counts as of the tile before J (published before
                        // that tile's last barrier and untouched since: the same choice in every wave)
                        Jn = next_tile(1);
                        P8(2)
                        if (Jn >= 0) issue_slab(Jn, 0, 0, (sidx + 1) & 1);
                    }
                    P8(4)
                    mfma_slab(sidx & 1, acc[sl]);
                    // the rows' next block of operands (the same rows' block 0 for the next tile): requested behind the block's last
                    // MFMAs, waited for together with the slab request at the barrier
                    if (sl == 3) (void)load_rows(kb + 1 < nkb ? kb + 1 : 0);
                    P8(0)
                    ++sidx;
                    slab_endk();
                    P8(1)
                }
            }
            // the tile's four slabs: test against the rows' thresholds, survivors in, merge -- slab after slab, each against the
            // lists as the one before left them
            pend = true;
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                pJ = J; pslab = sl; prj = rj4[sl]; ppar = tpar;
                test_only(acc[sl]);
                insert_merge(acc[sl]);
            }
            pend = false;
            publish();
            ++tdone;
            tpar ^= 1;
            P8(6)
            if (Jn < 0) break;
            J = Jn;
        }
        slab_endk();   // (every wave has left the ring and the lists: the selection takes them back)
        P8(6)
    };

    // ---- phase A: the row tile against itself (gives every row K finite candidates); query rows are not part of the
    // data set and start from the ranked tiles directly
    if constexpr (JOIN) {
        const int nu = a.ucount[bt];
        const int nchunks = (nu + ST_T - 1) / ST_T;
        run(nchunks, [&](int q) { return q; }, [&](int) { return 0.f; });
        processed = nchunks;
    } else if (!a.query) {
        run(1, [&](int) { return I; }, [&](int) { return 0.f; });
    }

    if constexpr (!JOIN) {
    // ---- phase B: all other column tiles, exactly as k_st_knn ranks and selects them (streamed.hip): rank key and
    // valid bound of every column tile into a scratch row, then rounds of {3-level radix selection of the next ST_KEEP
    // tiles in (key, tile) order, collect, sort, stream}
    float *skey = a.scr_key + (size_t)bt * a.nt_all;
    float *slb = a.scr_lb + (size_t)bt * a.nt_all;
    if (!a.pre_ranked)   // (k_st_rank_pairs has filled the scratch rows: streamed.hip)
    for (int J = threadIdx.x; J < a.nt_all; J += STBK_THREADS) {
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
    uint32_t *hist = reinterpret_cast<uint32_t *>(&sh.cand_d[0][0]);   // 4096 bins; cand_d is idle between runs
    SelBufK &sb = *reinterpret_cast<SelBufK *>(&sh.ring[0]);           // the ring is idle between runs too
    static_assert(sizeof(sh.cand_d) >= 4096 * sizeof(uint32_t), "histogram does not fit");
    uint32_t done_bits = 0;   // (done_bits, done_j): key bits / index of the last tile already considered
    int done_j = -1;
    for (;;) {
        const float thrmax = thrmax_now();
        uint32_t prefix = 0;
        uint32_t want = ST_KEEP;
        bool all = false;
        for (int level = 0; level < 3 && !all; ++level) {
            const int shift = level == 0 ? 20 : level == 1 ? 8 : 0;
            const int nbins = level == 2 ? 256 : 4096;
            const uint32_t pmask = level == 0 ? 0u : level == 1 ? 0xfff00000u : 0xffffff00u;
            for (int q = threadIdx.x; q < nbins; q += STBK_THREADS) hist[q] = 0;
            __syncthreads();
            for (int J = threadIdx.x; J < a.nt_all; J += STBK_THREADS) {
                const uint32_t kb = __float_as_uint(skey[J]);
                const float lb = slb[J];
                const bool after_done = kb > done_bits || (kb == done_bits && J > done_j);
                if (kb < 0x7f800000u && after_done && lb * lb < thrmax && (kb & pmask) == prefix)
                    atomicAdd(&hist[(kb >> shift) & (nbins - 1)], 1u);
            }
            __syncthreads();
            // first bin whose cumulative count reaches `want`: thread t owns bins [per t, per (t+1)) (threads beyond the
            // bins own none); exclusive scan over the threads, then the owner of the crossing walks its bins
            const int per = nbins >= STBK_THREADS ? nbins / STBK_THREADS : 1;
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
            if (sh.sel_bin < 0) all = true;
            else { prefix |= (uint32_t)sh.sel_bin << shift; want -= sh.sel_before; }
            __syncthreads();
        }
        const uint32_t cut_bits = all ? 0x7f7fffffu : prefix;   // take keys <= cut (ties resolved by the sort below)
        if (threadIdx.x == 0) sh.nsurv = 0;
        __syncthreads();
        for (int J = threadIdx.x; J < a.nt_all; J += STBK_THREADS) {
            const uint32_t kb = __float_as_uint(skey[J]);
            const float lb = slb[J];
            const bool after_done = kb > done_bits || (kb == done_bits && J > done_j);
            if (kb < 0x7f800000u && after_done && lb * lb < thrmax && kb <= cut_bits) {
                const int slot = atomicAdd(&sh.nsurv, 1);
                if (slot < ST_SURV) { sb.surv_lb[slot] = __uint_as_float(kb); sb.surv_vb[slot] = lb; sb.surv_j[slot] = J; }
            }
        }
        __syncthreads();
        int ns = min(sh.nsurv, ST_SURV);
        if (ns == 0) break;
        {   // sort by (rank key, J): bitonic over ST_SURV slots
            for (int q = threadIdx.x; q < ST_SURV; q += STBK_THREADS)
                if (q >= ns) { sb.surv_lb[q] = INFINITY; sb.surv_j[q] = 0x7fffffff; }
            __syncthreads();
            for (int k2 = 2; k2 <= ST_SURV; k2 <<= 1)
                for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
                    for (int q = threadIdx.x; q < ST_SURV; q += STBK_THREADS) {
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
        if (ns > ST_KEEP && more) ns = ST_KEEP;
        const uint32_t round_last_bits = __float_as_uint(sb.surv_lb[ns - 1]);
        const int round_last_j = sb.surv_j[ns - 1];
        // the round's tiles leave the ring before the stream takes it back
        for (int q = threadIdx.x; q < ns; q += STBK_THREADS) { sh.run_j[q] = sb.surv_j[q]; sh.run_vb[q] = sb.surv_vb[q]; }
        __syncthreads();   // hist (cand_d) and the sort buffers (ring) are idle again: the stream may run
        run(ns, [&](int q) { return sh.run_j[q]; }, [&](int q) { return sh.run_vb[q]; });
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
        float *ex = KMAX <= ST_SLAB + 1 ? &sh.cand_d[0][0] : &sh.ring[0];   // [ST_T][KMAX] exact d^2 (cand_d / the ring are idle now)
        for (int q = threadIdx.x; q < ST_T * KL; q += STBK_THREADS) {
            const int row = q / KL, e = q - row * KL;
            const int32_t cc = sh.list_c[row][e];
            float d2 = INFINITY;
            bool known = false;
            if constexpr (JOIN) {
                // (a column the row already listed before this pass: its exact d^2 is in the previous lists -- knnbf.hip)
                if (cc != 0x7fffffff) {
                    const int32_t *ol = a.lists_all + ((size_t)grow0 + row) * K;
                    const float *od = a.out_d2 + ((size_t)bt * ST_T + row) * K;
                    for (int t = 0; t < K; ++t)
                        if (ol[t] == cc) { known = true; d2 = od[t]; }
                }
            }
            if (!known && cc != 0x7fffffff && sh.list_d[row][e] < INFINITY) {
                const float4 *x = reinterpret_cast<const float4 *>(a.Rs + (size_t)(grow0 + row) * dimr);
                const float4 *y = reinterpret_cast<const float4 *>(a.Xs + (size_t)cc * dimr);
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 8
                for (int t = 0; t < dimr / 4; ++t) {
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
        for (int q = threadIdx.x; q < ST_T * K; q += STBK_THREADS) {
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

template <int KMAX> static int launchk(annchor_ctx *c, const KnnArgs &a, bool join)
{
    const size_t lds = sizeof(KnnSharedK<128, KMAX>);
    ANN_REQUIRE(c, lds <= 160 * 1024, ANNCHOR_ELIMIT, "streamed k-NN (k-blocked split-fp16 form) needs %zu B of LDS", lds);
    if (join) {
        ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)k_st_knnbk<KMAX, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_st_knnbk<KMAX, true><<<a.tile_count, STBK_THREADS, lds, c->stream>>>(a);
    } else {
        ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)k_st_knnbk<KMAX, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_st_knnbk<KMAX, false><<<a.tile_count, STBK_THREADS, lds, c->stream>>>(a);
    }
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}

// The tile phase / a join pass for rows of padded dimension 256 .. 1024 (a multiple of 128) when K + 2 <= 32 list entries and the
// split copy of the columns exists; *handled = false otherwise (padded dim 256: the caller's exact-f32 kernel takes it).  Lists of 16 / 32 / 64 entries: up to 62 neighbours.
int ann_stream_launch_knnbk(annchor_ctx *c, const KnnArgs &a0, int dim_padded, bool *handled, bool join)
{
    *handled = false;
    static const bool bk128 = getenv("ANNCHOR_ST_KERNEL") && !strcmp(getenv("ANNCHOR_ST_KERNEL"), "bk");   // A/B: the k-blocked kernel on 128 dimensions (one block)
    if (dim_padded < (bk128 ? 128 : 256) || dim_padded > 1024 || (dim_padded & 127) || a0.K + ST_BF_MARGIN > ST_KMAX_BIG || !a0.Xb || !a0.rsb || !a0.cvec) return ANNCHOR_OK;
    *handled = true;
    KnnArgs a = a0;
    a.dimr = dim_padded;
    // (64-entry lists: 125 KB of LDS, one workgroup per CU -- up to 62 neighbours + self)
    return a.K + ST_BF_MARGIN <= 16 ? launchk<16>(c, a, join) : a.K + ST_BF_MARGIN <= ST_KMAX ? launchk<ST_KMAX>(c, a, join) : launchk<ST_KMAX_BIG>(c, a, join);
}
