// knnh.hip -- the tile phase of the streamed k-NN build as a two-stage filter (k_st_knnh, round 6): fp16 matrix products
// decide which column MAY belong to a row's list, float32 differences decide whether it does.
//
// What the round's measurements say about k_st_knnbf (knnbf.hip): a SIMD issues for ONE of its two waves at a time -- beside an
// MFMA stream only the streaming wave's own vector instructions issue (tools/microbench/pingpong.hip) -- so a slab costs the sum of
// what both waves issue: 768 matrix-pipe cycles (24 MFMAs: hi.hi + hi.lo + lo.hi of the split operands) + ~127 vector
// instructions + ~30 LDS instructions, twice (PMC: 2035 cycles per wave and slab on a SIMD).  Dropping two of the three MFMAs in a
// timing experiment took 15 ms off the 70 ms tile phase: the matrix pipe's share is real time, not hidden.  Hence:
//   * STAGE 1, on the matrix cores: x.y from the fp16 hi halves alone -- 8 MFMAs per 32 x 32 x 128 block instead of 24, half the
//     operand bytes through HBM, L2 and LDS (256 B per column).  The error of that product is bounded: |x.y - hi.hi| <=
//     2^-10 (1 + 2^-11) |x||y| <= beta (|x|^2 + |y|^2) / 2 with beta = 1.05 x 2^-10 (both hi halves are within 2^-11 relative of their
//     floats; the 5 % cover the float32 accumulation of 128 exact products and the arithmetic of the test), so
//         d^2(x, y) < thr   implies   hi.hi - (1 - beta) |y|^2 / 2  >  ((1 - beta) |x|^2 - thr) / 2 :
//     the accumulators start from the column term, the row term is one LDS word per row, the test is ONE v_cmp per row into a wave
//     mask, in the MFMAs' shadow.  A column that fails it is provably not among the row's K nearest so far.
//   * STAGE 2, on the vector ALUs, for what passes (a few columns per wave and 64-column slab once the lists are warm): the exact
//     float32 sum (x - y)^2 of the ORIGINAL rows -- the reference's own arithmetic (np.linalg.norm(x - y), annchor/distances.py:8-13)
//     -- by the whole wave (two coalesced 512-byte loads, a DPP reduction), then a sorted insertion into the row's list (16 lanes
//     hold it; position = popcount of the comparison ballot; a DPP row shift makes room).  The loads of a slab's survivors are
//     issued after its test and consumed one slab later: their latency overlaps a slab of streaming.
// The lists therefore hold EXACT float32 distances at all times: no margin entries, no re-ranking epilogue, no guard -- whatever
// the conditioning of the data, the selection is the reference's.  Ill-conditioned data (|x|^2 >> d^2) only lets more columns
// through stage 1: slower, not wrong.
// Warm start: the first tiles of a row tile -- every column passes until the lists are full and tight -- are k_st_knnbf's
// (knnbf.hip: approximate values straight from the MFMAs, batched merges): the launcher runs it with a budget of STH_WARM
// tiles, this kernel starts from its lists (exact after its epilogue), marks the tiles it evaluated as done and goes on in the
// same ranking order with what is left of the budget.
// Stream mechanics: 64-column slabs of hi halves (16 KB) by LDS-DMA into a ring of three slots; one workgroup barrier per slab.
// After a slab's stream a wave waits for everything it has outstanding -- its pieces of the NEXT slab and the survivor loads, both
// a whole slab old --, evaluates those survivors, and only then requests its pieces of the slab after the next and the new
// survivors' rows: nothing younger than what a wait is for is ever in flight (see vm_wait_all).
#include "streamed.h"

#define STH_THREADS 256
#define ST_BF_MARGIN_H 2   // (the warm-up kernel's list margin: its shapes are this kernel's)
#define STH_COLS 64        // columns per slab
#ifndef STH_Q
#define STH_Q 4            // exact evaluations in flight per wave (deferred by one slab): one batch.  (8: 52.6 ms against 51.4 at C3 --
                           // sixteen more registers and twice the code for a second batch that is rarely full; 12 spills: 87 ms)
#endif
#define STH_QCAP 64        // survivors a stream can queue (one lane of two registers each); more: the slab is rebuilt from the accumulators
#define STH_BETA 1.0254e-3f   // 1.05 x 2^-10
#define STH_SLACK 0.125f      // absolute slack of the test in scaled units (subnormal fp16 halves: <= 2^-25 per coordinate)

typedef _Float16 f16x8h __attribute__((ext_vector_type(8)));
typedef float f32x4h __attribute__((ext_vector_type(4)));

// ST_PROFILE builds: per-wave cycle sums by segment (a.prof[0..7], printed by knn_tile_phase):
//   0 slab barrier   1 stream (operand reads, MFMAs, tests, queueing)   2 wait for the outstanding loads   3 exact evaluations + insertions
//   4 requests (next slab's pieces, survivors' rows)   5 tile choice + publication   6 many survivors: evaluated at once, waited for
//   7 selection rounds, the rest
#ifdef ST_PROFILE
__device__ __forceinline__ long long sth_now()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return (long long)t;
}
#define PH(i) { const long long pf_n = sth_now(); pf[i] += pf_n - pf_t; pf_t = pf_n; }
#elif defined(STH_MARK)
#define PH(i) asm volatile("; STHMARK " #i ::: "memory");
#else
#define PH(i)
#endif

template <int KS> struct KnnSharedH {
    float ring[3 * STH_COLS * 64];   // FIRST (LDS-DMA destinations below 64 KB): three slots of 64 columns x 128 fp16.  Between runs: the
                                     // selection's sort buffers and histogram
    float list_d[ST_T][KS + 1];      // exact d^2 (original units), ascending by (d^2, column)
    int32_t list_c[ST_T][KS + 1];
    float thr[ST_T];                 // the row's K-th exact d^2 (-1: padding row, never a candidate)
    float hb[ST_T];                  // ((1 - beta) |x_row|^2 - thr scale^2) / 2 - slack: the row side of the stage-1 test
    float rrow[ST_T];                // |scale (x_row - c)|^2
    float loI[64], hiI[64], midI[64];
    float run_vb[ST_KEEP];
    int32_t run_j[ST_KEEP];
    uint32_t run_ev[ST_KEEP / 32];
    float wave_thr[2][4];
    int wave_ins[2][4];
    int nsurv, sel_bin;
    uint32_t sel_before;
    int red[4];
};
struct SelBufH {   // candidate tiles of a selection round (aliases the ring, idle between runs)
    float surv_lb[ST_SURV];
    float surv_vb[ST_SURV];
    int32_t surv_j[ST_SURV];
};

template <int KS> __global__ __launch_bounds__(STH_THREADS, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_st_knnh(KnnArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smemh[];
    using Sh = KnnSharedH<KS>;
    using SelBuf = SelBufH;
    Sh &sh = *reinterpret_cast<Sh *>(smemh);
    constexpr int DIM = 128, G = DIM / 16, NI = 4, OPS = NI + 1;
    static_assert(sizeof(sh.ring) <= 65536, "LDS-DMA destinations must stay below 64 KB");
    static_assert(sizeof(SelBuf) == 12288 && sizeof(sh.ring) >= 12288 + 16384, "selection buffers + histogram alias the ring");
    static_assert(KS == 16, "a row's list lives in one 16-lane DPP row");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rg = wave;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smemh;
    int bt;
    {   // XCD-banded row-tile assignment (block b runs on XCD b % 8): neighbours in the k-d order share an L2
        const int nb_ = gridDim.x, q = nb_ >> 3, r = nb_ & 7, x = blockIdx.x & 7, y = blockIdx.x >> 3;
        bt = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    }
    const int I = a.tile_begin + bt;
    const int64_t grow0 = (int64_t)I * ST_T;
    const int K = a.K;
    const int col = lane & 31, half = lane >> 5;
    const int rowbase = rg * 32;
    const int rowq = rowbase + 4 * half;   // C layout: row = rowq + (r & 3) + 8 (r >> 2), column = lane & 31
    // ---- row operand: lane holds row (lane & 31), dimensions 16 g + 8 half .. + 7 of k-step g, fp16 hi halves of the centred, scaled values
    f16x8h ah[G];
    const float scale = a.cvec[DIM];
    const float sc2 = scale * scale;
    float rr_c;
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
                const float x = xu[j] * scale;
                ah[g][j] = (_Float16)x;
                acc2 += x * x;
            }
        }
        rr_c = acc2 + __shfl_xor(acc2, 32);
    }
    auto hb_of = [&](float thr_orig, float rr_s) -> float {
        // (a row whose list is not full yet takes everything: -3e38, not -inf -- the accumulators start from column term - row term, and
        // a padding column's term is -inf)
        return thr_orig < 0.f ? INFINITY : (thr_orig < INFINITY ? 0.5f * ((1.f - STH_BETA) * rr_s - thr_orig * sc2) - STH_SLACK : -3.0e38f);
    };
    if (threadIdx.x < ST_T) {
        // the lists as the warm-up left them (k_st_knnbf's epilogue: exact d^2, original units)
        const int row = threadIdx.x;
        const bool real = a.rr[grow0 + row] < INFINITY;
        // (all of a row's entries requested before the first is looked at: one entry at a time, 2 x 16 dependent global round trips
        // opened every workgroup)
        float dv[KS];
        int32_t cv[KS];
        const float *pd = a.out_d2 + ((size_t)bt * ST_T + row) * K;
        const int32_t *pc = a.out_col + ((size_t)bt * ST_T + row) * K;
#pragma unroll
        for (int q = 0; q < KS; ++q) { dv[q] = pd[min(q, K - 1)]; cv[q] = pc[min(q, K - 1)]; }
        float last = INFINITY;
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            const bool have = q < K;
            sh.list_d[row][q] = have ? dv[q] : INFINITY;
            sh.list_c[row][q] = have ? cv[q] : 0x7fffffff;
            if (have) last = dv[q];
        }
        sh.thr[row] = real ? last : -1.f;
    }
    if (lane < 32) sh.rrow[rowbase + lane] = a.rr[grow0 + rowbase + lane] < INFINITY ? rr_c : INFINITY;
    if ((int)threadIdx.x < a.na) {
        sh.loI[threadIdx.x] = a.rlo[(size_t)threadIdx.x * a.nt_r + I];
        sh.hiI[threadIdx.x] = a.rhi[(size_t)threadIdx.x * a.nt_r + I];
        sh.midI[threadIdx.x] = a.rmid[(size_t)threadIdx.x * a.nt_r + I];
    }
    if (threadIdx.x < 8) sh.wave_ins[threadIdx.x >> 2][threadIdx.x & 3] = 0;
    if (threadIdx.x == 0) sh.nsurv = 0;
    if (threadIdx.x < 4) sh.red[threadIdx.x] = 0;
    // ---- the column tiles the warm-up evaluated: never candidates again (their rank key becomes +inf), counted against the budget
    float *skey = a.scr_key + (size_t)bt * a.nt_all;
    float *slb = a.scr_lb + (size_t)bt * a.nt_all;
    uint32_t *ebits = a.eval_bits ? a.eval_bits + (size_t)bt * a.eval_words : nullptr;
    __syncthreads();
    {
        int mine = 0;
        if (ebits)
            for (int w = threadIdx.x; w < a.eval_words; w += STH_THREADS) {
                uint32_t bits = ebits[w];
                mine += __popc(bits);
                while (bits) {
                    const int b = __builtin_ctz(bits);
                    bits &= bits - 1;
                    if (32 * w + b < a.nt_all) skey[32 * w + b] = INFINITY;
                }
            }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
        if (lane == 0) sh.red[wave] = mine;
    }
    if (threadIdx.x < ST_T) sh.hb[threadIdx.x] = hb_of(sh.thr[threadIdx.x], sh.rrow[threadIdx.x]);
    __syncthreads();
    int processed = sh.red[0] + sh.red[1] + sh.red[2] + sh.red[3];   // column tiles evaluated so far (uniform), warm-up included
    const int processed0 = processed;
    {   // the thresholds the first choice of a tile reads
        float t = lane < 32 ? sh.thr[rowbase + lane] : -1.f;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) t = fmaxf(t, __shfl_xor(t, off));
        if (lane == 0) sh.wave_thr[0][rg] = t;
    }
#ifdef ST_PROFILE
    long long pf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long pf_t = sth_now();
#endif
    int ins = 0;         // list insertions of this wave (uniform)
    int tdone = 0;       // column tiles completed and published by this kernel (uniform); tile n publishes into slot (n + 1) & 1
    int win_start = processed, win_ins = 0;
    bool dried = false;
    __syncthreads();

    // ---------------------------------------------------------------- requests
    uint32_t loff[NI];   // piece i of this wave: 64 units of 16 bytes, unit U = (rg NI + i) 64 + lane = (column U / 16, unit U % 16)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int u = (rg * NI + i) * 64 + lane;
        const int c = u >> 4, x = u & 15;
        // (the hi half of column c starts its 512-byte row of the split copy; piece i is requested with the instruction offset 1024 i,
        // which moves its LDS destination AND its source: taken off here -- c >= 4 i, so the difference stays >= 0)
        loff[i] = (uint32_t)(c * (DIM * 4) + ((x ^ (c & 15)) << 4)) - 1024u * i;
    }
    const uint32_t coloff = (uint32_t)col * 4;   // (the norms' loads: an offset register nothing ever overwrites, see issue)
    const char *xb = reinterpret_cast<const char *>(a.Xb);
    auto scalar_ptr = [](const void *ptr) -> const char * {
        const uint64_t v = (uint64_t)(uintptr_t)ptr;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return reinterpret_cast<const char *>((uintptr_t)(((uint64_t)hi << 32) | lo));
    };
    auto scalar_ptrf = [](const float *ptr) -> const float * {
        const uint64_t v = (uint64_t)(uintptr_t)ptr;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return reinterpret_cast<const float *>((uintptr_t)(((uint64_t)hi << 32) | lo));
    };
    int slot_cur = 0;   // ring slot of the slab being streamed (uniform): 0, 1, 2, 0 ..; the slab after the next goes to the slot before it
    // the columns' squared norms (scaled, centred) come as plain loads with the slab's requests: (nl0, nl1) the newest request's,
    // (nn0, nn1) the next slab's, moved there after the wait that covers them (lane: columns col and 32 + col)
    float nn0 = 0.f, nn1 = 0.f, nl0 = 0.f, nl1 = 0.f;
    // (buffer loads, executed on every path -- beyond the last tile the offset is out of range: zeros, no access --: a load the compiler
    // sees must not sit inside a branch, see issue_slots; an inline-asm load's destination would be the compiler's to copy before the
    // data has landed)
    const __amdgpu_buffer_rsrc_t srd_rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(scalar_ptrf(a.rsb)), 0, a.nt_all * ST_T * 4, 0x00020000);
    auto norms_of = [&](int Jv, int slab, float &d0, float &d1) __attribute__((always_inline)) {
        const uint32_t off = Jv >= 0 ? (uint32_t)((Jv * ST_T + slab * STH_COLS) * 4) + coloff : 0xFFFFFF00u;
        d0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_rsb, off, 0, 0));
        d1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_rsb, off + 128, 0, 0));
    };
    auto issue = [&](int Jv, int slab, int slotv) __attribute__((always_inline)) {
        const int J = __builtin_amdgcn_readfirstlane(Jv);
        const int slot = __builtin_amdgcn_readfirstlane(slotv);
        const char *src = scalar_ptr(xb + ((size_t)J * ST_T + slab * STH_COLS) * (DIM * 4));
        const uint32_t dst = lds0 + (uint32_t)(slot * (STH_COLS * 256) + rg * NI * 1024);
        // What a wave does AFTER these requests must not depend on the memory pipeline having taken them: a profile of the first form
        // showed ~900 cycles per slab between the last request and the barrier -- m0 rewritten between the four LDS-DMA instructions
        // and restored after them (each write waits until the instruction before it has been dispatched), and the two norm loads'
        // address registers, temporaries the next vector instructions overwrote (a write-after-read wait on a load still queued
        // behind the four DMA instructions).  Now: the norm loads apart (norms_of below: buffer loads on every path);
        // m0 written ONCE (nothing else in the kernel uses it: not restored), the four destinations by the instruction offset.
        asm volatile("s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %4\n\t"
                     "global_load_lds_dwordx4 %1, %4 offset:1024\n\t"
                     "global_load_lds_dwordx4 %2, %4 offset:2048\n\t"
                     "global_load_lds_dwordx4 %3, %4 offset:3072"
                     : : "v"(loff[0]), "v"(loff[1]), "v"(loff[2]), "v"(loff[3]), "s"(src), "s"(dst) : "memory");   // (m0: no other user in this kernel -- checked in the ISA)
    };

    // ---------------------------------------------------------------- stage 2: exact evaluation + insertion
    // Survivors are evaluated FOUR at a time, one per 16-lane row of the wave: lane e of row g holds dimensions 8 e .. 8 e + 7 of
    // survivor g's row and column (two 16-byte loads each: the 16 lanes read 512 contiguous bytes), the 16-lane sum is four DPP
    // steps and leaves d^2 in every lane of the row, which also holds that survivor's list -- the insertion needs nothing from a
    // scalar register.  (First form: one survivor per wave pass, 64 lanes x 2 dimensions, a six-step reduction, v_readlane, ~58
    // vector instructions per survivor; the wave's vector instructions, not the matrix pipe, were what a slab cost.)  STH_Q / 4
    // such batches are in flight.
    constexpr int NB = STH_Q / 4;
    f32x4h xa[NB], xc[NB], ya[NB], yc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) { xa[b] = f32x4h{0.f, 0.f, 0.f, 0.f}; xc[b] = xa[b]; ya[b] = xa[b]; yc[b] = xa[b]; }
    int qpk = 0;                   // the stream's queue: lane n = survivor n of the slab under test, row in the tile | column in the slab << 8
    int nq = 0;                    // (uniform)
    int fpk = 0, fbase = 0, nfl = 0;   // the evaluations in flight (their slab's first global column)
    bool hq_stale = true;
    bool tm_stale = true;   // (uniform) an insertion since the wave's worst K-th distance was last taken
    float wave_tm = 0.f;    // (uniform) that distance
    float hqr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) hqr[r] = 0.f;
    const int e16 = lane & 15, g16 = lane >> 4;
    // the lane's survivor of batch `bi` of the queue QP starting at entry b: (valid, row, column)
    auto my_survivor = [&](int b, int n, int bi, int QP, int QB, int &row, int32_t &cc) -> bool {
        const int idx = b + 4 * bi + g16;
        const int pk = __builtin_amdgcn_ds_bpermute(idx << 2, QP);
        const bool valid = 4 * bi + g16 < n;
        row = valid ? (pk & 0xff) : 0;                 // (lanes without a survivor read row 0 / the slab's first column: valid memory, ignored)
        cc = valid ? QB + (pk >> 8) : QB;
        return valid;
    };
    // loads of survivors b .. b + n - 1 of the queue (QP: packed entries, QB: their slab's first column), n <= STH_Q
    // These loads are executed whether or not there is a survivor, as BUFFER loads: a lane without one asks for an offset beyond
    // its descriptor's range and gets zeros without a memory access.  As plain loads behind `if (n > 0)` they made the compiler
    // copy the four registers where the branches meet and put `s_waitcnt vmcnt(0)` in front of the copies -- on EVERY path, right
    // behind the slab's LDS-DMA requests: each wave waited out the memory latency of the slab it had just requested, every slab
    // (the profile's "requests" segment: ~900 of a slab's ~4700 ticks, and the waves it kept from the barrier).  No branch around a
    // load, no meeting point: the compiler waits for them where they are read, after vm_wait_all.
    // (rows: the row tile's 64 KB of a.Rs; columns: the survivors' slab, 64 columns of a.Xs from QB on)
    const __amdgpu_buffer_rsrc_t srd_rows = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(scalar_ptrf(a.Rs + (size_t)grow0 * DIM)), 0, ST_T * DIM * 4, 0x00020000);
    auto issue_slots = [&](int b, int n, int QP, int QB) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t srd_cols = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(scalar_ptrf(a.Xs + (size_t)__builtin_amdgcn_readfirstlane(QB) * DIM)), 0, STH_COLS * DIM * 4, 0x00020000);
#pragma unroll
        for (int bi = 0; bi < NB; ++bi) {
            uint32_t vx = 0xFFFFFF00u, vy = 0xFFFFFF00u;   // (out of range: zeros, no access)
            if (__builtin_expect(4 * bi < n, 0)) {   // (uniform)
                const int idx = b + 4 * bi + g16;
                const int pk = __builtin_amdgcn_ds_bpermute(idx << 2, QP);
                const bool valid = 4 * bi + g16 < n;
                vx = valid ? (uint32_t)((pk & 0xff) * (DIM * 4) + 32 * e16) : vx;
                vy = valid ? (uint32_t)((pk >> 8) * (DIM * 4) + 32 * e16) : vy;
            }
            xa[bi] = __builtin_bit_cast(f32x4h, __builtin_amdgcn_raw_buffer_load_b128(srd_rows, vx, 0, 0));
            xc[bi] = __builtin_bit_cast(f32x4h, __builtin_amdgcn_raw_buffer_load_b128(srd_rows, vx + 16, 0, 0));
            ya[bi] = __builtin_bit_cast(f32x4h, __builtin_amdgcn_raw_buffer_load_b128(srd_cols, vy, 0, 0));
            yc[bi] = __builtin_bit_cast(f32x4h, __builtin_amdgcn_raw_buffer_load_b128(srd_cols, vy + 16, 0, 0));
        }
    };
    // one round of insertions: row `row`, exact d^2 `d2` (+inf: nothing), column cc -- per 16-lane row, all four rows at once
    auto insert_round = [&](int row, float d2, int32_t cc) __attribute__((always_inline)) {
        const float ld = e16 < K ? sh.list_d[row][e16] : INFINITY;
        const int32_t lc = e16 < K ? sh.list_c[row][e16] : 0x7fffffff;
        const bool before = e16 < K && (ld < d2 || (ld == d2 && lc < cc));
        const unsigned long long mk = __ballot(before);
        const int pos = __popc((uint32_t)(mk >> (lane & 48)) & 0xffffu);   // entries of the lane's own row that stay ahead
        const bool ok = pos < K && d2 < INFINITY;
        const float pd = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ld), 0x111, 0xf, 0xf, false));   // row_shr:1
        const int32_t pc = __builtin_amdgcn_update_dpp(0, lc, 0x111, 0xf, 0xf, false);
        const float nd = e16 > pos ? pd : (e16 == pos ? d2 : ld);
        const int32_t nc = e16 > pos ? pc : (e16 == pos ? cc : lc);
        if (ok && e16 >= pos && e16 < K) { sh.list_d[row][e16] = nd; sh.list_c[row][e16] = nc; }
        if (ok && e16 == K - 1) { sh.thr[row] = nd; sh.hb[row] = hb_of(nd, sh.rrow[row]); }
        {   // (uniform: the wave's insertion count and "its rows' thresholds have changed" -- nothing to gather at the end of a tile)
            const unsigned long long okm = __ballot(ok && e16 == 0);
            ins += __popcll(okm);
            if (okm) { hq_stale = true; tm_stale = true; }
        }
    };
    // slots of n survivors (b .. of the queue): exact d^2, sorted insertion
    auto consume_slots = [&](int b, int n, int QP, int QB) __attribute__((always_inline)) {
#pragma unroll
        for (int bi = 0; bi < NB; ++bi)
            if (__builtin_expect(4 * bi < n, 0)) {   // (uniform)
                int row; int32_t cc;
                const bool valid = my_survivor(b, n, bi, QP, QB, row, cc);
                float t;
                {
                    const float d0 = xa[bi].x - ya[bi].x, d1 = xa[bi].y - ya[bi].y, d2_ = xa[bi].z - ya[bi].z, d3 = xa[bi].w - ya[bi].w;
                    const float d4 = xc[bi].x - yc[bi].x, d5 = xc[bi].y - yc[bi].y, d6 = xc[bi].z - yc[bi].z, d7 = xc[bi].w - yc[bi].w;
                    t = ((d0 * d0 + d1 * d1) + (d2_ * d2_ + d3 * d3)) + ((d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7));
                }
                // 16-lane sum by DPP (xor 1, xor 2 inside the quads, mirrors inside 8 and 16 lanes): every lane of the row has it
                t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xf, 0xf, false));
                t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x4E, 0xf, 0xf, false));
                t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x141, 0xf, 0xf, false));
                t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x140, 0xf, 0xf, false));
                const float d2 = valid ? t : INFINITY;
                // two survivors of a batch with the same row must not insert at once: each would shift a list it has not seen the
                // other's entry in.  (Rare once the lists are warm.)  Then the four rows take turns.
                const int r0 = __builtin_amdgcn_readlane(row, 0), r1 = __builtin_amdgcn_readlane(row, 16), r2 = __builtin_amdgcn_readlane(row, 32),
                          r3 = __builtin_amdgcn_readlane(row, 48);
                const int nb = min(4, n - 4 * bi);
                const bool clash = (nb > 1 && r0 == r1) || (nb > 2 && (r0 == r2 || r1 == r2)) || (nb > 3 && (r0 == r3 || r1 == r3 || r2 == r3));
                if (__builtin_expect(!clash, 1)) {   // (uniform)
                    insert_round(row, d2, cc);
                } else {
                    for (int g = 0; g < nb; ++g) {
                        insert_round(row, g16 == g ? d2 : INFINITY, cc);
                        wave_fence_lds();
                    }
                }
            }
    };
    // Every vector-memory wait of the kernel is "all of them", as the BUILTIN: the compiler's own wait bookkeeping sees it (it does
    // not see the LDS-DMA requests, inline asm; a wait it does not know of leaves the survivors' loads "pending" in its books and it
    // then waits for EVERYTHING -- the requests just made included -- before it reuses a register).  So nothing younger than what is
    // waited for may be outstanding at a wait: the requests of a slab are made AFTER the wait of the slab before.
    auto vm_wait_all = [&]() __attribute__((always_inline)) { __builtin_amdgcn_s_waitcnt(0x0F70); };   // vmcnt(0), expcnt / lgkmcnt untouched
    // everything in the queue QP[0 .. n), now: batches of STH_Q, each waited for
    auto drain_sync = [&](int n, int QP, int QB) __attribute__((always_inline)) {
        for (int b = 0; b < n; b += STH_Q) {
            const int m = min(STH_Q, n - b);
            issue_slots(b, m, QP, QB);
            vm_wait_all();
            consume_slots(b, m, QP, QB);
        }
    };
    // a survivor mask of the slab under test (lanes = columns of group g2, both half-waves' rows r) into the queue
    int pc0 = 0;   // first global column of the slab under test (uniform)
    auto push = [&](unsigned long long m, int g2, int r) __attribute__((always_inline)) {
        while (m) {   // (uniform)
            const int l = (int)__builtin_ctzll(m);
            m &= m - 1;
            const int row = rowbase + 4 * (l >> 5) + (r & 3) + 8 * (r >> 2);
            const int cl = 32 * g2 + (l & 31);
            if (!a.query && (int64_t)pc0 + cl == grow0 + row) continue;   // a point is not its own neighbour
            qpk = lane == nq ? (row | (cl << 8)) : qpk;   // (v_writelane cannot take value and lane from two scalar registers)
            ++nq;
        }
    };

    // ---------------------------------------------------------------- stage 1: one slab
    // 16 MFMAs of this wave's 32 rows against the 64 columns in ring slot `slot` into (c0, c1), which start from (column term - row
    // term); the test of the slab before (accumulators p0, p1, first column pc0) in their shadow: the running maximum of each
    // lane's 32 accumulators (v_max3), one compare against zero at the end; the survivors of a slab that has any are queued
    auto stream = [&](int slot, f32x16 &c0, f32x16 &c1, const f32x16 &p0, const f32x16 &p1, bool pend) __attribute__((always_inline)) {
        const float rj0 = nn0, rj1 = nn1;
        if (__builtin_expect(hq_stale, 0)) {   // (uniform) the thresholds of the lane's 16 rows: read again after an insertion of this wave
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 h4 = *reinterpret_cast<const float4 *>(&sh.hb[rowq + 8 * q]);
                hqr[4 * q] = h4.x; hqr[4 * q + 1] = h4.y; hqr[4 * q + 2] = h4.z; hqr[4 * q + 3] = h4.w;
            }
            hq_stale = false;
        }
        const float4 *base0 = reinterpret_cast<const float4 *>(&sh.ring[slot * (STH_COLS * 64)]) + col * 16;
        const float4 *base1 = base0 + 32 * 16;
        const int gsw = half ^ (col & 15);
        float4 b0[G], b1[G];
#pragma unroll
        for (int g = 0; g < G; ++g) b0[g] = base0[(2 * g) ^ gsw];
#pragma unroll
        for (int g = 0; g < G; ++g) b1[g] = base1[(2 * g) ^ gsw];
        const float n0 = -0.5f * (1.f - STH_BETA) * rj0, n1 = -0.5f * (1.f - STH_BETA) * rj1;
        float mx0 = 0.f, mx1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) c0[r] = n0 - hqr[r];   // column term - row term: the test of this slab is "any accumulator > 0"
#pragma unroll
        for (int m = 0; m < 2 * G; m += 2) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int mm = m + h2;
                if (mm < G) c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mm], __builtin_bit_cast(f16x8h, b0[mm]), c0, 0, 0, 0);
                else c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mm - G], __builtin_bit_cast(f16x8h, b1[mm - G]), c1, 0, 0, 0);
                if (mm == 0) {   // the second group's accumulators start in the first MFMAs' shadow
#pragma unroll
                    for (int r = 0; r < 16; ++r) c1[r] = n1 - hqr[r];
                }
            }
            {   // two steps of each group's running maximum behind this pair of MFMAs (a run's first slab: zeros)
                const int k = m >> 1;   // 0 .. 7
                if (k == 0) { mx0 = __builtin_fmaxf(__builtin_fmaxf(p0[0], p0[1]), p0[2]); mx1 = __builtin_fmaxf(__builtin_fmaxf(p1[0], p1[1]), p1[2]); }
                else if (k < 7) { mx0 = __builtin_fmaxf(__builtin_fmaxf(mx0, p0[2 * k + 1]), p0[2 * k + 2]); mx1 = __builtin_fmaxf(__builtin_fmaxf(mx1, p1[2 * k + 1]), p1[2 * k + 2]); }
                else mx0 = __builtin_fmaxf(__builtin_fmaxf(mx0, p0[15]), __builtin_fmaxf(mx1, p1[15]));
            }
            if (m < G) asm volatile("" : "+v"(c0));
            else asm volatile("" : "+v"(c1));
        }
        // (86 % of the slabs have no survivor at all once the lists are warm -- 0.43 per wave and slab at C3 --: ONE compare and one
        // branch per slab; the rows' masks are taken behind it.  First form: a v_cmp per row into a scalar pair, four masks ORed per
        // branch -- 32 compares, 24 s_or, 8 branches per slab, and the masks kept alive for the cold blocks spilled scalar registers
        // into vector lanes in the hot loop.)
        if (pend && __builtin_expect(__ballot(mx0 > 0.f) != 0, 0)) {
#pragma unroll
            for (int ti = 0; ti < 32; ++ti) {
                const int g2 = ti >> 4, r = ti & 15;
                const unsigned long long mk = __ballot((g2 ? p1[r] : p0[r]) > 0.f);
                if (mk) push(mk, g2, r);
            }
        }
    };
    // after a slab's stream: the survivor loads of one slab ago (and this wave's pieces of the next slab) have landed; evaluate
    // them; then the survivors the stream has just queued: few -> their loads now, their evaluation after the next slab;
    // many -> all of them now; more than the queue holds -> the slab is gone through again from the accumulators
    auto post = [&](int dJ, int dslab, const f32x16 &p0, const f32x16 &p1, bool pend) __attribute__((always_inline)) {
        PH(1)
        vm_wait_all();   // this wave's pieces of the next slab (requested a slab ago), its norms and the survivors' rows (likewise)
        PH(2)
        consume_slots(0, nfl, fpk, fbase);
        nfl = 0;
        PH(3)
        nn0 = nl0; nn1 = nl1;   // (landed: the next slab's norms)
        norms_of(dJ, dslab, nl0, nl1);
        if (dJ >= 0) issue(dJ, dslab, slot_cur == 0 ? 2 : slot_cur - 1);   // (that slot held the slab before this one: every wave has passed this slab's barrier)
        if (!pend) nq = 0;   // (a run's first slab: nothing was under test)
        int n_async = 0;     // survivors whose rows are requested now and evaluated after the next slab
        if (__builtin_expect(nq > STH_QCAP, 0)) {
            // (the first tiles after a cold start, ill-conditioned data) per-lane row masks from the accumulators, queue by queue
            nq = 0;
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                uint32_t lp = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) lp |= ((g2 ? p1[r] : p0[r]) > 0.f ? 1u : 0u) << r;
                unsigned long long lanes = __ballot(lp != 0);
                while (lanes) {   // (uniform)
                    const int l = (int)__builtin_ctzll(lanes);
                    lanes &= lanes - 1;
                    uint32_t pl = (uint32_t)__builtin_amdgcn_readlane((int)lp, l);
                    while (pl) {
                        const int r = __builtin_ctz(pl);
                        pl &= pl - 1;
                        const int row = rowbase + 4 * (l >> 5) + (r & 3) + 8 * (r >> 2);
                        const int cl = 32 * g2 + (l & 31);
                        if (!a.query && (int64_t)pc0 + cl == grow0 + row) continue;
                        qpk = lane == nq ? (row | (cl << 8)) : qpk;
                        if (++nq == STH_QCAP) { drain_sync(nq, qpk, pc0); nq = 0; }
                    }
                }
            }
            drain_sync(nq, qpk, pc0);
            PH(6)
        } else if (__builtin_expect(nq > STH_Q, 0)) {
            PH(4)
            drain_sync(nq, qpk, pc0);
            PH(6)
        } else {
            n_async = nq;
        }
        // (ONE place, behind the branches: the four registers are defined here on every path -- nothing to merge, nothing to copy)
        issue_slots(0, n_async, qpk, pc0);
        fpk = qpk; fbase = pc0; nfl = n_async;
        nq = 0;
        PH(4)
    };
    // the wave's insertion count and its rows' worst K-th distance, at the end of a tile: both uniform, the distance taken again only
    // after an insertion (one tile in five once the lists are warm).  (First form: every tile an LDS read of the 32 thresholds, eight DPP
    // steps, six v_readlane -- with next_tile's own round trips a fifth of a slab's time.)
    auto publish = [&]() __attribute__((always_inline)) {
        if (__builtin_expect(tm_stale, 0)) {   // (uniform)
            float t = lane < 32 ? sh.thr[rowbase + lane] : -1.f;
            t = fmaxf(t, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, t), __builtin_bit_cast(int, t), 0xB1, 0xf, 0xf, false)));
            t = fmaxf(t, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, t), __builtin_bit_cast(int, t), 0x4E, 0xf, 0xf, false)));
            t = fmaxf(t, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, t), __builtin_bit_cast(int, t), 0x141, 0xf, 0xf, false)));
            t = fmaxf(t, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, t), __builtin_bit_cast(int, t), 0x140, 0xf, 0xf, false)));
            wave_tm = fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t), 0)),
                            __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t), 16)));
            tm_stale = false;
        }
        if (lane == 0) { sh.wave_ins[(tdone + 1) & 1][wave] = ins; sh.wave_thr[(tdone + 1) & 1][rg] = wave_tm; }
    };
    auto thrmax_now = [&]() {
        const float *w = sh.wave_thr[tdone & 1];
        return fmaxf(fmaxf(w[0], w[1]), fmaxf(w[2], w[3]));
    };
    // (tried, twice -- here and in knnbf.hip's three-slot form: per-wave progress flags in LDS instead of the barrier, so that the waves
    // need not meet after the part of a slab whose length differs between them, the survivors.  63.8 ms against 55.0: a polling wave
    // keeps issuing -- LDS reads, compares, s_sleep -- on a SIMD whose other wave could use every slot, a wave at s_barrier does not.)
    auto slab_barrier = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    f32x16 a0, a1, b0a, b1a;   // the two 32-column groups of the even / odd slab of a tile
    // One run of the stream over a list of tiles in rank order: entry q is (tile jl(q), valid bound vb(q)); entries whose bound has
    // fallen behind the thresholds are skipped.  Uniform: every wave takes the same path.
    auto run = [&](int ns, auto jl, auto vb) __attribute__((always_inline)) {
        int q = 0;
        if (threadIdx.x == 0)
            for (int t = 0; t < ST_KEEP / 32; ++t) sh.run_ev[t] = 0;
        // (entry q of the list is read when entry q - 1 is taken: the choice of a tile then waits for the thresholds alone, not for
        // a chain of LDS round trips)
        int cj = ns > 0 ? jl(0) : 0;
        float cvb = ns > 0 ? vb(0) : 0.f;
        auto next_tile = [&](int in_stream) -> int {
            if (a.early_window > 0 && !dried) {
                const int done = processed - in_stream;
                if (__builtin_expect(done - win_start >= a.early_window, 0)) {
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
                const int J = cj;
                const float lb = cvb;
                ++q;
                {   // (read whether or not there is a next entry: a conditional read is waited for where it is made)
                    const int qn = min(q, ns - 1);
                    cj = jl(qn); cvb = vb(qn);
                }
                if (lb * lb < tm) {
                    ++processed;
                    if (ebits && threadIdx.x == 0) atomicOr(&sh.run_ev[(q - 1) >> 5], 1u << ((q - 1) & 31));   // (flushed at the end of the run; no value comes back: nothing to wait for)
                    return __builtin_amdgcn_readfirstlane(J);
                }
            }
            return -1;
        };
        PH(7)
        int J = next_tile(0);
        if (J < 0) return;
        // fill: both slabs of the first tile (the ring is idle: a workgroup barrier precedes every run)
        norms_of(J, 0, nn0, nn1);
        norms_of(J, 1, nl0, nl1);
        issue(J, 0, slot_cur);
        issue(J, 1, slot_cur == 2 ? 0 : slot_cur + 1);
        vm_wait_all();
        bool pend = false;
        nq = 0; nfl = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) { b0a[r] = 0.f; b1a[r] = 0.f; }   // (the first slab's "slab before": nothing passes)
        for (;;) {
            // ---- slab 0: the tile after J is chosen (thresholds / insertion counts as published at the end of the tile before J)
            // and its first slab requested
            PH(5)
            slab_barrier();
            PH(0)
            // (the choice of the next tile: after the stream -- its LDS round trips then find the LDS quiet -- not in front of it;
            // tried between the stream's operand reads and its MFMAs: 49.0 ms against 47.8)
            stream(slot_cur, a0, a1, b0a, b1a, pend);
            PH(1)
            const int Jn = next_tile(1);
            PH(5)
            post(Jn, 0, b0a, b1a, pend);
            pend = true; pc0 = J * ST_T;
            slot_cur = slot_cur == 2 ? 0 : slot_cur + 1;
            // ---- slab 1
            slab_barrier();
            PH(0)
            stream(slot_cur, b0a, b1a, a0, a1, true);
            post(Jn, 1, a0, a1, true);
            pc0 = J * ST_T + STH_COLS;
            slot_cur = slot_cur == 2 ? 0 : slot_cur + 1;
            publish();
            ++tdone;
            if (Jn < 0) break;
            J = Jn;
        }
        // ---- tail: the evaluations in flight, then the last slab's test without a stream to hide it in
        vm_wait_all();
        consume_slots(0, nfl, fpk, fbase);
        nfl = 0;
        nq = STH_QCAP + 1;   // (the slab is gone through from its accumulators)
        post(-1, 0, b0a, b1a, true);
        publish();
        ++tdone;
        __syncthreads();
        if (ebits)
            for (int t = threadIdx.x; t < ns; t += STH_THREADS)
                if ((sh.run_ev[t >> 5] >> (t & 31)) & 1u) {
                    const int Jt = jl(t);
                    atomicOr(&ebits[Jt >> 5], 1u << (Jt & 31));
                }
    };

    // ---- the remaining column tiles, exactly as k_st_knnbf ranks and selects them (knnbf.hip; the same code): rounds of {3-level
    // radix selection of the next ST_KEEP tiles in (key, tile) order, collect, sort, stream}.  (The scratch rows were filled by
    // k_st_rank_pairs -- the launcher insists --; the in-kernel ranking below is kept for the same reason it is there.)
    if (!a.pre_ranked)   // (k_st_rank_pairs has filled the scratch rows: streamed.hip)
    for (int J = threadIdx.x; J < a.nt_all; J += STH_THREADS) {
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
    uint32_t *hist = reinterpret_cast<uint32_t *>(&sh.ring[12288 / 4]);   // 4096 bins, behind the sort buffers in the idle ring
    SelBuf &sb = *reinterpret_cast<SelBuf *>(&sh.ring[0]);           // the ring is idle between runs too
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
            for (; q + (SW - 1) * STH_THREADS < cl_n; q += SW * STH_THREADS) {
                uint32_t kb[SW], lbb[SW], jj[SW];
#pragma unroll
                for (int u = 0; u < SW; ++u) { kb[u] = clk[q + u * STH_THREADS]; lbb[u] = clk[ST_CL_CAP + q + u * STH_THREADS]; jj[u] = clk[2 * ST_CL_CAP + q + u * STH_THREADS]; }
#pragma unroll
                for (int u = 0; u < SW; ++u) f(kb[u], __uint_as_float(lbb[u]), (int)jj[u]);
            }
            for (; q < cl_n; q += STH_THREADS) f(clk[q], __uint_as_float(clk[ST_CL_CAP + q]), (int)clk[2 * ST_CL_CAP + q]);
        } else {
            int J = threadIdx.x;
            for (; J + (SW - 1) * STH_THREADS < a.nt_all; J += SW * STH_THREADS) {
                uint32_t kb[SW];
                float lbv[SW];
#pragma unroll
                for (int u = 0; u < SW; ++u) { kb[u] = __float_as_uint(skey[J + u * STH_THREADS]); lbv[u] = slb[J + u * STH_THREADS]; }
#pragma unroll
                for (int u = 0; u < SW; ++u) f(kb[u], lbv[u], J + u * STH_THREADS);
            }
            for (; J < a.nt_all; J += STH_THREADS) f(__float_as_uint(skey[J]), slb[J], J);
        }
    };
    // first bin whose cumulative count reaches `want` among nbins bins of `hist`: thread t owns bins [per t, per (t+1)); the
    // bin in sh.sel_bin (-1: the total stays below `want`), the count before it in sh.sel_before
    auto find_bin = [&](int nbins, uint32_t want) {
        const int per = nbins >= STH_THREADS ? nbins / STH_THREADS : 1;
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
                for (int q = threadIdx.x; q < 4096; q += STH_THREADS) hist[q] = 0;
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
                for (int q = threadIdx.x; q < nbins; q += STH_THREADS) hist[q] = 0;
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
            for (int q = threadIdx.x; q < sortn; q += STH_THREADS)
                if (q >= ns) { sb.surv_lb[q] = INFINITY; sb.surv_j[q] = 0x7fffffff; }
            __syncthreads();
            for (int k2 = 2; k2 <= sortn; k2 <<= 1)
                for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
                    for (int q = threadIdx.x; q < sortn; q += STH_THREADS) {
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
        for (int q = threadIdx.x; q < ns; q += STH_THREADS) { sh.run_j[q] = sb.surv_j[q]; sh.run_vb[q] = sb.surv_vb[q]; }
        __syncthreads();   // the histogram and the sort buffers (both in the ring) are idle again: the stream may run
        run(ns, [&](int q) { return sh.run_j[q]; }, [&](int q) { return sh.run_vb[q]; });
        done_bits = round_last_bits;
        done_j = round_last_j;
        __syncthreads();
        if (dried) break;
        if (processed >= a.max_tiles) break;
        if (!more) break;   // the selection saw every eligible tile
    }
    __syncthreads();
    // ---- the lists ARE the result: exact float32 d^2 (original units), ascending by (d^2, column)
    for (int q = threadIdx.x; q < ST_T * K; q += STH_THREADS) {
        const int row = q / K, e = q - row * K;
        const float d2 = sh.list_d[row][e];
        a.out_d2[((size_t)bt * ST_T + row) * K + e] = d2;
        a.out_col[((size_t)bt * ST_T + row) * K + e] = d2 < INFINITY ? sh.list_c[row][e] : 0x7fffffff;
    }
    if (threadIdx.x == 0) atomicAdd(a.evals, (unsigned long long)(processed - processed0));
#ifdef ST_PROFILE
    PH(7)
    if (lane == 0 && a.prof)
        for (int i = 0; i < 8; ++i) atomicAdd(a.prof + i, (unsigned long long)pf[i]);
#endif
}

// The tile phase through the two-stage kernel when the shape fits it: graph builds (not queries) at padded dimension 128 with at most
// 14 neighbours kept per row, the split copy present; *handled = false sends the caller to k_st_knnbf alone.  Two launches:
// k_st_knnbf (through `warm`) with a budget of STH_WARM + 1 tiles per row tile and the evaluated tiles recorded, then k_st_knnh.
#define STH_WARM 32
// (the tile phase records the evaluated tiles for every build this returns true for, joins or not: the same kernels -- the same
// float32 sums -- whichever entry point the build came through)
bool ann_stream_knnh_fits(const KnnArgs &a, int dim_padded)
{
    static const char *kern = getenv("ANNCHOR_ST_KERNEL");   // default: this kernel; "bf4" / "bf3" / "4wave" / "bk" select the others for A/B runs
    if (kern && strcmp(kern, "h")) return false;
    return dim_padded == 128 && a.K + ST_BF_MARGIN_H <= 16 && a.Xb && a.rsb && a.cvec && !a.query;
}
int ann_stream_launch_knnh(annchor_ctx *c, const KnnArgs &a0, int dim_padded, bool *handled, int (*warm)(annchor_ctx *, const KnnArgs &, int, bool *, bool))
{
    *handled = false;
    if (!ann_stream_knnh_fits(a0, dim_padded) || !a0.eval_bits || !a0.pre_ranked || a0.eval_halves != 1) return ANNCHOR_OK;
    static const int warm_tiles = getenv("ANNCHOR_STH_WARM") ? std::max(0, atoi(getenv("ANNCHOR_STH_WARM"))) : STH_WARM;   // (A/B runs)
    KnnArgs w = a0;
    w.max_tiles = std::min(a0.max_tiles, warm_tiles + 1);
    bool ok = false;
    ANN_TRY(warm(c, w, dim_padded, &ok, false));
    if (!ok) return ANNCHOR_OK;
    *handled = true;
    if (a0.max_tiles <= warm_tiles + 1) return ANNCHOR_OK;   // the warm-up was the whole budget
    static const size_t lds_pad = getenv("ANNCHOR_STH_LDS_PAD") ? (size_t)atoi(getenv("ANNCHOR_STH_LDS_PAD")) : 0;   // (experiments: > 80 KB in all = one workgroup per CU)
    const size_t lds = sizeof(KnnSharedH<16>) + lds_pad;
    ANN_REQUIRE(c, lds <= 160 * 1024, ANNCHOR_ELIMIT, "streamed k-NN (two-stage form) needs %zu B of LDS", lds);
    ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)k_st_knnh<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ProfScope ps(c, "stream_tile_two_stage_kernel", 0.0);   // (inside stream_tile_gemm_topk: k_st_knnh alone, without the warm-up)
    k_st_knnh<16><<<a0.tile_count, STH_THREADS, lds, c->stream>>>(a0);
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}
