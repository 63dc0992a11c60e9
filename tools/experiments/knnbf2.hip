// knnbf2.hip -- k_st_knnbf2: the tile phase of the streamed k-NN build with TWO adjacent row tiles per workgroup sharing ONE
// column stream (round 5).
//
// k_st_knnbf (knnbf.hip) streams 64 KB of column operands per tile pair for 128 rows: at full matrix-pipe rate that would be
// ~13 TB/s of fabric traffic, so the kernel sits on the fabric (C3: 298 GB per launch = 580 x the data set, matrix pipe 38 %
// busy; at N = 8 x 10^6, where nothing of the stream stays in the Infinity Cache, a tile pair costs a third more than at
// N = 10^6).  Neighbours in the k-d order rank nearly the same column tiles near the top of their lists.  Here a 512-thread
// workgroup owns row tiles 2p and 2p + 1: waves 0-3 are group A (one 128-row tile), waves 4-7 group B, one wave of each group
// on every SIMD, ONE operand ring.  Every group keeps everything that decides its search to itself -- rank keys, bounds,
// lists, thresholds, early stop, budget, evaluated-tile bits -- exactly as in k_st_knnbf; what is shared is the stream:
//   * per selection round each group selects and sorts its next ST_KEEP column tiles by its own rank key;
//   * the two round lists are merged: a tile in both lists appears once, at the better of its two positions;
//   * the merged list is streamed; a group takes part in a tile iff the tile is in ITS list and passes ITS prune test (bound
//     against its thresholds, its early stop, its budget) -- a tile nobody wants is not fetched; a group that sits a tile out
//     flushes its pending slab and waits at the slab barriers, and its SIMD slots go to the other group's waves (beside an
//     idle partner a wave streams MFMAs at the full rate: tools/microbench/pingpong.hip).
// So a row tile evaluates tiles of its own ranking only, in an order perturbed inside a round by its sibling's ranking; the
// fetched bytes are those of the UNION of the two evaluated sets instead of their sum.  Decisions are taken by every wave, for
// both groups, from barrier-separated LDS state with the fixed one-tile lag of k_st_knnbf: runs are reproducible.
// The epilogue (exact float32 re-ranking of the K + 2 kept columns, the guard count) is k_st_knnbf's, per group.
// Shapes: padded dim <= 128, K + 2 <= 16, a graph build (no queries); everything else runs k_st_knnbf / k_st_knn.
#include "../../annchor_amd/csrc/streamed.h"

#define STB2_THREADS 512
#define ST_BF_MARGIN 2   // (as in knnbf.hip)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ST_PROFILE builds: per-wave cycle sums by segment (a.prof[0..7], printed by knn_tile_phase):
//   0 operand reads + MFMA stream (+ shadow test)   1 survivor inserts + list merge   2 waits in front of a slab (requests, barrier)
//   3 choice of the next tile   4 selection rounds   5 merge of the two round lists   6 prologue + ranking   7 epilogue, rest
#ifdef ST_PROFILE
__device__ __forceinline__ long long st2_now()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return (long long)t;
}
#define Q8(i) { const long long pf_n = st2_now(); pf[i] += pf_n - pf_t; pf_t = pf_n; }
#else
#define Q8(i)
#endif

template <int DIM, int KMAX> struct GroupB2 {
    float cand_d[ST_T][ST_SLAB + 1];   // (between runs: the selection's 4096-bin histogram; at the end: exact distances)
    uint8_t cand_c[ST_T][ST_SLAB + 4];
    float list_d[ST_T][KMAX + 1];
    int32_t list_c[ST_T][KMAX + 1];
    float thr[ST_T];
    float hb[ST_T];
    float rrow[ST_T];
    int cnt[ST_T];
    float loI[64], hiI[64], midI[64];
    float wave_thr[2][4];
    int wave_ins[2][4];
    int nsurv;
    int sel_bin;
    uint32_t sel_before;
    int ns_round;     // entries of the group's round list (0: nothing left)
    int more;         // the selection was cut: later tiles remain
};
struct SelBuf2 {   // candidate tiles of a group's selection round (the two groups' buffers alias the operand ring, idle between runs)
    float surv_lb[ST_SURV];
    float surv_vb[ST_SURV];
    int32_t surv_j[ST_SURV];
};
#define STB2_SLOTS 4   // ring slots: three slabs in flight while one is consumed
template <int DIM, int KMAX> struct KnnSharedB2 {
    static constexpr int SLABF = ST_SLAB * DIM;
    static constexpr int RINGF = STB2_SLOTS * SLABF * 4 >= 2 * 12288 ? STB2_SLOTS * SLABF : 2 * 12288 / 4;
    float ring[RINGF];   // FIRST (LDS-DMA destinations below 64 KB); slot = slab counter & 3; between runs the two SelBuf2
    float rsq[STB2_SLOTS][64];   // squared norms of a slot's 32 columns (LDS-DMA, lanes 32..63 repeat them)
    GroupB2<DIM, KMAX> g[2];
    int32_t run_j[2 * ST_KEEP];          // the merged round list: tile,
    float run_vb[2][2 * ST_KEEP];        // its valid bound in each group's list (+inf: not in that list)
    int run_n;
    // (merge scratch: 2 ST_KEEP 64-bit keys over g[0].cand_d, idle between the selection's histogram and the stream)
};

template <int UPC> __device__ __forceinline__ int unit_swz2(int col) { return UPC >= 16 ? (col & 15) : ((col >> 1) & (UPC - 1)); }

__device__ __forceinline__ void slab_end2()
{
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// ascending bitonic sort of P (a power of two, <= 2 ST_KEEP) 64-bit keys in LDS by the whole workgroup
__device__ __forceinline__ void sort64_2(unsigned long long *v, int P)
{
    for (int k2 = 2; k2 <= P; k2 <<= 1)
        for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
            for (int t = threadIdx.x; t < (P >> 1); t += STB2_THREADS) {
                const int q = ((t & ~(j2 - 1)) << 1) | (t & (j2 - 1));
                const int p2 = q | j2;
                const unsigned long long x = v[q], y = v[p2];
                const bool up = (q & k2) == 0;
                if ((x > y) == up) { v[q] = y; v[p2] = x; }
            }
            __syncthreads();
        }
}

template <int DIM, int KMAX> __global__ __launch_bounds__(STB2_THREADS, 1) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_st_knnbf2(KnnArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smemb2[];
    KnnSharedB2<DIM, KMAX> &sh = *reinterpret_cast<KnnSharedB2<DIM, KMAX> *>(smemb2);
    constexpr int UPC = DIM / 4;
    constexpr int NV = UPC / 2;
    constexpr int NPIECE = UPC * ST_SLAB / 64;                 // 1 KB pieces per slab: 16 / 8 / 4
    constexpr int NI = NPIECE / 2;                             // pieces per loading wave: waves 2 s, 2 s + 1 own ring slot s
    constexpr int SLAB_BYTES = ST_SLAB * DIM * 4;
    static_assert(sizeof(sh.ring) + sizeof(sh.rsq) <= 65536 + sizeof(sh.rsq) && sizeof(sh.ring) <= 65536, "LDS-DMA destinations: the ring below 64 KB");
    static_assert(sizeof(sh.g[0].cand_d) >= 2 * ST_KEEP * sizeof(unsigned long long), "merge scratch does not fit");
    static_assert(2 * sizeof(SelBuf2) <= sizeof(sh.ring) && sizeof(SelBuf2) == 12288, "selection buffers alias the ring");
    static_assert(KMAX <= ST_SLAB + 1, "the exact re-ranking reuses cand_d with row stride KMAX");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0..7
    const int grp = wave >> 2;            // 0: row tile 2p, 1: row tile 2p + 1
    const int rg = wave & 3;              // the wave's 32-row group inside its tile
    const int gt = threadIdx.x & 255;     // thread inside the group
    GroupB2<DIM, KMAX> &gs = sh.g[grp];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smemb2;
    int bp;
    {   // XCD-banded pair assignment (block b runs on XCD b % 8): neighbours in the k-d order share an L2
        const int nb_ = gridDim.x, q = nb_ >> 3, r = nb_ & 7, x = blockIdx.x & 7, y = blockIdx.x >> 3;
        bp = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    }
    const int bt = 2 * bp + grp;                         // this group's row tile inside the launch
    const bool gvalid[2] = {2 * bp < a.tile_count, 2 * bp + 1 < a.tile_count};
    const bool isB = grp != 0;
    const bool valid = isB ? gvalid[1] : gvalid[0];
    const int Ig[2] = {a.tile_begin + 2 * bp, a.tile_begin + 2 * bp + 1};
    const int I = isB ? Ig[1] : Ig[0];
    const int64_t grow0 = valid ? (int64_t)I * ST_T : 0;
    const int K = a.K;
    const int KL = min(KMAX, K + ST_BF_MARGIN);
    const int col = lane & 31, half = lane >> 5;
    const int rowbase = rg * 32;
    const int rowq = rowbase + 4 * half;
    constexpr int G = DIM / 16;
    f16x8 ah[G], al[G];
    const float scale = a.cvec[DIM];
    const float inv_scale2 = 1.f / (scale * scale);
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
                const float x = fminf(fmaxf(xu[j] * scale, -60000.f), 60000.f);
                const _Float16 h = (_Float16)x;
                ah[g][j] = h;
                al[g][j] = (_Float16)(x - (float)h);
                acc2 += x * x;
            }
        }
        rr_c = acc2 + __shfl_xor(acc2, 32);
    }
    if (gt < ST_T) {
        const int row = gt;
        const bool real = valid && a.rr[grow0 + row] < INFINITY;
        gs.cnt[row] = 0;
        gs.thr[row] = real ? INFINITY : -1.f;   // padding rows (and the rows of a missing sibling tile) never accept candidates
        for (int q = 0; q < KMAX; ++q) { gs.list_d[row][q] = INFINITY; gs.list_c[row][q] = 0x7fffffff; }
    }
    if (lane < 32) {
        const bool real = valid && a.rr[grow0 + rowbase + lane] < INFINITY;
        gs.rrow[rowbase + lane] = real ? rr_c : INFINITY;
    }
    if (gt < a.na) {
        gs.loI[gt] = valid ? a.rlo[(size_t)gt * a.nt_r + I] : 0.f;
        gs.hiI[gt] = valid ? a.rhi[(size_t)gt * a.nt_r + I] : 0.f;
        gs.midI[gt] = valid ? a.rmid[(size_t)gt * a.nt_r + I] : 0.f;
    }
    if (gt < 8) { gs.wave_ins[gt >> 2][gt & 3] = 0; gs.wave_thr[gt >> 2][gt & 3] = valid ? INFINITY : -1.f; }
    if (gt == 0) { gs.nsurv = 0; gs.ns_round = 0; gs.more = 0; }
    int ins = 0;                         // list insertions counted by this lane
    // state of BOTH groups, tracked by every wave (uniform): the stream's decisions are everybody's
    int processed[2] = {0, 0};           // column tiles scheduled so far
    int tdone[2] = {0, 0};               // column tiles completed and published; tile n publishes into slot n & 1
    int win_start[2] = {0, 0}, win_ins[2] = {0, 0};
    bool dried[2] = {false, false};
    bool fin[2] = {!gvalid[0], !gvalid[1]};   // nothing left to select
    bool pendg[2] = {false, false};      // the group's last slab waits for its test
    uint32_t *ebits = (valid && a.eval_bits) ? a.eval_bits + (size_t)bt * a.eval_words : nullptr;
#ifdef ST_PROFILE
    long long pf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long pf_t = st2_now();
#endif
    __syncthreads();
    if (gt < ST_T) {
        const int row = gt;
        const float t = gs.thr[row];
        gs.hb[row] = t < 0.f ? INFINITY : (t < INFINITY ? 0.5f * (gs.rrow[row] - t) : -INFINITY);
    }
    __syncthreads();

    // ---------------------------------------------------------------- the pieces of a phase
    // Ring slot `slot` belongs to waves 2 slot and 2 slot + 1: they request the slab that goes there (NI pieces of 1 KB each and,
    // the first of the two, the columns' squared norms), and only THEY wait for it -- before the barrier in front of the slab's
    // turn, three slabs later.  No wave has any other vector-memory load in the loop (the norms come through LDS): a wait of
    // the compiler's for one of its own loads would cover the requests, which it does not know of, as well.
    uint32_t loff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int u = ((wave & 1) * NI + i) * 64 + lane;
        const int c = u / UPC, x = u % UPC;
        loff[i] = (uint32_t)(c * DIM * 4 + ((x ^ unit_swz2<UPC>(c)) << 4));
    }
    const char *xb = reinterpret_cast<const char *>(a.Xb);
    const uint32_t rsq0 = lds0 + (uint32_t)(uintptr_t)((unsigned char *)&sh.rsq[0][0] - smemb2);
    auto issue_slab = [&](int J, int slab, int sidx) {
        const int slot = sidx & (STB2_SLOTS - 1);
        if ((wave >> 1) != slot) return;
        const char *src = xb + ((size_t)J * ST_T + slab * ST_SLAB) * (DIM * 4);
        const uint32_t dst = lds0 + (uint32_t)(slot * SLAB_BYTES + (wave & 1) * NI * 1024);
        unsigned keep;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(loff[i]), "s"(src), "s"(dst + (uint32_t)(i * 1024)) : "memory");
        if ((wave & 1) == 0) {
            const uint32_t voff = (uint32_t)((lane & 31) * 4);
            const char *nsrc = reinterpret_cast<const char *>(a.rsb + (int64_t)J * ST_T + slab * ST_SLAB);
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(nsrc), "s"(rsq0 + (uint32_t)(slot * 256)) : "memory");
        }
    };
    // in front of slab `sidx`'s turn: its two loading waves wait for their requests, everybody for its LDS traffic, then the
    // barrier -- which also hands the slot of the slab before it back to ITS loaders
    auto slab_sync = [&](int sidx) {
        __builtin_amdgcn_sched_barrier(0);
        if ((wave >> 1) == (sidx & (STB2_SLOTS - 1))) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    f32x16 acc0, acc1;
    float rj_c = 0.f;
    bool pend = false;
    int pJ = 0, pslab = 0;
    float prj = 0.f;
    uint32_t ppass = 0;
    auto stream_slab = [&](int sidx, f32x16 &accC, const f32x16 &accP) {
        const int slot = sidx & (STB2_SLOTS - 1);
        rj_c = sh.rsq[slot][col];
        const float4 *base = reinterpret_cast<const float4 *>(&sh.ring[slot * (ST_SLAB * DIM)]) + col * UPC;
        const int gsw = half ^ unit_swz2<UPC>(col);
        float hq[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 h4 = *reinterpret_cast<const float4 *>(&gs.hb[rowq + 8 * q]);
            hq[4 * q] = h4.x; hq[4 * q + 1] = h4.y; hq[4 * q + 2] = h4.z; hq[4 * q + 3] = h4.w;
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        float4 b[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) b[v] = base[(2 * v) ^ gsw];
        __builtin_amdgcn_sched_group_barrier(0x100, NV, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) accC[r] = 0.f;
        const float hrj = 0.5f * prj;
        uint32_t pass = 0;
        constexpr int NM = 3 * G;
        constexpr int TPM = (16 + NM - 1) / NM;
#pragma unroll
        for (int g = 0; g < G; ++g) {
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
                asm volatile("" : "+v"(accC), "+v"(pass));
            }
        }
        ppass = pend ? pass : 0u;
    };
    auto insert_merge = [&](const f32x16 &accP) {
        const int J = pJ, slab = pslab;
        const float rj = prj;
        uint32_t pass = ppass;
        const bool self_tile = (int64_t)J * ST_T == grow0;
        if (pass) {
            if (self_tile) {
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
                const float d2 = fmaxf(gs.rrow[rowl] + rj - 2.f * ag, 0.f);
                const int slot = atomicAdd(&gs.cnt[rowl], 1);
                gs.cand_d[rowl][slot] = d2;
                gs.cand_c[rowl][slot] = (uint8_t)col;
            }
        }
        wave_fence_lds();
        {
            constexpr int GL = KMAX;
            constexpr int NG = 64 / GL;
            const int grpi = lane / GL, e = lane % GL;
            const int mycnt = lane < 32 ? gs.cnt[rowbase + lane] : 0;
            unsigned long long todo = __ballot(mycnt > 0);
            const int32_t col0 = (int32_t)(J * ST_T + slab * ST_SLAB);
            while (todo) {
                int rsel = -1;
#pragma unroll
                for (int k = 0; k < NG; ++k) {
                    const int r = todo ? (int)__builtin_ctzll(todo) : -1;
                    if (todo) todo &= todo - 1;
                    rsel = grpi == k ? r : rsel;
                }
                const int row = rowbase + max(rsel, 0);
                const bool live = rsel >= 0;
                const int nc = live ? gs.cnt[row] : 0;
                float ld = (live && e < KL) ? gs.list_d[row][e] : INFINITY;
                int32_t lc = (live && e < KL) ? gs.list_c[row][e] : 0x7fffffff;
                int q = 0;
                while (__ballot(q < nc)) {
                    const bool on = q < nc;
                    const float d = on ? gs.cand_d[row][q] : INFINITY;
                    const int32_t cc = on ? col0 + gs.cand_c[row][q] : 0x7fffffff;
                    const bool before = e < KL && (ld < d || (ld == d && lc < cc));
                    const unsigned long long bb = __ballot(before);
                    const unsigned long long gmask = (GL == 64) ? ~0ull : ((1ull << GL) - 1);
                    const int pos = __popcll((bb >> (grpi * GL)) & gmask);
                    float pd;
                    int32_t pc;
                    if constexpr (GL == 16) {
                        pd = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ld), 0x111, 0xf, 0xf, false));
                        pc = __builtin_amdgcn_update_dpp(0, lc, 0x111, 0xf, 0xf, false);
                    } else {
                        pd = __shfl_up(ld, 1, GL);
                        pc = __shfl_up(lc, 1, GL);
                    }
                    if (on && pos < KL) {
                        ld = e > pos ? pd : (e == pos ? d : ld);
                        lc = e > pos ? pc : (e == pos ? cc : lc);
                        ins += (e == 0 && pos < K) ? 1 : 0;
                    }
                    ++q;
                }
                if (live) {
                    if (e < KL) { gs.list_d[row][e] = ld; gs.list_c[row][e] = lc; }
                    if (e == KL - 1) { gs.thr[row] = ld; gs.hb[row] = 0.5f * (gs.rrow[row] - ld); }
                    if (e == 0) gs.cnt[row] = 0;
                }
            }
        }
        wave_fence_lds();
    };
    auto test_only = [&](const f32x16 &accP) {
        uint32_t pass = 0;
        const float hrj = 0.5f * prj;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 h4 = *reinterpret_cast<const float4 *>(&gs.hb[rowq + 8 * q]);
            const float hv[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) pass |= (accP[4 * q + e] > hv[e] + hrj ? 1u : 0u) << (4 * q + e);
        }
        ppass = pend ? pass : 0u;
    };
    // the wave's insertion count and its rows' worst k-th distance into the slot of the group's next completed tile
    auto publish = [&]() {
        int wins = ins;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) wins += __shfl_xor(wins, off);
        float t = lane < 32 ? gs.thr[rowbase + lane] : -1.f;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) t = fmaxf(t, __shfl_xor(t, off));
        const int td = isB ? tdone[1] : tdone[0];
        if (lane == 0) { gs.wave_ins[(td + 1) & 1][rg] = wins; gs.wave_thr[(td + 1) & 1][rg] = t * inv_scale2; }
    };
    auto thrmax_of = [&](int g) {
        const float *w = sh.g[g].wave_thr[tdone[g] & 1];
        return fmaxf(fmaxf(w[0], w[1]), fmaxf(w[2], w[3]));
    };
    // the pending slab's test and merge without a stream to hide them in
    auto flush = [&]() {
        if (pend) { test_only(acc1); insert_merge(acc1); pend = false; }
    };

    // One run of the stream over the merged list sh.run_j / run_vb[0..1] (n entries).  Uniform: every wave takes the same path
    // and tracks both groups' counters.
    int fetched = 0;   // column tiles this workgroup streamed (uniform)
    auto run = [&](int n) {
        int q = 0;
        bool cur_p[2] = {false, false};   // who takes part in the tile in the stream
        // the next tile somebody wants; *p: who.  in_stream: a tile is in the stream (its participants' counts include it)
        auto next_tile = [&](bool in_stream, bool *p) -> int {
            float tm[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                if (a.early_window > 0 && !dried[g] && !fin[g]) {
                    const int done = processed[g] - ((in_stream && cur_p[g]) ? 1 : 0);
                    if (done - win_start[g] >= a.early_window) {
                        int cur = 0;
#pragma unroll
                        for (int w = 0; w < 4; ++w) cur += sh.g[g].wave_ins[tdone[g] & 1][w];
                        if (cur - win_ins[g] < a.early_tau) dried[g] = true;
                        else { win_start[g] = done; win_ins[g] = cur; }
                    }
                }
                tm[g] = thrmax_of(g);
            }
            while (q < n) {
                const int J = sh.run_j[q];
                bool any = false;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const float lb = sh.run_vb[g][q];
                    p[g] = !fin[g] && !dried[g] && processed[g] < a.max_tiles && lb < INFINITY && lb * lb < tm[g];
                    any = any || p[g];
                }
                ++q;
                if (any) {
                    ++fetched;
#pragma unroll
                    for (int g = 0; g < 2; ++g)
                        if (p[g]) ++processed[g];
                    if ((isB ? p[1] : p[0]) && ebits && gt == 0) atomicOr(&ebits[J >> 5], 1u << (J & 31));
                    return J;
                }
            }
            p[0] = p[1] = false;
            return -1;
        };
        int J = next_tile(false, cur_p);
        if (J < 0) return;
        int sidx = 0;   // slab counter of the run: slab s sits in ring slot s & 3
        issue_slab(J, 0, 0);
        issue_slab(J, 1, 1);
        issue_slab(J, 2, 2);
        pend = false;
        auto step = [&](int Jc, int sl, int si, f32x16 &accC, const f32x16 &accP) {
            stream_slab(si, accC, accP);
            Q8(0)
            if (pend) insert_merge(accP);
            pend = true; pJ = Jc; pslab = sl; prj = rj_c;
            Q8(1)
        };
        for (;;) {
            const bool part = isB ? cur_p[1] : cur_p[0];
            // slab s is computed behind barrier s; the request that goes out behind barrier s is slab s + 3's (into the slot of
            // slab s - 1, which everybody left before the barrier): the NEXT tile is therefore chosen during slab 1 -- from the
            // thresholds and insertion counts published at the end of the tile before J, as k_st_knnbf does during slab 3
            Q8(1)
            slab_sync(sidx);
            Q8(2)
            issue_slab(J, 3, sidx + 3);
            if (part) step(J, 0, sidx, acc0, acc1); else flush();   // (a group sitting the tile out merges its last slab now)
            Q8(1)
            slab_sync(sidx + 1);
            Q8(2)
            bool np[2];
            const int Jn = next_tile(true, np);
            Q8(3)
            if (Jn >= 0) issue_slab(Jn, 0, sidx + 4);
            if (part) step(J, 1, sidx + 1, acc1, acc0);
            slab_sync(sidx + 2);
            Q8(2)
            if (Jn >= 0) issue_slab(Jn, 1, sidx + 5);
            if (part) step(J, 2, sidx + 2, acc0, acc1);
            slab_sync(sidx + 3);
            Q8(2)
            if (Jn >= 0) issue_slab(Jn, 2, sidx + 6);
            if (part) { step(J, 3, sidx + 3, acc1, acc0); publish(); }
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                if (cur_p[g]) { ++tdone[g]; pendg[g] = true; } else pendg[g] = false;
            }
            sidx += 4;
            cur_p[0] = np[0]; cur_p[1] = np[1];
            if (Jn < 0) break;
            J = Jn;
        }
        // ---- tail: the last slab's test and merge, and the thresholds the next selection reads
        if (pend) { flush(); publish(); }
#pragma unroll
        for (int g = 0; g < 2; ++g)
            if (pendg[g]) { ++tdone[g]; pendg[g] = false; }
        slab_end2();
    };

    Q8(6)
    // ---- phase A: the pair's own two tiles, both groups against both (a row tile's sibling is its nearest tile)
    if (threadIdx.x == 0) {
        int n = 0;
        for (int g = 0; g < 2; ++g)
            if (gvalid[g]) { sh.run_j[n] = Ig[g]; sh.run_vb[0][n] = gvalid[0] ? 0.f : INFINITY; sh.run_vb[1][n] = gvalid[1] ? 0.f : INFINITY; ++n; }
        sh.run_n = n;
    }
    __syncthreads();
    run(sh.run_n);

    // ---- phase B: all other column tiles, ranked and selected per group exactly as k_st_knnbf does, streamed merged
    float *skey = a.scr_key + (size_t)(valid ? bt : 0) * a.nt_all;
    float *slb = a.scr_lb + (size_t)(valid ? bt : 0) * a.nt_all;
    if (valid)
        for (int J = gt; J < a.nt_all; J += 256) {
            float lb = 0.f, lbc = 0.f;
            for (int an = 0; an < a.na; ++an) {
                const float lj = a.lo[(size_t)an * a.nt_all + J], hj = a.hi[(size_t)an * a.nt_all + J];
                const float gap = fmaxf(gs.loI[an] - hj, lj - gs.hiI[an]);
                lb = fmaxf(lb, gap - 4e-6f * (fabsf(hj) + fabsf(gs.hiI[an])));
                const float dm = a.mid[(size_t)an * a.nt_all + J] - gs.midI[an];
                lbc += dm * dm;
            }
            const bool own = (J == Ig[0] && gvalid[0]) || (J == Ig[1] && gvalid[1]);   // (both evaluated in phase A)
            skey[J] = (own || !(lbc < INFINITY)) ? INFINITY : lbc;
            slb[J] = lb;
        }
    __syncthreads();
    Q8(6)
    uint32_t *hist = reinterpret_cast<uint32_t *>(&gs.cand_d[0][0]);
    SelBuf2 &sb = *reinterpret_cast<SelBuf2 *>(reinterpret_cast<unsigned char *>(&sh.ring[0]) + (size_t)grp * sizeof(SelBuf2));
    static_assert(sizeof(gs.cand_d) >= 4096 * sizeof(uint32_t), "histogram does not fit");
    uint32_t done_bits = 0;   // (own group's selection cursor)
    int done_j = -1;
    for (;;) {
        // ---- selection, both groups in lockstep (the same barriers whatever a group still has to do)
        const bool act = !(isB ? fin[1] : fin[0]);
        const float thrmax = isB ? thrmax_of(1) : thrmax_of(0);
        uint32_t prefix = 0;
        uint32_t want = ST_KEEP;
        bool all = false;
        for (int level = 0; level < 3; ++level) {
            const bool lv = act && !all;
            const int shift = level == 0 ? 20 : level == 1 ? 8 : 0;
            const int nbins = level == 2 ? 256 : 4096;
            const uint32_t pmask = level == 0 ? 0u : level == 1 ? 0xfff00000u : 0xffffff00u;
            if (lv)
                for (int q = gt; q < nbins; q += 256) hist[q] = 0;
            __syncthreads();
            if (lv)
                for (int J = gt; J < a.nt_all; J += 256) {
                    const uint32_t kb = __float_as_uint(skey[J]);
                    const float lb = slb[J];
                    const bool after_done = kb > done_bits || (kb == done_bits && J > done_j);
                    if (kb < 0x7f800000u && after_done && lb * lb < thrmax && (kb & pmask) == prefix)
                        atomicAdd(&hist[(kb >> shift) & (nbins - 1)], 1u);
                }
            __syncthreads();
            const int per = nbins >= 256 ? nbins / 256 : 1;
            const bool owner = lv && gt * per < nbins;
            uint32_t mine = 0;
            if (owner)
                for (int q = 0; q < per; ++q) mine += hist[gt * per + q];
            uint32_t incl = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t up = __shfl_up(incl, off);
                if (lane >= off) incl += up;
            }
            uint32_t *wtot = reinterpret_cast<uint32_t *>(&sb.surv_lb[0]);   // 4 wave totals of the group
            if (gt == 0) gs.sel_bin = -1;
            if (lane == 63) wtot[rg] = incl;
            __syncthreads();
            uint32_t before = incl - mine;
            for (int w2 = 0; w2 < rg; ++w2) before += wtot[w2];
            if (owner && before < want && before + mine >= want) {
                uint32_t ac = before;
                int q = gt * per;
                for (;; ++q) { if (ac + hist[q] >= want) break; ac += hist[q]; }
                gs.sel_bin = q;
                gs.sel_before = ac;
            }
            __syncthreads();
            if (lv) {
                if (gs.sel_bin < 0) all = true;
                else { prefix |= (uint32_t)gs.sel_bin << shift; want -= gs.sel_before; }
            }
            __syncthreads();
        }
        const uint32_t cut_bits = all ? 0x7f7fffffu : prefix;
        if (gt == 0) gs.nsurv = 0;
        __syncthreads();
        if (act)
            for (int J = gt; J < a.nt_all; J += 256) {
                const uint32_t kb = __float_as_uint(skey[J]);
                const float lb = slb[J];
                const bool after_done = kb > done_bits || (kb == done_bits && J > done_j);
                if (kb < 0x7f800000u && after_done && lb * lb < thrmax && kb <= cut_bits) {
                    const int slot = atomicAdd(&gs.nsurv, 1);
                    if (slot < ST_SURV) { sb.surv_lb[slot] = __uint_as_float(kb); sb.surv_vb[slot] = lb; sb.surv_j[slot] = J; }
                }
            }
        __syncthreads();
        int ns = act ? min(gs.nsurv, ST_SURV) : 0;
        {   // sort by (rank key, J): bitonic over ST_SURV slots, every group (an idle one sorts padding)
            for (int q = gt; q < ST_SURV; q += 256)
                if (q >= ns) { sb.surv_lb[q] = INFINITY; sb.surv_j[q] = 0x7fffffff; }
            __syncthreads();
            for (int k2 = 2; k2 <= ST_SURV; k2 <<= 1)
                for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
                    for (int q = gt; q < ST_SURV; q += 256) {
                        const int p2 = q ^ j2;
                        if (p2 > q) {
                            const bool up = (q & k2) == 0;
                            const float lq = sb.surv_lb[q], lp = sb.surv_lb[p2];
                            const int jq = sb.surv_j[q], jp = sb.surv_j[p2];
                            const bool gtr = lq > lp || (lq == lp && jq > jp);
                            if (gtr == up) {
                                sb.surv_lb[q] = lp; sb.surv_lb[p2] = lq; sb.surv_j[q] = jp; sb.surv_j[p2] = jq;
                                const float t = sb.surv_vb[q]; sb.surv_vb[q] = sb.surv_vb[p2]; sb.surv_vb[p2] = t;
                            }
                        }
                    }
                    __syncthreads();
                }
        }
        bool more = act && !all;
        if (ns > ST_KEEP) { ns = ST_KEEP; more = true; }   // (the merged list holds 2 ST_KEEP entries: a larger round is cut here)
        uint32_t round_last_bits = done_bits;
        int round_last_j = done_j;
        if (ns > 0) { round_last_bits = __float_as_uint(sb.surv_lb[ns - 1]); round_last_j = sb.surv_j[ns - 1]; }
        if (gt == 0) { gs.ns_round = ns; gs.more = more ? 1 : 0; }
        __syncthreads();
        const int nsg[2] = {sh.g[0].ns_round, sh.g[1].ns_round};
        const bool moreg[2] = {sh.g[0].more != 0, sh.g[1].more != 0};
        if (nsg[0] == 0) fin[0] = true;
        if (nsg[1] == 0) fin[1] = true;
        Q8(4)
        if (fin[0] && fin[1]) break;   // (uniform)
        // ---- merge the two round lists: one entry per tile, at the better of its two positions
        {
            const SelBuf2 *sbg[2] = {reinterpret_cast<const SelBuf2 *>(&sh.ring[0]),
                                     reinterpret_cast<const SelBuf2 *>(reinterpret_cast<unsigned char *>(&sh.ring[0]) + sizeof(SelBuf2))};
            const int n2 = nsg[0] + nsg[1];
            int P = 2;
            while (P < n2) P <<= 1;
            unsigned long long *mkey = reinterpret_cast<unsigned long long *>(&sh.g[0].cand_d[0][0]);
            for (int t = threadIdx.x; t < P; t += STB2_THREADS) {
                unsigned long long k = ~0ull;
                if (t < nsg[0]) k = ((unsigned long long)(uint32_t)sbg[0]->surv_j[t] << 32) | (unsigned long long)t;
                else if (t < n2) k = ((unsigned long long)(uint32_t)sbg[1]->surv_j[t - nsg[0]] << 32) | 0x80000000ull | (unsigned long long)(t - nsg[0]);
                mkey[t] = k;
            }
            __syncthreads();
            sort64_2(mkey, P);   // by tile, group A's entry first
            unsigned long long mk[(2 * ST_KEEP + STB2_THREADS - 1) / STB2_THREADS];
#pragma unroll
            for (int e = 0; e < (2 * ST_KEEP + STB2_THREADS - 1) / STB2_THREADS; ++e) {
                const int t = e * STB2_THREADS + threadIdx.x;
                unsigned long long out = ~0ull;
                if (t < n2) {
                    const unsigned long long k = mkey[t];
                    const uint32_t J = (uint32_t)(k >> 32);
                    const bool first = t == 0 || (uint32_t)(mkey[t - 1] >> 32) != J;
                    if (first) {
                        const bool isB = (k & 0x80000000ull) != 0;
                        uint32_t pa = isB ? 0x7ffu : (uint32_t)(k & 0x7ffu), pb = isB ? (uint32_t)(k & 0x7ffu) : 0x7ffu;
                        if (!isB && t + 1 < n2 && (uint32_t)(mkey[t + 1] >> 32) == J) pb = (uint32_t)(mkey[t + 1] & 0x7ffu);
                        const uint32_t mp = min(pa, pb);
                        out = ((unsigned long long)mp << 54) | ((unsigned long long)J << 22) | ((unsigned long long)pa << 11) | (unsigned long long)pb;
                    }
                }
                mk[e] = out;
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < (2 * ST_KEEP + STB2_THREADS - 1) / STB2_THREADS; ++e) {
                const int t = e * STB2_THREADS + threadIdx.x;
                if (t < P) mkey[t] = mk[e];
            }
            if (threadIdx.x == 0) sh.run_n = 0;
            __syncthreads();
            sort64_2(mkey, P);   // by (better position, tile); the second entries of shared tiles (~0) go to the end
            for (int t = threadIdx.x; t < n2; t += STB2_THREADS) {
                const unsigned long long k = mkey[t];
                if (k != ~0ull) {
                    const uint32_t pa = (uint32_t)(k >> 11) & 0x7ffu, pb = (uint32_t)k & 0x7ffu;
                    sh.run_j[t] = (int32_t)((k >> 22) & 0xffffffffull);
                    sh.run_vb[0][t] = pa != 0x7ffu ? sbg[0]->surv_vb[pa] : INFINITY;
                    sh.run_vb[1][t] = pb != 0x7ffu ? sbg[1]->surv_vb[pb] : INFINITY;
                    atomicAdd(&sh.run_n, 1);
                }
            }
            __syncthreads();   // the round's tiles have left the ring: the stream may take it back
        }
        Q8(5)
        run(sh.run_n);
        Q8(1)
        if (ns > 0) { done_bits = round_last_bits; done_j = round_last_j; }
#pragma unroll
        for (int g = 0; g < 2; ++g)
            if (dried[g] || processed[g] >= a.max_tiles || !moreg[g]) fin[g] = true;
        __syncthreads();
        if (fin[0] && fin[1]) break;
    }
    __syncthreads();
    // ---- exact re-ranking (k_st_knnbf's epilogue, per group)
    if (gt == 0) gs.nsurv = 0;
    {
        float *ex = &gs.cand_d[0][0];
        if (valid)
            for (int q = gt; q < ST_T * KL; q += 256) {
                const int row = q / KL, e = q - row * KL;
                const int32_t cc = gs.list_c[row][e];
                float d2 = INFINITY;
                if (cc != 0x7fffffff && gs.list_d[row][e] < INFINITY) {
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
        if (valid && gt < ST_T) {
            const int row = gt;
            float eps = 0.f;
            int nfin = 0;
            for (int e = 0; e < KL; ++e) {
                const float ap = gs.list_d[row][e] * inv_scale2, exv = ex[row * KMAX + e];
                if (exv < INFINITY) { eps = fmaxf(eps, fabsf(ap - exv)); ++nfin; }
            }
            for (int e = 1; e < KL; ++e) {
                const float d = ex[row * KMAX + e];
                const int32_t cc = gs.list_c[row][e];
                int p = e;
                while (p > 0 && (d < ex[row * KMAX + p - 1] || (d == ex[row * KMAX + p - 1] && cc < gs.list_c[row][p - 1]))) {
                    ex[row * KMAX + p] = ex[row * KMAX + p - 1];
                    gs.list_c[row][p] = gs.list_c[row][p - 1];
                    --p;
                }
                ex[row * KMAX + p] = d;
                gs.list_c[row][p] = cc;
            }
            if (nfin > K && ex[row * KMAX + K - 1] + 2.f * eps > gs.list_d[row][KL - 1] * inv_scale2) atomicAdd(&gs.nsurv, 1);
        }
        __syncthreads();
        if (valid)
            for (int q = gt; q < ST_T * K; q += 256) {
                const int row = q / K, e = q - row * K;
                const float d2 = ex[row * KMAX + e];
                a.out_d2[((size_t)bt * ST_T + row) * K + e] = d2;
                a.out_col[((size_t)bt * ST_T + row) * K + e] = d2 < INFINITY ? gs.list_c[row][e] : 0x7fffffff;
            }
    }
#ifdef ST_PROFILE
    Q8(7)
    if (lane == 0 && a.prof)
        for (int i = 0; i < 8; ++i) atomicAdd(a.prof + i, (unsigned long long)pf[i]);
#endif
    if (threadIdx.x == 0) atomicAdd(a.evals + 4, (unsigned long long)fetched);
    if (valid && gt == 0) {
        atomicAdd(a.evals, (unsigned long long)(isB ? processed[1] : processed[0]));
        if (gs.nsurv) atomicAdd(a.evals + 3, (unsigned long long)gs.nsurv);
    }
}

template <int DIM, int KMAX> static int launchb2(annchor_ctx *c, const KnnArgs &a)
{
    const size_t lds = sizeof(KnnSharedB2<DIM, KMAX>);
    ANN_REQUIRE(c, lds <= 160 * 1024, ANNCHOR_ELIMIT, "streamed k-NN (paired split-fp16 form) needs %zu B of LDS", lds);
    ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)k_st_knnbf2<DIM, KMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    k_st_knnbf2<DIM, KMAX><<<(a.tile_count + 1) / 2, STB2_THREADS, lds, c->stream>>>(a);
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}

// The tile phase with two row tiles per workgroup when the shape fits (graph build, padded dim <= 128, K + 2 <= 16, the split
// copy of the columns exists); *handled = false sends the caller on to k_st_knnbf.
int ann_stream_launch_knnbf2(annchor_ctx *c, const KnnArgs &a, int dim_padded, bool *handled)
{
    *handled = false;
    if (a.query || a.K + ST_BF_MARGIN > 16 || !a.Xb || !a.rsb || !a.cvec || a.tile_count < 2) return ANNCHOR_OK;
    *handled = true;
    switch (dim_padded) {
    case 32: return launchb2<32, 16>(c, a);
    case 64: return launchb2<64, 16>(c, a);
    case 128: return launchb2<128, 16>(c, a);
    default: *handled = false; return ANNCHOR_OK;
    }
}
