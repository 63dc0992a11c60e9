// lev.hip -- Levenshtein distance on gfx950: bit-parallel Myers/Hyyro as a systolic
// array across the lanes of a wavefront.
//
// Replaces, for f = levenshtein (reference annchor/distances.py:16-20 ->
// Levenshtein.distance), the evaluator get_exact(f, X, IJ) of
// annchor/utils.py:110-177.
//
// Mapping (CDNA4-first, not a translation of any CPU code):
//   * The shorter string of a pair is the bit-vector "pattern", cut into 32-bit
//     words; word w of pair slot g lives in lane g*G + w (G = words of the longest
//     string in the data set).  P = 64/G pairs share one wavefront.
//   * The longer string is the "text".  At iteration k lane w processes text columns
//     2(k-w) and 2(k-w)+1: their four horizontal carry bits travel down the lanes through
//     one `wave_shr:1` DPP move per iteration (no LDS, no shuffle unit); the loop body is
//     branch-free and its LDS reads (text pair, two match masks) are software-pipelined
//     two iterations deep, so only the DPP carry chain is loop-carried.
//   * Per-pattern match masks PM[symbol][word] sit in LDS (alphabet * G * 4 B per
//     pair slot, e.g. 2 KB for a-z and 600-char strings); every lane fetches one
//     dword per step, bank-conflict free within a symbol.
//   * The text itself is staged once into LDS with 16-byte coalesced reads.
// Work per pair = words(pattern) * len(text) word-steps of ~25 integer VALU ops:
// the kernel is integer-ALU bound (the data set is L2 resident), not HBM bound.
#include <cstdlib>

#include "common.h"
#include <atomic>
#include <type_traits>

#define LEV_THREADS 256
#define LEV_WAVES (LEV_THREADS / ANN_WAVE)

__device__ __forceinline__ uint32_t dpp_wave_shr1(uint32_t v)
{
    // lane l receives lane l-1's value; lane 0 keeps `v` (overwritten by the feed)
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

struct LevArgs {
    const uint8_t *sym;
    const int32_t *soff;
    const int32_t *slen;
    const int2 *ij;
    const int32_t *idx;
    const int32_t *anchor;
    int64_t n;
    double *out;
    double *RA;
    uint8_t *ncm;
    int G;         // lanes (32-bit words) per pair slot
    int P;         // pair slots per wave
    int alphabet;
    int text_stride;  // bytes of LDS text per slot (multiple of 16)
    int pm_bytes;     // bytes of PM per wave (multiple of 16)
    int wave_bytes;   // pm_bytes + P * text_stride + 256
};

// LDS traffic of one wave is ordered by the hardware; this only stops the compiler
// from moving LDS accesses of different lanes across the phase boundary.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// LDS read, and validity is tested once per column.  ~14 instructions per step.
struct LevArgsR {
    LevArgs b;
    int GL;        // lanes per slot = ceil(words / R)
    int pm_stride; // words per symbol row of a slot's PM table (GL * R)
    // fused max-min pick (see PairSource): the anchor of this one-to-all launch is the first
    // arg-max of the running minimum, derived by every wave for itself
    const double *pick_row;
    double *pick_runmin;
    int32_t *pick_out;
    int pick_reset, pick_nx;
    // two slot classes in one launch (k_lev_f): pairs whose shorter string has <= GL0 words run P0 = 64 / GL0 to a
    // wave, the rest P = 64 / GL.  perm[0 .. *n0) = list positions of the short class, perm[n-1 .. *n0]
    // (from the back) those of the long class; perm == nullptr: one class (GL, P), identity order.
    const int32_t *perm;
    const int32_t *n0;
    int GL0, P0;
};

#define LEVR_PAD 64   // text entries of padding either side of a slot's text (>= lanes per slot)

template <int R> __global__ __launch_bounds__(ANN_WAVE) void k_lev_r(LevArgsR ar)
{
    // one wave per workgroup: waves never synchronise with each other, and small workgroups
    // let the LDS allocator pack as many waves per CU as the tables allow
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const LevArgs &a = ar.b;
    const int lane = threadIdx.x;
    const int GL = ar.GL, P = a.P, A = a.alphabet, PS = ar.pm_stride;
    const int g = lane / GL;         // pair slot of this lane
    const int w = lane - g * GL;     // lane index inside the slot: owns words w*R .. w*R+R-1
    const bool slot_ok = g < P;
    uint32_t *pm_g = reinterpret_cast<uint32_t *>(smem) + (size_t)(slot_ok ? g : 0) * A * PS;
    // text of a slot: uint16 entries = byte offset of the symbol's PM row, LEVR_PAD entries of
    // padding either side so that the pipelined reads of lanes outside their column range
    // stay inside the slot (their values are never used)
    uint16_t *txt_g = reinterpret_cast<uint16_t *>(smem + a.pm_bytes + (size_t)(slot_ok ? g : 0) * a.text_stride);
    int *ssum = reinterpret_cast<int *>(smem + a.pm_bytes + (size_t)P * a.text_stride);
    const int64_t n_tasks = (a.n + P - 1) / P;
    const uint32_t row_bytes = (uint32_t)PS * 4u;
    typedef uint32_t vecR __attribute__((ext_vector_type(R)));

    if (slot_ok)
        for (int e = w * 8; e < a.text_stride / 2; e += GL * 8) *reinterpret_cast<uint4 *>(txt_g + e) = make_uint4(0, 0, 0, 0);

    int picked = -1;
    if (ar.pick_row) {
        // every wave: runmin and its first arg-max over the (small) data set, loads batched;
        // workgroup 0 also stores the running minimum and the anchor.  Other workgroups may read
        // runmin[j] before or after that store: min(min(r, d), d) == min(r, d).
        double bv = -INFINITY;
        int bi = 0x7fffffff;
        const int nx = ar.pick_nx;
        for (int j0 = 0; j0 < nx; j0 += 64 * 8) {
            double d[8], rm[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = min(j0 + e * 64 + lane, nx - 1);
                d[e] = ar.pick_row[j];
                rm[e] = ar.pick_reset ? 0.0 : ar.pick_runmin[j];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = j0 + e * 64 + lane;
                if (j < nx) {
                    const double v = ar.pick_reset ? d[e] : fmin(rm[e], d[e]);
                    if (blockIdx.x == 0) ar.pick_runmin[j] = v;
                    argmax_combine(bv, bi, v, j);
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(bv, off);
            const int oi = __shfl_xor(bi, off);
            argmax_combine(bv, bi, ov, oi);
        }
        picked = bi;
        if (blockIdx.x == 0 && lane == 0) *ar.pick_out = bi;
    }

    for (int64_t task = blockIdx.x; task < n_tasks; task += gridDim.x) {
        const int64_t t_pair = task * P + g;
        const bool active = slot_ok && t_pair < a.n;
        int si = 0, sj = 0;
        int64_t opos = t_pair;
        if (active) {
            if (a.anchor) { si = picked >= 0 ? picked : *a.anchor; sj = (int)t_pair; }
            else {
                int64_t q = a.idx ? a.idx[t_pair] : t_pair;
                int2 p = a.ij[q];
                si = p.x; sj = p.y;
                if (a.idx) opos = q;
            }
        }
        const int li = active ? a.slen[si] : 0, lj = active ? a.slen[sj] : 0;
        // pattern = LONGER string (its words spread over the slot's lanes, which are reserved
        // anyway), text = shorter one: the column loop runs min(li, lj) + lanes steps
        const bool swap = li < lj;
        const int ps = swap ? sj : si, ts = swap ? si : sj;
        const int m = swap ? lj : li, n = swap ? li : lj;
        const uint8_t *pat = a.sym + (active ? a.soff[ps] : 0);
        const uint8_t *tex = a.sym + (active ? a.soff[ts] : 0);
        const int Wp = (m + 31) >> 5;          // pattern words
        const int Gp = (Wp + R - 1) / R;       // lanes that hold pattern words
        if (slot_ok && w == 0) ssum[g] = 0;
        if (slot_ok)
            for (int c = 0; c < A; ++c) {
#pragma unroll
                for (int r = 0; r < R; ++r) pm_g[c * PS + w * R + r] = 0u;
            }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int wi = w * R + r;
            if (active && wi < Wp) {
                const uint4 *p16 = reinterpret_cast<const uint4 *>(pat + wi * 32);
                const uint4 q0 = p16[0], q1 = p16[1];
                const uint32_t wd[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                const int valid = min(32, m - wi * 32);
#pragma unroll
                for (int k = 0; k < 32; ++k) {
                    const uint32_t c = (wd[k >> 2] >> ((k & 3) * 8)) & 0xffu;
                    if (k < valid) atomicOr(&pm_g[c * PS + wi], 1u << k);
                }
            }
        }
        if (active) {
            const int chunks = (n + 15) >> 4;
            for (int ch = w; ch < chunks; ch += GL) {
                const uint4 q = reinterpret_cast<const uint4 *>(tex)[ch];
                const uint32_t wd[4] = {q.x, q.y, q.z, q.w};
                uint32_t o[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t c0 = (wd[k >> 1] >> ((k & 1) * 16)) & 0xffu, c1 = (wd[k >> 1] >> ((k & 1) * 16 + 8)) & 0xffu;
                    o[k] = (c0 * row_bytes) | ((c1 * row_bytes) << 16);
                }
                uint4 *dst = reinterpret_cast<uint4 *>(txt_g + LEVR_PAD + ch * 16);
                dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
                dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
            }
        }
        wave_lds_fence();

        uint32_t vp[R], vn[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { vp[r] = 0xffffffffu; vn[r] = 0u; }
        const uint32_t un = (active && m > 0) ? (uint32_t)n : 0u;
        int max_steps = (active && m > 0) ? n + Gp - 1 : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) max_steps = max(max_steps, __shfl_xor(max_steps, off));
        max_steps = __builtin_amdgcn_readfirstlane(max_steps);
        const unsigned char *pm_w = reinterpret_cast<const unsigned char *>(pm_g + w * R);
        const uint16_t *tp = txt_g + LEVR_PAD - w;   // tp[k] = row offset of this lane's symbol at iteration k
        // 2-deep LDS pipeline: symbol of iteration k+2 and match masks of iteration k+1 are in
        // flight while iteration k computes
        uint32_t c1 = tp[1];
        vecR eq = *reinterpret_cast<const vecR *>(pm_w + tp[0]);
        uint32_t out_hp = 0, out_hn = 0;   // hp / hn of this lane's last word in the previous iteration
        auto column = [&](int k, auto checked) {
            const uint32_t c2 = tp[k + 2];
            const vecR eq_n = *reinterpret_cast<const vecR *>(pm_w + c1);
            uint32_t hp_up = dpp_wave_shr1(out_hp), hn_up = dpp_wave_shr1(out_hn);
            __builtin_amdgcn_sched_barrier(0);
            // the row above the first pattern row: D[0][j] - D[0][j-1] = +1
            hp_up = (w == 0) ? 0x80000000u : hp_up;
            hn_up = (w == 0) ? 0u : hn_up;
            const bool valid = !decltype(checked)::value || (uint32_t)(k - w) < un;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t x = eq[r] | (hn_up >> 31);
                const uint32_t d0 = (((x & vp[r]) + vp[r]) ^ vp[r]) | x | vn[r];
                const uint32_t hp = vn[r] | ~(d0 | vp[r]);
                const uint32_t hn = d0 & vp[r];
                const uint32_t hps = __builtin_amdgcn_alignbit(hp, hp_up, 31);   // (hp << 1) | carry from the word below
                const uint32_t hns = __builtin_amdgcn_alignbit(hn, hn_up, 31);
                const uint32_t nvp = hns | ~(d0 | hps), nvn = hps & d0;
                vp[r] = valid ? nvp : vp[r];
                vn[r] = valid ? nvn : vn[r];
                hp_up = hp;
                hn_up = hn;
            }
            out_hp = hp_up;
            out_hn = hn_up;
            __builtin_amdgcn_sched_barrier(0);
            eq = eq_n;
            c1 = c2;
        };
        // Lane w works on column k - w.  Between k = GL - 1 (every lane has started) and the
        // shortest text of the wave's pairs (no lane has finished) every column is a real one:
        // that stretch runs without the per-column validity test.
        int k_lo = min(GL - 1, max_steps), k_hi = (active && m > 0) ? n : 0x7fffffff;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) k_hi = min(k_hi, __shfl_xor(k_hi, off));
        k_hi = max(k_lo, min(__builtin_amdgcn_readfirstlane(k_hi), max_steps));
        int k = 0;
        for (; k < k_lo; ++k) column(k, std::true_type());
        // two columns per trip: the register hand-overs (eq <- eq_n, c1 <- c2) between them are
        // renames in straight-line code, and the loop bookkeeping is paid once per two columns
        for (; k + 2 <= k_hi; k += 2) { column(k, std::false_type()); column(k + 1, std::false_type()); }
        for (; k < k_hi; ++k) column(k, std::false_type());
        for (; k < max_steps; ++k) column(k, std::true_type());
        if (active) {
            int part = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int wi = w * R + r;
                const uint32_t rows = wi < Wp - 1 ? 0xffffffffu : (wi == Wp - 1 ? (0xffffffffu >> (31 - ((m - 1) & 31))) : 0u);
                part += __popc(vp[r] & rows) - __popc(vn[r] & rows);
            }
            if (part) atomicAdd(&ssum[g], part);
        }
        wave_lds_fence();
        if (active && w == 0) {
            const double d = (double)(n + ssum[g]);
            if (a.out) a.out[t_pair] = d;
            if (a.RA) { a.RA[opos] = d; a.ncm[opos] = 0; }
        }
        wave_lds_fence();
    }
}

template <int R> static int launch_r(annchor_ctx *c, LevArgs a, int64_t npairs, const PairSource &src)
{
    const int W = (c->maxlen + 31) / 32 > 0 ? (c->maxlen + 31) / 32 : 1;
    LevArgsR ar;
    ar.GL = (W + R - 1) / R;
    ar.pm_stride = ar.GL * R;
    a.G = ar.GL;
    a.P = 64 / ar.GL;
    a.pm_bytes = (int)((((size_t)a.P * a.alphabet * ar.pm_stride * 4) + 15) & ~(size_t)15);
    a.text_stride = 2 * (2 * LEVR_PAD + ((c->maxlen + 15) & ~15) + 16);
    a.wave_bytes = a.pm_bytes + a.P * a.text_stride + 256;
    ar.b = a;
    ar.pick_row = nullptr; ar.pick_runmin = nullptr; ar.pick_out = nullptr; ar.pick_reset = 0; ar.pick_nx = 0;
    // fused anchor pick: each of the launch's waves scans all nx distances once, worth it while
    // that is small against the launch's own work (and the separate arg-max launch it replaces)
    if (src.anchor && src.pick_fused && c->nx <= 8192) {
        *src.pick_fused = true;
        if (src.pick_row) {
            ar.pick_row = src.pick_row; ar.pick_runmin = src.pick_runmin; ar.pick_out = src.pick_out;
            ar.pick_reset = src.pick_reset; ar.pick_nx = (int)c->nx;
        }
    }
    const size_t lds = (size_t)a.wave_bytes;
    ANN_REQUIRE(c, lds <= 160 * 1024, ANNCHOR_ELIMIT, "alphabet %d x length %d needs %zu B of LDS (> 160 KiB)", c->alphabet,
                c->maxlen, lds);
    if (lds > 64 * 1024)
        ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)k_lev_r<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t blocks = (npairs + a.P - 1) / a.P;
    const int64_t max_blocks = (int64_t)c->prop.multiProcessorCount * 32;
    if (blocks > max_blocks) blocks = max_blocks;
    k_lev_r<R><<<(int)blocks, ANN_WAVE, lds, c->stream>>>(ar);
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}


// ---------------------------------------------------------------------------------------
// k_lev_f: one word per lane like k_lev_r<1>, with the per-column instruction count cut from 19
// to 15 VALU and a bank-conflict-free match-mask table:
//   * the carry-in fix of a slot's first lane (top DP row: hp = 1, hn = 0) rides on the DPP move
//     itself -- `v_or_b32_dpp` / `v_and_b32_dpp` with a per-lane constant second operand -- instead
//     of two v_cndmask after two v_mov_dpp;
//   * the recurrence is written in v_bitop3_b32 terms (gfx950: any 3-input boolean function):
//     12 ALU ops per column instead of the 14 the compiler derives from the textbook form;
//   * PM[half][symbol][lane % 32]: a lane's match-mask words sit in bank lane % 32 for every
//     symbol, so the 32 lanes an LDS cycle serves never collide whatever symbols they look up
//     (the slot-major table had 63 % of its LDS-busy cycles in bank conflicts, profiles/r01_pmc_lev.json).
__device__ __forceinline__ uint32_t dpp_shr1_or(uint32_t v, uint32_t m)
{
    // lane l: v[l-1] | m[l]; lane 0 (no source lane, bound_ctrl): 0 | m[0]
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true) | m;
}
__device__ __forceinline__ uint32_t dpp_shr1_and(uint32_t v, uint32_t m)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true) & m;
}

#define LEVF_ROW 128   // bytes per symbol row of one half-wave's PM table (32 lanes x 4 B)

__global__ __launch_bounds__(ANN_WAVE) void k_lev_f(LevArgsR ar)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const LevArgs &a = ar.b;
    const int lane = threadIdx.x;
    const int A = a.alphabet;
    // PM: [2 halves][A symbols][32 lanes] uint32; this lane's column
    unsigned char *pm_col = smem + (size_t)(lane >> 5) * A * LEVF_ROW + (size_t)(lane & 31) * 4;
    const int Pmax = ar.perm ? ar.P0 : a.P;
    int *ssum = reinterpret_cast<int *>(smem + a.pm_bytes + (size_t)Pmax * a.text_stride);

    int picked = -1;
    if (ar.pick_row) {   // fused max-min pick, as in k_lev_r
        double bv = -INFINITY;
        int bi = 0x7fffffff;
        const int nx = ar.pick_nx;
        for (int j0 = 0; j0 < nx; j0 += 64 * 8) {
            double d[8], rm[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = min(j0 + e * 64 + lane, nx - 1);
                d[e] = ar.pick_row[j];
                rm[e] = ar.pick_reset ? 0.0 : ar.pick_runmin[j];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = j0 + e * 64 + lane;
                if (j < nx) {
                    const double v = ar.pick_reset ? d[e] : fmin(rm[e], d[e]);
                    if (blockIdx.x == 0) ar.pick_runmin[j] = v;
                    argmax_combine(bv, bi, v, j);
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(bv, off);
            const int oi = __shfl_xor(bi, off);
            argmax_combine(bv, bi, ov, oi);
        }
        picked = bi;
        if (blockIdx.x == 0 && lane == 0) *ar.pick_out = bi;
    }
    const int64_t n_short = ar.perm ? (int64_t)*ar.n0 : 0;
  for (int cls = ar.perm ? 0 : 1; cls < 2; ++cls) {
    // class 0: short patterns, P0 slots of GL0 lanes; class 1: the data set's general layout
    const int GL = cls == 0 ? ar.GL0 : ar.GL, P = cls == 0 ? ar.P0 : a.P;
    const int64_t n_cls = ar.perm ? (cls == 0 ? n_short : a.n - n_short) : a.n;
    const int g = lane / GL;         // pair slot of this lane
    const int w = lane - g * GL;     // word of the slot's pattern this lane owns
    const bool slot_ok = g < P;
    // text of a slot: one byte per symbol (its dense code), LEVR_PAD bytes of padding either side so that
    // the pipelined reads of lanes outside their column range stay inside the slot (values never used)
    uint8_t *txt_g = smem + a.pm_bytes + (size_t)(slot_ok ? g : 0) * a.text_stride;
    const int64_t n_tasks = (n_cls + P - 1) / P;
    if (slot_ok)
        for (int e = w * 16; e < a.text_stride; e += GL * 16) *reinterpret_cast<uint4 *>(txt_g + e) = make_uint4(0, 0, 0, 0);
    // carry-in constants of this lane: the first lane of a slot sees the row above the pattern
    // (horizontal delta +1: hp carry 1, hn carry 0) instead of the previous slot's last word
    uint32_t hp_or = w == 0 ? 0x80000000u : 0u;
    uint32_t hn_and = w == 0 ? 0u : 0xffffffffu;
    asm volatile("" : "+v"(hp_or), "+v"(hn_and));   // opaque: keeps them operands of v_or_b32_dpp / v_and_b32_dpp (not a select)

    for (int64_t task = blockIdx.x; task < n_tasks; task += gridDim.x) {
        const int64_t t_cls = task * P + g;
        const bool active = slot_ok && t_cls < n_cls;
        int64_t t_pair = t_cls;
        if (active && ar.perm) t_pair = cls == 0 ? ar.perm[t_cls] : ar.perm[a.n - 1 - t_cls];
        int si = 0, sj = 0;
        int64_t opos = t_pair;
        if (active) {
            if (a.anchor) { si = picked >= 0 ? picked : *a.anchor; sj = (int)t_pair; }
            else {
                int64_t q = a.idx ? a.idx[t_pair] : t_pair;
                int2 p = a.ij[q];
                si = p.x; sj = p.y;
                if (a.idx) opos = q;
            }
        }
        const int li = active ? a.slen[si] : 0, lj = active ? a.slen[sj] : 0;
        // the pattern (bit-vector side) is the longer string -- fewer text columns to walk -- except in the
        // short class, whose point is that the SHORTER string fits the narrow slot
        const bool swap = cls == 0 ? li > lj : li < lj;
        const int ps = swap ? sj : si, ts = swap ? si : sj;
        const int m = swap ? lj : li, n = swap ? li : lj;
        const uint8_t *pat = a.sym + (active ? a.soff[ps] : 0);
        const uint8_t *tex = a.sym + (active ? a.soff[ts] : 0);
        const int Wp = (m + 31) >> 5;          // pattern words = lanes that hold pattern words
        if (slot_ok && w == 0) ssum[g] = 0;
        // every lane clears its own column (lanes outside the slots too: their reads must stay defined)
        for (int c = 0; c < A; ++c) *reinterpret_cast<uint32_t *>(pm_col + (size_t)c * LEVF_ROW) = 0u;
        if (active && w < Wp) {
            const uint4 *p16 = reinterpret_cast<const uint4 *>(pat + w * 32);
            const uint4 q0 = p16[0], q1 = p16[1];
            const uint32_t wd[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            const int valid = min(32, m - w * 32);
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const uint32_t c = (wd[k >> 2] >> ((k & 3) * 8)) & 0xffu;
                // the column is private to this lane; LDS OR without return: the 32 updates go out back to back
                // instead of 32 dependent read-modify-write round trips (370 -> 329 us per 65 536 pairs)
                if (k < valid) atomicOr(reinterpret_cast<uint32_t *>(pm_col + (size_t)c * LEVF_ROW), 1u << k);
            }
        }
        if (active) {
            const int chunks = (n + 15) >> 4;
            for (int ch = w; ch < chunks; ch += GL)
                reinterpret_cast<uint4 *>(txt_g + LEVR_PAD)[ch] = reinterpret_cast<const uint4 *>(tex)[ch];
        }
        wave_lds_fence();

        uint32_t vp = 0xffffffffu, vn = 0u;
        const uint32_t un = (active && m > 0) ? (uint32_t)n : 0u;
        int max_steps = (active && m > 0) ? n + Wp - 1 : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) max_steps = max(max_steps, __shfl_xor(max_steps, off));
        max_steps = __builtin_amdgcn_readfirstlane(max_steps);
        const uint8_t *tp = txt_g + LEVR_PAD - w;   // tp[k] = this lane's symbol at iteration k (row = symbol * LEVF_ROW: one v_lshl_add)
        uint32_t c1 = tp[1];
        uint32_t eq = *reinterpret_cast<const uint32_t *>(pm_col + (uint32_t)tp[0] * LEVF_ROW);
        uint32_t out_hp = 0, out_hn = 0;   // hp / hn of this lane's word in the previous iteration
        auto column = [&](int k, auto checked) {
            const uint32_t c2 = tp[k + 2];
            const uint32_t eq_n = *reinterpret_cast<const uint32_t *>(pm_col + c1 * LEVF_ROW);
            const uint32_t hp_up = dpp_shr1_or(out_hp, hp_or), hn_up = dpp_shr1_and(out_hn, hn_and);
            __builtin_amdgcn_sched_barrier(0);
            const bool valid = !decltype(checked)::value || (uint32_t)(k - w) < un;
            const uint32_t c = hn_up >> 31;
            const uint32_t x = eq | c;
            const uint32_t t = __builtin_amdgcn_bitop3_b32(c, eq, vp, 0xa8);       // (c | eq) & vp
            const uint32_t sm = t + vp;
            const uint32_t d0p = __builtin_amdgcn_bitop3_b32(sm, vp, x, 0xbe);   // (sm ^ vp) | x
            const uint32_t hp = __builtin_amdgcn_bitop3_b32(vn, d0p, vp, 0xf1);    // vn | ~(d0p | vp)
            const uint32_t d0 = d0p | vn;
            const uint32_t hn = d0 & vp;
            const uint32_t hps = __builtin_amdgcn_alignbit(hp, hp_up, 31);          // (hp << 1) | carry from the word above
            const uint32_t hns = __builtin_amdgcn_alignbit(hn, hn_up, 31);
            const uint32_t nvp = __builtin_amdgcn_bitop3_b32(hns, d0, hps, 0xf1);  // hns | ~(d0 | hps)
            const uint32_t nvn = hps & d0;
            vp = valid ? nvp : vp;
            vn = valid ? nvn : vn;
            out_hp = hp;
            out_hn = hn;
            __builtin_amdgcn_sched_barrier(0);
            eq = eq_n;
            c1 = c2;
        };
        int k_lo = min(GL - 1, max_steps), k_hi = (active && m > 0) ? n : 0x7fffffff;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) k_hi = min(k_hi, __shfl_xor(k_hi, off));
        k_hi = max(k_lo, min(__builtin_amdgcn_readfirstlane(k_hi), max_steps));
        int k = 0;
        for (; k < k_lo; ++k) column(k, std::true_type());
        for (; k + 2 <= k_hi; k += 2) { column(k, std::false_type()); column(k + 1, std::false_type()); }
        for (; k < k_hi; ++k) column(k, std::false_type());
        for (; k < max_steps; ++k) column(k, std::true_type());
        if (active) {
            const uint32_t rows = w < Wp - 1 ? 0xffffffffu : (w == Wp - 1 ? (0xffffffffu >> (31 - ((m - 1) & 31))) : 0u);
            const int part = __popc(vp & rows) - __popc(vn & rows);
            if (part) atomicAdd(&ssum[g], part);
        }
        wave_lds_fence();
        if (active && w == 0) {
            const double d = (double)(n + ssum[g]);
            if (a.out) a.out[t_pair] = d;
            if (a.RA) { a.RA[opos] = d; a.ncm[opos] = 0; }
        }
        wave_lds_fence();
    }
  }
}

// Splits a pair list into the two slot classes of k_lev_f: list position t goes to the front of perm when
// the shorter string of its pair has <= gl0 words, to the back otherwise (block-wise ranges reserved with
// two atomic cursors: the order inside a class is arbitrary, every pair's result is stored by position).
__global__ __launch_bounds__(256) void k_lev_classify(const int2 *__restrict__ ij, const int32_t *__restrict__ idx,
                                                      const int32_t *__restrict__ slen, int64_t n, int gl0,
                                                      int32_t *__restrict__ perm, int32_t *__restrict__ cursors,
                                                      int32_t *__restrict__ next_cursors)
{
    // (the counters of the NEXT classification are zeroed here: its slot is idle -- whoever read it finished before this
    // launch started -- and a memset launch per classification is saved)
    if (blockIdx.x == 0 && threadIdx.x < 2) next_cursors[threadIdx.x] = 0;
    __shared__ int wcnt[2][4];
    __shared__ int base[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool is_short = false, is_long = false;
    if (t < n) {
        const int2 p = ij[idx ? idx[t] : t];
        const int m = min(slen[p.x], slen[p.y]);
        is_short = ((m + 31) >> 5) <= gl0;
        is_long = !is_short;
    }
    const unsigned long long bs = __ballot(is_short), bl = __ballot(is_long);
    if (lane == 0) { wcnt[0][wave] = __popcll(bs); wcnt[1][wave] = __popcll(bl); }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int tot = wcnt[threadIdx.x][0] + wcnt[threadIdx.x][1] + wcnt[threadIdx.x][2] + wcnt[threadIdx.x][3];
        base[threadIdx.x] = tot ? atomicAdd(&cursors[threadIdx.x], tot) : 0;
    }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
    if (is_short) {
        int r = base[0] + __popcll(bs & below);
        for (int q = 0; q < wave; ++q) r += wcnt[0][q];
        perm[r] = (int32_t)t;
    } else if (is_long) {
        int r = base[1] + __popcll(bl & below);
        for (int q = 0; q < wave; ++q) r += wcnt[1][q];
        perm[n - 1 - r] = (int32_t)t;
    }
}

// the class counters of the next k_lev_classify launch (zeroed by the previous one, or here on first use) and the slot after it
static int lev_next_cursors(annchor_ctx *c, int32_t **cur, int32_t **nxt)
{
    if (!c->lev_cursors.p) {
        ANN_TRY(ann_reserve(c, c->lev_cursors, 4 * sizeof(int32_t)));
        ANN_CHECK_HIP(c, hipMemsetAsync(c->lev_cursors.p, 0, 4 * sizeof(int32_t), c->stream));
        c->lev_cursor_epoch = 0;
    }
    *cur = c->lev_cursors.as<int32_t>() + 2 * (c->lev_cursor_epoch & 1);
    *nxt = c->lev_cursors.as<int32_t>() + 2 * ((c->lev_cursor_epoch + 1) & 1);
    ++c->lev_cursor_epoch;
    return ANNCHOR_OK;
}

static int launch_f(annchor_ctx *c, LevArgs a, int64_t npairs, const PairSource &src)
{
    const int W = (c->maxlen + 31) / 32 > 0 ? (c->maxlen + 31) / 32 : 1;
    LevArgsR ar;
    ar.GL = W;
    ar.pm_stride = 32;
    a.G = ar.GL;
    a.P = 64 / ar.GL;
    a.pm_bytes = 2 * a.alphabet * LEVF_ROW;
    a.text_stride = 2 * LEVR_PAD + ((c->maxlen + 15) & ~15) + 16;
    // Slot classes: with strings of up to W words a wave holds P = 64 / W pairs, yet the pattern is the
    // SHORTER string of a pair; pairs whose pattern has <= GL0 = 64 / (P + 1) words run P + 1 to a wave.
    // Worth one classification launch when the list is long and enough pairs qualify (share of pairs with
    // a short string on either side, from the length census of annchor_set_strings).
    ar.perm = nullptr; ar.n0 = nullptr; ar.GL0 = 0; ar.P0 = 0;
    static const int cls_min = getenv("ANNCHOR_LEV_CLASS_MIN") ? atoi(getenv("ANNCHOR_LEV_CLASS_MIN")) : 4096;
    if (!src.anchor && c->lev_gl0 > 0 && npairs >= cls_min && npairs < (1ll << 31)) {
        const double fp = 1.0 - (1.0 - c->lev_frac0) * (1.0 - c->lev_frac0);
        const int P0 = 64 / c->lev_gl0;
        const double cost = fp * (double)a.P / (double)P0 + (1.0 - fp);   // waves needed, relative to one class
        if (cost < 0.93) {
            ANN_TRY(ann_reserve(c, c->lev_perm, sizeof(int32_t) * ((size_t)npairs + 4)));
            int32_t *perm = c->lev_perm.as<int32_t>();
            int32_t *cursors = nullptr, *next_cursors = nullptr;
            ANN_TRY(lev_next_cursors(c, &cursors, &next_cursors));
            k_lev_classify<<<ann_blocks(npairs, 256), 256, 0, c->stream>>>(src.ij, src.idx, a.slen, npairs, c->lev_gl0, perm,
                                                                           cursors, next_cursors);
            ar.perm = perm; ar.n0 = cursors; ar.GL0 = c->lev_gl0; ar.P0 = P0;
        }
    }
    a.wave_bytes = a.pm_bytes + (ar.perm ? ar.P0 : a.P) * a.text_stride + 256;
    ar.b = a;
    ar.pick_row = nullptr; ar.pick_runmin = nullptr; ar.pick_out = nullptr; ar.pick_reset = 0; ar.pick_nx = 0;
    if (src.anchor && src.pick_fused && c->nx <= 8192) {
        *src.pick_fused = true;
        if (src.pick_row) {
            ar.pick_row = src.pick_row; ar.pick_runmin = src.pick_runmin; ar.pick_out = src.pick_out;
            ar.pick_reset = src.pick_reset; ar.pick_nx = (int)c->nx;
        }
    }
    const size_t lds = (size_t)a.wave_bytes;
    ANN_REQUIRE(c, lds <= 160 * 1024, ANNCHOR_ELIMIT, "alphabet %d x length %d needs %zu B of LDS (> 160 KiB)", c->alphabet,
                c->maxlen, lds);
    if (lds > 64 * 1024)
        ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)k_lev_f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t blocks = (npairs + a.P - 1) / a.P;
    const int64_t max_blocks = (int64_t)c->prop.multiProcessorCount * 32;
    if (blocks > max_blocks) blocks = max_blocks;
    k_lev_f<<<(int)blocks, ANN_WAVE, lds, c->stream>>>(ar);
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}

struct LevArgsA {
    const uint8_t *sym;
    const int32_t *soff;
    const int32_t *slen;
    const int32_t *anchor;    // anchor of this launch when it is not picked here
    int64_t n;                // targets: pair t = (anchor, t)
    double *out;
    int alphabet, text_stride, pm_bytes, fb_stride;
    const double *pick_row;
    double *pick_runmin;
    int32_t *pick_out;
    int pick_reset, pick_nx;
};

// ---------------------------------------------------------------------------------------
// k_lev_a2: k_lev_a with the waves packed for ONE wave per SIMD.  k_lev_a's nx waves of half-length chains ran as
// long as the nx / 3 full-length ones they replaced (36.8 vs 38.6 us per round at C2: 1600 waves on 1024 SIMDs put
// two on most SIMDs, and one of these waves keeps its SIMD's issue port ~60 % busy on its own).  Here a wave holds
// FOUR half-chains -- two pairs, slots of 16 lanes -- whenever the shorter string of a pair fits 16 words
// (<= 512 symbols: it becomes the bit-vector pattern, the longer one the text); pairs of two longer strings keep a wave
// to themselves (slots of 32 lanes).  Strings are listed short ones first (lev_order, built when they are bound), so
// wave b knows its pairs without knowing the anchor: C2 = 635 + 330 = 965 waves <= 1024 SIMDs, each walking
// ~max(len) / 2 + 16 columns.
struct LevArgsA2 {
    LevArgsA a;
    const int32_t *order;   // string ids, <= 16-word strings first
    int n_short;            // how many of them
    int buf_stride;         // bytes per staged string
    int pad;                // zero bytes either side of a staged string: a lane outside its column range reads up to
                            // max(len) / 2 + 34 symbols past either end (never used, but they index the match-mask table)
};

__global__ __launch_bounds__(ANN_WAVE) void k_lev_a2(LevArgsA2 aa)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const LevArgsA &a = aa.a;
    const int lane = threadIdx.x, A = a.alphabet;
    const int ws = (aa.n_short + 1) >> 1;                 // waves of the packed class
    const bool packed = (int)blockIdx.x < ws;
    const int GL = packed ? 16 : 32;
    const int slot = lane / GL, w = lane - slot * GL, pair = slot >> 1, half = slot & 1;
    const int LP = packed ? 32 : 64;                      // lanes per pair
    unsigned char *pm_col = smem + (size_t)(lane >> 5) * A * LEVF_ROW + (size_t)(lane & 31) * 4;
    uint8_t *bufs = smem + a.pm_bytes;                    // [0], [1]: the wave's strings t0 / t1, [2]: the anchor
    int16_t *FB = reinterpret_cast<int16_t *>(bufs + 3 * aa.buf_stride);   // [pair][half][fb_stride]
    for (int e = lane * 16; e < 3 * aa.buf_stride; e += 64 * 16) *reinterpret_cast<uint4 *>(bufs + e) = make_uint4(0, 0, 0, 0);
    uint32_t hp_or = w == 0 ? 0x80000000u : 0u;
    uint32_t hn_and = w == 0 ? 0u : 0xffffffffu;
    asm volatile("" : "+v"(hp_or), "+v"(hn_and));
    // ---- this wave's strings
    int tq[2] = {-1, -1};
    if (packed) {
        tq[0] = aa.order[2 * blockIdx.x];
        if (2 * (int)blockIdx.x + 1 < aa.n_short) tq[1] = aa.order[2 * blockIdx.x + 1];
    } else {
        tq[0] = aa.order[aa.n_short + ((int)blockIdx.x - ws)];
    }
    int ln[2] = {0, 0};
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if (tq[q] >= 0) {
            ln[q] = a.slen[tq[q]];
            const uint8_t *src = a.sym + a.soff[tq[q]];
            for (int ch = lane; ch < ((ln[q] + 15) >> 4); ch += 64)
                reinterpret_cast<uint4 *>(bufs + q * aa.buf_stride + aa.pad)[ch] = reinterpret_cast<const uint4 *>(src)[ch];
        }
    // ---- the anchor (fused max-min pick: every wave for itself, the strings' loads above in flight meanwhile)
    int si;
    if (a.pick_row) {
        double bv = -INFINITY;
        int bi = 0x7fffffff;
        const int nx = a.pick_nx;
        for (int j0 = 0; j0 < nx; j0 += 64 * 16) {
            double d[16], rm[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int j = min(j0 + e * 64 + lane, nx - 1);
                d[e] = a.pick_row[j];
                rm[e] = a.pick_reset ? 0.0 : a.pick_runmin[j];
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int j = j0 + e * 64 + lane;
                if (j < nx) {
                    const double v = a.pick_reset ? d[e] : fmin(rm[e], d[e]);
                    if (blockIdx.x == 0) a.pick_runmin[j] = v;
                    argmax_combine(bv, bi, v, j);
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(bv, off);
            const int oi = __shfl_xor(bi, off);
            argmax_combine(bv, bi, ov, oi);
        }
        si = bi;
        if (blockIdx.x == 0 && lane == 0) *a.pick_out = bi;
    } else {
        si = *a.anchor;
    }
    const int la = a.slen[si];
    {
        const uint8_t *src = a.sym + a.soff[si];
        for (int ch = lane; ch < ((la + 15) >> 4); ch += 64)
            reinterpret_cast<uint4 *>(bufs + 2 * aa.buf_stride + aa.pad)[ch] = reinterpret_cast<const uint4 *>(src)[ch];
    }
    wave_lds_fence();
    // ---- roles of this lane's pair: packed class -- pattern = the shorter string (fits the 16-lane slot); otherwise
    // pattern = the longer one (fewer text columns)
    const bool have = pair < 2 && tq[pair < 2 ? pair : 0] >= 0 && (packed || pair == 0);
    const int lt = have ? ln[pair] : 0;
    const bool t_is_pattern = packed ? (lt <= la) : (lt > la);
    const int m = have ? (t_is_pattern ? lt : la) : 0;          // pattern length
    const int n = have ? (t_is_pattern ? la : lt) : 0;          // text length
    const uint8_t *pat = bufs + (t_is_pattern ? pair : 2) * aa.buf_stride + aa.pad;
    const uint8_t *txt = bufs + (t_is_pattern ? 2 : pair) * aa.buf_stride + aa.pad;
    const int Wp = (m + 31) >> 5;
    for (int c = 0; c < A; ++c) *reinterpret_cast<uint32_t *>(pm_col + (size_t)c * LEVF_ROW) = 0u;
    if (have && w < Wp) {
        const int valid = min(32, m - w * 32);
        uint32_t sy[32];
        if (half == 0) {
            const uint4 *p16 = reinterpret_cast<const uint4 *>(pat + w * 32);
            const uint4 q0 = p16[0], q1 = p16[1];
            const uint32_t wd[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
            for (int k = 0; k < 32; ++k) sy[k] = (wd[k >> 2] >> ((k & 3) * 8)) & 0xffu;
        } else {
#pragma unroll
            for (int k = 0; k < 32; ++k) sy[k] = pat[max(m - 1 - (w * 32 + k), 0)];
        }
#pragma unroll
        for (int k = 0; k < 32; ++k)
            if (k < valid) atomicOr(reinterpret_cast<uint32_t *>(pm_col + (size_t)sy[k] * LEVF_ROW), 1u << k);
    }
    wave_lds_fence();

    const int h = (n + 1) >> 1;
    const uint32_t un = (have && m > 0) ? (uint32_t)(half ? n - h : h) : 0u;
    int max_steps = (have && m > 0) ? h + Wp - 1 : 0;
    int k_lo = (have && m > 0) ? Wp - 1 : 0, k_hi = (have && m > 0) ? n - h : 0x7fffffff;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        max_steps = max(max_steps, __shfl_xor(max_steps, off));
        k_lo = max(k_lo, __shfl_xor(k_lo, off));
        k_hi = min(k_hi, __shfl_xor(k_hi, off));
    }
    max_steps = __builtin_amdgcn_readfirstlane(max_steps);
    k_lo = min(__builtin_amdgcn_readfirstlane(k_lo), max_steps);
    k_hi = max(k_lo, min(__builtin_amdgcn_readfirstlane(k_hi), max_steps));
    const int dir = half ? -1 : 1;
    const uint8_t *tp = txt + (half ? n - 1 + w : -w);
    uint32_t vp = 0xffffffffu, vn = 0u;
    uint32_t c1 = tp[dir];
    uint32_t eq = *reinterpret_cast<const uint32_t *>(pm_col + (uint32_t)tp[0] * LEVF_ROW);
    uint32_t out_hp = 0, out_hn = 0;
    const uint8_t *tnext = tp + 2 * dir;
    auto column = [&](int k, auto checked) {
        const uint32_t c2 = *tnext;
        tnext += dir;
        const uint32_t eq_n = *reinterpret_cast<const uint32_t *>(pm_col + c1 * LEVF_ROW);
        const uint32_t hp_up = dpp_shr1_or(out_hp, hp_or), hn_up = dpp_shr1_and(out_hn, hn_and);
        __builtin_amdgcn_sched_barrier(0);
        const bool valid = !decltype(checked)::value || (uint32_t)(k - w) < un;
        const uint32_t c = hn_up >> 31;
        const uint32_t x = eq | c;
        const uint32_t tt = __builtin_amdgcn_bitop3_b32(c, eq, vp, 0xa8);
        const uint32_t sm = tt + vp;
        const uint32_t d0p = __builtin_amdgcn_bitop3_b32(sm, vp, x, 0xbe);
        const uint32_t hp = __builtin_amdgcn_bitop3_b32(vn, d0p, vp, 0xf1);
        const uint32_t d0 = d0p | vn;
        const uint32_t hn = d0 & vp;
        const uint32_t hps = __builtin_amdgcn_alignbit(hp, hp_up, 31);
        const uint32_t hns = __builtin_amdgcn_alignbit(hn, hn_up, 31);
        const uint32_t nvp = __builtin_amdgcn_bitop3_b32(hns, d0, hps, 0xf1);
        const uint32_t nvn = hps & d0;
        vp = valid ? nvp : vp;
        vn = valid ? nvn : vn;
        out_hp = hp;
        out_hn = hn;
        __builtin_amdgcn_sched_barrier(0);
        eq = eq_n;
        c1 = c2;
    };
    int k = 0;
    for (; k < k_lo; ++k) column(k, std::true_type());
    for (; k + 2 <= k_hi; k += 2) { column(k, std::false_type()); column(k + 1, std::false_type()); }
    for (; k < k_hi; ++k) column(k, std::false_type());
    for (; k < max_steps; ++k) column(k, std::true_type());

    // ---- F / B' (prefix sums of the vertical deltas down this half's rows), then min_i F[i] + B'[m - i] per pair
    const uint32_t rows = !have ? 0u : (w < Wp - 1 ? 0xffffffffu : (w == Wp - 1 ? (0xffffffffu >> (31 - ((m - 1) & 31))) : 0u));
    const int part = __popc(vp & rows) - __popc(vn & rows);
    int incl = part;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
        const int o = __shfl_up(incl, off, 32);
        if (w >= off) incl += o;
    }
    int val = (int)un + incl - part;
    if (m == 0) val = half ? n - h : h;
    int16_t *fbp = FB + (size_t)(pair < 2 ? pair : 0) * 2 * a.fb_stride;
    int16_t *fb = fbp + (size_t)half * a.fb_stride;
    if (have && w == 0) fb[0] = (int16_t)val;
#pragma unroll
    for (int b = 0; b < 32; ++b) {
        val += (int)((vp >> b) & 1u) - (int)((vn >> b) & 1u);
        if ((rows >> b) & 1u) fb[w * 32 + b + 1] = (int16_t)val;
    }
    wave_lds_fence();
    int best = 0x7fffffff;
    const int lp = lane & (LP - 1);
    if (have)
        for (int i = lp; i <= m; i += LP) best = min(best, (int)fbp[i] + (int)fbp[a.fb_stride + m - i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        if (off < LP) best = min(best, __shfl_xor(best, off));
    if (have && lp == 0) a.out[tq[pair]] = (double)best;
}

// k_lev_p2: the packed latency kernel for SHORT pair lists (a sampling step's 5000 pairs, a small refinement): the same two
// pairs per wave / forward-backward split as k_lev_a2, each pair with its own two strings; pairs come from the class
// permutation k_lev_classify builds (short patterns first).  Such a launch is 1250 four-pair waves on 1024 SIMDs for k_lev_f --
// twice a 500-column chain where two waves share a SIMD -- and 2500 half-chain waves here.
struct LevArgsP2 {
    const uint8_t *sym;
    const int32_t *soff;
    const int32_t *slen;
    const int2 *ij;
    const int32_t *idx;
    const int32_t *perm;      // list positions, short-pattern pairs first, the others from the back
    const int32_t *cursors;   // [0] pairs of the packed class, [1] the others
    int64_t n;
    double *out;
    double *RA;
    uint8_t *ncm;
    int alphabet, pm_bytes, fb_stride, buf_stride, pad;
};

__global__ __launch_bounds__(ANN_WAVE) void k_lev_p2(LevArgsP2 aa)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const LevArgsP2 &a = aa;
    const int lane = threadIdx.x, A = a.alphabet;
    const int n_short = a.cursors[0], n_long = a.cursors[1];
    const int ws = (n_short + 1) >> 1;                    // tasks of the packed class (two pairs each), then one per long pair
    // (the class sizes are known on the device only: the grid is the host's upper bound, a wave takes every gridDim-th task)
    for (int task = blockIdx.x; task < ws + n_long; task += gridDim.x) {
    const bool packed = task < ws;
    const int GL = packed ? 16 : 32;
    const int slot = lane / GL, w = lane - slot * GL, pair = slot >> 1, half = slot & 1;
    const int LP = packed ? 32 : 64;                      // lanes per pair
    unsigned char *pm_col = smem + (size_t)(lane >> 5) * A * LEVF_ROW + (size_t)(lane & 31) * 4;
    uint8_t *bufs = smem + a.pm_bytes;                    // [2 q], [2 q + 1]: the two strings of the wave's pair q
    int16_t *FB = reinterpret_cast<int16_t *>(bufs + 4 * aa.buf_stride);   // [pair][half][fb_stride]
    for (int e = lane * 16; e < 4 * aa.buf_stride; e += 64 * 16) *reinterpret_cast<uint4 *>(bufs + e) = make_uint4(0, 0, 0, 0);
    uint32_t hp_or = w == 0 ? 0x80000000u : 0u;
    uint32_t hn_and = w == 0 ? 0u : 0xffffffffu;
    asm volatile("" : "+v"(hp_or), "+v"(hn_and));
    // ---- this wave's pairs: list positions from the class permutation (short patterns first, long ones from the back)
    int tq[2] = {-1, -1};
    if (packed) {
        tq[0] = a.perm[2 * task];
        if (2 * task + 1 < n_short) tq[1] = a.perm[2 * task + 1];
    } else {
        tq[0] = a.perm[n_short + (task - ws)];
    }
    int si[2] = {0, 0}, sj[2] = {0, 0}, li[2] = {0, 0}, lj[2] = {0, 0};
    int64_t opos[2] = {0, 0};
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if (tq[q] >= 0) {
            opos[q] = a.idx ? (int64_t)a.idx[tq[q]] : (int64_t)tq[q];
            const int2 pq = a.ij[opos[q]];
            si[q] = pq.x; sj[q] = pq.y;
        }
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if (tq[q] >= 0) {
            li[q] = a.slen[si[q]]; lj[q] = a.slen[sj[q]];
            const uint8_t *s0 = a.sym + a.soff[si[q]], *s1 = a.sym + a.soff[sj[q]];
            for (int ch = lane; ch < ((li[q] + 15) >> 4); ch += 64)
                reinterpret_cast<uint4 *>(bufs + (2 * q) * aa.buf_stride + aa.pad)[ch] = reinterpret_cast<const uint4 *>(s0)[ch];
            for (int ch = lane; ch < ((lj[q] + 15) >> 4); ch += 64)
                reinterpret_cast<uint4 *>(bufs + (2 * q + 1) * aa.buf_stride + aa.pad)[ch] = reinterpret_cast<const uint4 *>(s1)[ch];
        }
    wave_lds_fence();
    // ---- roles of this lane's pair: packed class -- pattern = the shorter string (fits the 16-lane slot); otherwise
    // pattern = the longer one (fewer text columns)
    const bool have = pair < 2 && tq[pair < 2 ? pair : 0] >= 0 && (packed || pair == 0);
    const int pq = pair < 2 ? pair : 0;
    const int l0 = have ? li[pq] : 0, l1 = have ? lj[pq] : 0;
    // packed class: pattern = the shorter string (fits the 16-lane slot); otherwise the longer one (fewer text columns)
    const bool first_is_pattern = packed ? (l0 <= l1) : (l0 > l1);
    const int m = have ? (first_is_pattern ? l0 : l1) : 0;          // pattern length
    const int n = have ? (first_is_pattern ? l1 : l0) : 0;          // text length
    const uint8_t *pat = bufs + (2 * pq + (first_is_pattern ? 0 : 1)) * aa.buf_stride + aa.pad;
    const uint8_t *txt = bufs + (2 * pq + (first_is_pattern ? 1 : 0)) * aa.buf_stride + aa.pad;
    const int Wp = (m + 31) >> 5;
    for (int c = 0; c < A; ++c) *reinterpret_cast<uint32_t *>(pm_col + (size_t)c * LEVF_ROW) = 0u;
    if (have && w < Wp) {
        const int valid = min(32, m - w * 32);
        uint32_t sy[32];
        if (half == 0) {
            const uint4 *p16 = reinterpret_cast<const uint4 *>(pat + w * 32);
            const uint4 q0 = p16[0], q1 = p16[1];
            const uint32_t wd[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
            for (int k = 0; k < 32; ++k) sy[k] = (wd[k >> 2] >> ((k & 3) * 8)) & 0xffu;
        } else {
#pragma unroll
            for (int k = 0; k < 32; ++k) sy[k] = pat[max(m - 1 - (w * 32 + k), 0)];
        }
#pragma unroll
        for (int k = 0; k < 32; ++k)
            if (k < valid) atomicOr(reinterpret_cast<uint32_t *>(pm_col + (size_t)sy[k] * LEVF_ROW), 1u << k);
    }
    wave_lds_fence();

    const int h = (n + 1) >> 1;
    const uint32_t un = (have && m > 0) ? (uint32_t)(half ? n - h : h) : 0u;
    int max_steps = (have && m > 0) ? h + Wp - 1 : 0;
    int k_lo = (have && m > 0) ? Wp - 1 : 0, k_hi = (have && m > 0) ? n - h : 0x7fffffff;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        max_steps = max(max_steps, __shfl_xor(max_steps, off));
        k_lo = max(k_lo, __shfl_xor(k_lo, off));
        k_hi = min(k_hi, __shfl_xor(k_hi, off));
    }
    max_steps = __builtin_amdgcn_readfirstlane(max_steps);
    k_lo = min(__builtin_amdgcn_readfirstlane(k_lo), max_steps);
    k_hi = max(k_lo, min(__builtin_amdgcn_readfirstlane(k_hi), max_steps));
    const int dir = half ? -1 : 1;
    const uint8_t *tp = txt + (half ? n - 1 + w : -w);
    uint32_t vp = 0xffffffffu, vn = 0u;
    uint32_t c1 = tp[dir];
    uint32_t eq = *reinterpret_cast<const uint32_t *>(pm_col + (uint32_t)tp[0] * LEVF_ROW);
    uint32_t out_hp = 0, out_hn = 0;
    const uint8_t *tnext = tp + 2 * dir;
    auto column = [&](int k, auto checked) {
        const uint32_t c2 = *tnext;
        tnext += dir;
        const uint32_t eq_n = *reinterpret_cast<const uint32_t *>(pm_col + c1 * LEVF_ROW);
        const uint32_t hp_up = dpp_shr1_or(out_hp, hp_or), hn_up = dpp_shr1_and(out_hn, hn_and);
        __builtin_amdgcn_sched_barrier(0);
        const bool valid = !decltype(checked)::value || (uint32_t)(k - w) < un;
        const uint32_t c = hn_up >> 31;
        const uint32_t x = eq | c;
        const uint32_t tt = __builtin_amdgcn_bitop3_b32(c, eq, vp, 0xa8);
        const uint32_t sm = tt + vp;
        const uint32_t d0p = __builtin_amdgcn_bitop3_b32(sm, vp, x, 0xbe);
        const uint32_t hp = __builtin_amdgcn_bitop3_b32(vn, d0p, vp, 0xf1);
        const uint32_t d0 = d0p | vn;
        const uint32_t hn = d0 & vp;
        const uint32_t hps = __builtin_amdgcn_alignbit(hp, hp_up, 31);
        const uint32_t hns = __builtin_amdgcn_alignbit(hn, hn_up, 31);
        const uint32_t nvp = __builtin_amdgcn_bitop3_b32(hns, d0, hps, 0xf1);
        const uint32_t nvn = hps & d0;
        vp = valid ? nvp : vp;
        vn = valid ? nvn : vn;
        out_hp = hp;
        out_hn = hn;
        __builtin_amdgcn_sched_barrier(0);
        eq = eq_n;
        c1 = c2;
    };
    int k = 0;
    for (; k < k_lo; ++k) column(k, std::true_type());
    for (; k + 2 <= k_hi; k += 2) { column(k, std::false_type()); column(k + 1, std::false_type()); }
    for (; k < k_hi; ++k) column(k, std::false_type());
    for (; k < max_steps; ++k) column(k, std::true_type());

    // ---- F / B' (prefix sums of the vertical deltas down this half's rows), then min_i F[i] + B'[m - i] per pair
    const uint32_t rows = !have ? 0u : (w < Wp - 1 ? 0xffffffffu : (w == Wp - 1 ? (0xffffffffu >> (31 - ((m - 1) & 31))) : 0u));
    const int part = __popc(vp & rows) - __popc(vn & rows);
    int incl = part;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
        const int o = __shfl_up(incl, off, 32);
        if (w >= off) incl += o;
    }
    int val = (int)un + incl - part;
    if (m == 0) val = half ? n - h : h;
    int16_t *fbp = FB + (size_t)(pair < 2 ? pair : 0) * 2 * a.fb_stride;
    int16_t *fb = fbp + (size_t)half * a.fb_stride;
    if (have && w == 0) fb[0] = (int16_t)val;
#pragma unroll
    for (int b = 0; b < 32; ++b) {
        val += (int)((vp >> b) & 1u) - (int)((vn >> b) & 1u);
        if ((rows >> b) & 1u) fb[w * 32 + b + 1] = (int16_t)val;
    }
    wave_lds_fence();
    int best = 0x7fffffff;
    const int lp = lane & (LP - 1);
    if (have)
        for (int i = lp; i <= m; i += LP) best = min(best, (int)fbp[i] + (int)fbp[a.fb_stride + m - i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        if (off < LP) best = min(best, __shfl_xor(best, off));
    if (have && lp == 0) {
        const double d = (double)best;
        if (a.out) a.out[tq[pq]] = d;
        if (a.RA) { a.RA[opos[pq]] = d; a.ncm[opos[pq]] = 0; }
    }
    wave_lds_fence();
    }   // task
}


// ---------------------------------------------------------------------------------------
// k_lev_ap: EVERY round of the max-min picker (pickers.py:44-50) in one launch.  The rounds are a chain -- round r + 1's
// anchor is the arg-max over round r's distances -- and as 15 launches of k_lev_a2 each of them pays a dispatch gap, the
// staging of the wave's own strings, a match-mask build and an arg-max scan of the whole row (~15 of 26 us per round at C2).
// Here the waves of k_lev_a2 (two short strings or one long string per wave, one wave per SIMD) stay resident:
//   * a wave's own strings are ALWAYS the bit-vector patterns (they fit their slots by construction), the anchor is the text:
//     the match-mask tables are built once, a round stages only the anchor's symbols, and every wave of a round walks
//     the same number of columns (len(anchor) / 2 + words) -- nobody waits at the barrier for a longer text;
//   * a wave owns the running minima of its own points (registers): the arg-max is a max over one 32-bit key per wave
//     [running minimum : 16 | 0xffff - point : 16] (np.argmax: first maximal index), and the grid barrier IS that
//     reduction -- every wave stores (round tag, key) to its arrival slot; <= 16 collector waves poll 64 slots each until they
//     carry the round's tag and publish the maximum, everybody polls the collectors' slots; no atomics (12.5 ns each on one
//     address, serialised);
//   * anchorRank (picker.hip: k_fill_i32 + k_anchor_rank) falls out: a wave knows every round's anchor.
// All waves must be resident (the launcher checks the occupancy).  Should they not be -- another process holding CUs -- a
// wave that polls longer than the time limit raises the abort word and everybody leaves; the RESCUE instantiation,
// enqueued behind every persistent launch, then redoes the rounds as ONE workgroup (no residency assumption) and returns
// at once otherwise.
struct LevArgsAP {
    const uint8_t *sym;
    const int32_t *soff;
    const int32_t *slen;
    const int32_t *order;         // string ids, <= 16-word strings first
    double *Dt;                   // [na][nx]
    int32_t *A;                   // [na]
    int32_t *rank;                // [nx]: round of the LAST occurrence in A, -1 otherwise
    unsigned long long *slots;    // [4][slot_stride]: (tag << 32) | key; arrival slots of even / odd rounds, then the collectors' partial maxima
    uint32_t *abort_epoch;        // epoch of the last launch that gave up
    uint16_t *rescue_min;         // [nx]: running minima of the rescue form
    long long timeout_ticks;      // wall_clock64 ticks (100 MHz)
    long long *dbg;               // -DLEV_AP_PROFILE: [wave][round][8] s_memtime stamps
    uint32_t epoch;
    int n_short, nx, na, first, ntasks, slot_stride;
    int alphabet, pm_bytes, fb_stride, buf_stride, pad, wave_bytes;
};

template <bool RESCUE> __global__ __launch_bounds__(RESCUE ? 512 : ANN_WAVE) void k_lev_ap(LevArgsAP a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    __shared__ uint32_t s_best;
    __shared__ int32_t s_A[64];
    {
        const uint32_t ab = __hip_atomic_load(a.abort_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (RESCUE ? ab != a.epoch : ab == a.epoch) return;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    unsigned char *smem = smem_all + (size_t)wv * a.wave_bytes;
    const int A = a.alphabet, nx = a.nx;
    unsigned char *pm_col = smem + (size_t)(lane >> 5) * A * LEVF_ROW + (size_t)(lane & 31) * 4;
    uint8_t *bufs = smem + a.pm_bytes;                    // [0], [1]: the wave's strings, [2]: the anchor
    int16_t *FB = reinterpret_cast<int16_t *>(bufs + 3 * a.buf_stride);   // [pair][half][fb_stride]
    const int ws = (a.n_short + 1) >> 1;                  // tasks of the packed class
    // ---- the task's shape (persistent form: bound once, before round 0)
    int tq0 = -1, tq1 = -1, w = 0, pair = 0, half = 0, LP = 64, m = 0, Wp = 0, mytq = -1;
    bool have = false;
    uint32_t hp_or = 0, hn_and = 0, rows = 0;
    uint32_t rm0 = 0xffffu, rm1 = 0xffffu;               // running minima of the wave's points (persistent form)
    int rk0 = -1, rk1 = -1;
    int si = a.first;
    for (int r = 0; r < a.na; ++r) {
        if (RESCUE) {
            if (threadIdx.x == 0) { s_best = 0; s_A[r] = si; }
            __syncthreads();
        }
#ifdef LEV_AP_PROFILE
#define AP_STAMP(i) do { if (!RESCUE && lane == 0) a.dbg[((size_t)blockIdx.x * 64 + r) * 8 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define AP_STAMP(i) do { } while (0)
#endif
        AP_STAMP(0);
        if ((RESCUE ? threadIdx.x : blockIdx.x + lane) == 0) a.A[r] = si;
        const int la = a.slen[si];
        const uint8_t *asrc = a.sym + a.soff[si];
        uint32_t mykey = 0;
        for (int vb = RESCUE ? wv : (int)blockIdx.x; vb < a.ntasks; vb += RESCUE ? nwv : a.ntasks) {
            if (RESCUE || r == 0) {
                const bool packed = vb < ws;
                const int GL = packed ? 16 : 32;
                const int slot = lane / GL;
                w = lane - slot * GL; pair = slot >> 1; half = slot & 1; LP = packed ? 32 : 64;
                hp_or = w == 0 ? 0x80000000u : 0u;
                hn_and = w == 0 ? 0u : 0xffffffffu;
                asm volatile("" : "+v"(hp_or), "+v"(hn_and));
                tq0 = tq1 = -1;
                if (packed) {
                    tq0 = a.order[2 * vb];
                    if (2 * vb + 1 < a.n_short) tq1 = a.order[2 * vb + 1];
                } else {
                    tq0 = a.order[a.n_short + (vb - ws)];
                }
                for (int e = lane * 16; e < 3 * a.buf_stride; e += 64 * 16) *reinterpret_cast<uint4 *>(bufs + e) = make_uint4(0, 0, 0, 0);
                int ln0 = 0, ln1 = 0;
                {
                    ln0 = a.slen[tq0];
                    const uint8_t *src = a.sym + a.soff[tq0];
                    for (int ch = lane; ch < ((ln0 + 15) >> 4); ch += 64)
                        reinterpret_cast<uint4 *>(bufs + a.pad)[ch] = reinterpret_cast<const uint4 *>(src)[ch];
                }
                if (tq1 >= 0) {
                    ln1 = a.slen[tq1];
                    const uint8_t *src = a.sym + a.soff[tq1];
                    for (int ch = lane; ch < ((ln1 + 15) >> 4); ch += 64)
                        reinterpret_cast<uint4 *>(bufs + a.buf_stride + a.pad)[ch] = reinterpret_cast<const uint4 *>(src)[ch];
                }
                wave_lds_fence();
                have = pair == 0 || (packed && pair == 1 && tq1 >= 0);
                mytq = !have ? -1 : (pair == 0 ? tq0 : tq1);
                m = !have ? 0 : (pair == 0 ? ln0 : ln1);          // the wave's own string is the pattern
                Wp = (m + 31) >> 5;
                const uint8_t *pat = bufs + pair * a.buf_stride + a.pad;
                for (int c = 0; c < A; ++c) *reinterpret_cast<uint32_t *>(pm_col + (size_t)c * LEVF_ROW) = 0u;
                if (have && w < Wp) {
                    const int valid = min(32, m - w * 32);
                    uint32_t sy[32];
                    if (half == 0) {
                        const uint4 *p16 = reinterpret_cast<const uint4 *>(pat + w * 32);
                        const uint4 q0 = p16[0], q1 = p16[1];
                        const uint32_t wd[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
                        for (int k = 0; k < 32; ++k) sy[k] = (wd[k >> 2] >> ((k & 3) * 8)) & 0xffu;
                    } else {
#pragma unroll
                        for (int k = 0; k < 32; ++k) sy[k] = pat[max(m - 1 - (w * 32 + k), 0)];
                    }
#pragma unroll
                    for (int k = 0; k < 32; ++k)
                        if (k < valid) atomicOr(reinterpret_cast<uint32_t *>(pm_col + (size_t)sy[k] * LEVF_ROW), 1u << k);
                }
                rows = !have ? 0u : (w < Wp - 1 ? 0xffffffffu : (w == Wp - 1 ? (0xffffffffu >> (31 - ((m - 1) & 31))) : 0u));
            }
            // ---- the round's text: the anchor (stale bytes behind a shorter anchor are symbols of the previous one: they index
            // the match-mask table and are never used)
            for (int ch = lane; ch < ((la + 15) >> 4); ch += 64)
                reinterpret_cast<uint4 *>(bufs + 2 * a.buf_stride + a.pad)[ch] = reinterpret_cast<const uint4 *>(asrc)[ch];
            wave_lds_fence();
            AP_STAMP(1);
            const uint8_t *txt = bufs + 2 * a.buf_stride + a.pad;
            const int n = have ? la : 0;
            const int h = (n + 1) >> 1;
            const uint32_t un = (have && m > 0) ? (uint32_t)(half ? n - h : h) : 0u;
            int max_steps = (have && m > 0) ? h + Wp - 1 : 0;
            int k_lo = (have && m > 0) ? Wp - 1 : 0, k_hi = (have && m > 0) ? n - h : 0x7fffffff;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                max_steps = max(max_steps, __shfl_xor(max_steps, off));
                k_lo = max(k_lo, __shfl_xor(k_lo, off));
                k_hi = min(k_hi, __shfl_xor(k_hi, off));
            }
            max_steps = __builtin_amdgcn_readfirstlane(max_steps);
            k_lo = min(__builtin_amdgcn_readfirstlane(k_lo), max_steps);
            k_hi = max(k_lo, min(__builtin_amdgcn_readfirstlane(k_hi), max_steps));
            const int dir = half ? -1 : 1;
            const uint8_t *tp = txt + (half ? n - 1 + w : -w);
            uint32_t vp = 0xffffffffu, vn = 0u;
            // LDS reads run ahead of the recurrence: the match mask of column k + 2 and the symbol of column k + 5 are requested in
            // column k (a wave alone on its SIMD has only its own ~100 cycles per column to cover an LDS round trip: with one
            // column of lead it waited on most of them).  Three registers per queue, each column overwriting the one it has just
            // consumed: after three columns every value sits where it started, so the main loop (three columns per trip) has no
            // copies; the few checked columns at either end rotate by moves.
            //   E0 = mask(k), E1 = mask(k + 1), (E2 free);  S2 = symbol(k + 2), S0 = symbol(k + 3), S1 = symbol(k + 4)
            uint32_t E0 = *reinterpret_cast<const uint32_t *>(pm_col + (uint32_t)tp[0] * LEVF_ROW);
            uint32_t E1 = *reinterpret_cast<const uint32_t *>(pm_col + (uint32_t)tp[dir] * LEVF_ROW);
            uint32_t E2 = 0;
            uint32_t S2 = tp[2 * dir], S0 = tp[3 * dir], S1 = tp[4 * dir];
            const uint8_t *tnext = tp + 5 * dir;
            uint32_t out_hp = 0, out_hn = 0;
            auto column = [&](uint32_t &Ecur, uint32_t &Enew, uint32_t &Saddr, int k, auto checked) {
                const unsigned char *ea = pm_col + Saddr * LEVF_ROW;
                Saddr = *tnext;
                tnext += dir;
                Enew = *reinterpret_cast<const uint32_t *>(ea);
                const uint32_t eq = Ecur;
                const uint32_t hp_up = dpp_shr1_or(out_hp, hp_or), hn_up = dpp_shr1_and(out_hn, hn_and);
                __builtin_amdgcn_sched_barrier(0);
                const bool valid = !decltype(checked)::value || (uint32_t)(k - w) < un;
                const uint32_t c = hn_up >> 31;
                const uint32_t x = eq | c;
                const uint32_t tt = __builtin_amdgcn_bitop3_b32(c, eq, vp, 0xa8);
                const uint32_t sm = tt + vp;
                const uint32_t d0p = __builtin_amdgcn_bitop3_b32(sm, vp, x, 0xbe);
                const uint32_t hp = __builtin_amdgcn_bitop3_b32(vn, d0p, vp, 0xf1);
                const uint32_t d0 = d0p | vn;
                const uint32_t hn = d0 & vp;
                const uint32_t hps = __builtin_amdgcn_alignbit(hp, hp_up, 31);
                const uint32_t hns = __builtin_amdgcn_alignbit(hn, hn_up, 31);
                const uint32_t nvp = __builtin_amdgcn_bitop3_b32(hns, d0, hps, 0xf1);
                const uint32_t nvn = hps & d0;
                vp = valid ? nvp : vp;
                vn = valid ? nvn : vn;
                out_hp = hp;
                out_hn = hn;
                __builtin_amdgcn_sched_barrier(0);
            };
            auto column1 = [&](int k, auto checked) {   // one column, the queues rotated back into place
                column(E0, E2, S2, k, checked);
                const uint32_t e = E0; E0 = E1; E1 = E2; E2 = e;
                const uint32_t t = S2; S2 = S0; S0 = S1; S1 = t;
            };
            AP_STAMP(2);
            int k = 0;
            for (; k < k_lo; ++k) column1(k, std::true_type());
            for (; k + 3 <= k_hi; k += 3) {
                column(E0, E2, S2, k, std::false_type());
                column(E1, E0, S0, k + 1, std::false_type());
                column(E2, E1, S1, k + 2, std::false_type());
            }
            for (; k < k_hi; ++k) column1(k, std::false_type());
            for (; k < max_steps; ++k) column1(k, std::true_type());
            AP_STAMP(3);

            // ---- F / B' (prefix sums of the vertical deltas down this half's rows), then min_i F[i] + B'[m - i] per pair
            const int part = __popc(vp & rows) - __popc(vn & rows);
            int incl = part;
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) {
                const int o = __shfl_up(incl, off, 32);
                if (w >= off) incl += o;
            }
            int val = (int)un + incl - part;
            if (m == 0) val = half ? n - h : h;
            int16_t *fbp = FB + (size_t)pair * 2 * a.fb_stride;
            int16_t *fb = fbp + (size_t)half * a.fb_stride;
            if (have && w == 0) fb[0] = (int16_t)val;
#pragma unroll
            for (int b = 0; b < 32; ++b) {
                val += (int)((vp >> b) & 1u) - (int)((vn >> b) & 1u);
                if ((rows >> b) & 1u) fb[w * 32 + b + 1] = (int16_t)val;
            }
            wave_lds_fence();
            int best = 0x7fffffff;
            const int lp = lane & (LP - 1);
            if (have)
                for (int i = lp; i <= m; i += LP) best = min(best, (int)fbp[i] + (int)fbp[a.fb_stride + m - i]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
                if (off < LP) best = min(best, __shfl_xor(best, off));
            if (have && lp == 0) a.Dt[(size_t)r * nx + mytq] = (double)best;
            wave_lds_fence();   // (FB and the text are rewritten by the next task / round)
            AP_STAMP(4);
            // ---- running minima (pickers.py:47-50: over row 0 alone for the second anchor, over rows 1..r afterwards)
            if (RESCUE) {
                if (have && lp == 0) {
                    uint32_t v = (uint32_t)best;
                    if (r > 1) v = min(v, (uint32_t)a.rescue_min[mytq]);
                    a.rescue_min[mytq] = (uint16_t)v;
                    atomicMax(&s_best, (v << 16) | (uint32_t)(0xffff - mytq));
                }
            } else {
                const uint32_t d0 = (uint32_t)__shfl(best, 0), d1 = (uint32_t)__shfl(best, 32);
                rm0 = r <= 1 ? d0 : min(rm0, d0);
                rm1 = r <= 1 ? d1 : min(rm1, d1);
                mykey = (rm0 << 16) | (uint32_t)(0xffff - tq0);
                if (tq1 >= 0) mykey = max(mykey, (rm1 << 16) | (uint32_t)(0xffff - tq1));
                if (si == tq0) rk0 = r;
                if (si == tq1) rk1 = r;
            }
        }
        if (r + 1 == a.na) break;
        if (RESCUE) {
            __syncthreads();
            si = 0xffff - (int)(s_best & 0xffffu);
            __syncthreads();
        } else {
            // ---- arrive + arg-max in two hops: wave g < ceil(ntasks / 64) collects the slots of waves 64 g .. 64 g + 63 (one load
            // per lane) and publishes their maximum; everybody polls those <= 16 partial slots.  (Everybody polling every slot
            // -- 965 waves x 7.7 KB per poll on 120 cache lines -- cost 5.6 us per round; this is two short hops.)
            const uint32_t tag = a.epoch * 64u + (uint32_t)r;
            unsigned long long *sl = a.slots + (size_t)(r & 1) * a.slot_stride;
            unsigned long long *pl = sl + 2 * a.slot_stride;          // partial maxima [2][slot_stride] behind the arrival slots
            const int ngroups = (a.ntasks + 63) >> 6;
            if (lane == 0)
                __hip_atomic_store(sl + blockIdx.x, ((unsigned long long)tag << 32) | mykey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long t0 = wall_clock64();
            AP_STAMP(5);
            uint32_t mx;
            bool collected = (int)blockIdx.x >= ngroups;             // nothing to collect / done
            int polls = 0;
            for (;;) {
                if (!collected) {
                    const unsigned long long v = __hip_atomic_load(sl + min((int)blockIdx.x * 64 + lane, a.ntasks - 1), __ATOMIC_RELAXED,
                                                                   __HIP_MEMORY_SCOPE_AGENT);
                    if (__all((uint32_t)(v >> 32) == tag)) {
                        uint32_t gm = (uint32_t)v;
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) gm = max(gm, (uint32_t)__shfl_xor((int)gm, off));
                        if (lane == 0)
                            __hip_atomic_store(pl + blockIdx.x, ((unsigned long long)tag << 32) | gm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        collected = true;
                    }
                }
                if (collected) {
                    const unsigned long long v = __hip_atomic_load(pl + min(lane, ngroups - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    mx = (uint32_t)v;
                    if (__all((uint32_t)(v >> 32) == tag)) break;
                }
                // (the abort word and the clock every 8th poll: a second dependent load would double the poll period)
                if ((++polls & 7) == 0 || a.timeout_ticks == 0) {
                    if (__hip_atomic_load(a.abort_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.epoch) return;
                    if (wall_clock64() - t0 > a.timeout_ticks) {
                        if (lane == 0) __hip_atomic_store(a.abort_epoch, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        return;
                    }
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, off));
            si = 0xffff - (int)(mx & 0xffffu);
            AP_STAMP(6);
        }
    }
    // ---- anchorRank
    if (RESCUE) {
        __syncthreads();
        for (int j = threadIdx.x; j < nx; j += blockDim.x) a.rank[j] = -1;
        __syncthreads();
        if (threadIdx.x == 0)
            for (int r = 0; r < a.na; ++r) a.rank[s_A[r]] = r;   // a later occurrence overrides an earlier one (annchor.py:288-289)
    } else {
        // (the last round's anchor is in si; the loop above recorded rounds 0 .. na - 1 as it ran them)
        if (lane == 0) {
            a.rank[tq0] = rk0;
            if (tq1 >= 0) a.rank[tq1] = rk1;
        }
    }
}

static int launch_p2(annchor_ctx *c, const PairSource &src, double *d_out, double *d_RA, uint8_t *d_ncm)
{
    LevArgsP2 a;
    a.sym = c->sym.as<uint8_t>(); a.soff = c->soff.as<int32_t>(); a.slen = c->slen.as<int32_t>();
    a.ij = src.ij; a.idx = src.idx; a.n = src.n; a.out = d_out; a.RA = d_RA; a.ncm = d_ncm;
    a.alphabet = c->alphabet;
    a.pm_bytes = 2 * a.alphabet * LEVF_ROW;
    a.fb_stride = (c->maxlen + 2 + 7) & ~7;
    a.pad = ((c->maxlen / 2 + 48) + 15) & ~15;
    a.buf_stride = 2 * a.pad + ((c->maxlen + 15) & ~15) + 16;
    const size_t lds = (size_t)a.pm_bytes + 4 * (size_t)a.buf_stride + 4 * sizeof(int16_t) * a.fb_stride;
    ANN_REQUIRE(c, lds <= 160 * 1024, ANNCHOR_ELIMIT, "alphabet %d x length %d needs %zu B of LDS (> 160 KiB)", c->alphabet,
                c->maxlen, lds);
    ANN_TRY(ann_reserve(c, c->lev_perm, sizeof(int32_t) * (size_t)(src.n + 2)));
    int32_t *perm = c->lev_perm.as<int32_t>();
    int32_t *cursors = nullptr, *next_cursors = nullptr;
    ANN_TRY(lev_next_cursors(c, &cursors, &next_cursors));
    k_lev_classify<<<ann_blocks(src.n, 256), 256, 0, c->stream>>>(src.ij, src.idx, a.slen, src.n, 16, perm, cursors, next_cursors);
    a.perm = perm; a.cursors = cursors;
    if (lds > 64 * 1024)
        ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)k_lev_p2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // (the class sizes stay on the device; a pair is in the long class only when BOTH its strings have more than 16 words, and
    // the strings bound at most L (L + 1) / 2 such pairs when the list has no repeats -- a list that has more is still
    // complete: the waves stride over the tasks)
    const int64_t L = c->nx - c->lev_nshort;
    const int64_t long_bound = std::min<int64_t>(src.n, L * (L + 1) / 2);
    const int64_t grid = std::min<int64_t>(src.n, (src.n + 1) / 2 + long_bound + 8);
    k_lev_p2<<<(int)grid, ANN_WAVE, lds, c->stream>>>(a);
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}

static int launch_a2(annchor_ctx *c, const PairSource &src, double *d_out)
{
    LevArgsA2 aa;
    LevArgsA &a = aa.a;
    a.sym = c->sym.as<uint8_t>(); a.soff = c->soff.as<int32_t>(); a.slen = c->slen.as<int32_t>();
    a.anchor = src.anchor; a.n = src.n; a.out = d_out; a.alphabet = c->alphabet;
    a.pm_bytes = 2 * a.alphabet * LEVF_ROW;
    a.text_stride = 2 * LEVR_PAD + ((c->maxlen + 15) & ~15) + 16;
    a.fb_stride = (c->maxlen + 2 + 7) & ~7;
    a.pick_row = nullptr; a.pick_runmin = nullptr; a.pick_out = nullptr; a.pick_reset = 0; a.pick_nx = 0;
    if (src.pick_fused && c->nx <= 8192) {
        *src.pick_fused = true;
        if (src.pick_row) {
            a.pick_row = src.pick_row; a.pick_runmin = src.pick_runmin; a.pick_out = src.pick_out;
            a.pick_reset = src.pick_reset; a.pick_nx = (int)c->nx;
        }
    }
    aa.order = c->lev_order.as<int32_t>();
    aa.n_short = c->lev_nshort;
    aa.pad = ((c->maxlen / 2 + 48) + 15) & ~15;
    aa.buf_stride = 2 * aa.pad + ((c->maxlen + 15) & ~15) + 16;
    size_t lds = (size_t)a.pm_bytes + 3 * (size_t)aa.buf_stride + 4 * sizeof(int16_t) * a.fb_stride;
    ANN_REQUIRE(c, lds <= 160 * 1024, ANNCHOR_ELIMIT, "alphabet %d x length %d needs %zu B of LDS (> 160 KiB)", c->alphabet,
                c->maxlen, lds);
    const int64_t blocks = (c->lev_nshort + 1) / 2 + (c->nx - c->lev_nshort);
    {
        // A latency-bound launch wants its waves on DIFFERENT SIMDs: a workgroup is one wave, so asking for a quarter
        // of a CU's LDS caps a CU at four of them -- one per SIMD -- instead of wherever the dispatcher packs them
        // (ANNCHOR_LEV_A2_LDS overrides the request; 0 = only what the tables need).
        const char *e = getenv("ANNCHOR_LEV_A2_LDS");
        const size_t want = e ? (size_t)atoll(e) : (size_t)40 * 1024;
        if (blocks <= (int64_t)c->prop.multiProcessorCount * 4 && want > lds && want <= 160 * 1024) lds = want;
    }
    if (lds > 64 * 1024)
        ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)k_lev_a2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    k_lev_a2<<<(int)blocks, ANN_WAVE, lds, c->stream>>>(aa);
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}

// A persistent launch that gives up (its waves were not all resident: another process holds CUs) costs its whole time limit before
// the rescue form runs -- 20 ms against a 0.3 ms launch.  The abort word is copied to the context's pinned tail behind every
// persistent launch and looked at after the next host wait; two launches of a process that gave up (time limit not forced by the
// environment) switch the persistent form off for the process, with one line on stderr.  annchor_lev_persist_state(1) re-arms.
static std::atomic<int> g_lev_ap_strikes{0};
static std::atomic<int> g_lev_ap_off{0};
void ann_lev_ap_probe(annchor_ctx *c)
{
    const uint32_t epoch = c->lev_ap_probe_epoch;
    c->lev_ap_probe_epoch = 0;
    if (!c->pin || !epoch) return;
    const unsigned char *tail = c->pin + (size_t)annchor_ctx::PIN_SLOTS * annchor_ctx::PIN_SLOT_BYTES + annchor_ctx::PIN_DL_BYTES;
    uint32_t seen;
    memcpy(&seen, tail, 4);
    if (seen != epoch) return;
    if (getenv("ANNCHOR_LEV_PERSIST_TIMEOUT_US") && !getenv("ANNCHOR_LEV_PERSIST_LEARN")) return;   // a forced time limit (tests) teaches nothing
    if (g_lev_ap_strikes.fetch_add(1) + 1 >= 2 && !g_lev_ap_off.exchange(1))
        fprintf(stderr, "annchor_hip: the persistent anchor launch gave up twice (its waves were not all resident: is the GPU shared?); "
                        "the picker's rounds run as separate launches from now on (annchor_lev_persist_state(1) re-arms)\n");
}
// set: 0 = off, 1 = on (strikes forgotten), anything else = query; returns 1 when the persistent form is armed
extern "C" int annchor_lev_persist_state(int set)
{
    if (set == 0) g_lev_ap_off.store(1);
    if (set == 1) { g_lev_ap_off.store(0); g_lev_ap_strikes.store(0); }
    return g_lev_ap_off.load() ? 0 : 1;
}

// All anchor rounds in one launch (k_lev_ap) when the data set allows it: byte alphabet, every string within a 32-lane slot,
// point ids and distances within 16 bits, and every wave of the launch resident at once.
int ann_lev_anchor_rounds(annchor_ctx *c, int32_t na, int32_t first, bool *done)
{
    *done = false;
    {
        const char *e = getenv("ANNCHOR_LEV_PERSIST");   // 0: the rounds one by one (A/B runs, tests); read per call
        if (e && atoi(e) == 0) return ANNCHOR_OK;
        if (g_lev_ap_off.load()) return ANNCHOR_OK;      // (it gave up twice in this process)
        const char *e_a = getenv("ANNCHOR_LEV_ANCHOR"), *e_r = getenv("ANNCHOR_LEV_R");
        if ((e_a && atoi(e_a) != 2) || e_r) return ANNCHOR_OK;   // a forced kernel variant means the per-round path
    }
    if (c->metric != ANNCHOR_METRIC_LEVENSHTEIN || c->sym_wide || !c->lev_order.p) return ANNCHOR_OK;
    if ((c->maxlen + 31) / 32 > 32 || c->nx > 8192 || na < 1 || na > 64) return ANNCHOR_OK;
    LevArgsAP a;
    a.ntasks = (c->lev_nshort + 1) / 2 + (int)(c->nx - c->lev_nshort);
    if (a.ntasks > 1024) return ANNCHOR_OK;
    a.alphabet = c->alphabet;
    a.pm_bytes = 2 * a.alphabet * LEVF_ROW;
    a.fb_stride = (c->maxlen + 2 + 7) & ~7;
    a.pad = ((c->maxlen / 2 + 48) + 15) & ~15;
    a.buf_stride = 2 * a.pad + ((c->maxlen + 15) & ~15) + 16;
    a.wave_bytes = (int)(((size_t)a.pm_bytes + 3 * (size_t)a.buf_stride + 4 * sizeof(int16_t) * a.fb_stride + 15) & ~(size_t)15);
    size_t lds = (size_t)a.wave_bytes;
    if (lds > 160 * 1024) return ANNCHOR_OK;
    // one wave per SIMD, as for k_lev_a2: a quarter of a CU's LDS per single-wave workgroup
    if (a.ntasks <= c->prop.multiProcessorCount * 4 && lds < (size_t)40 * 1024) lds = (size_t)40 * 1024;
    if (lds > 64 * 1024)
        ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)k_lev_ap<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0;
    ANN_CHECK_HIP(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_lev_ap<false>, ANN_WAVE, lds));
    if ((int64_t)per_cu * c->prop.multiProcessorCount < a.ntasks) return ANNCHOR_OK;
    // the rescue form: one workgroup of up to 8 waves
    int rwaves = (int)std::min<size_t>(8, (160 * 1024 - 1024) / (size_t)a.wave_bytes);
    if (rwaves < 1) return ANNCHOR_OK;
    const size_t rlds = (size_t)rwaves * a.wave_bytes;
    if (rlds > 64 * 1024)
        ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)k_lev_ap<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds));
    a.slot_stride = 1024;
    const size_t slot_bytes = sizeof(unsigned long long) * 4 * (size_t)a.slot_stride;
    if (!c->lev_ap.p) {
        ANN_TRY(ann_reserve(c, c->lev_ap, slot_bytes + 64));
        ANN_CHECK_HIP(c, hipMemsetAsync(c->lev_ap.p, 0, slot_bytes + 64, c->stream));
        c->lev_ap_epoch = 0;
    }
    ANN_TRY(ann_reserve(c, c->lev_ap_min, sizeof(uint16_t) * (size_t)c->nx));
    ANN_TRY(ann_reserve(c, c->anchorRank, sizeof(int32_t) * (size_t)c->nx));
    a.sym = c->sym.as<uint8_t>(); a.soff = c->soff.as<int32_t>(); a.slen = c->slen.as<int32_t>();
    a.order = c->lev_order.as<int32_t>();
    a.Dt = c->Dt.as<double>(); a.A = c->A.as<int32_t>(); a.rank = c->anchorRank.as<int32_t>();
    a.slots = c->lev_ap.as<unsigned long long>();
    a.abort_epoch = reinterpret_cast<uint32_t *>(c->lev_ap.as<unsigned char>() + slot_bytes);
    a.rescue_min = c->lev_ap_min.as<uint16_t>();
    {
        const char *e = getenv("ANNCHOR_LEV_PERSIST_TIMEOUT_US");   // (tests force the rescue form with 0)
        a.timeout_ticks = (e ? atoll(e) : 20000ll) * 100;
    }
    a.dbg = nullptr;
#ifdef LEV_AP_PROFILE
    {
        static long long *dbg = nullptr;
        if (!dbg) (void)hipMalloc(&dbg, sizeof(long long) * 1024 * 64 * 8);
        a.dbg = dbg;
    }
#endif
    a.epoch = ++c->lev_ap_epoch;
    if ((a.epoch & 0x3ffffffu) == 0) a.epoch = c->lev_ap_epoch = 1;   // (tag = epoch * 64 + round stays within 32 bits)
    a.n_short = c->lev_nshort; a.nx = (int)c->nx; a.na = na; a.first = first;
    {
        const double word_bytes = (double)na * (double)c->nx * (2.0 * c->maxlen + 8);
        ProfScope ps(c, "levenshtein_pairs", word_bytes);
        k_lev_ap<false><<<a.ntasks, ANN_WAVE, lds, c->stream>>>(a);
        k_lev_ap<true><<<1, rwaves * ANN_WAVE, rlds, c->stream>>>(a);
    }
    if (c->pin) {   // did it give up?  (looked at after the next host wait: ann_lev_ap_probe)
        unsigned char *tail = c->pin + (size_t)annchor_ctx::PIN_SLOTS * annchor_ctx::PIN_SLOT_BYTES + annchor_ctx::PIN_DL_BYTES;
        ANN_CHECK_HIP(c, hipMemcpyAsync(tail, a.abort_epoch, 4, hipMemcpyDeviceToHost, c->stream));
        c->lev_ap_probe_epoch = a.epoch;
    }
    ANN_CHECK_HIP(c, hipGetLastError());
#ifdef LEV_AP_PROFILE
    {
        static int calls = 0;
        if (++calls == 5) {
            std::vector<long long> h((size_t)1024 * 64 * 8);
            (void)hipStreamSynchronize(c->stream);
            (void)hipMemcpy(h.data(), a.dbg, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
            const char *nm[6] = {"stage anchor", "loop setup", "columns", "prefix+combine", "key", "barrier"};
            for (int b : {0, 1, 500, 700, a.ntasks - 1}) {
                fprintf(stderr, "k_lev_ap wave %d:", b);
                for (int ph = 0; ph < 6; ++ph) {
                    double sum = 0;
                    for (int r = 0; r + 1 < na; ++r) sum += (double)(h[((size_t)b * 64 + r) * 8 + ph + 1] - h[((size_t)b * 64 + r) * 8 + ph]);
                    fprintf(stderr, " %s %.0f", nm[ph], sum / (na - 1));
                }
                fprintf(stderr, " | round %.0f (s_memtime ticks per round)\n", (double)(h[((size_t)b * 64 + na - 2) * 8 + 6] - h[(size_t)b * 64 * 8]) / (na - 1));
            }
        }
    }
#endif
    *done = true;
    return ANNCHOR_OK;
}

// ---------------------------------------------------------------------------------------
// k_lev_w: alphabets beyond 256 distinct symbols (annchor_set_strings_u16: 16-bit dense codes, up to 65 536 -- any
// Unicode corpus).  A per-pattern match-mask table would be alphabet x 128 B of LDS per half-wave, so the match word of
// a column is computed instead: a lane keeps the 32 symbols of its pattern word in 16 registers (two 16-bit codes each)
// and compares them with the column's text symbol -- ~80 VALU instructions per word-column next to the 15 of the
// recurrence: about six times slower than k_lev_f, exact for any alphabet.  One slot class (the longer string is the
// pattern), text staged in LDS as 16-bit codes, no fused arg-max (the picker's separate launch takes over).
struct LevArgsW {
    const uint16_t *sym;
    const int32_t *soff;   // offsets in SYMBOLS, multiples of 8 (16-byte aligned starts)
    const int32_t *slen;
    const int2 *ij;
    const int32_t *idx;
    const int32_t *anchor;
    int64_t n;
    double *out;
    double *RA;
    uint8_t *ncm;
    int GL, P, text_stride;   // lanes per slot, slots per wave, uint16 entries of text per slot
};

__global__ __launch_bounds__(ANN_WAVE) void k_lev_w(LevArgsW a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x, GL = a.GL, P = a.P;
    const int g = lane / GL, w = lane - g * GL;
    const bool slot_ok = g < P;
    uint16_t *txt_g = reinterpret_cast<uint16_t *>(smem) + (size_t)(slot_ok ? g : 0) * a.text_stride;
    int *ssum = reinterpret_cast<int *>(smem + (size_t)P * a.text_stride * 2);
    const int64_t n_tasks = (a.n + P - 1) / P;
    uint32_t hp_or = w == 0 ? 0x80000000u : 0u;
    uint32_t hn_and = w == 0 ? 0u : 0xffffffffu;
    asm volatile("" : "+v"(hp_or), "+v"(hn_and));
    for (int64_t task = blockIdx.x; task < n_tasks; task += gridDim.x) {
        const int64_t t_pair = task * P + g;
        const bool active = slot_ok && t_pair < a.n;
        int si = 0, sj = 0;
        int64_t opos = t_pair;
        if (active) {
            if (a.anchor) { si = *a.anchor; sj = (int)t_pair; }
            else {
                const int64_t q = a.idx ? a.idx[t_pair] : t_pair;
                const int2 p = a.ij[q];
                si = p.x; sj = p.y;
                if (a.idx) opos = q;
            }
        }
        const int li = active ? a.slen[si] : 0, lj = active ? a.slen[sj] : 0;
        const bool swap = li < lj;                 // pattern = the longer string
        const int ps = swap ? sj : si, ts = swap ? si : sj;
        const int m = swap ? lj : li, n = swap ? li : lj;
        const uint16_t *pat = a.sym + (active ? a.soff[ps] : 0);
        const uint16_t *tex = a.sym + (active ? a.soff[ts] : 0);
        const int Wp = (m + 31) >> 5;
        if (slot_ok && w == 0) ssum[g] = 0;
        // this lane's 32 pattern symbols, two per register; slots past the pattern's end get a code no text symbol has
        uint32_t pr[16];
        {
            const uint4 *p16 = reinterpret_cast<const uint4 *>(pat + w * 32);
            const bool has = active && w < Wp;
            const uint4 z = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
            const uint4 q0 = has ? p16[0] : z, q1 = has ? p16[1] : z, q2 = has ? p16[2] : z, q3 = has ? p16[3] : z;
            const uint32_t v[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
            const int valid = has ? min(32, m - w * 32) : 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                uint32_t x = v[r];
                if (2 * r >= valid) x |= 0x0000ffffu;        // (0xffff is never a dense code: alphabets hold <= 65 535 symbols)
                if (2 * r + 1 >= valid) x |= 0xffff0000u;
                pr[r] = x;
            }
        }
        if (slot_ok)
            for (int e = w * 8; e < a.text_stride; e += GL * 8) *reinterpret_cast<uint4 *>(txt_g + e) = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
        wave_lds_fence();
        if (active) {
            const int chunks = (n + 7) >> 3;   // 8 symbols per 16 bytes
            for (int ch = w; ch < chunks; ch += GL)
                reinterpret_cast<uint4 *>(txt_g + LEVR_PAD)[ch] = reinterpret_cast<const uint4 *>(tex)[ch];
        }
        wave_lds_fence();
        uint32_t vp = 0xffffffffu, vn = 0u;
        const uint32_t un = (active && m > 0) ? (uint32_t)n : 0u;
        int max_steps = (active && m > 0) ? n + Wp - 1 : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) max_steps = max(max_steps, __shfl_xor(max_steps, off));
        max_steps = __builtin_amdgcn_readfirstlane(max_steps);
        const uint16_t *tp = txt_g + LEVR_PAD - w;
        uint32_t out_hp = 0, out_hn = 0;
        for (int k = 0; k < max_steps; ++k) {
            const uint32_t c = tp[k];
            const uint32_t cc = c | (c << 16);
            uint32_t eq = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t x = pr[r] ^ cc;
                eq |= ((x & 0xffffu) == 0u ? 1u : 0u) << (2 * r);
                eq |= ((x >> 16) == 0u ? 1u : 0u) << (2 * r + 1);
            }
            const uint32_t hp_up = dpp_shr1_or(out_hp, hp_or), hn_up = dpp_shr1_and(out_hn, hn_and);
            const bool valid = (uint32_t)(k - w) < un;
            const uint32_t cin = hn_up >> 31;
            const uint32_t x = eq | cin;
            const uint32_t d0 = (((x & vp) + vp) ^ vp) | x | vn;
            const uint32_t hp = vn | ~(d0 | vp);
            const uint32_t hn = d0 & vp;
            const uint32_t hps = __builtin_amdgcn_alignbit(hp, hp_up, 31);
            const uint32_t hns = __builtin_amdgcn_alignbit(hn, hn_up, 31);
            const uint32_t nvp = hns | ~(d0 | hps), nvn = hps & d0;
            vp = valid ? nvp : vp;
            vn = valid ? nvn : vn;
            out_hp = hp;
            out_hn = hn;
        }
        if (active) {
            const uint32_t rows = w < Wp - 1 ? 0xffffffffu : (w == Wp - 1 ? (0xffffffffu >> (31 - ((m - 1) & 31))) : 0u);
            const int part = __popc(vp & rows) - __popc(vn & rows);
            if (part) atomicAdd(&ssum[g], part);
        }
        wave_lds_fence();
        if (active && w == 0) {
            const double d = (double)(n + ssum[g]);
            if (a.out) a.out[t_pair] = d;
            if (a.RA) { a.RA[opos] = d; a.ncm[opos] = 0; }
        }
        wave_lds_fence();
    }
}

static int launch_w(annchor_ctx *c, const PairSource &src, double *d_out, double *d_RA, uint8_t *d_ncm)
{
    LevArgsW a;
    a.sym = c->sym.as<uint16_t>(); a.soff = c->soff.as<int32_t>(); a.slen = c->slen.as<int32_t>();
    a.ij = src.ij; a.idx = src.idx; a.anchor = src.anchor; a.n = src.n; a.out = d_out; a.RA = d_RA; a.ncm = d_ncm;
    const int W = std::max(1, (c->maxlen + 31) / 32);
    ANN_REQUIRE(c, W <= 64, ANNCHOR_ELIMIT, "strings longer than 2048 symbols are not supported by this build (max %d)", c->maxlen);
    a.GL = W; a.P = 64 / W;
    a.text_stride = 2 * LEVR_PAD + ((c->maxlen + 7) & ~7) + 8;
    const size_t lds = (size_t)a.P * a.text_stride * 2 + 256;
    ANN_REQUIRE(c, lds <= 160 * 1024, ANNCHOR_ELIMIT, "strings of %d symbols need %zu B of LDS", c->maxlen, lds);
    if (lds > 64 * 1024)
        ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)k_lev_w, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t blocks = (src.n + a.P - 1) / a.P;
    const int64_t max_blocks = (int64_t)c->prop.multiProcessorCount * 32;
    if (blocks > max_blocks) blocks = max_blocks;
    ProfScope ps(c, "levenshtein_pairs", (double)src.n * (4.0 * c->maxlen + 8));
    k_lev_w<<<(int)blocks, ANN_WAVE, lds, c->stream>>>(a);
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}

int ann_lev_launch(annchor_ctx *c, const PairSource &src, double *d_out, double *d_RA, uint8_t *d_ncm)
{
    if (src.n == 0) return ANNCHOR_OK;
    if (c->sym_wide) return launch_w(c, src, d_out, d_RA, d_ncm);   // > 256 distinct symbols: 16-bit codes
    LevArgs a;
    a.sym = c->sym.as<uint8_t>();
    a.soff = c->soff.as<int32_t>();
    a.slen = c->slen.as<int32_t>();
    a.ij = src.ij;
    a.idx = src.idx;
    a.anchor = src.anchor;
    a.n = src.n;
    a.out = d_out;
    a.RA = d_RA;
    a.ncm = d_ncm;
    int G = (c->maxlen + 31) / 32;
    if (G < 1) G = 1;
    ANN_REQUIRE(c, G <= 64, ANNCHOR_ELIMIT, "strings longer than 2048 symbols are not supported by this build (max %d)",
                c->maxlen);
    a.G = G;
    a.P = 64 / G;
    a.alphabet = c->alphabet;
    a.text_stride = ((c->maxlen + 15) & ~15) + 16;
    a.pm_bytes = (int)((((size_t)a.P * a.alphabet * G * 4) + 15) & ~(size_t)15);
    a.wave_bytes = a.pm_bytes + a.P * a.text_stride + 256;  // + per-slot sums (<= 64 ints)
    // kernel choice.  Live kernels: k_lev_ap (the picker's rounds as one persistent launch) with k_lev_a2 as its round-by-round
    // form, k_lev_f / k_lev_p2 (pair lists, strings up to 1024 symbols), k_lev_r<1> (longer strings), k_lev_w (wide alphabets).
    // Retired in round 5 (measured slower everywhere, kept until then for A/B runs): the two-columns-per-iteration kernel k_lev,
    // k_lev_r with 2 / 4 words per lane, the one-pair-per-wave anchor kernel k_lev_a.
    const char *env_r = getenv("ANNCHOR_LEV_R");   // tests: 1 forces k_lev_r<1> on short strings too; read per launch
    {
        const int W = (c->maxlen + 31) / 32;
        const int R = env_r && atoi(env_r) == 1 ? 1 : 9;   // default: k_lev_f (394 vs 415 us per 65 536 pairs, 34 vs 37.5 us per anchor round)
        // one-to-all launches (anchor rounds): the latency-shaped kernel, unless ANNCHOR_LEV_ANCHOR=0 / a forced variant
        const char *env_a = getenv("ANNCHOR_LEV_ANCHOR");   // read per launch, like ANNCHOR_LEV_R
        const bool anchor_split = !(env_a && atoi(env_a) == 0);
        if (src.anchor && d_out && !d_RA && R == 9 && W <= 32 && anchor_split && c->maxlen < 32000) {
            // (whole rounds of the picker; any other one-to-all launch takes the pair-list kernel below)
            if (src.n == c->nx && c->lev_order.p) {
                ProfScope ps(c, "levenshtein_pairs", (double)src.n * (2.0 * c->maxlen + 8));
                return launch_a2(c, src, d_out);
            }
        }
        if (R == 9 && W <= 32) {   // k_lev_f (strings up to 1024 symbols: a slot's lanes must map to distinct banks)
            ProfScope ps(c, "levenshtein_pairs", (double)src.n * (2.0 * c->maxlen + 8));
            // short pair lists are latency bound like the anchor rounds: the packed forward / backward kernel
            // (ANNCHOR_LEV_P2_MAX: longest list that takes it, 0 = never; read per launch)
            // k_lev_f runs its list in batches of one wave per SIMD (41 us each at C2's lengths, P pairs per wave); the packed
            // kernel costs ~30 us + 6 us per 1000 pairs (tools/p2_crossover.py): it wins on a list that fills less than ~0.37
            // of a batch (1000 pairs: 36 vs 41 us) and on one that has just spilled into its second batch (5000 pairs: 60 vs
            // 78 us), ties at 2500 / 8000 and loses from 12 000 on (99 vs 80 us).  ANNCHOR_LEV_P2_MAX forces it for every list
            // up to that length (0 = never); read per launch.
            const char *env_p = getenv("ANNCHOR_LEV_P2_MAX");
            const int pf = c->lev_gl0 > 0 ? std::max(1, 64 / c->lev_gl0) : a.P;
            const double cap = (double)c->prop.multiProcessorCount * 4.0 * pf;
            // (long strings only -- four pairs or fewer per k_lev_f wave: with short strings k_lev_f packs up to 16 pairs per
            // wave and its chains are short already)
            const bool p2 = env_p ? src.n <= atoll(env_p)
                                  : (pf <= 4 && ((double)src.n <= 0.37 * cap || ((double)src.n > cap && (double)src.n <= 1.83 * cap)));
            if (!src.anchor && src.ij && p2 && src.n >= 64 && c->maxlen < 32000) return launch_p2(c, src, d_out, d_RA, d_ncm);
            return launch_f(c, a, src.n, src);
        }
        // strings of more than 1024 symbols (or ANNCHOR_LEV_R=1): one word per lane, slots of up to 64 lanes
        ProfScope ps(c, "levenshtein_pairs", (double)src.n * (2.0 * c->maxlen + 8));
        return launch_r<1>(c, a, src.n, src);
    }
}
