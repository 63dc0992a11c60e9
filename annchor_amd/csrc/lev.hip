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
#include "common.h"

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

// LDS layout per wave: [P][alphabet][G] uint32 PM, then [P][text_stride] bytes text.
__global__ __launch_bounds__(LEV_THREADS) void k_lev(LevArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int G = a.G, P = a.P, A = a.alphabet;
    uint32_t *pm = reinterpret_cast<uint32_t *>(smem + (size_t)wave * a.wave_bytes);
    unsigned char *txt = smem + (size_t)wave * a.wave_bytes + a.pm_bytes;
    int *ssum = reinterpret_cast<int *>(txt + (size_t)a.P * a.text_stride);  // [P] per-slot popcount sums

    const int g = lane / G;          // pair slot of this lane
    const int w = lane - g * G;      // word index inside the slot
    const bool slot_ok = g < P;
    uint32_t *pm_g = pm + (size_t)(slot_ok ? g : 0) * A * G;
    unsigned char *txt_g = txt + (size_t)(slot_ok ? g : 0) * a.text_stride;

    const int64_t n_tasks = (a.n + P - 1) / P;
    const int64_t wave_global = (int64_t)blockIdx.x * LEV_WAVES + wave;
    const int64_t wave_stride = (int64_t)gridDim.x * LEV_WAVES;

    for (int64_t task = wave_global; task < n_tasks; task += wave_stride) {
        const int64_t t_pair = task * P + g;
        const bool active = slot_ok && t_pair < a.n;
        int si = 0, sj = 0;
        int64_t opos = t_pair;
        if (active) {
            if (a.anchor) { si = *a.anchor; sj = (int)t_pair; }
            else {
                int64_t q = a.idx ? a.idx[t_pair] : t_pair;
                int2 p = a.ij[q];
                si = p.x; sj = p.y;
                if (a.idx) opos = q;
            }
        }
        int li = active ? a.slen[si] : 0, lj = active ? a.slen[sj] : 0;
        // pattern = shorter string, text = longer
        const bool swap = li > lj;
        const int ps = swap ? sj : si, ts = swap ? si : sj;
        const int m = swap ? lj : li, n = swap ? li : lj;
        const uint8_t *pat = a.sym + (active ? a.soff[ps] : 0);
        const uint8_t *tex = a.sym + (active ? a.soff[ts] : 0);
        const int Wp = (m + 31) >> 5;

        if (slot_ok && w == 0) ssum[g] = 0;
        // ---- build PM column of this lane: zero, then OR in the 32 pattern symbols
        if (slot_ok)
            for (int c = 0; c < A; ++c) pm_g[c * G + w] = 0u;
        if (active && w < Wp) {
            const uint4 *p16 = reinterpret_cast<const uint4 *>(pat + w * 32);
            uint4 q0 = p16[0], q1 = p16[1];  // starts are 16B aligned and padded
            uint32_t wd[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            const int valid = min(32, m - w * 32);
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                uint32_t c = (wd[k >> 2] >> ((k & 3) * 8)) & 0xffu;
                if (k < valid) atomicOr(&pm_g[c * G + w], 1u << k);
            }
        }
        // ---- stage the text: lanes of the slot copy 16B chunks
        if (active) {
            const int chunks = (n + 15) >> 4;
            for (int ch = w; ch < chunks; ch += G)
                reinterpret_cast<uint4 *>(txt_g)[ch] = reinterpret_cast<const uint4 *>(tex)[ch];
        }
        wave_lds_fence();

        // ---- systolic sweep, two text columns per iteration.  At iteration k lane w handles
        // columns 2(k-w) and 2(k-w)+1; the two symbols and the four carry bits of the lane
        // above arrive in one register through one DPP move.  Validity is a function of
        // (k - w, n) alone, so no flag travels with the data and the loop body is branch-free.
        uint32_t vp = 0xffffffffu, vn = 0u;
        const uint32_t un = (active && m > 0) ? (uint32_t)n : 0u;
        int steps = (active && m > 0) ? ((n + 1) >> 1) + Wp - 1 : 0;
        int max_steps = steps;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) max_steps = max(max_steps, __shfl_xor(max_steps, off));
        max_steps = __builtin_amdgcn_readfirstlane(max_steps);
        const uint32_t *pm_w = pm_g + w;
        const uint16_t *txt2 = reinterpret_cast<const uint16_t *>(txt_g);
        const int tmax = (a.text_stride >> 1) - 1;
        // Two-deep software pipeline over LDS: the text pair of iteration k+2 and the match
        // masks of iteration k+1 are requested while iteration k computes, so neither LDS
        // latency sits on the loop-carried dependency (which is the DPP carry chain only).
        auto text_at = [&](int kk) -> uint32_t { return txt2[min(max(kk, 0), tmax)]; };
        uint32_t c0 = text_at(0 - w), c1 = text_at(1 - w);
        uint32_t eqA = pm_w[(c0 & 0xffu) * G], eqB = pm_w[(c0 >> 8) * G];
        uint32_t carry = 0;  // [0] hpA, [1] hnA, [2] hpB, [3] hnB of this lane's last iteration
        for (int k = 0; k < max_steps; ++k) {
            const uint32_t c2 = text_at(k + 2 - w);
            const uint32_t eqA_n = pm_w[(c1 & 0xffu) * G], eqB_n = pm_w[(c1 >> 8) * G];
            uint32_t in = dpp_wave_shr1(carry);
            // keep the three LDS requests above the arithmetic (hipcc otherwise rotates the loop
            // and waits for each request right where it was issued)
            __builtin_amdgcn_sched_barrier(0);
            in = (w == 0) ? 0x5u : in;  // top row of the DP: +1 horizontal delta, never -1
            const uint32_t col = (uint32_t)(k - w) * 2u;  // huge when k < w
            const bool vA = col < un, vB = (col + 1u) < un;
            // ---- column A
            uint32_t hpc = in & 1u, hnc = (in >> 1) & 1u;
            uint32_t x = eqA | hnc;
            uint32_t d0 = (((x & vp) + vp) ^ vp) | x | vn;
            uint32_t hp = vn | ~(d0 | vp);
            uint32_t hn = d0 & vp;
            const uint32_t hpoA = hp >> 31, hnoA = hn >> 31;
            hp = (hp << 1) | hpc;
            hn = (hn << 1) | hnc;
            uint32_t nvp = hn | ~(d0 | hp), nvn = hp & d0;
            vp = vA ? nvp : vp; vn = vA ? nvn : vn;
            // ---- column B
            hpc = (in >> 2) & 1u; hnc = (in >> 3) & 1u;
            x = eqB | hnc;
            d0 = (((x & vp) + vp) ^ vp) | x | vn;
            hp = vn | ~(d0 | vp);
            hn = d0 & vp;
            const uint32_t hpoB = hp >> 31, hnoB = hn >> 31;
            hp = (hp << 1) | hpc;
            hn = (hn << 1) | hnc;
            nvp = hn | ~(d0 | hp); nvn = hp & d0;
            vp = vB ? nvp : vp; vn = vB ? nvn : vn;
            carry = hpoA | (hnoA << 1) | (hpoB << 2) | (hnoB << 3);
            __builtin_amdgcn_sched_barrier(0);  // consume the prefetched values only down here
            eqA = eqA_n; eqB = eqB_n; c1 = c2;
        }
        // D[m][n] = D[0][n] + sum of the vertical deltas of the last column
        //         = n + popcount(VP & rows) - popcount(VN & rows), summed over the slot's words
        if (active && w < Wp) {
            const uint32_t rows = (w == Wp - 1) ? (0xffffffffu >> (31 - ((m - 1) & 31))) : 0xffffffffu;
            const int part = __popc(vp & rows) - __popc(vn & rows);
            if (part) atomicAdd(&ssum[g], part);
        }
        wave_lds_fence();
        if (active && w == 0) {
            const double d = (double)(n + ssum[g]);  // m == 0: no word contributes, d = n
            if (a.out) a.out[t_pair] = d;
            if (a.RA) { a.RA[opos] = d; a.ncm[opos] = 0; }
        }
        wave_lds_fence();
    }
}

int ann_lev_launch(annchor_ctx *c, const PairSource &src, double *d_out, double *d_RA, uint8_t *d_ncm)
{
    if (src.n == 0) return ANNCHOR_OK;
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
    size_t lds = (size_t)a.wave_bytes * LEV_WAVES;
    ANN_REQUIRE(c, lds <= 160 * 1024, ANNCHOR_ELIMIT, "alphabet %d x length %d needs %zu B of LDS (> 160 KiB)", c->alphabet,
                c->maxlen, lds);
    if (lds > 64 * 1024)
        ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)k_lev, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t tasks = (src.n + a.P - 1) / a.P;
    int64_t blocks = (tasks + LEV_WAVES - 1) / LEV_WAVES;
    int max_blocks = c->prop.multiProcessorCount * 8;
    if (blocks > max_blocks) blocks = max_blocks;
    // algorithmic work: one byte per symbol of both strings is all that must be read
    ProfScope ps(c, "levenshtein_pairs", (double)src.n * (2.0 * c->maxlen + 8));
    k_lev<<<(int)blocks, LEV_THREADS, lds, c->stream>>>(a);
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}
