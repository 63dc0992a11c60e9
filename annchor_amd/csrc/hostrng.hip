// hostrng.hip -- host-side (no device code): NumPy's legacy RandomState stream for the
// stratified sampler.
//
// The reference draws its samples with `np.random.seed(random_seed + loop_num)` followed
// by one `np.random.choice(ixmask, size, replace=False)` per bin
// (annchor/utils.py:543-578, inside njit; see DESIGN.md section 5 for what is and is not
// pinned about that stream).  NumPy's legacy choice-without-replacement is
// `ixmask[permutation(len(ixmask))[:size]]`, i.e. a full Fisher-Yates shuffle of
// arange(len) driven by MT19937 with masked rejection sampling.  This file restates
// that published algorithm (Matsumoto & Nishimura's MT19937; NumPy's
// `random_interval` / `_shuffle_raw`) so that the host does not spend ~10 ms per
// iteration inside NumPy's generic shuffle; tests/test_host_logic.py checks it
// against np.random bit for bit.
#include <cstdint>
#include <vector>

#include "../../include/annchor_hip.h"

namespace {
struct MT {
    uint32_t key[624];
    uint32_t buf[624];  // tempered outputs of the current block
    int pos;
    void seed(uint32_t s)
    {
        for (int p = 0; p < 624; ++p) {
            key[p] = s;
            s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)p + 1u;
        }
        pos = 624;
    }
    void gen()
    {
        const uint32_t UP = 0x80000000u, LO = 0x7fffffffu, MA = 0x9908b0dfu;
        int i = 0;
        for (; i < 624 - 397; ++i) {
            uint32_t y = (key[i] & UP) | (key[i + 1] & LO);
            key[i] = key[i + 397] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
        }
        for (; i < 623; ++i) {
            uint32_t y = (key[i] & UP) | (key[i + 1] & LO);
            key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
        }
        uint32_t y = (key[623] & UP) | (key[0] & LO);
        key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
        for (int k = 0; k < 624; ++k) {
            uint32_t z = key[k];
            z ^= z >> 11;
            z ^= (z << 7) & 0x9d2c5680u;
            z ^= (z << 15) & 0xefc60000u;
            z ^= z >> 18;
            buf[k] = z;
        }
        pos = 0;
    }
    inline uint32_t next()
    {
        if (pos == 624) gen();
        return buf[pos++];
    }
    inline uint32_t interval(uint32_t max)  // uniform in [0, max], masked rejection
    {
        if (max == 0) return 0;
        uint32_t mask = max;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        uint32_t v;
        while ((v = next() & mask) > max) {}
        return v;
    }
};
}  // namespace

extern "C" int annchor_legacy_choice_ranks(uint32_t seed, const int64_t *counts, const int64_t *want, int32_t nbins,
                                           int64_t *ranks_out, int64_t *n_out)
{
    if (!counts || !want || !ranks_out || !n_out || nbins < 0) return ANNCHOR_EINVAL;
    MT mt;
    mt.seed(seed);
    static thread_local std::vector<uint32_t> J;       // J[i] = partner drawn for position i (i = c-1 .. 1)
    static thread_local std::vector<uint64_t> bits;    // bitmap of currently tracked positions (kept all-zero between bins)
    static thread_local std::vector<int32_t> slot_of;  // tracked position -> output slot (kept all -1 between bins)
    std::vector<uint32_t> pos;                         // output slot -> tracked position
    int64_t w = 0;
    for (int b = 0; b < nbins; ++b) {
        const int64_t c = counts[b], k = want[b];
        if (c < 0 || k < 0 || c >= (1ll << 31)) return ANNCHOR_ELIMIT;
        if (c < k) {  // utils.py:553-554: the whole bin, no draw
            for (int64_t t = 0; t < c; ++t) ranks_out[w++] = t;
            n_out[b] = c;
            continue;
        }
        // forward: the swap partners of the Fisher-Yates shuffle, in stream order.
        if (J.size() < (size_t)c + 1) J.resize((size_t)c + 1);
        // Branch-free form of the rejection loop, one power-of-two band of i at a time (the
        // mask is constant inside a band): write the masked draw, step to the next position
        // only when it was accepted (value <= i).  Same draws, same order as NumPy.
        for (uint32_t i = c >= 2 ? (uint32_t)(c - 1) : 0u; i >= 1;) {
            const uint32_t mask = 0xffffffffu >> __builtin_clz(i);
            const uint32_t lo = (mask >> 1) + 1;  // smallest i with this mask
            while (i >= lo) {
                if (mt.pos == 624) mt.gen();
                int p = mt.pos;
                while (p < 624 && i >= lo) {
                    const uint32_t v = mt.buf[p++] & mask;
                    J[i] = v;
                    i -= (v <= i);
                }
                mt.pos = p;
            }
        }
        // backward: only the first k output positions are needed.  Undo the swaps from the
        // last (i = 1) to the first (i = c-1), following just those k positions; a bitmap
        // (c bits, cache resident) filters the steps that touch none of them.
        if (bits.size() < (size_t)(c + 63) / 64) bits.resize((size_t)(c + 63) / 64, 0);
        if (slot_of.size() < (size_t)c) slot_of.resize((size_t)c, -1);
        pos.resize((size_t)k);
        for (int64_t t = 0; t < k; ++t) {
            pos[(size_t)t] = (uint32_t)t;
            slot_of[(size_t)t] = (int32_t)t;
            bits[(size_t)t >> 6] |= 1ull << (t & 63);
        }
        const uint64_t *bp = bits.data();
        for (int64_t i = 1; i < c; ++i) {
            const uint32_t j = J[(size_t)i];
            const bool ti = (bp[i >> 6] >> (i & 63)) & 1ull, tj = (bp[j >> 6] >> (j & 63)) & 1ull;
            if (!(ti | tj) || j == (uint32_t)i) continue;
            const int32_t si = slot_of[(size_t)i], sj = slot_of[j];
            slot_of[(size_t)i] = sj;
            slot_of[j] = si;
            if (si >= 0) pos[(size_t)si] = j;
            if (sj >= 0) pos[(size_t)sj] = (uint32_t)i;
            if (ti != tj) {
                bits[(size_t)i >> 6] ^= 1ull << (i & 63);
                bits[j >> 6] ^= 1ull << (j & 63);
            }
        }
        for (int64_t t = 0; t < k; ++t) {
            const uint32_t p = pos[(size_t)t];
            ranks_out[w++] = p;
            slot_of[p] = -1;               // restore the scratch invariants
            bits[p >> 6] = 0;
        }
        n_out[b] = k;
    }
    return ANNCHOR_OK;
}
