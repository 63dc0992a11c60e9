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
// `random_interval` / `_shuffle_raw`); tests/test_host_logic.py checks it against
// np.random bit for bit.
//
// Cost structure and how it is hidden:
//   1. the raw MT19937 output depends on the seed only -> annchor_legacy_prefetch()
//      generates it on a background thread while the GPU runs the stages before sampling;
//   2. the rejection scan that turns raw draws into swap partners J[i] is inherently
//      sequential across bins (a bin starts where the previous one stopped) but is a
//      branch-free compare/advance loop;
//   3. only the first `want` entries of each shuffled bin are needed: the swaps are undone
//      backwards for those entries alone (bitmap-filtered), on pooled helper threads while the
//      scan of the later bins proceeds, and on the calling thread once the scan is done.
// Measured on the MI355X box's host (EPYC 9575F), C2 bin sizes (1.25 M draws, 1.8 M stream
// words): scan 0.43 ms (AVX-512: 16 draws per compare, thresholds lagged two blocks, register
// compress; 1.5 ms portable), trace of the last big bin 0.25-0.35 ms on a pooled helper thread.
#include <immintrin.h>
#include <pthread.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/annchor_hip.h"

namespace {
struct MT {
    uint32_t key[624];
    void seed(uint32_t s)
    {
        for (int p = 0; p < 624; ++p) {
            key[p] = s;
            s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)p + 1u;
        }
    }
    // next 624 tempered outputs.  The recurrence reaches back 227 / forward 397 words, so it
    // vectorises 16 wide; compiled for AVX-512 / AVX2 / baseline and chosen at run time (the
    // producer thread's speed bounds the first sampling step of a fit: its consumer starts
    // ~1.3 ms after the stream does)
    __attribute__((always_inline)) inline void block_body(uint32_t *out)
    {
        const uint32_t UP = 0x80000000u, LO = 0x7fffffffu, MA = 0x9908b0dfu;
        int i = 0;
        for (; i < 624 - 397; ++i) {
            uint32_t y = (key[i] & UP) | (key[i + 1] & LO);
            key[i] = key[i + 397] ^ (y >> 1) ^ ((0u - (y & 1u)) & MA);
        }
        for (; i < 623; ++i) {
            uint32_t y = (key[i] & UP) | (key[i + 1] & LO);
            key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ ((0u - (y & 1u)) & MA);
        }
        uint32_t y = (key[623] & UP) | (key[0] & LO);
        key[623] = key[396] ^ (y >> 1) ^ ((0u - (y & 1u)) & MA);
        for (int k = 0; k < 624; ++k) {
            uint32_t z = key[k];
            z ^= z >> 11;
            z ^= (z << 7) & 0x9d2c5680u;
            z ^= (z << 15) & 0xefc60000u;
            z ^= z >> 18;
            out[k] = z;
        }
    }
    // AVX-512 by hand: 16 state words per step.  In-place update in three stretches (the recurrence reads key[i + 1]
    // and key[i + 397]: words of the previous block for i < 227, of this block from then on -- always >= 16 words away
    // from the ones being written, except the very last word, which wraps to key[0]), bit select / conditional xor as
    // one ternary-logic / masked op each; tempering in a second sweep.  ~0.1 ns per word against ~0.4 for what the
    // compiler makes of block_body: generation of a C2 stream (1.9 M words) 0.8 -> 0.3 ms on the box's EPYC 9575F,
    // which is what lets it hide behind the anchor rounds (tools/rng_inline.py).
    __attribute__((target("avx512f"))) void block_avx512(uint32_t *out)
    {
        const __m512i UPm = _mm512_set1_epi32((int)0x80000000u), MAv = _mm512_set1_epi32((int)0x9908b0dfu), one = _mm512_set1_epi32(1);
#define MT_STEP(I, SRC, LANES)                                                                                              \
    {                                                                                                                       \
        const __m512i a = _mm512_maskz_loadu_epi32((LANES), key + (I)), b = _mm512_maskz_loadu_epi32((LANES), key + (I) + 1); \
        const __m512i cc = _mm512_maskz_loadu_epi32((LANES), key + (SRC));                                                   \
        const __m512i y = _mm512_ternarylogic_epi32(UPm, a, b, 0xCA); /* UPm ? a : b == (a & UP) | (b & LO) */               \
        const __mmask16 odd = _mm512_test_epi32_mask(y, one);                                                               \
        __m512i r = _mm512_xor_si512(cc, _mm512_srli_epi32(y, 1));                                                           \
        r = _mm512_mask_xor_epi32(r, odd, r, MAv);                                                                          \
        _mm512_mask_storeu_epi32(key + (I), (LANES), r);                                                                    \
    }
        int i = 0;
        for (; i + 16 <= 227; i += 16) MT_STEP(i, i + 397, (__mmask16)0xffff)
        MT_STEP(i, i + 397, (__mmask16)((1u << (227 - i)) - 1u))               // 227 = 14 * 16 + 3
        for (i = 227; i + 16 <= 623; i += 16) MT_STEP(i, i - 227, (__mmask16)0xffff)
        MT_STEP(i, i - 227, (__mmask16)((1u << (623 - i)) - 1u))               // 623 - 227 = 24 * 16 + 12
#undef MT_STEP
        {
            const uint32_t y = (key[623] & 0x80000000u) | (key[0] & 0x7fffffffu);
            key[623] = key[396] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
        }
        const __m512i m7 = _mm512_set1_epi32((int)0x9d2c5680u), m15 = _mm512_set1_epi32((int)0xefc60000u);
        for (int k = 0; k < 624; k += 16) {                                      // 624 = 39 * 16
            __m512i z = _mm512_loadu_si512(key + k);
            z = _mm512_xor_si512(z, _mm512_srli_epi32(z, 11));
            z = _mm512_xor_si512(z, _mm512_and_si512(_mm512_slli_epi32(z, 7), m7));
            z = _mm512_xor_si512(z, _mm512_and_si512(_mm512_slli_epi32(z, 15), m15));
            z = _mm512_xor_si512(z, _mm512_srli_epi32(z, 18));
            _mm512_storeu_si512(out + k, z);
        }
    }
    __attribute__((target("avx2"))) void block_avx2(uint32_t *out) { block_body(out); }
    void block_base(uint32_t *out) { block_body(out); }
    void block(uint32_t *out)
    {
        static const int level = getenv("ANNCHOR_RNG_SCALAR") ? 0 : __builtin_cpu_supports("avx512f") ? 2 : __builtin_cpu_supports("avx2") ? 1 : 0;
        if (level == 2) block_avx512(out);
        else if (level == 1) block_avx2(out);
        else block_base(out);
    }
};

// Stream buffers are recycled: a fresh multi-megabyte allocation costs more in page faults
// than generating its contents.
std::mutex g_pool_mu;
std::vector<std::pair<size_t, uint32_t *>> g_pool;
uint32_t *pool_get(size_t words, size_t *got)
{
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (size_t i = 0; i < g_pool.size(); ++i)
            if (g_pool[i].first >= words) {
                uint32_t *p = g_pool[i].second;
                *got = g_pool[i].first;
                g_pool.erase(g_pool.begin() + (long)i);
                return p;
            }
    }
    *got = words;
    return new uint32_t[words];
}
void pool_put(uint32_t *p, size_t words)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (g_pool.size() < 8) g_pool.push_back({words, p}); else delete[] p;
}

// A seed's output stream, produced in 624-word blocks by a background thread (or on
// demand by the consumer when it runs ahead of / without a producer).
struct Stream {
    MT mt;
    struct Buf {
        uint32_t *p = nullptr;
        uint32_t *get() const { return p; }
    } buf;                            // pooled, uninitialised storage
    size_t buf_words = 0;
    std::atomic<size_t> ready{0};     // words valid in buf
    size_t cap_blocks = 0;
    std::thread producer;
    std::atomic<bool> producing{false};

    ~Stream()
    {
        if (producer.joinable()) producer.join();
        if (buf.p) pool_put(buf.p, buf_words);
    }

    void start(uint32_t seed, size_t ndraws, bool background)
    {
        mt.seed(seed);
        cap_blocks = (ndraws + 623) / 624 + 1;
        buf.p = pool_get(cap_blocks * 624, &buf_words);
        if (background) {
            producing = true;
            producer = std::thread([this] {
                for (size_t b = 0; b < cap_blocks; ++b) {
                    mt.block(buf.get() + b * 624);
                    ready.store((b + 1) * 624, std::memory_order_release);
                }
                producing = false;
            });
        }
    }
    // make words [0, upto) available (consumer side)
    inline void need(size_t upto)
    {
        if (ready.load(std::memory_order_acquire) >= upto) return;
        if (producer.joinable() && upto <= cap_blocks * 624) {
            while (ready.load(std::memory_order_acquire) < upto) std::this_thread::yield();
            return;
        }
        if (producer.joinable()) producer.join();   // beyond the prefetched range: extend inline
        while (ready.load(std::memory_order_relaxed) < upto) {
            const size_t r = ready.load(std::memory_order_relaxed);
            if (buf_words < r + 624) {
                size_t nw = 0;
                uint32_t *nb = pool_get(std::max(buf_words * 2, r + 624), &nw);
                memcpy(nb, buf.p, r * sizeof(uint32_t));
                pool_put(buf.p, buf_words);
                buf.p = nb;
                buf_words = nw;
            }
            mt.block(buf.get() + r);
            ready.store(r + 624, std::memory_order_relaxed);
        }
    }
};

std::mutex g_mu;
std::map<uint32_t, std::shared_ptr<Stream>> g_streams;
// Finished streams are kept (the stream is a pure function of the seed, and the default seed is
// the same for every Annchor): a later fit with the same seed finds its words already there.
// At most RNG_CACHE_ENTRIES streams of at most RNG_CACHE_MAX_WORDS words, least recently used out.
constexpr size_t RNG_CACHE_ENTRIES = 4, RNG_CACHE_MAX_WORDS = 16u << 20;   // 64 MB each
std::vector<std::pair<uint32_t, std::shared_ptr<Stream>>> g_cache;   // most recent last

std::shared_ptr<Stream> cache_take(uint32_t seed)   // g_mu held
{
    if (getenv("ANNCHOR_RNG_NO_CACHE")) { g_cache.clear(); return nullptr; }
    for (size_t i = 0; i < g_cache.size(); ++i)
        if (g_cache[i].first == seed) {
            auto st = g_cache[i].second;
            g_cache.erase(g_cache.begin() + (long)i);
            return st;
        }
    return nullptr;
}
void cache_put(uint32_t seed, const std::shared_ptr<Stream> &st)   // g_mu held
{
    if (getenv("ANNCHOR_RNG_NO_CACHE")) return;   // (read per call: bench.py reports the fit time without the cache too)
    if (st->producing.load() || st->ready.load() > RNG_CACHE_MAX_WORDS) return;
    if (st->producer.joinable()) st->producer.join();
    g_cache.push_back({seed, st});
    if (g_cache.size() > RNG_CACHE_ENTRIES) g_cache.erase(g_cache.begin());
}

// Per-bin scratch kept across calls (grow-only; `flag` all zero and `slot_of` all -1
// between uses) so that a call neither allocates nor first-touches megabytes.
struct BinScratch {
    std::vector<uint32_t> J;
    std::vector<uint8_t> flag;       // flag[x] = 1: position x is tracked (the trace's test: one load and an OR per partner)
    std::vector<uint64_t> hbits;     // bins too long for `flag` to stay in cache: a hashed bit set of the positions ever tracked
    std::vector<int32_t> slot_of;
};
std::vector<std::unique_ptr<BinScratch>> g_scratch;
std::mutex g_call_mu;

// AVX-512 paths are compiled with per-function target attributes and chosen at run time;
// ANNCHOR_RNG_SCALAR=1 forces the portable loops (tests run both).
bool use_avx512()
{
    static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") &&
                           !(getenv("ANNCHOR_RNG_SCALAR") && atoi(getenv("ANNCHOR_RNG_SCALAR")));
    return ok;
}

// Swap partners are stored in stream (chronological) order: Jc[t] is the partner drawn for
// index i = c - 1 - t.
inline uint32_t partner(const uint32_t *Jc, int64_t c, int64_t i) { return Jc[c - 1 - i]; }

struct Tracer {
    uint8_t *fl;
    int32_t *slot_of;
    uint32_t *pos;
    inline void step(int64_t i, uint32_t j)
    {
        const bool ti = fl[i] != 0, tj = fl[j] != 0;
        if (!(ti | tj) || j == (uint32_t)i) return;
        const int32_t si = slot_of[(size_t)i], sj = slot_of[j];
        slot_of[(size_t)i] = sj;
        slot_of[j] = si;
        if (si >= 0) pos[(size_t)si] = j;
        if (sj >= 0) pos[(size_t)sj] = (uint32_t)i;
        if (ti != tj) {
            fl[i] ^= 1;
            fl[j] ^= 1;
        }
    }
};

// Undo the Fisher-Yates swaps for the first k output positions only.
void trace_prefix(BinScratch *sc, int64_t c, int64_t k, int64_t *out)
{
    const uint32_t *Jc = sc->J.data();
    if (sc->flag.size() < (size_t)c + 64) sc->flag.resize((size_t)c + 64, 0);
    if (sc->slot_of.size() < (size_t)c) sc->slot_of.resize((size_t)c, -1);
    std::vector<uint32_t> pos((size_t)k);
    Tracer T{sc->flag.data(), sc->slot_of.data(), pos.data()};
    for (int64_t t = 0; t < k; ++t) {
        pos[(size_t)t] = (uint32_t)t;
        T.slot_of[(size_t)t] = (int32_t)t;
        T.fl[(size_t)t] = 1;
    }
    int64_t i = 1;
    for (; i < c && i < k; ++i) T.step(i, partner(Jc, c, i));
    {
        // steps i >= k: position i is untracked until its own step (earlier steps only touch
        // smaller positions), so a step matters only if its partner is tracked.  Sixteen
        // partners are tested per branch against one flag byte per position (a load and an OR
        // each; the bit map this replaced cost a shift pair more per partner and its 8x smaller
        // footprint bought nothing: 0.41 -> 0.25 ms for a bin of 620 000 pairs on the box, which
        // is the critical path of a draw; a summary level in front of the bit map made it slower;
        // an AVX-512 gather measured 3x slower on Zen 5); the rare group with a hit is replayed exactly.
        const uint8_t *fl = T.fl;
        // Long bins (the flag bytes of 18 M pairs are 18 MB: every test a cache miss, ~5 ns per step): the partners are
        // tested against a hashed set of 2^20 bits (128 KB: L2) holding every position that was EVER tracked (k at the start,
        // one more per real hit: ~k ln(c / k)); only a partner whose hashed bit is set looks at the exact flags.
        static const int64_t hashed_min = getenv("ANNCHOR_RNG_HASHED_MIN") ? atoll(getenv("ANNCHOR_RNG_HASHED_MIN")) : (1ll << 20);   // (C2 bins of 620 000 pairs: flags 3.16 ms per fit, hashed 4.2; bins of 3.8 M: 56 vs 50 ms per fit)
        if (c >= hashed_min) {
            constexpr int HB = 20;
            if (sc->hbits.size() < ((size_t)1 << (HB - 6))) sc->hbits.resize((size_t)1 << (HB - 6), 0);
            uint64_t *hb = sc->hbits.data();
            auto hslot = [](uint32_t x) -> uint32_t { return (uint32_t)(((uint64_t)x * 0x9E3779B97F4A7C15ull) >> (64 - HB)); };
            std::vector<uint32_t> touched;
            touched.reserve((size_t)k * 16);
            for (int64_t t2 = 0; t2 < k; ++t2) { const uint32_t h = hslot(pos[(size_t)t2]); hb[h >> 6] |= 1ull << (h & 63); touched.push_back(h >> 6); }
            for (; i + 16 <= c; i += 16) {
                const uint32_t *q = Jc + (c - 1 - i - 15);
                uint64_t any = 0;
                for (int u = 0; u < 16; ++u) { const uint32_t h = hslot(q[u]); any |= hb[h >> 6] >> (h & 63); }
                if (!(any & 1ull)) continue;
                for (int64_t s2 = i; s2 < i + 16; ++s2) {
                    const uint32_t j = partner(Jc, c, s2);
                    const uint32_t h = hslot(j);
                    if (!((hb[h >> 6] >> (h & 63)) & 1ull) || !fl[j] || j == (uint32_t)s2) continue;
                    T.step(s2, j);   // position s2 is tracked from here on
                    const uint32_t h2 = hslot((uint32_t)s2);
                    hb[h2 >> 6] |= 1ull << (h2 & 63);
                    touched.push_back(h2 >> 6);
                }
            }
            for (uint32_t w : touched) hb[w] = 0;
        }
        for (; i + 16 <= c; i += 16) {
            const uint32_t *q = Jc + (c - 1 - i - 15);
            uint32_t any = 0;
            for (int u = 0; u < 16; ++u) any |= fl[q[u]];
            if (any)
                for (int64_t s2 = i; s2 < i + 16; ++s2) T.step(s2, partner(Jc, c, s2));
        }
    }
    for (; i < c; ++i) T.step(i, partner(Jc, c, i));
    for (int64_t t = 0; t < k; ++t) {
        const uint32_t p = pos[(size_t)t];
        out[t] = p;
        T.slot_of[p] = -1;   // restore the scratch invariants
        T.fl[p] = 0;
    }
}

// Forward rejection scan for one bin: partners for i = c-1 .. 1 in stream order into Jc.
// One power-of-two band of i at a time (constant mask inside a band): write the masked draw,
// advance only when it was accepted (value <= i).  Same draws as NumPy.
// Windows of W draws: while i moves by at most W inside a window, a draw with value <= i-W
// is accepted and one with value > i is rejected whatever the exact i is, so the W
// masks/compares are independent of the running index; the (rare) window holding a value in
// (i-W, i] falls through to the exact loop.
#define PUB_EVERY 16384   // stream words between two publications of the streamed trace's progress
struct Scan {
    Stream *st;
    size_t cur;
    // (streamed trace: the number of partner words written so far -- all bins -- is published every PUB_EVERY words for a kernel
    // that reads them from pinned memory while the scan moves on; nullptr otherwise)
    unsigned long long *pub = nullptr;
    unsigned long long pub_base = 0;
    inline void publish(size_t t) const { if (pub) __atomic_store_n(pub, pub_base + t, __ATOMIC_RELEASE); }
    inline const uint32_t *fill(size_t *avail)
    {
        *avail = st->ready.load(std::memory_order_acquire);
        if (*avail <= cur) { st->need(cur + 624); *avail = st->ready.load(std::memory_order_acquire); }
        return st->buf.get();
    }
};

// STORE = false: only advance through the bin's draws (pass 1 of the parallel scheme: where
// does the next bin start in the stream?)
template <bool STORE> void scan_bin_scalar(Scan &S, int64_t c, uint32_t *Jc)
{
    size_t t = 0;
    for (uint32_t i = c >= 2 ? (uint32_t)(c - 1) : 0u; i >= 1;) {
        const uint32_t mask = 0xffffffffu >> __builtin_clz(i);
        const uint32_t lo = (mask >> 1) + 1;  // smallest i with this mask
        while (i >= lo) {
            size_t avail;
            const uint32_t *buf = S.fill(&avail);
            size_t p = S.cur;
            if (STORE && S.pub) {
                S.publish(t);
                if (avail > p + PUB_EVERY) avail = p + PUB_EVERY;
            }
            while (p + 8 <= avail && i >= lo + 8) {
                const uint32_t hi_t = i, lo_t = i - 8;
                uint32_t v[8];
                uint32_t unsure = 0;
                for (int u = 0; u < 8; ++u) {
                    v[u] = buf[p + u] & mask;
                    unsure |= (uint32_t)(v[u] > lo_t) & (uint32_t)(v[u] <= hi_t);
                }
                if (unsure) break;
                for (int u = 0; u < 8; ++u) {
                    if (STORE) Jc[t] = v[u];
                    const uint32_t acc = (v[u] <= lo_t);
                    t += acc;
                    i -= acc;
                }
                p += 8;
            }
            for (int u = 0; u < 8 && p < avail && i >= lo; ++u) {  // exact steps (window with an unsure draw / band edge)
                const uint32_t v = buf[p++] & mask;
                if (STORE) Jc[t] = v;
                const uint32_t acc = (v <= i);
                t += acc;
                i -= acc;
            }
            S.cur = p;
        }
    }
}

template <bool STORE> __attribute__((target("avx512f,avx512bw,popcnt"))) void scan_bin_avx512(Scan &S, int64_t c, uint32_t *Jc)
{
    size_t t = 0;
    for (uint32_t i = c >= 2 ? (uint32_t)(c - 1) : 0u; i >= 1;) {
        const uint32_t mask = 0xffffffffu >> __builtin_clz(i);
        const uint32_t lo = (mask >> 1) + 1;
        const __m512i vmask = _mm512_set1_epi32((int)mask);
        while (i >= lo) {
            size_t avail;
            const uint32_t *buf = S.fill(&avail);
            size_t p = S.cur;
            if (STORE && S.pub) {   // a piece of the stream at a time: the words written are published between two pieces
                S.publish(t);
                if (avail > p + PUB_EVERY) avail = p + PUB_EVERY;
            }
            // The thresholds of a 16-draw block come from the running index TWO blocks back (i2):
            // the index now is in [i2 - 32, i2] and stays within 16 of that inside the block, so a
            // draw <= i2 - 48 is accepted and one > i2 rejected whatever happened in between.  The
            // compare therefore does not wait for the previous block's popcount (GPR -> vector
            // broadcast, compare, mask -> GPR, popcount: ~30 cycles of latency per block when
            // chained; the chain now spans two blocks and overlaps).  A draw in (i2 - 48, i2]
            // sends the block to the exact loop below.
            uint32_t i1 = i, i2 = i;
            while (p + 16 <= avail && i >= lo + 16 && i >= 64) {
                const __m512i v = _mm512_and_si512(_mm512_loadu_si512(buf + p), vmask);
                const __mmask16 acc = _mm512_cmple_epu32_mask(v, _mm512_set1_epi32((int)(i2 - 48)));
                const __mmask16 rej = _mm512_cmpgt_epu32_mask(v, _mm512_set1_epi32((int)i2));
                if ((__mmask16)(acc | rej) != 0xffff) break;
                if (STORE) _mm512_storeu_si512(Jc + t, _mm512_maskz_compress_epi32(acc, v));   // Jc has 32 words of slack
                const uint32_t n = (uint32_t)__builtin_popcount(acc);
                t += n;
                i2 = i1;
                i1 = i;
                i -= n;
                p += 16;
            }
            for (int u = 0; u < 16 && p < avail && i >= lo; ++u) {
                const uint32_t v = buf[p++] & mask;
                if (STORE) Jc[t] = v;
                const uint32_t a = (v <= i);
                t += a;
                i -= a;
            }
            S.cur = p;
        }
    }
}

// Persistent helpers for the backward traces.  A thread created per call lands on a sleeping
// core with cold caches (measured on the box: the same draw takes 1.1-1.3 ms on a fresh thread,
// 0.75-0.8 ms on a warm one); these block on a condition variable between calls, are woken when
// a draw starts and spin on the bin states while the scan runs.
// CPUs sharing the last-level cache with `cpu`, without `cpu` and its SMT sibling(s) (sysfs; empty if it cannot be read).
static cpu_set_t l3_siblings_of(int cpu, bool *ok)
{
    cpu_set_t set;
    CPU_ZERO(&set);
    *ok = false;
    auto read_list = [&](const char *leaf, cpu_set_t *out) -> bool {
        char path[160];
        snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/%s", cpu, leaf);
        FILE *f = fopen(path, "r");
        if (!f) return false;
        char buf[512];
        const bool got = fgets(buf, sizeof buf, f) != nullptr;
        fclose(f);
        if (!got) return false;
        for (char *p = buf; *p;) {   // "0-7,128-135"
            char *e;
            const long a = strtol(p, &e, 10);
            if (e == p) break;
            long b = a;
            if (*e == '-') { p = e + 1; b = strtol(p, &e, 10); }
            for (long c2 = a; c2 <= b && c2 < CPU_SETSIZE; ++c2) CPU_SET((int)c2, out);
            p = (*e == ',') ? e + 1 : e;
            if (*e != ',') break;
        }
        return true;
    };
    cpu_set_t smt;
    CPU_ZERO(&smt);
    if (!read_list("cache/index3/shared_cpu_list", &set)) return set;
    read_list("topology/thread_siblings_list", &smt);
    CPU_SET(cpu, &smt);
    for (int c2 = 0; c2 < CPU_SETSIZE; ++c2)
        if (CPU_ISSET(c2, &smt)) CPU_CLR(c2, &set);
    *ok = CPU_COUNT(&set) > 0;
    return set;
}

struct HelperPool {
    std::atomic<int> poster_cpu{-1};   // where the thread that posted the current job runs (ANNCHOR_RNG_PIN: helpers follow its L3)
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<uint64_t> gen{0};
    const std::function<void()> *job = nullptr;
    std::atomic<int> inside{0};
    int nthreads = 0;
    // Optionally (ANNCHOR_RNG_SPIN_US > 0) a helper keeps polling for the next job for that long
    // before it blocks.  Off by default: on the box's host (EPYC 9575F) 150-fit medians were 5.95 ms
    // without polling and 6.15 ms with 5 ms of it -- the polling cores cost the calling thread more
    // boost clock than their warm caches give back.
    static int64_t spin_ns()
    {
        static const int64_t v = 1000 * (getenv("ANNCHOR_RNG_SPIN_US") ? atoll(getenv("ANNCHOR_RNG_SPIN_US")) : 0);
        return v;
    }
    void start(int n)
    {
        std::lock_guard<std::mutex> lk(mu);
        for (; nthreads < n; ++nthreads)
            std::thread([this] {
                uint64_t seen = 0;
                for (;;) {
                    // poll, then block
                    const auto t0 = std::chrono::steady_clock::now();
                    while (gen.load(std::memory_order_acquire) == seen) {
                        _mm_pause();
                        if (std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() >= spin_ns()) {
                            std::unique_lock<std::mutex> lk(mu);
                            cv.wait(lk, [&] { return gen.load(std::memory_order_acquire) != seen; });
                            break;
                        }
                    }
                    const std::function<void()> *j;
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        seen = gen.load(std::memory_order_acquire);
                        j = job;
                        if (!j) continue;          // the job is already over: nothing to join
                        inside.fetch_add(1, std::memory_order_acq_rel);
                    }
                    {
                        // the partner arrays a helper traces were just written by the posting thread: stay inside its last-level
                        // cache (a helper woken on another CCX reads them across the fabric -- fits then take 4.1 ms instead of 3.3)
                        // (box, 23 fits per run, alternating: 3.26-3.45 ms pinned against 3.29-3.91 free; ANNCHOR_RNG_PIN=0 leaves them free)
                        static const bool pin = getenv("ANNCHOR_RNG_PIN") ? atoi(getenv("ANNCHOR_RNG_PIN")) != 0 : true;
                        thread_local int pinned_for = -1;   // the CPU the current mask was derived from
                        const int pc = poster_cpu.load(std::memory_order_acquire);
                        if (pin && pc >= 0 && pc < CPU_SETSIZE && pc != pinned_for) {
                            // (re-derive only when the poster moved; same L3 -> same mask, and the syscall is skipped then)
                            static std::atomic<int> l3_of[CPU_SETSIZE];   // first CPU of cpu's L3 domain + 1 (0 = not looked up yet)
                            thread_local int l3_pinned = -1;
                            int id = l3_of[pc].load(std::memory_order_relaxed) - 1;
                            bool ok = false;
                            cpu_set_t set;
                            if (id < 0 || id != l3_pinned) {
                                set = l3_siblings_of(pc, &ok);
                                if (ok) {
                                    // (the domain's id: its lowest CPU, counting the poster and its SMT sibling back in)
                                    int first = pc;
                                    for (int c2 = 0; c2 < pc; ++c2) if (CPU_ISSET(c2, &set)) { first = c2; break; }
                                    id = first;
                                    l3_of[pc].store(id + 1, std::memory_order_relaxed);
                                    if (id != l3_pinned) { sched_setaffinity(0, sizeof set, &set); l3_pinned = id; }
                                }
                            }
                            pinned_for = pc;
                        }
                    }
                    (*j)();
                    inside.fetch_sub(1, std::memory_order_acq_rel);
                }
            }).detach();
    }
    void post(const std::function<void()> *j)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            job = j;
            poster_cpu.store(sched_getcpu(), std::memory_order_release);
            gen.fetch_add(1, std::memory_order_acq_rel);
        }
        cv.notify_all();
    }
    // no helper may enter the job any more; wait for those inside to leave it
    void finish()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            job = nullptr;
        }
        while (inside.load(std::memory_order_acquire) != 0) _mm_pause();
    }
};
HelperPool *g_helpers = new HelperPool();   // leaked on purpose: its threads outlive static destruction
}  // namespace

extern "C" int annchor_legacy_prefetch(uint32_t seed, int64_t ndraws)
{
    if (ndraws <= 0 || ndraws > (1ll << 33)) return ANNCHOR_EINVAL;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (getenv("ANNCHOR_RNG_NO_CACHE")) g_cache.clear();
        for (auto &e : g_cache)
            if (e.first == seed && e.second->ready.load() >= (size_t)ndraws) return ANNCHOR_OK;   // already generated
    }
    auto s = std::make_shared<Stream>();
    s->start(seed, (size_t)ndraws, true);
    std::lock_guard<std::mutex> lk(g_mu);
    g_streams[seed] = s;   // a previous stream of the same seed is joined and dropped
    return ANNCHOR_OK;
}

// The same stream produced HERE, on the calling thread, before the call returns.  A fit() calls this while the GPU
// runs a stage the host would otherwise wait for (the anchor rounds; the model fit of an iteration): the calling thread's
// core is warm and clocked up -- 1.9 M words take ~0.4 ms on it, ~2 ms on a freshly woken producer thread.
extern "C" int annchor_legacy_generate(uint32_t seed, int64_t ndraws)
{
    if (ndraws <= 0 || ndraws > (1ll << 33)) return ANNCHOR_EINVAL;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (getenv("ANNCHOR_RNG_NO_CACHE")) g_cache.clear();
        for (auto &e : g_cache)
            if (e.first == seed && e.second->ready.load() >= (size_t)ndraws) return ANNCHOR_OK;   // already generated
        auto it = g_streams.find(seed);
        if (it != g_streams.end() && it->second->cap_blocks * 624 >= (size_t)ndraws) return ANNCHOR_OK;   // being / already produced
    }
    auto s = std::make_shared<Stream>();
    s->start(seed, (size_t)ndraws, false);
    s->need((size_t)ndraws);
    std::lock_guard<std::mutex> lk(g_mu);
    g_streams[seed] = s;
    return ANNCHOR_OK;
}

// ... in pieces: words [0, upto) of the stream on the calling thread (the stream is created at the first call; the draw that
// consumes it extends it inline should a piece be missing).  ctx.hip parks these pieces at a context's host waits.
int ann_legacy_generate_upto(uint32_t seed, int64_t ndraws, int64_t upto)
{
    if (ndraws <= 0 || ndraws > (1ll << 33)) return ANNCHOR_EINVAL;
    std::shared_ptr<Stream> s;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (getenv("ANNCHOR_RNG_NO_CACHE")) g_cache.clear();
        for (auto &e : g_cache)
            if (e.first == seed && e.second->ready.load() >= (size_t)ndraws) return ANNCHOR_OK;   // already generated
        auto it = g_streams.find(seed);
        if (it != g_streams.end() && it->second->cap_blocks * 624 >= (size_t)ndraws) s = it->second;
        else {
            s = std::make_shared<Stream>();
            s->start(seed, (size_t)ndraws, false);
            g_streams[seed] = s;
        }
    }
    if (s->producer.joinable()) return ANNCHOR_OK;   // a producer thread has it
    s->need((size_t)std::min(upto, ndraws));
    return ANNCHOR_OK;
}

// The forward half of the draw alone, for the device-side trace (csrc/drawtrace.hip): the swap partners of every bin that is
// shuffled (counts[b] >= want[b]) in stream order into J + joff[b] (counts[b] words, 32 words of slack behind them); after_bin is
// called as soon as a bin's partners are complete -- the caller queues their upload while the scan moves on.  Same stream, same
// draws as annchor_legacy_choice_ranks.
int ann_legacy_scan(uint32_t seed, const int64_t *counts, const int64_t *want, int32_t nbins, uint32_t *J, const int64_t *joff,
                    void (*after_bin)(int, void *), void *user, unsigned long long *progress)
{
    std::shared_ptr<Stream> st;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_streams.find(seed);
        if (it != g_streams.end()) { st = it->second; g_streams.erase(it); }
        else st = cache_take(seed);
    }
    if (!st) {
        st = std::make_shared<Stream>();
        int64_t tot = 0;
        for (int b = 0; b < nbins; ++b) tot += counts[b] >= want[b] ? counts[b] : 0;
        st->start(seed, (size_t)(tot + tot / 2 + 1024), false);
    }
    std::lock_guard<std::mutex> call_lk(g_call_mu);
    Scan S{st.get(), 0};
    S.pub = progress;   // (*progress: partner words complete, counted over the shuffled bins in order -- c - 1 per bin)
    for (int b = 0; b < nbins; ++b) {
        if (counts[b] < want[b] || counts[b] < 2) continue;   // utils.py:553-554: the whole bin, no draw (one element: nothing to draw)
        if (use_avx512()) scan_bin_avx512<true>(S, counts[b], J + joff[b]);
        else scan_bin_scalar<true>(S, counts[b], J + joff[b]);
        S.pub_base += (unsigned long long)(counts[b] - 1);
        S.publish(0);
        if (after_bin) after_bin(b, user);
    }
    std::lock_guard<std::mutex> lk(g_mu);
    cache_put(seed, st);
    return ANNCHOR_OK;
}

extern "C" int annchor_legacy_choice_ranks(uint32_t seed, const int64_t *counts, const int64_t *want, int32_t nbins,
                                           int64_t *ranks_out, int64_t *n_out)
{
    if (!counts || !want || !ranks_out || !n_out || nbins < 0) return ANNCHOR_EINVAL;
    static const bool timing = getenv("ANNCHOR_RNG_TIMING") != nullptr;   // stderr breakdown of one call
    const auto t_entry = std::chrono::steady_clock::now();
    auto ms_since = [&](std::chrono::steady_clock::time_point t0) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    std::shared_ptr<Stream> st;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_streams.find(seed);
        if (it != g_streams.end()) { st = it->second; g_streams.erase(it); }   // prefetched for this call
        else st = cache_take(seed);                                           // generated by an earlier call
    }
    if (!st) {
        st = std::make_shared<Stream>();
        int64_t tot = 0;
        for (int b = 0; b < nbins; ++b) tot += counts[b] >= want[b] ? counts[b] : 0;
        st->start(seed, (size_t)(tot + tot / 2 + 1024), false);
    }
    std::lock_guard<std::mutex> call_lk(g_call_mu);
    struct Keep {   // on every way out: the (finished) stream goes to the per-seed cache
        uint32_t seed;
        std::shared_ptr<Stream> &st;
        ~Keep()
        {
            std::lock_guard<std::mutex> lk(g_mu);
            cache_put(seed, st);
        }
    } keep{seed, st};
    while (g_scratch.size() < (size_t)nbins) g_scratch.emplace_back(new BinScratch());
    Scan S{st.get(), 0};  // next unread word of the stream
    std::vector<int64_t> offs((size_t)nbins + 1, 0);
    for (int b = 0; b < nbins; ++b) {
        if (counts[b] < 0 || want[b] < 0 || counts[b] >= (1ll << 31)) return ANNCHOR_ELIMIT;
        n_out[b] = counts[b] < want[b] ? counts[b] : want[b];
        offs[(size_t)b + 1] = offs[(size_t)b] + n_out[b];
    }
    // Large draws (pair lists of 10^7..10^8 candidates: the shuffles cover every not-computed
    // pair): two passes.  Pass 1 runs through the stream counting only -- which word does each
    // bin start at? -- then the bins, now independent, are scanned (with stores) and traced
    // by one thread each.  The sequential part drops to the count-only scan.
    {
        // (Round 2 scanned twice above 4 M draws -- a count pass to find where every bin starts, then scan + trace of the bins in
        // parallel.  With the faster traces and the pinned helpers the single sequential scan that hands each bin to a helper as
        // soon as it is scanned wins at every size measured: the populations are uneven -- one bin of 16 M pairs of 27 M -- and the
        // count pass (4 ms there) only delays that bin's trace.  N = 10 000: 74 -> 67 ms per fit, N = 16 000: 224 -> 194 ms.  Kept
        // behind ANNCHOR_RNG_PAR_MIN for experiments.)
        static const int64_t par_threshold = getenv("ANNCHOR_RNG_PAR_MIN") ? atoll(getenv("ANNCHOR_RNG_PAR_MIN")) : (1ll << 60);
        int64_t total = 0;
        int nlive = 0;
        for (int b = 0; b < nbins; ++b)
            if (counts[b] >= want[b]) { total += counts[b]; ++nlive; }
        if (total >= par_threshold && nlive > 1) {
            std::vector<size_t> start((size_t)nbins, 0);
            for (int b = 0; b < nbins; ++b) {
                const int64_t c = counts[b], k = want[b];
                if (c < k) {
                    int64_t *out = ranks_out + offs[(size_t)b];
                    for (int64_t t = 0; t < c; ++t) out[t] = t;
                    continue;
                }
                start[(size_t)b] = S.cur;
                if (use_avx512()) scan_bin_avx512<false>(S, c, nullptr);
                else scan_bin_scalar<false>(S, c, nullptr);
            }
            const double t_count = ms_since(t_entry);
            for (int b = 0; b < nbins; ++b) {   // grow the scratch before the threads start
                BinScratch *sc = g_scratch[(size_t)b].get();
                if (counts[b] >= want[b] && sc->J.size() < (size_t)counts[b] + 32) sc->J.resize((size_t)counts[b] + 32);
            }
            std::atomic<int> next{0};
            auto work = [&] {
                for (;;) {
                    const int b = next.fetch_add(1);
                    if (b >= nbins) return;
                    if (counts[b] < want[b]) continue;
                    Scan S2{st.get(), start[(size_t)b]};   // every word of the bin is already generated
                    BinScratch *sc = g_scratch[(size_t)b].get();
                    const double t0 = timing ? ms_since(t_entry) : 0.0;
                    if (use_avx512()) scan_bin_avx512<true>(S2, counts[b], sc->J.data());
                    else scan_bin_scalar<true>(S2, counts[b], sc->J.data());
                    const double t1 = timing ? ms_since(t_entry) : 0.0;
                    trace_prefix(sc, counts[b], want[b], ranks_out + offs[(size_t)b]);
                    if (timing)
                        fprintf(stderr, "[rng]   bin %d (%lld): scan %.3f -> %.3f, trace -> %.3f ms\n", b, (long long)counts[b], t0, t1, ms_since(t_entry));
                }
            };
            // the persistent helpers (warm, inside this thread's last-level cache) instead of one fresh thread per bin: a
            // thread created per call lands on a sleeping core somewhere on the socket
            static const bool fresh = getenv("ANNCHOR_RNG_PAR_FRESH") != nullptr;
            if (fresh) {
                std::vector<std::thread> pool;
                for (int t = 1; t < nlive; ++t) pool.emplace_back(work);
                work();
                for (auto &th : pool) th.join();
            } else {
                const std::function<void()> job = work;
                g_helpers->start(std::min(nlive - 1, 6));
                g_helpers->post(&job);
                work();
                g_helpers->finish();
            }
            if (timing)
                fprintf(stderr, "[rng] parallel: count pass %.3f ms, all bins done %.3f ms, words %zu\n", t_count, ms_since(t_entry), S.cur);
            return ANNCHOR_OK;
        }
    }
    // Backward traces: the pooled helper threads take scanned bins from the front while this
    // thread is still scanning; when the scan is done this thread takes bins from the back.  (A
    // thread created per call, let alone per bin, lands on a sleeping core: see HelperPool.)
    enum { PENDING = 0, READY = 1, TAKEN = 2, SKIP = 3 };
    std::unique_ptr<std::atomic<int>[]> state(new std::atomic<int>[(size_t)nbins + 1]);
    for (int b = 0; b < nbins; ++b) state[(size_t)b].store(PENDING, std::memory_order_relaxed);
    auto trace_bin = [&](int b) {
        const double t0 = timing ? ms_since(t_entry) : 0.0;
        trace_prefix(g_scratch[(size_t)b].get(), counts[b], want[b], ranks_out + offs[(size_t)b]);
        if (timing) fprintf(stderr, "[rng]   bin %d (%lld): trace %.3f -> %.3f ms\n", b, (long long)counts[b], t0, ms_since(t_entry));
    };
    const std::function<void()> claim_front = [&] {
        for (int b = 0; b < nbins; ++b) {
            int s;
            while ((s = state[(size_t)b].load(std::memory_order_acquire)) == PENDING) _mm_pause();
            int want_s = READY;
            if (s == READY && state[(size_t)b].compare_exchange_strong(want_s, TAKEN, std::memory_order_acq_rel)) trace_bin(b);
        }
    };
    static const int n_helpers = getenv("ANNCHOR_RNG_HELPERS") ? atoi(getenv("ANNCHOR_RNG_HELPERS")) : 3;
    const bool pooled = nbins > 1 && n_helpers > 0;
    if (pooled) {
        g_helpers->start(n_helpers);
        g_helpers->post(&claim_front);
    }
    for (int b = 0; b < nbins; ++b) {
        const int64_t c = counts[b], k = want[b];
        int64_t *out = ranks_out + offs[(size_t)b];
        if (c < k) {  // utils.py:553-554: the whole bin, no draw
            for (int64_t t = 0; t < c; ++t) out[t] = t;
            state[(size_t)b].store(SKIP, std::memory_order_release);
            continue;
        }
        // forward: swap partners of the shuffle in stream order (sequential across bins: a bin
        // starts where the previous one stopped)
        BinScratch *sc = g_scratch[(size_t)b].get();
        if (sc->J.size() < (size_t)c + 32) sc->J.resize((size_t)c + 32);
        if (use_avx512()) scan_bin_avx512<true>(S, c, sc->J.data());
        else scan_bin_scalar<true>(S, c, sc->J.data());
        state[(size_t)b].store(READY, std::memory_order_release);
    }
    const double t_scan = ms_since(t_entry);
    for (int b = nbins - 1; b >= 0; --b) {
        int want_s = READY;
        if (state[(size_t)b].compare_exchange_strong(want_s, TAKEN, std::memory_order_acq_rel)) trace_bin(b);
    }
    const double t_own = ms_since(t_entry);
    if (pooled) g_helpers->finish();
    if (timing)
        fprintf(stderr, "[rng] scan done %.3f ms, own traces done %.3f ms, helper joined %.3f ms, words %zu, producer %s\n", t_scan,
                t_own, ms_since(t_entry), S.cur, st->producing.load() ? "still running" : "finished");
    return ANNCHOR_OK;
}

// ---- the draw on a persistent worker thread (see include/annchor_hip.h)
namespace {
struct DrawJob {
    uint32_t seed;
    std::vector<int64_t> counts, want, ranks, n_out;
    int rc = 0;
    bool done = false;
};
struct DrawWorker {
    std::mutex mu;
    std::condition_variable cv, done_cv;
    std::vector<DrawJob *> queue;
    std::atomic<int> pending{0};
    bool started = false;
    void submit(DrawJob *j)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!started) {
                started = true;
                std::thread([this] {
                    for (;;) {
                        DrawJob *job;
                        {
                            // poll before blocking, like the trace helpers: the next draw of a running
                            // series of fits is a few milliseconds away
                            const auto t0 = std::chrono::steady_clock::now();
                            while (pending.load(std::memory_order_acquire) == 0 &&
                                   std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() <
                                       HelperPool::spin_ns())
                                _mm_pause();
                            std::unique_lock<std::mutex> lk(mu);
                            cv.wait(lk, [&] { return !queue.empty(); });
                            job = queue.front();
                            queue.erase(queue.begin());
                            pending.fetch_sub(1, std::memory_order_acq_rel);
                        }
                        job->rc = annchor_legacy_choice_ranks(job->seed, job->counts.data(), job->want.data(), (int32_t)job->counts.size(),
                                                              job->ranks.data(), job->n_out.data());
                        {
                            std::lock_guard<std::mutex> lk(mu);
                            job->done = true;
                        }
                        done_cv.notify_all();
                    }
                }).detach();
            }
            queue.push_back(j);
            pending.fetch_add(1, std::memory_order_acq_rel);
        }
        cv.notify_one();
    }
    void wait(DrawJob *j)
    {
        std::unique_lock<std::mutex> lk(mu);
        done_cv.wait(lk, [&] { return j->done; });
    }
};
DrawWorker *g_draw = new DrawWorker();   // leaked on purpose, like the helper pool

// fork(): the child has none of the parent's threads.  Give it fresh (thread-less) pools so that its
// first draw starts its own instead of waiting for workers that do not exist.
struct ForkGuard {
    ForkGuard()
    {
        pthread_atfork(nullptr, nullptr, [] {
            g_helpers = new HelperPool();
            g_draw = new DrawWorker();
        });
    }
} g_fork_guard;
}  // namespace

extern "C" int annchor_legacy_choice_begin(uint32_t seed, const int64_t *counts, const int64_t *want, int32_t nbins, void **ticket)
{
    if (!counts || !want || !ticket || nbins < 0) return ANNCHOR_EINVAL;
    DrawJob *j = new DrawJob();
    j->seed = seed;
    j->counts.assign(counts, counts + nbins);
    j->want.assign(want, want + nbins);
    int64_t tot = 0;
    for (int b = 0; b < nbins; ++b) tot += counts[b] < want[b] ? (counts[b] > 0 ? counts[b] : 0) : (want[b] > 0 ? want[b] : 0);
    j->ranks.resize((size_t)tot + 1);
    j->n_out.resize((size_t)nbins + 1);
    g_draw->submit(j);
    *ticket = j;
    return ANNCHOR_OK;
}

extern "C" int annchor_legacy_choice_end(void *ticket, int64_t *ranks_out, int64_t *n_out)
{
    if (!ticket) return ANNCHOR_EINVAL;
    DrawJob *j = static_cast<DrawJob *>(ticket);
    g_draw->wait(j);
    const int rc = j->rc;
    if (rc == ANNCHOR_OK && ranks_out && n_out) {
        const size_t nb = j->counts.size();
        int64_t tot = 0;
        for (size_t b = 0; b < nb; ++b) { n_out[b] = j->n_out[b]; tot += j->n_out[b]; }
        memcpy(ranks_out, j->ranks.data(), sizeof(int64_t) * (size_t)tot);
    }
    delete j;
    return rc == ANNCHOR_OK && !(ranks_out && n_out) ? ANNCHOR_EINVAL : rc;
}
