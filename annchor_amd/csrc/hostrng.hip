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
//      backwards for those entries alone (bitmap-filtered), one worker thread per bin.
#include <atomic>
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
    // next 624 tempered outputs
    void block(uint32_t *out)
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

// Per-bin scratch kept across calls (grow-only; `bits` all zero and `slot_of` all -1
// between uses) so that a call neither allocates nor first-touches megabytes.
struct BinScratch {
    std::vector<uint32_t> J;
    std::vector<uint64_t> bits;
    std::vector<int32_t> slot_of;
};
std::vector<std::unique_ptr<BinScratch>> g_scratch;
std::mutex g_call_mu;

// Undo the Fisher-Yates swaps (partners J[1..c-1]) for the first k output positions only.
void trace_prefix(BinScratch *sc, int64_t c, int64_t k, int64_t *out)
{
    const uint32_t *J = sc->J.data();
    if (sc->bits.size() < (size_t)(c + 63) / 64) sc->bits.resize((size_t)(c + 63) / 64, 0);
    if (sc->slot_of.size() < (size_t)c) sc->slot_of.resize((size_t)c, -1);
    std::vector<uint64_t> &bits = sc->bits;
    std::vector<int32_t> &slot_of = sc->slot_of;
    std::vector<uint32_t> pos((size_t)k);
    for (int64_t t = 0; t < k; ++t) {
        pos[(size_t)t] = (uint32_t)t;
        slot_of[(size_t)t] = (int32_t)t;
        bits[(size_t)t >> 6] |= 1ull << (t & 63);
    }
    uint64_t *bp = bits.data();
    for (int64_t i = 1; i < c; ++i) {
        const uint32_t j = J[i];
        const bool ti = (bp[i >> 6] >> (i & 63)) & 1ull, tj = (bp[j >> 6] >> (j & 63)) & 1ull;
        if (!(ti | tj) || j == (uint32_t)i) continue;
        const int32_t si = slot_of[(size_t)i], sj = slot_of[j];
        slot_of[(size_t)i] = sj;
        slot_of[j] = si;
        if (si >= 0) pos[(size_t)si] = j;
        if (sj >= 0) pos[(size_t)sj] = (uint32_t)i;
        if (ti != tj) {
            bp[i >> 6] ^= 1ull << (i & 63);
            bp[j >> 6] ^= 1ull << (j & 63);
        }
    }
    for (int64_t t = 0; t < k; ++t) {
        const uint32_t p = pos[(size_t)t];
        out[t] = p;
        slot_of[p] = -1;   // restore the scratch invariants
        bp[p >> 6] = 0;
    }
}
}  // namespace

extern "C" int annchor_legacy_prefetch(uint32_t seed, int64_t ndraws)
{
    if (ndraws <= 0 || ndraws > (1ll << 33)) return ANNCHOR_EINVAL;
    auto s = std::make_shared<Stream>();
    s->start(seed, (size_t)ndraws, true);
    std::lock_guard<std::mutex> lk(g_mu);
    g_streams[seed] = s;   // a previous stream of the same seed is joined and dropped
    return ANNCHOR_OK;
}

extern "C" int annchor_legacy_choice_ranks(uint32_t seed, const int64_t *counts, const int64_t *want, int32_t nbins,
                                           int64_t *ranks_out, int64_t *n_out)
{
    if (!counts || !want || !ranks_out || !n_out || nbins < 0) return ANNCHOR_EINVAL;
    std::shared_ptr<Stream> st;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_streams.find(seed);
        if (it != g_streams.end()) { st = it->second; g_streams.erase(it); }  // a stream is consumed once
    }
    if (!st) {
        st = std::make_shared<Stream>();
        int64_t tot = 0;
        for (int b = 0; b < nbins; ++b) tot += counts[b] >= want[b] ? counts[b] : 0;
        st->start(seed, (size_t)(tot + tot / 2 + 1024), false);
    }
    std::lock_guard<std::mutex> call_lk(g_call_mu);
    while (g_scratch.size() < (size_t)nbins) g_scratch.emplace_back(new BinScratch());
    size_t cur = 0;  // next unread word of the stream
    std::vector<std::thread> workers;
    std::vector<int64_t> offs((size_t)nbins + 1, 0);
    for (int b = 0; b < nbins; ++b) {
        if (counts[b] < 0 || want[b] < 0 || counts[b] >= (1ll << 31)) return ANNCHOR_ELIMIT;
        n_out[b] = counts[b] < want[b] ? counts[b] : want[b];
        offs[(size_t)b + 1] = offs[(size_t)b] + n_out[b];
    }
    for (int b = 0; b < nbins; ++b) {
        const int64_t c = counts[b], k = want[b];
        int64_t *out = ranks_out + offs[(size_t)b];
        if (c < k) {  // utils.py:553-554: the whole bin, no draw
            for (int64_t t = 0; t < c; ++t) out[t] = t;
            continue;
        }
        // forward: swap partners of the shuffle in stream order.  Branch-free rejection, one
        // power-of-two band of i at a time (constant mask inside a band): write the masked
        // draw, advance only when it was accepted (value <= i).  Same draws as NumPy.
        BinScratch *sc = g_scratch[(size_t)b].get();
        std::vector<uint32_t> &J = sc->J;
        if (J.size() < (size_t)c + 1) J.resize((size_t)c + 1);
        for (uint32_t i = c >= 2 ? (uint32_t)(c - 1) : 0u; i >= 1;) {
            const uint32_t mask = 0xffffffffu >> __builtin_clz(i);
            const uint32_t lo = (mask >> 1) + 1;  // smallest i with this mask
            while (i >= lo) {
                size_t avail = st->ready.load(std::memory_order_acquire);
                if (avail <= cur) { st->need(cur + 624); avail = st->ready.load(std::memory_order_acquire); }
                const uint32_t *buf = st->buf.get();
                size_t p = cur;
                // Windows of 8 draws: while i moves by at most 8 inside a window, a draw with
                // value <= i-8 is accepted and one with value > i is rejected whatever the
                // exact i is, so the 8 masks/compares are independent of the running index;
                // the (rare) window holding a value in (i-8, i] falls through to the exact loop.
                while (p + 8 <= avail && i >= lo + 8) {
                    const uint32_t hi_t = i, lo_t = i - 8;
                    uint32_t v[8];
                    uint32_t unsure = 0;
                    for (int t = 0; t < 8; ++t) {
                        v[t] = buf[p + t] & mask;
                        unsure |= (uint32_t)(v[t] > lo_t) & (uint32_t)(v[t] <= hi_t);
                    }
                    if (unsure) break;
                    for (int t = 0; t < 8; ++t) {
                        J[i] = v[t];
                        i -= (v[t] <= lo_t);
                    }
                    p += 8;
                }
                for (int t = 0; t < 8 && p < avail && i >= lo; ++t) {  // exact steps (window with an unsure draw / band edge)
                    const uint32_t v = buf[p++] & mask;
                    J[i] = v;
                    i -= (v <= i);
                }
                cur = p;
            }
        }
        // backward trace of the first k positions on a worker thread
        workers.emplace_back([sc, c, k, out] { trace_prefix(sc, c, k, out); });
    }
    for (auto &w : workers) w.join();
    return ANNCHOR_OK;
}
