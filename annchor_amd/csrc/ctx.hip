// ctx.hip -- context lifecycle, memory, data-set upload, state access, profiling.
#include <cstdarg>
#include <cstdlib>

#include "common.h"
#include <algorithm>
#include <chrono>
#include <mutex>

static std::string g_create_err;
void ann_stream_release(annchor_ctx *c);
void ann_enemies_release(annchor_ctx *c);
void ann_comm_release(annchor_ctx *c);
hipError_t ann_comm_guarded_sync(annchor_ctx *c, hipStream_t stream, const char *where);   // comm.hip

const char *ann_set_err(annchor_ctx *c, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    // (a collective's timeout is the cause of whatever HIP reports after the communicator's abort: keep saying so)
    if (c && c->comm_aborted && c->err.rfind("collective timed out", 0) == 0) return c->err.c_str();
    if (c) c->err = buf; else g_create_err = buf;
    return c ? c->err.c_str() : g_create_err.c_str();
}

// Large device allocations are kept for the next context instead of going back to the driver: a create / fit / close
// cycle at 127 M pairs allocates and frees ~10 buffers of 0.5-1 GB, and every fifth cycle or so one of those hipMalloc calls
// took 1.4 s on this stack (free memory constant; tools/repeat_fit.py) against 2 ms otherwise.  Blocks of 16 MB and more,
// at most pool_cap() bytes of them per process; a request takes the smallest cached block of its device that is large enough
// and not more than a quarter larger.  annchor_release_parked() returns them to the driver; ANNCHOR_NO_CTX_POOL=1 disables it.
namespace {
struct PoolBlock { void *p; size_t bytes; int device; };
std::mutex g_pool_mu;
std::vector<PoolBlock> g_pool;
size_t g_pool_bytes = 0;
constexpr size_t POOL_MIN_BLOCK = (size_t)16 << 20;
// the cap: half of the device's memory (144 GB here; ANNCHOR_CTX_POOL_GB overrides).  One context on a thinned list of 6 x 10^8
// pairs owns ~60 GB: with the first cap of 48 GB part of it went back to the driver at every close, and one fit in ten waited
// 4 s for its allocations (tools/lev100k_profile.py)
size_t pool_cap()
{
    static const size_t cap = [] {
        if (const char *e = getenv("ANNCHOR_CTX_POOL_GB")) return (size_t)atoll(e) << 30;
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return (size_t)48 << 30; }
        return tot / 2;
    }();
    return cap;
}
bool pool_enabled()
{
    static const bool off = getenv("ANNCHOR_NO_CTX_POOL") != nullptr;
    return !off;
}
void *pool_take(int device, size_t want, size_t *got)
{
    if (want < POOL_MIN_BLOCK || !pool_enabled()) return nullptr;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    int best = -1;
    for (int i = 0; i < (int)g_pool.size(); ++i)
        if (g_pool[(size_t)i].device == device && g_pool[(size_t)i].bytes >= want && g_pool[(size_t)i].bytes <= want + want / 4 &&
            (best < 0 || g_pool[(size_t)i].bytes < g_pool[(size_t)best].bytes))
            best = i;
    if (best < 0) return nullptr;
    void *p = g_pool[(size_t)best].p;
    *got = g_pool[(size_t)best].bytes;
    g_pool_bytes -= *got;
    g_pool.erase(g_pool.begin() + best);
    return p;
}
}  // namespace
// hipMalloc through the pool (want rounded by the caller); *got = the block's real size
int ann_dev_alloc(annchor_ctx *c, void **p, size_t want, size_t *got)
{
    *got = want;
    static const bool trace_all = getenv("ANNCHOR_ALLOC_TRACE") && atoi(getenv("ANNCHOR_ALLOC_TRACE")) >= 2;
    if ((*p = pool_take(c->device, want, got)) != nullptr) {
        if (trace_all && want >= POOL_MIN_BLOCK) fprintf(stderr, "  pool hit  %9.1f MB -> block %9.1f MB (pool now %.1f GB)\n", (double)want / 1e6, (double)*got / 1e6, (double)g_pool_bytes / 1e9);
        return ANNCHOR_OK;
    }
    *got = want;
    if (trace_all && want >= POOL_MIN_BLOCK) fprintf(stderr, "  pool MISS %9.1f MB (pool now %.1f GB)\n", (double)want / 1e6, (double)g_pool_bytes / 1e9);
    if (hipMalloc(p, want) == hipSuccess) return ANNCHOR_OK;
    (void)hipGetLastError();
    *p = nullptr;
    {
        // out of memory with blocks parked in the pool: give them back to the driver and ask once more
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (PoolBlock &pb : g_pool) { (void)hipSetDevice(pb.device); (void)hipFree(pb.p); }
        g_pool.clear();
        g_pool_bytes = 0;
        (void)hipSetDevice(c->device);
    }
    ANN_CHECK_HIP(c, hipMalloc(p, want));
    return ANNCHOR_OK;
}

void ann_dev_free(annchor_ctx *c, void *p, size_t bytes)
{
    if (!p) return;
    if (bytes >= POOL_MIN_BLOCK && pool_enabled()) {
        // (the hipFree this replaces waited for the device: kernels queued on this context's stream may still read the block,
        // and the next taker may be another context with another stream)
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (g_pool_bytes + bytes <= pool_cap()) {
            g_pool.push_back({p, bytes, c->device});
            g_pool_bytes += bytes;
            static const bool trace_all = getenv("ANNCHOR_ALLOC_TRACE") && atoi(getenv("ANNCHOR_ALLOC_TRACE")) >= 2;
            if (trace_all) fprintf(stderr, "  to pool   %9.1f MB\n", (double)bytes / 1e6);
            return;
        }
    }
    (void)hipFree(p);
}

int ann_reserve(annchor_ctx *c, DevBuf &b, size_t bytes)
{
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return ANNCHOR_OK;
    size_t want = (bytes + 255) & ~(size_t)255;
    if (c->arena && c->arena_off + want <= c->arena_size) {
        if (b.p && !b.in_arena) ann_dev_free(c, b.p, b.cap);
        b.p = c->arena + c->arena_off;  // a previous (smaller) arena slice is simply abandoned
        c->arena_off += want;
        b.cap = want;
        b.in_arena = true;
        return ANNCHOR_OK;
    }
    static const bool trace = getenv("ANNCHOR_ALLOC_TRACE") != nullptr;   // stderr line per device allocation slower than 2 ms
    const auto t0 = std::chrono::steady_clock::now();
    if (b.p && !b.in_arena) ann_dev_free(c, b.p, b.cap);
    const auto t1 = std::chrono::steady_clock::now();
    b.p = nullptr;
    b.cap = 0;
    b.in_arena = false;
    size_t got = 0;
    ANN_TRY(ann_dev_alloc(c, &b.p, want, &got));
    want = got;
    b.cap = want;
    {
        // every member of the context that ever got an allocation of its own is released by annchor_destroy (a hand-kept list
        // there had fallen 18 buffers behind: ~1.3 GB leaked per context at 127 M pairs, and the device's allocator answered
        // the next contexts' gigabyte requests in 0.3-1.8 s instead of 2 ms)
        const char *lo = reinterpret_cast<const char *>(c), *me = reinterpret_cast<const char *>(&b);
        if (me >= lo && me < lo + sizeof(annchor_ctx) && std::find(c->own_allocs.begin(), c->own_allocs.end(), &b) == c->own_allocs.end())
            c->own_allocs.push_back(&b);
    }
    if (trace) {
        const double f = std::chrono::duration<double, std::milli>(t1 - t0).count();
        const double m = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
        if (f + m > 2.0) fprintf(stderr, "alloc %.1f MB: free %.2f ms, malloc %.2f ms\n", (double)want / 1e6, f, m);
    }
    return ANNCHOR_OK;
}

int ann_arena_init(annchor_ctx *c, int64_t nx)
{
    // pair-list state is ~100 B per candidate pair (worst case: all pairs) + O(nx) rows
    double pairs = 0.5 * (double)nx * (double)(nx - 1);
    if (pairs > 32e6) pairs = 32e6;  // larger problems fall back to per-buffer allocations beyond the slab
    size_t bytes = (size_t)(pairs * 112.0) + (size_t)nx * 4096 + ((size_t)64 << 20);
    if (c->arena) {
        // one slab per context; a slab inherited from a parked context (see annchor_destroy) is kept
        // when it is large enough and nothing has been carved from it yet
        if (c->arena_size >= bytes || c->arena_off != 0) return ANNCHOR_OK;
        ann_dev_free(c, c->arena, c->arena_size);
        c->arena = nullptr;
        c->arena_size = 0;
    }
    void *p = nullptr;
    size_t got = 0;
    if ((p = pool_take(c->device, bytes, &got)) != nullptr) bytes = got;
    else if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return ANNCHOR_OK; }  // optional
    c->arena = (char *)p;
    c->arena_size = bytes;
    c->arena_off = 0;
    return ANNCHOR_OK;
}

// The small pieces of state the kernels expect to find ZERO the first time they run (sticky flags, the selection tables, the tie
// histogram, the classification cursors, the arrival slots of the persistent anchor launch): carved from the slab back to back
// when the data set is bound and cleared by ONE memset there, instead of six first-use memsets inside every first fit of a
// context (~5 us of host time each, most of them in front of a kernel the GPU is waiting for).  Without a slab (or once it is
// full) the first-use memsets remain.
int ann_prewarm_state(annchor_ctx *c)
{
    if (!c->arena || c->state_prewarmed) return ANNCHOR_OK;
    const size_t slot_bytes = sizeof(unsigned long long) * 4 * 1024 + 64;   // lev.hip: ann_lev_anchor_rounds
    struct Piece { DevBuf *b; size_t bytes; };
    Piece pieces[] = {{&c->dev_flags, sizeof(int32_t) * 16}, {&c->lev_cursors, 4 * sizeof(int32_t)}, {&c->sel_state, ann_sel_state_bytes()},
                      {&c->sel2, ann_sel2_table_bytes()},    {&c->tie_hist, ann_tie_hist_bytes()},   {&c->lev_ap, slot_bytes}};
    size_t need = 0;
    for (auto &q : pieces) {
        if (q.b->p) return ANNCHOR_OK;   // (already in use: leave everything to the first-use path)
        need += (q.bytes + 255) & ~(size_t)255;
    }
    if (c->arena_off + need > c->arena_size) return ANNCHOR_OK;
    char *first = c->arena + c->arena_off;
    for (auto &q : pieces) ANN_TRY(ann_reserve(c, *q.b, q.bytes));
    ANN_CHECK_HIP(c, hipMemsetAsync(first, 0, need, c->stream));
    c->dev_flags_clean = true;
    c->lev_cursor_epoch = 0;
    c->gn_err_clean = true;
    c->sel2_clean = c->sel2.p;
    c->tie_hist_clean = c->tie_hist.p;
    c->lev_ap_epoch = 0;
    c->state_prewarmed = true;
    return ANNCHOR_OK;
}

int ann_h2d(annchor_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return ANNCHOR_OK;
    if (c->pin && bytes <= annchor_ctx::PIN_SLOT_BYTES) {
        // small upload: copy into a pinned ring slot, enqueue, return -- the caller's buffer is
        // free again at once and the stream orders the copy before the kernels that follow
        const int sl = c->pin_next;
        c->pin_next = (sl + 1) % annchor_ctx::PIN_SLOTS;
        if (c->pin_busy[sl]) ANN_CHECK_HIP(c, hipEventSynchronize(c->pin_ev[sl]));
        unsigned char *slot = c->pin + (size_t)sl * annchor_ctx::PIN_SLOT_BYTES;
        memcpy(slot, src, bytes);
        ANN_CHECK_HIP(c, hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, c->stream));
        ANN_CHECK_HIP(c, hipEventRecord(c->pin_ev[sl], c->stream));
        c->pin_busy[sl] = true;
        return ANNCHOR_OK;
    }
    ANN_CHECK_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    ANN_CHECK_HIP(c, ann_sync(c, __func__));  // pageable host memory: do not retain the pointer
    return ANNCHOR_OK;
}

// Every host wait of the pair-list form goes through here: ANNCHOR_SYNC_TRACE=1 names them on stderr (tools/wait_census.py).
hipError_t ann_sync(annchor_ctx *c, const char *where)
{
    static const bool trace = getenv("ANNCHOR_SYNC_TRACE") != nullptr;
    static const bool timing = getenv("ANNCHOR_SYNC_TIMING") != nullptr;   // monotonic-clock stamps of every wait (tools/host_timeline.py)
    if (trace) fprintf(stderr, "sync %s\n", where);
    const long long t_in = timing ? ann_now_ns() : 0;
    if (c->idle_gen_n > c->idle_gen_done) {
        // host work parked for exactly this moment: the stream has just been given work and the caller is about to wait for it
        c->idle_gen_done = std::min(c->idle_gen_n, c->idle_gen_done + c->idle_gen_chunk);
        (void)ann_legacy_generate_upto(c->idle_gen_seed, c->idle_gen_n, c->idle_gen_done);
    }
    const long long t_w = timing ? ann_now_ns() : 0;
    const hipError_t rc = c->comm ? ann_comm_guarded_sync(c, c->stream, where) : hipStreamSynchronize(c->stream);
    if (timing) fprintf(stderr, "T wait %s %lld %lld %lld\n", where, t_in, t_w, ann_now_ns());
    if (c->lev_ap_probe_epoch) ann_lev_ap_probe(c);
    return rc;
}

// The legacy sampler's MT19937 stream of `seed` (it depends on the seed only) produced on the calling thread at this context's NEXT
// host waits, `chunk` words per wait (<= 0: all at the first), i.e. after the caller has enqueued whatever it enqueues before it
// must wait: fit() parks the first draw's stream behind the anchor rounds -- annchor_build_locality's kernels are queued before the
// generation starts, and the second piece runs while the GPU emits the pair list and the features.
extern "C" int annchor_legacy_generate_at_next_wait(annchor_ctx *c, uint32_t seed, int64_t ndraws, int64_t chunk)
{
    if (!c || ndraws < 0) return ANNCHOR_EINVAL;
    c->idle_gen_seed = seed;
    c->idle_gen_n = ndraws;
    c->idle_gen_done = 0;
    c->idle_gen_chunk = chunk > 0 ? chunk : ndraws;
    return ANNCHOR_OK;
}

int ann_d2h(annchor_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return ANNCHOR_OK;
    if (c->pin && bytes <= annchor_ctx::PIN_DL_BYTES) {
        // (a pageable destination costs the runtime ~100 us of staging for a few hundred KB: the graph,
        // the sample's feature rows)
        unsigned char *slot = c->pin + (size_t)annchor_ctx::PIN_SLOTS * annchor_ctx::PIN_SLOT_BYTES;
        ANN_CHECK_HIP(c, hipMemcpyAsync(slot, src, bytes, hipMemcpyDeviceToHost, c->stream));
        ANN_CHECK_HIP(c, ann_sync(c, __func__));
        memcpy(dst, slot, bytes);
        return ANNCHOR_OK;
    }
    ANN_CHECK_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    ANN_CHECK_HIP(c, ann_sync(c, __func__));
    return ANNCHOR_OK;
}

// two small downloads, one wait
int ann_d2h2(annchor_ctx *c, void *dst1, const void *src1, size_t bytes1, void *dst2, const void *src2, size_t bytes2)
{
    const size_t off2 = (bytes1 + 63) & ~(size_t)63;
    if (c->pin && off2 + bytes2 <= annchor_ctx::PIN_DL_BYTES) {
        unsigned char *slot = c->pin + (size_t)annchor_ctx::PIN_SLOTS * annchor_ctx::PIN_SLOT_BYTES;
        ANN_CHECK_HIP(c, hipMemcpyAsync(slot, src1, bytes1, hipMemcpyDeviceToHost, c->stream));
        ANN_CHECK_HIP(c, hipMemcpyAsync(slot + off2, src2, bytes2, hipMemcpyDeviceToHost, c->stream));
        ANN_CHECK_HIP(c, ann_sync(c, __func__));
        memcpy(dst1, slot, bytes1);
        memcpy(dst2, slot + off2, bytes2);
        return ANNCHOR_OK;
    }
    ANN_TRY(ann_d2h(c, dst1, src1, bytes1));
    return ann_d2h(c, dst2, src2, bytes2);
}

// a small download the host waits for while MORE work is queued behind it: the copy goes to the pinned slot, an event marks
// its end, `then` enqueues what follows (it must not wait), the host waits for the event only
int ann_d2h_then(annchor_ctx *c, void *dst, const void *src, size_t bytes, int (*then)(annchor_ctx *))
{
    if (!c->pin || bytes > annchor_ctx::PIN_DL_BYTES) {
        ANN_TRY(ann_d2h(c, dst, src, bytes));
        return then(c);
    }
    if (!c->dl_ev) ANN_CHECK_HIP(c, hipEventCreate(&c->dl_ev));
    unsigned char *slot = c->pin + (size_t)annchor_ctx::PIN_SLOTS * annchor_ctx::PIN_SLOT_BYTES;
    ANN_CHECK_HIP(c, hipMemcpyAsync(slot, src, bytes, hipMemcpyDeviceToHost, c->stream));
    ANN_CHECK_HIP(c, hipEventRecord(c->dl_ev, c->stream));
    const int rc = then(c);
    static const bool trace = getenv("ANNCHOR_SYNC_TRACE") != nullptr;
    static const bool timing = getenv("ANNCHOR_SYNC_TIMING") != nullptr;
    if (trace) fprintf(stderr, "sync %s\n", __func__);
    const long long t_w = timing ? ann_now_ns() : 0;
    ANN_CHECK_HIP(c, hipEventSynchronize(c->dl_ev));
    if (timing) fprintf(stderr, "T wait %s %lld %lld %lld\n", __func__, t_w, t_w, ann_now_ns());
    memcpy(dst, slot, bytes);
    // (the probe's copy was queued before this download's: it has landed; a launch queued by `then` is a later one)
    return rc;
}

// ------------------------------------------------------------------ profiling
int ann_prof_entry(annchor_ctx *c, const char *name)
{
    for (size_t i = 0; i < c->prof.size(); ++i)
        if (c->prof[i].name == name || strcmp(c->prof[i].name, name) == 0) return (int)i;
    ProfEntry e;
    e.name = name;
    c->prof.push_back(e);
    return (int)c->prof.size() - 1;
}

static void prof_drain(annchor_ctx *c)
{
    for (auto &p : c->pending) {
        float ms = 0;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess)
            c->prof[p.entry].ms += ms;
        c->ev_pool.push_back(p.a);
        c->ev_pool.push_back(p.b);
    }
    c->pending.clear();
}

ProfScope::ProfScope(annchor_ctx *ctx, const char *name, double alg_bytes) : c(ctx)
{
    if (!c->prof_on) return;
    if (c->prof_mode == 2) {   // metric kernels only
        const size_t n = strlen(name);
        if (n < 6 || strcmp(name + n - 6, "_pairs") != 0) return;
    }
    entry = ann_prof_entry(c, name);
    c->prof[entry].launches += 1;
    c->prof[entry].alg_bytes += alg_bytes;
    if (entry == c->prof_group_entry) { entry = -1; return; }   // timed by the enclosing ProfGroup
    auto get = [&](hipEvent_t *e) {
        if (!c->ev_pool.empty()) { *e = c->ev_pool.back(); c->ev_pool.pop_back(); return true; }
        return hipEventCreate(e) == hipSuccess;
    };
    if (!get(&a) || !get(&b)) { entry = -1; return; }
    (void)hipEventRecord(a, c->stream);
}

ProfGroup::ProfGroup(annchor_ctx *ctx, const char *name) : c(ctx)
{
    if (!c->prof_on || c->prof_group_entry >= 0) return;
    if (c->prof_mode == 2) {   // metric kernels only
        const size_t n = strlen(name);
        if (n < 6 || strcmp(name + n - 6, "_pairs") != 0) return;
    }
    entry = ann_prof_entry(c, name);
    auto get = [&](hipEvent_t *e) {
        if (!c->ev_pool.empty()) { *e = c->ev_pool.back(); c->ev_pool.pop_back(); return true; }
        return hipEventCreate(e) == hipSuccess;
    };
    if (!get(&a) || !get(&b)) { entry = -1; return; }
    (void)hipEventRecord(a, c->stream);
    c->prof_group_entry = entry;
}

ProfGroup::~ProfGroup()
{
    if (entry < 0) return;
    (void)hipEventRecord(b, c->stream);
    c->pending.push_back({entry, a, b});
    c->prof_group_entry = -1;
}

ProfScope::~ProfScope()
{
    if (entry < 0) return;
    (void)hipEventRecord(b, c->stream);
    c->pending.push_back({entry, a, b});
    if (c->pending.size() > 4096) prof_drain(c);
}

extern "C" int annchor_prof_enable(annchor_ctx *c, int32_t on)
{
    if (!c) return ANNCHOR_EINVAL;
    c->prof_on = on != 0;
    c->prof_mode = on;
    return ANNCHOR_OK;
}

extern "C" int annchor_prof_reset(annchor_ctx *c)
{
    if (!c) return ANNCHOR_EINVAL;
    prof_drain(c);
    for (auto &e : c->prof) { e.ms = 0; e.launches = 0; e.alg_bytes = 0; }
    return ANNCHOR_OK;
}

extern "C" int annchor_prof_get(annchor_ctx *c, int32_t max_entries, const char **names, double *ms,
                                int64_t *launches, double *alg_bytes)
{
    if (!c) return ANNCHOR_EINVAL;
    (void)ann_sync(c, __func__);
    prof_drain(c);
    int n = 0;
    for (auto &e : c->prof) {
        if (n >= max_entries) break;
        names[n] = e.name;
        ms[n] = e.ms;
        launches[n] = e.launches;
        alg_bytes[n] = e.alg_bytes;
        ++n;
    }
    return n;
}

// ------------------------------------------------------------------ lifecycle
// What a context needs from the runtime besides its buffers.  A destroyed context parks these for
// the next one on the same device: hipStreamCreate / hipStreamDestroy / hipHostMalloc / hipHostFree
// and the slab's hipMalloc / hipFree add up to ~4 ms per context on this stack
// (tools/lifecycle_time.py), against a 5.5 ms fit.  Parked shells are never freed (process exit).
struct CtxShell {
    int device = 0;
    hipDeviceProp_t prop;
    hipStream_t stream = nullptr;
    unsigned char *pin = nullptr;
    hipEvent_t pin_ev[annchor_ctx::PIN_SLOTS] = {};
    hipEvent_t call_a = nullptr, call_b = nullptr;
    std::vector<hipEvent_t> ev_pool;
    char *arena = nullptr;
    size_t arena_size = 0;
};
static std::mutex g_shell_mu;
static std::vector<CtxShell> g_shells;
static constexpr size_t SHELL_MAX = 4;
static constexpr size_t SHELL_ARENA_MAX = (size_t)2 << 30;

extern "C" int annchor_create(int device, annchor_ctx **out)
{
    if (!out) return ANNCHOR_EINVAL;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        ann_set_err(nullptr, "no HIP device available (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return ANNCHOR_ENODEV;
    }
    if (device < 0 || device >= count) {
        ann_set_err(nullptr, "device %d out of range (have %d)", device, count);
        return ANNCHOR_EINVAL;
    }
    annchor_ctx *c = new annchor_ctx();
    c->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess) {
        ann_set_err(nullptr, "device init failed: %s", hipGetErrorString(e));
        delete c;
        return ANNCHOR_EHIP;
    }
    {
        // a parked shell of this device (stream, pinned staging, events, device slab): creating and
        // destroying those costs ~4 ms per context, more than half a C2 fit
        std::lock_guard<std::mutex> lk(g_shell_mu);
        for (size_t i = 0; i < g_shells.size(); ++i)
            if (g_shells[i].device == device) {
                CtxShell &sh = g_shells[i];
                c->prop = sh.prop; c->stream = sh.stream; c->pin = sh.pin;
                for (int k = 0; k < annchor_ctx::PIN_SLOTS; ++k) c->pin_ev[k] = sh.pin_ev[k];
                c->call_a = sh.call_a; c->call_b = sh.call_b;
                c->ev_pool.swap(sh.ev_pool);
                c->arena = sh.arena; c->arena_size = sh.arena_size; c->arena_off = 0;
                g_shells.erase(g_shells.begin() + (long)i);
                *out = c;
                return ANNCHOR_OK;
            }
    }
    if ((e = hipGetDeviceProperties(&c->prop, device)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
        ann_set_err(nullptr, "device init failed: %s", hipGetErrorString(e));
        delete c;
        return ANNCHOR_EHIP;
    }
    (void)hipEventCreate(&c->call_a);
    (void)hipEventCreate(&c->call_b);
    if (getenv("ANNCHOR_NO_PIN") ||
        hipHostMalloc((void **)&c->pin, annchor_ctx::PIN_SLOTS * annchor_ctx::PIN_SLOT_BYTES + annchor_ctx::PIN_DL_BYTES + annchor_ctx::PIN_TAIL_BYTES, hipHostMallocDefault) != hipSuccess)
        c->pin = nullptr;   // fall back to pageable transfers
    for (int i = 0; i < annchor_ctx::PIN_SLOTS; ++i) (void)hipEventCreateWithFlags(&c->pin_ev[i], hipEventDisableTiming);
    *out = c;
    return ANNCHOR_OK;
}

extern "C" void annchor_destroy(annchor_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)ann_sync(c, __func__);
    prof_drain(c);
    ann_comm_release(c);
    ann_stream_release(c);
    ann_enemies_release(c);
    for (DevBuf *b : c->own_allocs)
        if (b->p && !b->in_arena) { ann_dev_free(c, b->p, b->cap); b->p = nullptr; b->cap = 0; }
    c->own_allocs.clear();
    if (c->dl_ev) { c->ev_pool.push_back(c->dl_ev); c->dl_ev = nullptr; }
    {
        // park the shell for the next context of this device (at most SHELL_MAX of them, slabs up to
        // SHELL_ARENA_MAX; ANNCHOR_NO_CTX_POOL=1 turns the parking off)
        static const bool no_pool = getenv("ANNCHOR_NO_CTX_POOL") != nullptr;
        std::lock_guard<std::mutex> lk(g_shell_mu);
        if (!no_pool && g_shells.size() < SHELL_MAX && c->stream && c->pin) {
            CtxShell sh;
            sh.device = c->device; sh.prop = c->prop; sh.stream = c->stream; sh.pin = c->pin;
            for (int k = 0; k < annchor_ctx::PIN_SLOTS; ++k) sh.pin_ev[k] = c->pin_ev[k];
            sh.call_a = c->call_a; sh.call_b = c->call_b;
            sh.ev_pool.swap(c->ev_pool);
            sh.arena = c->arena; sh.arena_size = c->arena_size;
            if (sh.arena && sh.arena_size > SHELL_ARENA_MAX) { ann_dev_free(c, sh.arena, sh.arena_size); sh.arena = nullptr; sh.arena_size = 0; }
            g_shells.push_back(std::move(sh));
            delete c;
            return;
        }
    }
    if (c->arena) ann_dev_free(c, c->arena, c->arena_size);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    for (int i = 0; i < annchor_ctx::PIN_SLOTS; ++i)
        if (c->pin_ev[i]) (void)hipEventDestroy(c->pin_ev[i]);
    if (c->pin) (void)hipHostFree(c->pin);
    if (c->call_a) (void)hipEventDestroy(c->call_a);
    if (c->call_b) (void)hipEventDestroy(c->call_b);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

// free the parked context shells (device slabs, pinned staging, streams); contexts in use are untouched
extern "C" int annchor_parked_bytes(int device, int64_t *bytes)
{
    if (!bytes) return ANNCHOR_EINVAL;
    size_t tot = 0;
    {
        std::lock_guard<std::mutex> lk(g_shell_mu);
        for (const CtxShell &sh : g_shells) if (sh.device == device && sh.arena) tot += sh.arena_size;
    }
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (const PoolBlock &pb : g_pool) if (pb.device == device) tot += pb.bytes;
    }
    *bytes = (int64_t)tot;
    return ANNCHOR_OK;
}

extern "C" int annchor_release_parked(void)
{
    std::vector<CtxShell> shells;
    {
        std::lock_guard<std::mutex> lk(g_shell_mu);
        shells.swap(g_shells);
    }
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (PoolBlock &pb : g_pool) { (void)hipSetDevice(pb.device); (void)hipFree(pb.p); }
        g_pool.clear();
        g_pool_bytes = 0;
    }
    for (CtxShell &sh : shells) {
        (void)hipSetDevice(sh.device);
        if (sh.arena) (void)hipFree(sh.arena);
        for (hipEvent_t e : sh.ev_pool) (void)hipEventDestroy(e);
        for (int i = 0; i < annchor_ctx::PIN_SLOTS; ++i)
            if (sh.pin_ev[i]) (void)hipEventDestroy(sh.pin_ev[i]);
        if (sh.pin) (void)hipHostFree(sh.pin);
        if (sh.call_a) (void)hipEventDestroy(sh.call_a);
        if (sh.call_b) (void)hipEventDestroy(sh.call_b);
        if (sh.stream) (void)hipStreamDestroy(sh.stream);
    }
    return (int)shells.size();
}

// PCI bus id of a device ("0000:c1:00.0"): lets a launcher bind its process to the CPUs of the
// GPU's NUMA node (/sys/bus/pci/devices/<id>/numa_node) before it creates contexts
extern "C" int annchor_device_pci_bus_id(int device, char *buf, int buflen)
{
    if (!buf || buflen < 16) return ANNCHOR_EINVAL;
    return hipDeviceGetPCIBusId(buf, buflen, device) == hipSuccess ? ANNCHOR_OK : ANNCHOR_EHIP;
}

// free / total device memory in bytes (the host sizes the pair-list form from it: ~130 B per candidate pair)
extern "C" int annchor_device_mem_info(int device, int64_t *free_bytes, int64_t *total_bytes)
{
    if (!free_bytes || !total_bytes) return ANNCHOR_EINVAL;
    size_t f = 0, t = 0;
    if (hipSetDevice(device) != hipSuccess || hipMemGetInfo(&f, &t) != hipSuccess) { (void)hipGetLastError(); return ANNCHOR_EHIP; }
    *free_bytes = (int64_t)f;
    *total_bytes = (int64_t)t;
    return ANNCHOR_OK;
}

extern "C" const char *annchor_last_error(annchor_ctx *c) { return c ? c->err.c_str() : g_create_err.c_str(); }
extern "C" const char *annchor_create_error(void) { return g_create_err.c_str(); }

extern "C" int annchor_device_name(annchor_ctx *c, char *buf, int buflen)
{
    if (!c || !buf || buflen <= 0) return ANNCHOR_EINVAL;
    snprintf(buf, (size_t)buflen, "%s (%s, %d CUs)", c->prop.name, c->prop.gcnArchName, c->prop.multiProcessorCount);
    return ANNCHOR_OK;
}

extern "C" int annchor_synchronize(annchor_ctx *c)
{
    if (!c) return ANNCHOR_EINVAL;
    ANN_CHECK_HIP(c, ann_sync(c, __func__));
    return ANNCHOR_OK;
}

extern "C" int annchor_last_kernel_ms(annchor_ctx *c, float *ms)
{
    if (!c || !ms) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->call_timed, ANNCHOR_EINVAL, "no timed call yet");
    ANN_CHECK_HIP(c, hipEventSynchronize(c->call_b));
    ANN_CHECK_HIP(c, hipEventElapsedTime(ms, c->call_a, c->call_b));
    return ANNCHOR_OK;
}

// ------------------------------------------------------------------- data set
static void reset_pipeline(annchor_ctx *c)
{
    c->na = c->nA = 0;
    c->n = 0; c->have_bitmap = false;
    c->have_features = c->have_RA = false; c->sel_prepared = false;
    c->nsamp = c->ncand = c->nnext = 0;
}

extern "C" int annchor_set_opaque(annchor_ctx *c, int64_t nx)
{
    if (!c) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, nx > 1 && nx < (1ll << 31), ANNCHOR_ELIMIT, "nx=%lld out of range", (long long)nx);
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    ANN_TRY(ann_arena_init(c, nx));
    ANN_TRY(ann_prewarm_state(c));
    c->metric = ANNCHOR_METRIC_NONE;
    c->nx = nx;
    reset_pipeline(c);
    return ANNCHOR_OK;
}

// Strings over more than 256 distinct symbols: 16-bit dense codes 0 .. alphabet - 1, alphabet <= 65 535 (code 0xffff is the
// kernel's "no symbol").  Evaluated by k_lev_w (lev.hip): match words computed per column instead of looked up.
extern "C" int annchor_set_strings_u16(annchor_ctx *c, const uint16_t *symbols, const int64_t *offs, const int32_t *lens, int64_t nx,
                                       int32_t alphabet)
{
    if (!c || !symbols || !offs || !lens) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, nx > 1 && nx < (1ll << 31), ANNCHOR_ELIMIT, "nx=%lld out of range", (long long)nx);
    ANN_REQUIRE(c, alphabet >= 1 && alphabet <= 65535, ANNCHOR_ELIMIT, "alphabet=%d not in 1..65535", alphabet);
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    std::vector<int32_t> o((size_t)nx);
    size_t total = 0;
    int maxlen = 0;
    for (int64_t s = 0; s < nx; ++s) {
        ANN_REQUIRE(c, lens[s] >= 0, ANNCHOR_EINVAL, "negative length at %lld", (long long)s);
        o[(size_t)s] = (int32_t)total;
        total += ((size_t)lens[s] + 7) & ~(size_t)7;      // 16-byte aligned starts
        if (lens[s] > maxlen) maxlen = lens[s];
        ANN_REQUIRE(c, total < (1ull << 30), ANNCHOR_ELIMIT, "string pool exceeds 2 GiB");
    }
    ANN_REQUIRE(c, maxlen <= 2048, ANNCHOR_ELIMIT, "string length %d exceeds the supported 2048 (16-bit symbols)", maxlen);
    std::vector<uint16_t> pool(total + 64, 0xffffu);
    for (int64_t s = 0; s < nx; ++s) {
        for (int32_t k = 0; k < lens[s]; ++k)
            ANN_REQUIRE(c, symbols[offs[s] + k] < alphabet, ANNCHOR_EINVAL, "symbol code out of range in string %lld", (long long)s);
        memcpy(pool.data() + o[(size_t)s], symbols + offs[s], sizeof(uint16_t) * (size_t)lens[s]);
    }
    ANN_TRY(ann_arena_init(c, nx));
    ANN_TRY(ann_prewarm_state(c));
    ANN_TRY(ann_reserve(c, c->sym, pool.size() * sizeof(uint16_t)));
    ANN_TRY(ann_reserve(c, c->soff, sizeof(int32_t) * (size_t)nx));
    ANN_TRY(ann_reserve(c, c->slen, sizeof(int32_t) * (size_t)nx));
    ANN_TRY(ann_h2d(c, c->sym.p, pool.data(), pool.size() * sizeof(uint16_t)));
    ANN_TRY(ann_h2d(c, c->soff.p, o.data(), sizeof(int32_t) * (size_t)nx));
    ANN_TRY(ann_h2d(c, c->slen.p, lens, sizeof(int32_t) * (size_t)nx));
    c->metric = ANNCHOR_METRIC_LEVENSHTEIN;
    c->nx = nx;
    c->alphabet = alphabet;
    c->maxlen = maxlen;
    c->sym_wide = true;
    c->lev_gl0 = 0; c->lev_frac0 = 0.0; c->lev_nshort = 0;
    reset_pipeline(c);
    return ANNCHOR_OK;
}

extern "C" int annchor_set_strings(annchor_ctx *c, const uint8_t *symbols, const int64_t *offs,
                                   const int32_t *lens, int64_t nx, int32_t alphabet)
{
    if (!c || !symbols || !offs || !lens) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, nx > 1 && nx < (1ll << 31), ANNCHOR_ELIMIT, "nx=%lld out of range", (long long)nx);
    ANN_REQUIRE(c, alphabet >= 1 && alphabet <= 256, ANNCHOR_ELIMIT, "alphabet=%d not in 1..256", alphabet);
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    // repack with 16-byte aligned starts so that lanes can use wide loads
    std::vector<int32_t> o((size_t)nx);
    size_t total = 0;
    int maxlen = 0;
    for (int64_t s = 0; s < nx; ++s) {
        ANN_REQUIRE(c, lens[s] >= 0, ANNCHOR_EINVAL, "negative length at %lld", (long long)s);
        o[(size_t)s] = (int32_t)total;
        total += ((size_t)lens[s] + 15) & ~(size_t)15;
        if (lens[s] > maxlen) maxlen = lens[s];
        ANN_REQUIRE(c, total < (1ull << 31), ANNCHOR_ELIMIT, "string pool exceeds 2 GiB");
    }
    ANN_REQUIRE(c, maxlen <= 32 * 64 * 8, ANNCHOR_ELIMIT, "string length %d exceeds the supported 16384", maxlen);
    std::vector<uint8_t> pool(total + 16, 0);
    for (int64_t s = 0; s < nx; ++s) {
        for (int32_t k = 0; k < lens[s]; ++k)
            ANN_REQUIRE(c, symbols[offs[s] + k] < alphabet, ANNCHOR_EINVAL, "symbol code out of range in string %lld",
                        (long long)s);
        memcpy(pool.data() + o[(size_t)s], symbols + offs[s], (size_t)lens[s]);
    }
    ANN_TRY(ann_arena_init(c, nx));
    ANN_TRY(ann_prewarm_state(c));
    ANN_TRY(ann_reserve(c, c->sym, pool.size()));
    ANN_TRY(ann_reserve(c, c->soff, sizeof(int32_t) * (size_t)nx));
    ANN_TRY(ann_reserve(c, c->slen, sizeof(int32_t) * (size_t)nx));
    ANN_TRY(ann_h2d(c, c->sym.p, pool.data(), pool.size()));
    ANN_TRY(ann_h2d(c, c->soff.p, o.data(), sizeof(int32_t) * (size_t)nx));
    ANN_TRY(ann_h2d(c, c->slen.p, lens, sizeof(int32_t) * (size_t)nx));
    c->metric = ANNCHOR_METRIC_LEVENSHTEIN;
    c->nx = nx;
    c->alphabet = alphabet;
    c->maxlen = maxlen;
    c->sym_wide = false;
    {   // anchor rounds (lev.hip, k_lev_a2): strings of <= 16 words first
        std::vector<int32_t> ord((size_t)nx);
        int64_t ns = 0;
        for (int64_t s = 0; s < nx; ++s) ns += lens[s] <= 512;
        int64_t a = 0, b = ns;
        for (int64_t s = 0; s < nx; ++s) ord[(size_t)(lens[s] <= 512 ? a++ : b++)] = (int32_t)s;
        ANN_TRY(ann_reserve(c, c->lev_order, sizeof(int32_t) * (size_t)nx));
        ANN_TRY(ann_h2d(c, c->lev_order.p, ord.data(), sizeof(int32_t) * (size_t)nx));
        c->lev_nshort = (int)ns;
    }
    {   // slot classes of the Levenshtein kernel (lev.hip, k_lev_f): P pairs per wave for the longest string,
        // P + 1 for pairs whose shorter string has <= 64 / (P + 1) words
        const int W = (maxlen + 31) / 32 > 0 ? (maxlen + 31) / 32 : 1;
        c->lev_gl0 = 0;
        c->lev_frac0 = 0.0;
        if (W <= 32) {
            const int gl0 = 64 / (64 / W + 1);
            if (gl0 >= 1) {
                int64_t cnt = 0;
                for (int64_t s = 0; s < nx; ++s) cnt += (lens[s] + 31) / 32 <= gl0;
                c->lev_gl0 = gl0;
                c->lev_frac0 = (double)cnt / (double)nx;
            }
        }
    }
    reset_pipeline(c);
    return ANNCHOR_OK;
}

static int set_points(annchor_ctx *c, const void *X, int64_t nx, int32_t dim, size_t esz, int metric)
{
    if (!c || !X) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, nx > 1 && nx < (1ll << 31), ANNCHOR_ELIMIT, "nx=%lld out of range", (long long)nx);
    ANN_REQUIRE(c, dim >= 1 && dim <= 65536, ANNCHOR_ELIMIT, "dim=%d out of range", dim);
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    size_t bytes = esz * (size_t)nx * (size_t)dim;
    ANN_TRY(ann_arena_init(c, nx));
    ANN_TRY(ann_prewarm_state(c));
    ANN_TRY(ann_reserve(c, c->pts, bytes));
    ANN_TRY(ann_h2d(c, c->pts.p, X, bytes));
    c->metric = metric;
    c->nx = nx;
    c->dim = dim;
    reset_pipeline(c);
    return ANNCHOR_OK;
}

extern "C" int annchor_set_points_f32(annchor_ctx *c, const float *X, int64_t nx, int32_t dim)
{
    return set_points(c, X, nx, dim, sizeof(float), ANNCHOR_METRIC_EUCLIDEAN_F32);
}

extern "C" int annchor_set_points_f64(annchor_ctx *c, const double *X, int64_t nx, int32_t dim)
{
    return set_points(c, X, nx, dim, sizeof(double), ANNCHOR_METRIC_EUCLIDEAN_F64);
}

extern "C" int annchor_set_points_cosine_f32(annchor_ctx *c, const float *X, int64_t nx, int32_t dim)
{
    return set_points(c, X, nx, dim, sizeof(float), ANNCHOR_METRIC_COSINE_F32);
}

extern "C" int annchor_set_points_cosine_f64(annchor_ctx *c, const double *X, int64_t nx, int32_t dim)
{
    return set_points(c, X, nx, dim, sizeof(double), ANNCHOR_METRIC_COSINE_F64);
}

extern "C" int annchor_set_histograms(annchor_ctx *c, const double *hist, int64_t nx, int32_t nbins,
                                      const double *cost)
{
    if (!c || !hist || !cost) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, nx > 1 && nx < (1ll << 31), ANNCHOR_ELIMIT, "nx=%lld out of range", (long long)nx);
    ANN_REQUIRE(c, nbins >= 1 && nbins <= 1024, ANNCHOR_ELIMIT, "nbins=%d: this build supports 1..1024 bins", nbins);
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    int maxs = 0;
    bool integral = true;
    double maxsum = 0, maxval = 0;
    for (int64_t s = 0; s < nx; ++s) {
        int k = 0;
        double sum = 0;
        for (int b = 0; b < nbins; ++b) {
            const double v = hist[s * nbins + b];
            if (v > maxval) maxval = v;
            k += v != 0;
            sum += v;
            if (v < 0 || v != (double)(int64_t)v) integral = false;
        }
        if (sum > maxsum) maxsum = sum;
        ANN_REQUIRE(c, k > 0, ANNCHOR_EINVAL, "histogram %lld is empty", (long long)s);
        if (k > maxs) maxs = k;
    }
    // more than 64 bins: sparse histograms only (at most 32 non-zero entries each, kept as (bin, mass) lists: the two supports of a
    // solve are then at most 64 nodes, one lane each); the metric test below must hold as well
    ANN_REQUIRE(c, nbins <= 64 || maxs <= 32, ANNCHOR_ELIMIT,
                "histograms of %d bins: at most 32 non-zero entries each are supported beyond 64 bins (largest support here: %d)", nbins, maxs);
    ANN_TRY(ann_arena_init(c, nx));
    ANN_TRY(ann_prewarm_state(c));
    ANN_TRY(ann_reserve(c, c->cost, sizeof(double) * (size_t)nbins * nbins));
    if (nbins <= 64) {
        ANN_TRY(ann_reserve(c, c->hist, sizeof(double) * (size_t)nx * nbins));
        ANN_TRY(ann_h2d(c, c->hist.p, hist, sizeof(double) * (size_t)nx * nbins));
    } else {
        std::vector<int32_t> hb((size_t)nx * 32, 0), hc((size_t)nx, 0);
        std::vector<double> hv((size_t)nx * 32, 0.0);
        for (int64_t s = 0; s < nx; ++s) {
            int k = 0;
            for (int b = 0; b < nbins; ++b) {
                const double v = hist[s * nbins + b];
                if (v != 0) { hb[(size_t)s * 32 + k] = b; hv[(size_t)s * 32 + k] = v; ++k; }
            }
            hc[(size_t)s] = k;
        }
        ANN_TRY(ann_reserve(c, c->hs_bin, sizeof(int32_t) * hb.size()));
        ANN_TRY(ann_reserve(c, c->hs_val, sizeof(double) * hv.size()));
        ANN_TRY(ann_reserve(c, c->hs_cnt, sizeof(int32_t) * hc.size()));
        ANN_TRY(ann_h2d(c, c->hs_bin.p, hb.data(), sizeof(int32_t) * hb.size()));
        ANN_TRY(ann_h2d(c, c->hs_val.p, hv.data(), sizeof(double) * hv.size()));
        ANN_TRY(ann_h2d(c, c->hs_cnt.p, hc.data(), sizeof(int32_t) * hc.size()));
    }
    ANN_TRY(ann_h2d(c, c->cost.p, cost, sizeof(double) * (size_t)nbins * nbins));
    c->metric = ANNCHOR_METRIC_WASSERSTEIN;
    c->nx = nx;
    c->nbins = nbins;
    c->max_support = maxs;
    c->hist_integral = integral && maxsum * maxsum < 2147483647.0;
    c->hist_fits_i16 = c->hist_integral && maxval * maxsum < 32767.0;   // a flow is at most (a mass) x (the other histogram's sum)
    // Is the ground cost a metric on the bins (zero diagonal, triangle inequality)?  Then mass two histograms hold on the same
    // bin stays where it is in some optimal plan, and the solver works on the two differences only (csrc/emd.hip).  The
    // inequality is tested with a relative slack of 2^-40: Euclidean costs computed in floating point miss it by an ulp on
    // collinear bins, and a violation of eps changes the optimum by at most eps x the moved mass.
    bool metric_cost = true;
    for (int i = 0; i < nbins && metric_cost; ++i) {
        if (cost[i * nbins + i] != 0.0) metric_cost = false;
        for (int j = 0; j < nbins && metric_cost; ++j) {
            const double cij = cost[i * nbins + j];
            if (!(cij >= 0.0)) { metric_cost = false; break; }
            for (int k = 0; k < nbins; ++k) {
                const double via = cost[i * nbins + k] + cost[k * nbins + j];
                if (cij > via + 9.094947017729282e-13 * via) { metric_cost = false; break; }
            }
        }
    }
    ANN_REQUIRE(c, nbins <= 64 || metric_cost, ANNCHOR_ELIMIT,
                "histograms of %d bins need a metric ground cost (zero diagonal, triangle inequality): the solver for other costs takes up to 64 bins", nbins);
    c->cost_is_metric = metric_cost;
    c->cost_max = 0.0;
    for (int i = 0; i < nbins * nbins; ++i) c->cost_max = std::max(c->cost_max, fabs(cost[i]));
    reset_pipeline(c);
    return ANNCHOR_OK;
}

// ----------------------------------------------------------- metric boundary
int ann_metric_launch(annchor_ctx *c, const PairSource &src, double *d_out, double *d_RA, uint8_t *d_ncm)
{
    switch (c->metric) {
    case ANNCHOR_METRIC_LEVENSHTEIN: return ann_lev_launch(c, src, d_out, d_RA, d_ncm);
    case ANNCHOR_METRIC_EUCLIDEAN_F32:
    case ANNCHOR_METRIC_EUCLIDEAN_F64:
    case ANNCHOR_METRIC_COSINE_F32:
    case ANNCHOR_METRIC_COSINE_F64: return ann_euclid_launch(c, src, d_out, d_RA, d_ncm);
    case ANNCHOR_METRIC_WASSERSTEIN: return ann_emd_launch(c, src, d_out, d_RA, d_ncm);
    default: ann_set_err(c, "no device metric bound to this context"); return ANNCHOR_EINVAL;
    }
}

__global__ void k_ij64_to_int2(const int64_t *__restrict__ ij, int2 *__restrict__ out, int64_t n)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = make_int2((int)ij[2 * t], (int)ij[2 * t + 1]);
}

extern "C" int annchor_metric_pairs(annchor_ctx *c, const int64_t *ij, int64_t n, double *out)
{
    if (!c || (n > 0 && (!ij || !out))) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->metric != ANNCHOR_METRIC_NONE, ANNCHOR_EINVAL, "no device metric bound to this context");
    ANN_REQUIRE(c, n >= 0 && n < (1ll << 31), ANNCHOR_ELIMIT, "n=%lld out of range", (long long)n);
    if (n == 0) return ANNCHOR_OK;
    for (int64_t t = 0; t < 2 * n; ++t)
        ANN_REQUIRE(c, ij[t] >= 0 && ij[t] < c->nx, ANNCHOR_EINVAL, "pair index %lld out of range", (long long)ij[t]);
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    ANN_TRY(ann_reserve(c, c->stage_in, sizeof(int64_t) * 2 * (size_t)n));
    ANN_TRY(ann_reserve(c, c->tmp0, sizeof(int2) * (size_t)n));
    ANN_TRY(ann_reserve(c, c->stage_out, sizeof(double) * (size_t)n));
    ANN_TRY(ann_h2d(c, c->stage_in.p, ij, sizeof(int64_t) * 2 * (size_t)n));
    k_ij64_to_int2<<<ann_blocks(n, 256), 256, 0, c->stream>>>(c->stage_in.as<int64_t>(), c->tmp0.as<int2>(), n);
    PairSource src;
    src.ij = c->tmp0.as<int2>();
    src.n = n;
    ANN_CHECK_HIP(c, hipEventRecord(c->call_a, c->stream));
    ANN_TRY(ann_metric_launch(c, src, c->stage_out.as<double>(), nullptr, nullptr));
    ANN_CHECK_HIP(c, hipEventRecord(c->call_b, c->stream));
    c->call_timed = true;
    ANN_CHECK_HIP(c, hipGetLastError());
    return ann_d2h(c, out, c->stage_out.p, sizeof(double) * (size_t)n);
}
