// common.h -- shared declarations for libannchor_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <chrono>

#include "../../include/annchor_hip.h"

#define ANN_WAVE 64

// grow-only device buffer
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool in_arena = false;  // carved from the context arena (not individually freed)
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct ProfEntry {
    const char *name;
    double ms = 0;
    int64_t launches = 0;
    double alg_bytes = 0;
};

struct PendingEvent {
    int entry;
    hipEvent_t a, b;
};

struct annchor_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    hipDeviceProp_t prop;

    // ---- one slab per context, sized from nx when the data set is bound, from which the
    // pipeline buffers are carved: fit() then runs without hipMalloc/hipFree calls
    char *arena = nullptr;
    size_t arena_size = 0, arena_off = 0;

    // ---- data set
    int metric = ANNCHOR_METRIC_NONE;
    int64_t nx = 0;
    DevBuf sym, soff, slen;  // strings: symbols (uint8, 16B-aligned starts), int32 offsets, int32 lens
    int alphabet = 0, maxlen = 0;
    bool sym_wide = false;   // symbols are 16-bit codes (annchor_set_strings_u16: alphabets of 257..65535 symbols); soff counts symbols
    // Levenshtein slot packing: patterns of <= lev_gl0 words fit one more pair per wave than the data
    // set's longest string allows (0: no such class); lev_frac0 = share of the strings that short
    int lev_gl0 = 0;
    double lev_frac0 = 0.0;
    DevBuf lev_order;        // int32 [nx]: string ids, strings of <= 16 words first (k_lev_a2: two such pairs share a wave)
    int lev_nshort = 0;
    DevBuf lev_ap;           // k_lev_ap (all anchor rounds in one launch): u64 [2][1024] arrival slots, then the abort word
    DevBuf lev_ap_min;       // uint16 [nx]: running minima of the rescue form
    uint32_t lev_ap_epoch = 0;
    uint32_t lev_ap_probe_epoch = 0;   // a copy of the abort word is on its way to the pinned tail: looked at after the next host wait (lev.hip)
    DevBuf lev_cursors;      // int32 [2][2]: class counters of k_lev_classify, two slots used in turn (a call zeroes the NEXT call's slot)
    int lev_cursor_epoch = 0;
    DevBuf lev_perm;         // int32 [n] pair positions, short patterns first / long ones from the back; + 2 counters
    DevBuf pts;              // points (f32 or f64) row-major [nx, dim]
    int dim = 0;
    DevBuf hist, cost, supp; // histograms f64 [nx, nbins] (nbins <= 64), cost [nbins, nbins], the exact-OT kernels' flags / counters
    DevBuf hs_bin, hs_val, hs_cnt;   // nbins > 64: the non-zero entries of every histogram, int32 [nx][32] bins (ascending), f64 [nx][32] masses, int32 [nx]
    int nbins = 0, max_support = 0;
    double cost_max = 0.0;       // largest ground cost
    int emd_epoch = 0;           // launches of the exact-OT kernels (their two work counters take turns)
    bool cost_is_metric = false; // ground cost: zero diagonal + triangle inequality (common mass of two histograms cancels)
    bool hist_integral = false;  // all masses integer valued and (row sum)^2 < 2^31: exact int32 flows
    bool hist_fits_i16 = false;  // ... and (largest mass) x (largest row sum) < 2^15: every flow fits int16

    // ---- the fitted model + residual lists downloaded with the graph (model.hip: ann_model_prefetch_*): what
    // annchor_model_download_with_errors returns without another wait
    std::vector<unsigned char> model_cache;
    bool model_cache_valid = false;
    size_t model_cache_errs = 0;   // bytes of residual lists in the cache

    // ---- the refinement launch parked behind the next sampling step's statistics (annchor_park_refine)
    bool park_refine = false;
    hipEvent_t dl_ev = nullptr;   // marks the end of a download the host waits for while later work is already queued

    // ---- host work parked for this context's next host wait (annchor_legacy_generate_at_next_wait)
    uint32_t idle_gen_seed = 0;
    int64_t idle_gen_n = 0, idle_gen_done = 0, idle_gen_chunk = 0;

    // ---- in-library RCCL communicator (comm.hip): ncclComm_t, set by annchor_comm_init
    void *comm = nullptr;
    int comm_world = 1, comm_rank = 0;
    // a second communicator on its own stream for ONE large all-gather that overlaps the engine stream's work (the raw rows of the
    // row-sharded build beside the anchor rounds and the k-d order); comm_side_pending: the engine stream has not waited for it yet
    void *comm_side = nullptr;
    hipStream_t comm_side_stream = nullptr;
    hipEvent_t comm_side_ev = nullptr, comm_main_ev = nullptr;
    bool comm_side_pending = false;
    // dead-peer guard: host waits of a context with a communicator run under a watchdog (comm.hip) that aborts the communicators
    // when a wait lasts longer than comm_timeout_s (a peer that never enters a collective leaves RCCL's kernel spinning forever)
    double comm_timeout_s = 300.0;
    bool comm_aborted = false;
    void *comm_watch = nullptr;

    // ---- anchors
    int na = 0, nA = 0;
    DevBuf Dt;          // double [na][nx]   (anchor-major: lane-coalesced over points)
    DevBuf A;           // int32 [na]
    DevBuf anchorRank;  // int32 [nx]  rank of the LAST occurrence in A, -1 if not an anchor
    DevBuf runmin, redval, redidx;

    // ---- locality
    DevBuf sid, cA, thr;          // uint64 [nx][sid_nw], int32 [nx], int32 [nx]
    int sid_nw = 1;               // 64-bit words per anchor mask (ann_sid_words(na))
    DevBuf Kbits, Kpref;          // uint64 [nx][kw], uint32 [nx][kw]
    DevBuf deg, low, rowstart;    // int32 [nx], int32 [nx], int64 [nx+1]
    DevBuf Iptr, Iidx;            // int64 [nx+1], int32 [2n]
    int64_t n = 0;                // number of candidate pairs
    DevBuf ij;                    // int2 [n]
    bool have_bitmap = false;     // Kbits / Kpref / low / rowstart describe the current pair list
    // column-ordered copy of the "column-like" half of the rows: T[colbase(i) + s] = RA[pair (j_s, i)],
    // j_s < i ascending, colbase(i) = Iptr[i] - rowstart[i]; rebuilt by ann_transpose_columns before a
    // row kernel runs over a large pair list (rowsel.h)
    DevBuf colT, colM;            // double [n], uint8 [n]

    // ---- per-pair state
    DevBuf lb, ub, dad, RA, prob;  // double [n]
    DevBuf anc, ncm, label;        // uint8 [n]
    bool have_features = false, have_RA = false;
    int64_t n_unc = -1;            // cached count of not-computed pairs (-1 = unknown: recount)
    int64_t n_unc_after_features = -1;   // ... what it will be once compute_features has marked the anchor pairs (from build_locality)

    // ---- samples
    DevBuf spos, sy;  // int32 [m], double [m]
    DevBuf draw_J, draw_next, draw_q1;   // the legacy draw's swap partners (uint32, stream order per bin) and the trace's next[] (features.hip)
    DevBuf sfeat, spred;   // double [m][4] feature rows / double [m] unclipped predictions of the samples (device-resident model fit)
    // ---- device-resident model of an iteration (model.hip): per-partition OLS coefficients, residual lists
    DevBuf model;          // DeviceModel
    DevBuf ols_scratch;    // double [nb][4][m]: a partition's centred design matrix and targets
    std::vector<DevBuf *> own_allocs;   // members of this context that ann_reserve gave an allocation of their own (freed by annchor_destroy)
    int prof_group_entry = -1;   // >= 0: a ProfGroup of that family is open (its inner scopes record no events)
    bool model_fitted = false;   // `model` holds this iteration's regression
    bool errs_on_device = false; // `errs` / `errptr` hold this iteration's sorted residuals (annchor_fit_errors_device)
    int model_nb = 0;
    DevBuf dev_flags;      // int32 [16] sticky error flags raised by kernels, read with the selection stage's final state:
                           // [0] sample step: a (bin, rank) entry did not exist
    bool dev_flags_clean = false;
    bool gn_err_clean = false;    // the sweep's error flag is known to be zero
    bool state_prewarmed = false; // ann_prewarm_state carved and cleared the small zero-initialised state
    DevBuf sstats;         // SamplerStats: quantiles, bin edges and bin counts of a sampling step (annchor_sampler_stats)
    DevBuf hs_key, hs_pos, hs_misc;   // hashed stratified sampling: per-partition candidate lists, counters / outputs
    int64_t nsamp = 0;

    // ---- selection
    DevBuf thresh;               // double [nx]
    DevBuf cand, next;           // int32 lists
    int64_t ncand = 0, nnext = 0;
    bool cand_marked = false;      // not_computed_mask already cleared for the current candidates
    DevBuf gl_val, gl_pos, gl_cnt, gl_ncomp, marked, markcount;  // guarantee_nmin scratch
    DevBuf gn_state;             // guarantee_nmin rounds: mark masks (2), out-of-list mark counts (3), change flags
    // selection stage split in two (annchor_select_prepare): thresholds + guarantee_nmin launched ahead, while the host
    // fits the error model; reset by everything that changes RefineApprox / the mask
    bool sel_prepared = false;
    int sel_k = 0, sel_nmin = 0;
    bool gn_pending = false;     // rounds launched, convergence not yet looked at
    int gn_round = 0, gn_L = 0;
    DevBuf sel_hist, sel_state, blk_cnt, blk_off;                // radix select / compaction scratch (sel_state: the cut state + the sweep's error flag)
    DevBuf sel_pass_state;                                       // the byte-pass selection's own state (scan.hip)
    DevBuf tie_lists, tie_hist;                                  // scrambled positions of the pairs on the two probability cuts; their histogram
    const void *tie_hist_clean = nullptr;                        // tie_hist known to be zero at this address
    DevBuf sel2, sel_bufA, sel_bufB, sel_seg;                             // filter-then-finish selection: tables, candidates
    const void *sel2_clean = nullptr;                            // sel2 tables known to be zero at this address
    DevBuf errs, errptr;
    DevBuf ecdf_index;           // per-label bucket index over the sorted errors (long lists)
    DevBuf cptr, cidx, cval;     // computed-neighbour CSR for update_bounds
    DevBuf c16, cbnd;            // its 2-byte key copy + per-list position of the first key >= 65 536 (k_update_bounds_bits<true>)
    DevBuf tmp0, tmp1, tmp2, tmp3;
    DevBuf scan_tmp;

    // ---- staging
    DevBuf stage_in, stage_out;

    // ---- profiling
    bool prof_on = false;
    int prof_mode = 0;             // 1: every kernel family, 2: metric kernels only (names ending in "_pairs")
    std::vector<hipEvent_t> ev_pool;   // recycled profiling events
    std::vector<ProfEntry> prof;
    std::vector<PendingEvent> pending;
    hipEvent_t call_a = nullptr, call_b = nullptr;
    bool call_timed = false;
    // pinned staging for small host transfers (a pageable hipMemcpy of a few bytes costs ~25 us on
    // this stack, the pinned path ~13 us; small uploads need no synchronisation at all)
    static constexpr int PIN_SLOTS = 8;
    static constexpr size_t PIN_SLOT_BYTES = 64 * 1024;
    static constexpr size_t PIN_DL_BYTES = 1024 * 1024;   // downloads up to this size go through pinned memory
    static constexpr size_t PIN_TAIL_BYTES = 64;          // behind the download region: the persistent anchor launch's abort word (lev.hip)
    unsigned char *pin = nullptr;            // PIN_SLOTS slots: ring for uploads; then PIN_DL_BYTES for downloads
    hipEvent_t pin_ev[PIN_SLOTS] = {};
    bool pin_busy[PIN_SLOTS] = {};
    int pin_next = 0;
};

const char *ann_set_err(annchor_ctx *c, const char *fmt, ...);

#define ANN_CHECK_HIP(c, expr)                                                                  \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            ann_set_err((c), "%s:%d: %s failed: %s", __FILE__, __LINE__, #expr,                 \
                        hipGetErrorString(_e));                                                 \
            return ANNCHOR_EHIP;                                                                \
        }                                                                                       \
    } while (0)

#define ANN_TRY(expr)                \
    do {                             \
        int _r = (expr);             \
        if (_r != ANNCHOR_OK) return _r; \
    } while (0)

#define ANN_REQUIRE(c, cond, code, ...)    \
    do {                                   \
        if (!(cond)) {                     \
            ann_set_err((c), __VA_ARGS__); \
            return (code);                 \
        }                                  \
    } while (0)

int ann_reserve(annchor_ctx *c, DevBuf &b, size_t bytes);
int ann_arena_init(annchor_ctx *c, int64_t nx);
int ann_prewarm_state(annchor_ctx *c);   // ctx.hip
void ann_lev_ap_probe(annchor_ctx *c);   // lev.hip: after a host wait -- did the last persistent anchor launch give up?
size_t ann_sel2_table_bytes();           // scan.hip
size_t ann_tie_hist_bytes();             // select.hip
size_t ann_sel_state_bytes();            // select.hip
int ann_h2d(annchor_ctx *c, void *dst, const void *src, size_t bytes);
hipError_t ann_sync(annchor_ctx *c, const char *where);
int ann_dev_alloc(annchor_ctx *c, void **p, size_t want, size_t *got);   // hipMalloc / hipFree through the process-wide block pool (ctx.hip)
void ann_dev_free(annchor_ctx *c, void *p, size_t bytes);
struct Sel2Epilogue;   // selstate.h: what the selection's finishing workgroup writes for the consumer chained behind it
int ann_kth_async(annchor_ctx *c, const double *vals, const uint8_t *flag, int64_t n, const int64_t *ks, int nk,
                  const unsigned long long **d_prefix, const int **d_unfinished, const Sel2Epilogue *epi = nullptr);
void ann_kth_async_done(annchor_ctx *c);
int ann_d2h(annchor_ctx *c, void *dst, const void *src, size_t bytes);
int ann_d2h_then(annchor_ctx *c, void *dst, const void *src, size_t bytes, int (*then)(annchor_ctx *));
int ann_d2h2(annchor_ctx *c, void *dst1, const void *src1, size_t bytes1, void *dst2, const void *src2, size_t bytes2);

// profiling scopes: one entry per kernel family
int ann_prof_entry(annchor_ctx *c, const char *name);
// One event pair around a run of launches of the same family (the 15 dependent anchor rounds): the scopes of that family
// opened inside only count their launch and bytes.  38 events per C2 fit -> 10, and no event packets between the rounds.
struct ProfGroup {
    annchor_ctx *c;
    int entry = -1;
    hipEvent_t a = nullptr, b = nullptr;
    ProfGroup(annchor_ctx *ctx, const char *name);
    ~ProfGroup();
};

struct ProfScope {
    annchor_ctx *c;
    int entry = -1;
    hipEvent_t a = nullptr, b = nullptr;
    ProfScope(annchor_ctx *ctx, const char *name, double alg_bytes = 0);
    ~ProfScope();
};

static inline int ann_blocks(int64_t n, int per_block) { return (int)((n + per_block - 1) / per_block); }

// ---- device helpers ------------------------------------------------------
// order-preserving map double -> uint64 (ascending)
__device__ __forceinline__ uint64_t ann_key_asc(double v)
{
    uint64_t u = (uint64_t)__double_as_longlong(v);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double ann_key_asc_inv(uint64_t k)
{
    uint64_t u = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

// Load at a clamped index (n > 0).  `t < n ? p[t] : x` compiles to a branch around every load
// with an s_waitcnt vmcnt(0) after each, i.e. one memory round trip per element; clamping keeps
// a batch of loads in flight and the caller masks the value.
template <typename T> __device__ __forceinline__ T ann_ldc(const T *__restrict__ p, int64_t t, int64_t n)
{
    return p[t < n ? t : n - 1];
}

// Streaming accesses: above this many pairs the per-pair arrays (8-24 B per pair each) exceed what L2 / MALL
// hold between kernels, and a kernel that writes or reads them once does better telling the caches so
// (non-temporal: bounds_dad_features 1.53 -> 1.07 ms at 127 M pairs); below it the next kernel finds them cached.
#define ANN_STREAM_MIN_PAIRS (8ll << 20)
template <typename T> __device__ __forceinline__ void ann_store(T *p, T v, bool stream)
{
    if (stream) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <typename T> __device__ __forceinline__ T ann_load(const T *p, bool stream)
{
    return stream ? __builtin_nontemporal_load(p) : *p;
}

// ---- the stratified regression model (regressors.py:39-103), host- or device-resident
#define MAXBINS 64
struct RegModel {
    double e[MAXBINS + 1];
    double w[MAXBINS][3];
    double c[MAXBINS];
    int nb;
};
// the model of one iteration as the device fits it (annchor_fit_regression_device / annchor_fit_errors_device)
struct DeviceModel {
    RegModel reg;
    int32_t status[MAXBINS];     // per partition: 0 ok, 1 (numerically) rank deficient, 2 fewer rows than columns, 3 list too long
    int32_t err_status;          // residual lists: 0 ok, 1 an empty partition, 2 a partition longer than the sorter takes
    int64_t rows[MAXBINS];       // samples per regression partition
    int64_t errptr[MAXBINS + 1]; // offsets of the partitions' sorted residuals in `errs`
};

// ---- internal cross-file entry points -------------------------------------
// metric on a device pair list: out[t] (and optionally RA[pos[t]] = d, ncm[pos[t]] = 0)
struct PairSource {
    const int2 *ij = nullptr;     // explicit pairs (when idx == nullptr: pair t = ij[t])
    const int32_t *idx = nullptr; // optional positions into ij: pair t = ij[idx[t]]
    const int32_t *anchor = nullptr; // one-to-all: pair t = (*anchor, t)
    int64_t n = 0;
    // Max-min picking fused into a one-to-all launch (optional).  The launch first derives its own
    // anchor from the previous round's distances: runmin = reset ? row : min(runmin, row), anchor =
    // first arg-max of runmin (pickers.py:47-50), stored to *pick_out (== anchor) -- every workgroup
    // computes it for itself, so the separate arg-max launch and its dispatch gap disappear.  A
    // metric launch that did this sets *pick_fused; otherwise the caller picks as before.
    const double *pick_row = nullptr;
    double *pick_runmin = nullptr;
    int32_t *pick_out = nullptr;
    int pick_reset = 0;
    bool *pick_fused = nullptr;
};

// sid[i]: the `locality` nearest anchors of point i as a bit mask of NW 64-bit words (NW = 1, 2 or 4: up to 256 anchors),
// stored [nx][NW]; shared nearest anchors of two points = popcount of the AND
#define ANN_MAX_ANCHORS 256
template <int NW> struct Sid { uint64_t w[NW]; };
template <int NW> __device__ __forceinline__ Sid<NW> sid_ld(const uint64_t *__restrict__ sid, int64_t i)
{
    Sid<NW> s;
#pragma unroll
    for (int w = 0; w < NW; ++w) s.w[w] = sid[i * NW + w];
    return s;
}
template <int NW> __device__ __forceinline__ Sid<NW> sid_zero()
{
    Sid<NW> s;
#pragma unroll
    for (int w = 0; w < NW; ++w) s.w[w] = 0ull;
    return s;
}
template <int NW> __device__ __forceinline__ int sid_common(const Sid<NW> &a, const Sid<NW> &b)
{
    int c = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) c += __popcll(a.w[w] & b.w[w]);
    return c;
}
static inline int ann_sid_words(int na) { return na <= 64 ? 1 : na <= 128 ? 2 : 4; }
// KERNEL_CALL(NW) for the context's mask width
#define ANN_SID_DISPATCH(nw, CALL) do { if ((nw) == 1) { CALL(1); } else if ((nw) == 2) { CALL(2); } else { CALL(4); } } while (0)

// np.argmax: first maximal index
__device__ __forceinline__ void argmax_combine(double &v, int &i, double ov, int oi)
{
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}
int ann_metric_launch(annchor_ctx *c, const PairSource &src, double *d_out, double *d_RA, uint8_t *d_ncm);
int ann_lev_launch(annchor_ctx *c, const PairSource &src, double *d_out, double *d_RA, uint8_t *d_ncm);
// every round of the max-min picker in ONE launch (lev.hip, k_lev_ap); *done = false: not taken, the caller runs the rounds one by one
int ann_lev_anchor_rounds(annchor_ctx *c, int32_t na, int32_t first, bool *done);
int ann_legacy_generate_upto(uint32_t seed, int64_t ndraws, int64_t upto);   // hostrng.hip
static inline long long ann_now_ns()
{
    return (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
int ann_legacy_scan(uint32_t seed, const int64_t *counts, const int64_t *want, int32_t nbins, uint32_t *J, const int64_t *joff,
                    void (*after_bin)(int, void *), void *user, unsigned long long *progress = nullptr);   // hostrng.hip
// model.hip: the device-fitted model, its flags and residual lists copied (async) to pinned memory at `at` (room bytes; *used = 0:
// nothing to fetch / no room) and, after the caller's wait, kept in the context for annchor_model_download_with_errors
int ann_model_prefetch_begin(annchor_ctx *c, unsigned char *at, size_t room, size_t *used);
void ann_model_prefetch_end(annchor_ctx *c, const unsigned char *at, size_t used);
int ann_euclid_launch(annchor_ctx *c, const PairSource &src, double *d_out, double *d_RA, uint8_t *d_ncm);
int ann_emd_launch(annchor_ctx *c, const PairSource &src, double *d_out, double *d_RA, uint8_t *d_ncm);

// generic device primitives (scan.hip)
int ann_exclusive_scan_i32_to_i64(annchor_ctx *c, const int32_t *in, int64_t *out, int64_t n);  // out has n+1
// k-th smallest (0-based) of double keys where flag != 0 (flag may be null = all)
int ann_kth_smallest(annchor_ctx *c, const double *vals, const uint8_t *flag, int64_t n, const int64_t *ks,
                     int nk, double *h_out);

int ann_set_anchor_flags(annchor_ctx *c, const int64_t *hA, int nA);
// predict + clip + merge + label over all pairs and the samples' unclipped predictions, from a model in DEVICE memory
int ann_dev_flags(annchor_ctx *c);   // reserve + zero the sticky flags once per context
int ann_predict_merge_device(annchor_ctx *c, const RegModel *d_model, int first_iteration, int is_metric);
