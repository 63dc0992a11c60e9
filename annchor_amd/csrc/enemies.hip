// enemies.hip -- the nearest-enemy graph on the device.
//
// Replaces Annchor.get_nearest_enemies (reference annchor/annchor.py:685-782) with get_check under the label filter
// (annchor/utils.py:454-491), adjust_check (utils.py:437-451) and get_IJs_from_check (utils.py:502-540):
//   candidates  per point: the ENEMIES (other label) sharing nearest anchors -- c_ij = popcount(sid_i & sid_j) >= thr_i,
//               thr_i = min(loc_thresh, (loc_min + 1)-th largest c_i. among enemies) -- symmetrised when a threshold
//               was lowered, minus the pairs fit() already holds: a second symmetric bitmap N next to the fitted keep
//               bitmap K; the sorted pair list and the per-point index of the new pairs come out of N through the same
//               prefix-popcount kernels as the fitted ones (locality.hip);
//   features    bounds / dad / anchor flag of the new pairs (the fit's own kernel), predicted distance clipped to
//               [lb, ub] from the fitted regression (annchor.py:724-728);
//   first       per point, over its fitted + new entries: the `first` (50) closest-looking enemies; the ones not yet
//               computed get their exact distance (annchor.py:744-761);
//   graph       per point: the nn nearest among computed enemies -- not-computed and same-label entries are pushed
//               behind by the row maximum (annchor.py:763-781).
// Ties resolve by list order (old entries by other endpoint, then new entries by other endpoint), as in the CPU restatement the tests compare with;
// the reference's argsorts are unstable.  The fitted RefineApprox / not_computed_mask are updated in place, as the
// reference updates its own; the new pairs live in a separate set of arrays (downloadable for the host's views).
#include "common.h"
#include "rowsel.h"

// kernels of locality.hip / features.hip reused on the enemy bitmap and pair list
#define LOC_THREADS 256
__global__ __launch_bounds__(LOC_THREADS) void k_row_prefix(const uint64_t *__restrict__ K, int64_t nx, int kw,
                                                           uint32_t *__restrict__ pref, int32_t *__restrict__ deg,
                                                           int32_t *__restrict__ low, int32_t *__restrict__ up);
__global__ __launch_bounds__(256) void k_emit_pairs(const uint64_t *__restrict__ K, const uint32_t *__restrict__ pref, int64_t nx, int kw,
                                                   const int32_t *__restrict__ low, const int64_t *__restrict__ rowstart,
                                                   const int64_t *__restrict__ Iptr, int2 *__restrict__ ij, int32_t *__restrict__ Iidx,
                                                   int stream, int rows_only);
__global__ __launch_bounds__(256) void k_features(const int2 *__restrict__ ij, int64_t n, const double *__restrict__ Dt,
                                                 int64_t nx, int na, const int32_t *__restrict__ cA,
                                                 const int32_t *__restrict__ anchorRank, double *__restrict__ lb,
                                                 double *__restrict__ ub, double *__restrict__ dad,
                                                 uint8_t *__restrict__ anc, uint8_t *__restrict__ ncm);

struct EnemyState {
    DevBuf y;                                  // int32 [nx] dense label codes
    DevBuf thr, flags;                         // int32 [nx]; int32 [4]: lowered, short row, -, -
    DevBuf N, Npref, deg, low, up, rowstart, Iptr;
    DevBuf ij, Iidx, lb, ub, dad, anc, ncm, RA;
    DevBuf todo, todo_ij, todo_d, todo_n;      // entries to evaluate: int32 position | new-flag in bit 31
    DevBuf out_i, out_d;
    int64_t n_new = 0, n_todo = 0;
    bool have_candidates = false, have_prediction = false;
};
static std::vector<std::pair<annchor_ctx *, EnemyState *>> g_en;

static EnemyState *en_state(annchor_ctx *c, bool create)
{
    for (auto &p : g_en)
        if (p.first == c) return p.second;
    if (!create) return nullptr;
    EnemyState *s = new EnemyState();
    g_en.push_back({c, s});
    return s;
}

void ann_enemies_release(annchor_ctx *c)
{
    for (size_t i = 0; i < g_en.size(); ++i)
        if (g_en[i].first == c) {
            EnemyState *s = g_en[i].second;
            DevBuf *bufs[] = {&s->y, &s->thr, &s->flags, &s->N, &s->Npref, &s->deg, &s->low, &s->up, &s->rowstart, &s->Iptr, &s->ij, &s->Iidx,
                              &s->lb, &s->ub, &s->dad, &s->anc, &s->ncm, &s->RA, &s->todo, &s->todo_ij, &s->todo_d, &s->todo_n, &s->out_i,
                              &s->out_d};
            for (DevBuf *b : bufs)
                if (b->p && !b->in_arena) ann_dev_free(c, b->p, b->cap);
            delete s;
            g_en.erase(g_en.begin() + (long)i);
            return;
        }
}

static int en_reserve(annchor_ctx *c, DevBuf &b, size_t bytes)
{
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return ANNCHOR_OK;
    if (b.p && !b.in_arena) ann_dev_free(c, b.p, b.cap);
    b.p = nullptr; b.cap = 0; b.in_arena = false;
    const size_t want0 = (bytes + 255) & ~(size_t)255;
    size_t got = 0;
    ANN_TRY(ann_dev_alloc(c, &b.p, want0, &got));
    b.cap = got;
    return ANNCHOR_OK;
}

// ------------------------------------------------------------------ candidates
// one block per row: histogram of c_ij over the ENEMIES j, then the (loc_min + 1)-th largest (utils.py:472-480 with f)
template <int NW>
__global__ __launch_bounds__(LOC_THREADS) void k_en_thresh(const uint64_t *__restrict__ sid, const int32_t *__restrict__ y, int64_t nx,
                                                          int loc_thresh, int loc_min, int32_t *__restrict__ thr, int32_t *__restrict__ flags)
{
    __shared__ uint32_t hist[ANN_MAX_ANCHORS + 1];
    for (int t = threadIdx.x; t < ANN_MAX_ANCHORS + 1; t += blockDim.x) hist[t] = 0;
    __syncthreads();
    const int64_t i = blockIdx.x;
    const Sid<NW> mi = sid_ld<NW>(sid, i);
    const int32_t yi = y[i];
    for (int64_t j = threadIdx.x; j < nx; j += blockDim.x)
        if (y[j] != yi) atomicAdd(&hist[sid_common<NW>(mi, sid_ld<NW>(sid, j))], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t ne = 0;
        for (int v = 0; v <= ANN_MAX_ANCHORS; ++v) ne += hist[v];
        const int64_t lm = loc_min < ne - 1 ? loc_min : ne - 1;
        int64_t cum = 0;
        int v = ANN_MAX_ANCHORS;
        for (; v >= 0; --v) {
            cum += hist[v];
            if (cum >= lm + 1) break;
        }
        if (v < 0) v = 0;
        thr[i] = v < loc_thresh ? v : loc_thresh;
        if (v < loc_thresh) atomicOr(&flags[0], 1);   // a threshold was lowered: adjust_check symmetrises
    }
}

// new-pair bits: wave per (row, 64-column word), lane = column
template <int NW>
__global__ __launch_bounds__(256) void k_en_keep_bits(const uint64_t *__restrict__ sid, const int32_t *__restrict__ y,
                                                     const int32_t *__restrict__ thr, const int32_t *__restrict__ flags,
                                                     const uint64_t *__restrict__ Kfit, int64_t nx, int kw, uint64_t *__restrict__ N)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t wave_count = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t items = nx * kw;
    const int lowered = flags[0];
    const int64_t per = (items + wave_count - 1) / wave_count;
    const int64_t t0 = wave_global * per, t1 = min(t0 + per, items);
    for (int64_t t = t0; t < t1; ++t) {
        const int64_t i = t / kw;
        const int w = (int)(t - i * kw);
        const Sid<NW> mi = sid_ld<NW>(sid, i);
        const int ti = thr[i];
        const int32_t yi = y[i];
        const uint64_t fit = Kfit[t];
        const int64_t j = (int64_t)w * 64 + lane;
        bool keep = false;
        if (j < nx && j != i && y[j] != yi && !((fit >> lane) & 1ull)) {
            const int cc = sid_common<NW>(mi, sid_ld<NW>(sid, j));
            const int tj = thr[j];
            // row of the smaller index decides; when a threshold was lowered the larger index's row counts too
            const int t_small = i < j ? ti : tj, t_large = i < j ? tj : ti;
            keep = cc >= t_small || (lowered && cc >= t_large);
        }
        const unsigned long long bits = __ballot(keep);
        if (lane == 0) N[t] = bits;
    }
}

extern "C" int annchor_enemies_candidates(annchor_ctx *c, const int32_t *y, int32_t loc_thresh, int32_t loc_min, int64_t *n_new)
{
    if (!c || !y || !n_new) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->have_bitmap && c->n > 0 && c->have_features, ANNCHOR_ESTATE, "nearest enemies need a fitted pair list");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    EnemyState *s = en_state(c, true);
    const int64_t nx = c->nx;
    const int kw = (int)((nx + 63) / 64);
    ANN_TRY(en_reserve(c, s->y, sizeof(int32_t) * (size_t)nx));
    ANN_TRY(en_reserve(c, s->thr, sizeof(int32_t) * (size_t)nx));
    ANN_TRY(en_reserve(c, s->flags, sizeof(int32_t) * 4));
    ANN_TRY(en_reserve(c, s->N, sizeof(uint64_t) * (size_t)nx * kw));
    ANN_TRY(en_reserve(c, s->Npref, sizeof(uint32_t) * (size_t)nx * kw));
    ANN_TRY(en_reserve(c, s->deg, sizeof(int32_t) * (size_t)nx));
    ANN_TRY(en_reserve(c, s->low, sizeof(int32_t) * (size_t)nx));
    ANN_TRY(en_reserve(c, s->up, sizeof(int32_t) * (size_t)nx));
    ANN_TRY(en_reserve(c, s->rowstart, sizeof(int64_t) * (size_t)(nx + 1)));
    ANN_TRY(en_reserve(c, s->Iptr, sizeof(int64_t) * (size_t)(nx + 1)));
    ANN_TRY(ann_h2d(c, s->y.p, y, sizeof(int32_t) * (size_t)nx));
    ANN_CHECK_HIP(c, hipMemsetAsync(s->flags.p, 0, sizeof(int32_t) * 4, c->stream));
    {
        ProfScope ps(c, "enemy_keep_bitmap", (double)nx * kw * 20.0);
#define ENT_CALL(NW) k_en_thresh<NW><<<(int)nx, LOC_THREADS, 0, c->stream>>>(c->sid.as<uint64_t>(), s->y.as<int32_t>(), nx, loc_thresh, loc_min, s->thr.as<int32_t>(), s->flags.as<int32_t>())
        ANN_SID_DISPATCH(c->sid_nw, ENT_CALL);
#undef ENT_CALL
#define ENK_CALL(NW) k_en_keep_bits<NW><<<(int)std::min<int64_t>(ann_blocks(nx * kw * 64, 256), (int64_t)c->prop.multiProcessorCount * 32), 256, 0, c->stream>>>( \
            c->sid.as<uint64_t>(), s->y.as<int32_t>(), s->thr.as<int32_t>(), s->flags.as<int32_t>(), c->Kbits.as<uint64_t>(), nx, kw, s->N.as<uint64_t>())
        ANN_SID_DISPATCH(c->sid_nw, ENK_CALL);
#undef ENK_CALL
        k_row_prefix<<<(int)nx, LOC_THREADS, 0, c->stream>>>(s->N.as<uint64_t>(), nx, kw, s->Npref.as<uint32_t>(), s->deg.as<int32_t>(),
                                                            s->low.as<int32_t>(), s->up.as<int32_t>());
    }
    ANN_TRY(ann_exclusive_scan_i32_to_i64(c, s->up.as<int32_t>(), s->rowstart.as<int64_t>(), nx));
    ANN_TRY(ann_exclusive_scan_i32_to_i64(c, s->deg.as<int32_t>(), s->Iptr.as<int64_t>(), nx));
    int64_t n = 0;
    ANN_TRY(ann_d2h(c, &n, s->rowstart.as<int64_t>() + nx, sizeof n));
    ANN_REQUIRE(c, n + c->n < (1ll << 30), ANNCHOR_ELIMIT, "%lld enemy pairs exceed the pair-list limit", (long long)n);
    s->n_new = n;
    const size_t nn = (size_t)std::max<int64_t>(n, 1);
    ANN_TRY(en_reserve(c, s->ij, sizeof(int2) * nn));
    ANN_TRY(en_reserve(c, s->Iidx, sizeof(int32_t) * 2 * nn));
    ANN_TRY(en_reserve(c, s->lb, 8 * nn));
    ANN_TRY(en_reserve(c, s->ub, 8 * nn));
    ANN_TRY(en_reserve(c, s->dad, 8 * nn));
    ANN_TRY(en_reserve(c, s->RA, 8 * nn));
    ANN_TRY(en_reserve(c, s->anc, nn));
    ANN_TRY(en_reserve(c, s->ncm, nn));
    if (n > 0) {
        ProfScope ps(c, "enemy_emit_features", (double)n * 50.0 + (double)nx * kw * 12.0);
        k_emit_pairs<<<(int)std::min<int64_t>(ann_blocks(nx * kw * 64, 256), (int64_t)c->prop.multiProcessorCount * 64), 256, 0, c->stream>>>(
            s->N.as<uint64_t>(), s->Npref.as<uint32_t>(), nx, kw, s->low.as<int32_t>(), s->rowstart.as<int64_t>(), s->Iptr.as<int64_t>(),
            s->ij.as<int2>(), s->Iidx.as<int32_t>(), 0, 0);
        k_features<<<ann_blocks(n, 256), 256, 0, c->stream>>>(s->ij.as<int2>(), n, c->Dt.as<double>(), nx, c->na, c->cA.as<int32_t>(),
                                                              c->anchorRank.as<int32_t>(), s->lb.as<double>(), s->ub.as<double>(),
                                                              s->dad.as<double>(), s->anc.as<uint8_t>(), s->ncm.as<uint8_t>());
    }
    ANN_CHECK_HIP(c, hipGetLastError());
    s->have_candidates = true;
    s->have_prediction = false;
    *n_new = n;
    return ANNCHOR_OK;
}

// ------------------------------------------------------------------ prediction of the new pairs
__global__ void k_en_predict(int64_t n, const RegModel *__restrict__ mp, const double *__restrict__ lb, const double *__restrict__ ub,
                             const double *__restrict__ dad, double *__restrict__ RA)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const RegModel &m = *mp;
    const double l = lb[p], u = ub[p], d = dad[p];
    int b = -1;
    for (int k = 0; k < m.nb; ++k)
        if (d > m.e[k] && d <= m.e[k + 1]) b = k;      // regressors.py:84-87
    double pr = b < 0 ? 0.0 : ((m.w[b][0] * l + m.w[b][1] * u) + m.w[b][2] * d) + m.c[b];
    RA[p] = fmin(fmax(pr, l), u);                      // np.clip(pred, lb, ub), annchor.py:728
}

__global__ void k_en_clip(int64_t n, const double *__restrict__ pred, const double *__restrict__ lb, const double *__restrict__ ub,
                          double *__restrict__ RA)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) RA[p] = fmin(fmax(pred[p], lb[p]), ub[p]);
}

// The fitted stratified regression (HOST coefficients: bins [nb + 1], W [nb][3], c [nb]) applied to the new pairs, or --
// pred != NULL -- a custom regression's predictions for them (host, float64 [n_new]); clipped to [lb, ub] either way.
extern "C" int annchor_enemies_predict(annchor_ctx *c, const double *bins, int32_t nb, const double *W, const double *cc, const double *pred)
{
    if (!c || (!pred && (!bins || !W || !cc))) return ANNCHOR_EINVAL;
    EnemyState *s = en_state(c, false);
    ANN_REQUIRE(c, s && s->have_candidates, ANNCHOR_ESTATE, "annchor_enemies_candidates first");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    const int64_t n = s->n_new;
    s->have_prediction = true;
    if (n == 0) return ANNCHOR_OK;
    if (pred) {
        ANN_TRY(ann_reserve(c, c->stage_in, sizeof(double) * (size_t)n));
        ANN_TRY(ann_h2d(c, c->stage_in.p, pred, sizeof(double) * (size_t)n));
        k_en_clip<<<ann_blocks(n, 256), 256, 0, c->stream>>>(n, c->stage_in.as<double>(), s->lb.as<double>(), s->ub.as<double>(), s->RA.as<double>());
    } else {
        ANN_REQUIRE(c, nb >= 1 && nb <= MAXBINS, ANNCHOR_ELIMIT, "1..%d partitions supported", MAXBINS);
        RegModel m;
        memset(&m, 0, sizeof m);
        m.nb = nb;
        for (int k = 0; k <= nb; ++k) m.e[k] = bins[k];
        for (int k = 0; k < nb; ++k) { m.w[k][0] = W[3 * k]; m.w[k][1] = W[3 * k + 1]; m.w[k][2] = W[3 * k + 2]; m.c[k] = cc[k]; }
        ANN_TRY(en_reserve(c, s->todo_n, sizeof(RegModel) + 64));
        ANN_TRY(ann_h2d(c, s->todo_n.as<char>() + 64, &m, sizeof m));
        k_en_predict<<<ann_blocks(n, 256), 256, 0, c->stream>>>(n, reinterpret_cast<const RegModel *>(s->todo_n.as<char>() + 64), s->lb.as<double>(),
                                                               s->ub.as<double>(), s->dad.as<double>(), s->RA.as<double>());
    }
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}

// ------------------------------------------------------------------ merged rows
// Row i = its fitted entries (positions into the fitted arrays) followed by its new entries (positions into the enemy arrays).
struct EnRows {
    const int64_t *Iptr_f;
    const int32_t *Iidx_f;
    const int2 *ij_f;
    const double *RA_f;
    const uint8_t *ncm_f;
    const int64_t *Iptr_n;
    const int32_t *Iidx_n;
    const int2 *ij_n;
    const double *RA_n;
    const uint8_t *ncm_n;
    const int32_t *y;
};

struct EnEntry {
    int32_t pos;     // position in the fitted or the enemy arrays
    int is_new;
    int other;
    double ra;
    bool unc;
};

__device__ __forceinline__ EnEntry en_entry(const EnRows &R, int64_t i, int64_t bf, int len_f, int64_t bn, int s)
{
    EnEntry e;
    if (s < len_f) {
        e.pos = R.Iidx_f[bf + s]; e.is_new = 0;
        const int2 q = R.ij_f[e.pos];
        e.other = q.x == (int)i ? q.y : q.x;
        e.ra = R.RA_f[e.pos]; e.unc = R.ncm_f[e.pos] != 0;
    } else {
        e.pos = R.Iidx_n[bn + (s - len_f)]; e.is_new = 1;
        const int2 q = R.ij_n[e.pos];
        e.other = q.x == (int)i ? q.y : q.x;
        e.ra = R.RA_n[e.pos]; e.unc = R.ncm_n[e.pos] != 0;
    }
    return e;
}

// per row: the `first` closest-looking enemies (RefineApprox ascending, ties in list order); the not-computed ones among
// them are appended to the todo list (annchor.py:744-761)
__global__ __launch_bounds__(ROW_THREADS) void k_en_first(EnRows R, int64_t nx, int first, int nn, int32_t *__restrict__ todo,
                                                         unsigned long long *__restrict__ n_todo, int32_t *__restrict__ flags)
{
    __shared__ RowSelShared sh;
    __shared__ uint32_t wsum[ROW_THREADS / 64];
    __shared__ uint32_t taken_s, ne_s;
    const int64_t i = blockIdx.x;
    const int64_t bf = R.Iptr_f[i], bn = R.Iptr_n[i];
    const int len_f = (int)(R.Iptr_f[i + 1] - bf), len = len_f + (int)(R.Iptr_n[i + 1] - bn);
    const int32_t yi = R.y[i];
    if (len <= nn) { if (threadIdx.x == 0) atomicOr(&flags[1], 1); return; }   // "a point has no more than nn candidates"
    const uint64_t KINF = ~0ull;
    auto key = [&](int s) -> uint64_t {
        const EnEntry e = en_entry(R, i, bf, len_f, bn, s);
        return R.y[e.other] != yi ? ann_key_asc(e.ra) : KINF;
    };
    if (threadIdx.x == 0) { ne_s = 0; taken_s = 0; }
    __syncthreads();
    uint32_t mine = 0;
    for (int s = threadIdx.x; s < len; s += ROW_THREADS) mine += key(s) != KINF;
    if (mine) atomicAdd(&ne_s, mine);
    __syncthreads();
    const int ne = (int)ne_s;
    if (ne == 0) return;
    const int want = min(first, ne);
    const uint64_t t = row_kth_key(sh, len, (uint32_t)(want - 1), key);
    // strictly below the cut: all of them; on the cut: in list order until `want` are taken
    uint32_t lt = 0;
    for (int s = threadIdx.x; s < len; s += ROW_THREADS) lt += key(s) < t;
    uint32_t lt_tot;
    (void)row_block_scan(lt, wsum, &lt_tot);
    const int room = want - (int)lt_tot;   // entries equal to the cut that still fit (>= 1)
    for (int s0 = 0; s0 < len; s0 += ROW_THREADS) {
        const int s = s0 + threadIdx.x;
        uint64_t kk = KINF;
        EnEntry e;
        e.pos = 0; e.is_new = 0; e.unc = false; e.other = 0; e.ra = 0;
        if (s < len) { e = en_entry(R, i, bf, len_f, bn, s); kk = R.y[e.other] != yi ? ann_key_asc(e.ra) : KINF; }
        const uint32_t eq = (s < len && kk == t && kk != KINF) ? 1u : 0u;
        uint32_t eq_tot;
        const uint32_t ex = row_block_scan(eq, wsum, &eq_tot);
        const uint32_t before = taken_s;
        const bool pick = (s < len) && kk != KINF && (kk < t || (eq && (int)(before + ex) < room));
        if (pick && e.unc) {
            const unsigned long long slot = atomicAdd(n_todo, 1ull);
            todo[slot] = e.pos | (e.is_new ? (int32_t)0x80000000 : 0);
        }
        __syncthreads();
        if (threadIdx.x == 0) taken_s = before + eq_tot;
        __syncthreads();
    }
}

__global__ void k_en_todo_pairs(const int32_t *__restrict__ todo, int64_t m, const int2 *__restrict__ ij_f, const int2 *__restrict__ ij_n,
                                int2 *__restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const int32_t v = todo[t];
    out[t] = v < 0 ? ij_n[v & 0x7fffffff] : ij_f[v];
}

__global__ void k_en_writeback(const int32_t *__restrict__ todo, const double *__restrict__ d, int64_t m, double *__restrict__ RA_f,
                               uint8_t *__restrict__ ncm_f, double *__restrict__ RA_n, uint8_t *__restrict__ ncm_n)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const int32_t v = todo[t];
    if (v < 0) { RA_n[v & 0x7fffffff] = d[t]; ncm_n[v & 0x7fffffff] = 0; }
    else { RA_f[v] = d[t]; ncm_f[v] = 0; }
}

static EnRows en_rows(annchor_ctx *c, EnemyState *s)
{
    EnRows R;
    R.Iptr_f = c->Iptr.as<int64_t>(); R.Iidx_f = c->Iidx.as<int32_t>(); R.ij_f = c->ij.as<int2>(); R.RA_f = c->RA.as<double>();
    R.ncm_f = c->ncm.as<uint8_t>();
    R.Iptr_n = s->Iptr.as<int64_t>(); R.Iidx_n = s->Iidx.as<int32_t>(); R.ij_n = s->ij.as<int2>(); R.RA_n = s->RA.as<double>();
    R.ncm_n = s->ncm.as<uint8_t>(); R.y = s->y.as<int32_t>();
    return R;
}

// The todo list of annchor.py:744-761.  Device metric (evaluate != 0): evaluated and written back here, *n_todo = its
// length (the caller's evaluation count).  Otherwise the pairs come back (todo_ij: HOST int64 [cap][2], cap >= nx * first)
// for the host metric and annchor_enemies_set_exact takes the values.
extern "C" int annchor_enemies_first(annchor_ctx *c, int32_t first, int32_t nn, int32_t evaluate, int64_t *todo_ij, int64_t cap,
                                     int64_t *n_todo)
{
    if (!c || !n_todo || (!evaluate && !todo_ij)) return ANNCHOR_EINVAL;
    EnemyState *s = en_state(c, false);
    ANN_REQUIRE(c, s && s->have_candidates && s->have_prediction, ANNCHOR_ESTATE, "annchor_enemies_candidates / _predict first");
    ANN_REQUIRE(c, c->have_RA, ANNCHOR_ESTATE, "RefineApprox not initialised");
    ANN_REQUIRE(c, first >= 1 && nn >= 1, ANNCHOR_EINVAL, "bad parameters");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    const int64_t nx = c->nx, maxtodo = nx * (int64_t)first;
    ANN_TRY(en_reserve(c, s->todo, sizeof(int32_t) * (size_t)maxtodo));
    ANN_TRY(en_reserve(c, s->todo_n, sizeof(RegModel) + 64));
    ANN_CHECK_HIP(c, hipMemsetAsync(s->todo_n.p, 0, 8, c->stream));
    {
        ProfScope ps(c, "enemy_first_lists", (double)(c->n + s->n_new) * 2 * 30.0);
        k_en_first<<<(int)nx, ROW_THREADS, 0, c->stream>>>(en_rows(c, s), nx, first, nn, s->todo.as<int32_t>(),
                                                          s->todo_n.as<unsigned long long>(), s->flags.as<int32_t>());
    }
    ANN_CHECK_HIP(c, hipGetLastError());
    unsigned long long m = 0;
    int32_t fl[4];
    ANN_TRY(ann_d2h2(c, &m, s->todo_n.p, sizeof m, fl, s->flags.p, sizeof fl));
    ANN_REQUIRE(c, !fl[1], ANNCHOR_EINVAL, "a point has no more than nn=%d candidates", nn);
    s->n_todo = (int64_t)m;
    *n_todo = (int64_t)m;
    if (m == 0) return ANNCHOR_OK;
    ANN_TRY(en_reserve(c, s->todo_ij, sizeof(int2) * (size_t)m));
    ANN_TRY(en_reserve(c, s->todo_d, sizeof(double) * (size_t)m));
    k_en_todo_pairs<<<ann_blocks((int64_t)m, 256), 256, 0, c->stream>>>(s->todo.as<int32_t>(), (int64_t)m, c->ij.as<int2>(), s->ij.as<int2>(),
                                                                       s->todo_ij.as<int2>());
    if (evaluate) {
        ANN_REQUIRE(c, c->metric != ANNCHOR_METRIC_NONE, ANNCHOR_EINVAL, "no device metric bound to this context");
        PairSource src;
        src.ij = s->todo_ij.as<int2>();
        src.n = (int64_t)m;
        ANN_TRY(ann_metric_launch(c, src, s->todo_d.as<double>(), nullptr, nullptr));
        k_en_writeback<<<ann_blocks((int64_t)m, 256), 256, 0, c->stream>>>(s->todo.as<int32_t>(), s->todo_d.as<double>(), (int64_t)m,
                                                                          c->RA.as<double>(), c->ncm.as<uint8_t>(), s->RA.as<double>(),
                                                                          s->ncm.as<uint8_t>());
        ANN_CHECK_HIP(c, hipGetLastError());
        c->n_unc = -1; c->sel_prepared = false;
        return ANNCHOR_OK;
    }
    ANN_REQUIRE(c, (int64_t)m <= cap, ANNCHOR_EINVAL, "todo buffer too small");
    std::vector<int2> h((size_t)m);
    ANN_TRY(ann_d2h(c, h.data(), s->todo_ij.p, sizeof(int2) * (size_t)m));
    for (size_t t = 0; t < (size_t)m; ++t) { todo_ij[2 * t] = h[t].x; todo_ij[2 * t + 1] = h[t].y; }
    return ANNCHOR_OK;
}

extern "C" int annchor_enemies_set_exact(annchor_ctx *c, const double *exact, int64_t m)
{
    if (!c || (m > 0 && !exact)) return ANNCHOR_EINVAL;
    EnemyState *s = en_state(c, false);
    ANN_REQUIRE(c, s && m == s->n_todo, ANNCHOR_EINVAL, "expected %lld exact distances", (long long)(s ? s->n_todo : 0));
    if (m == 0) return ANNCHOR_OK;
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    ANN_TRY(ann_h2d(c, s->todo_d.p, exact, sizeof(double) * (size_t)m));
    k_en_writeback<<<ann_blocks(m, 256), 256, 0, c->stream>>>(s->todo.as<int32_t>(), s->todo_d.as<double>(), m, c->RA.as<double>(),
                                                              c->ncm.as<uint8_t>(), s->RA.as<double>(), s->ncm.as<uint8_t>());
    ANN_CHECK_HIP(c, hipGetLastError());
    c->n_unc = -1; c->sel_prepared = false;
    return ANNCHOR_OK;
}

// per row: key d = RA (+ mx if not computed) (+ mx if same label), mx = row maximum of RA; the nn smallest by (d, list
// order); reported distance = RA (annchor.py:763-781)
__global__ __launch_bounds__(ROW_THREADS) void k_en_graph(EnRows R, int64_t nx, int nn, int64_t *__restrict__ oi, double *__restrict__ od)
{
    __shared__ double wmx[ROW_THREADS / 64];
    __shared__ unsigned long long wk[ROW_THREADS / 64];
    __shared__ int ws[ROW_THREADS / 64];
    __shared__ unsigned long long last_k;
    __shared__ int last_s;
    const int64_t i = blockIdx.x;
    const int64_t bf = R.Iptr_f[i], bn = R.Iptr_n[i];
    const int len_f = (int)(R.Iptr_f[i + 1] - bf), len = len_f + (int)(R.Iptr_n[i + 1] - bn);
    const int32_t yi = R.y[i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double mx = -INFINITY;
    for (int s = threadIdx.x; s < len; s += ROW_THREADS) mx = fmax(mx, en_entry(R, i, bf, len_f, bn, s).ra);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
    if (lane == 0) wmx[wave] = mx;
    __syncthreads();
    mx = fmax(fmax(wmx[0], wmx[1]), fmax(wmx[2], wmx[3]));
    auto key = [&](const EnEntry &e) -> unsigned long long {
        double d = e.ra;
        if (e.unc) d += mx;
        if (R.y[e.other] == yi) d += mx;
        return ann_key_asc(d);
    };
    if (threadIdx.x == 0) { last_k = 0; last_s = -1; }
    __syncthreads();
    for (int r = 0; r < nn; ++r) {
        const unsigned long long lk = last_k;
        const int ls = last_s;
        unsigned long long bk = ~0ull;
        int bs = 0x7fffffff;
        for (int s = threadIdx.x; s < len; s += ROW_THREADS) {
            const unsigned long long kk = key(en_entry(R, i, bf, len_f, bn, s));
            const bool after = r == 0 || kk > lk || (kk == lk && s > ls);
            if (after && (kk < bk || (kk == bk && s < bs))) { bk = kk; bs = s; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long ok = __shfl_xor(bk, off);
            const int os = __shfl_xor(bs, off);
            if (ok < bk || (ok == bk && os < bs)) { bk = ok; bs = os; }
        }
        __syncthreads();
        if (lane == 0) { wk[wave] = bk; ws[wave] = bs; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < ROW_THREADS / 64; ++w)
                if (wk[w] < bk || (wk[w] == bk && ws[w] < bs)) { bk = wk[w]; bs = ws[w]; }
            last_k = bk; last_s = bs;
            const EnEntry e = en_entry(R, i, bf, len_f, bn, bs);
            oi[i * nn + r] = e.other;
            od[i * nn + r] = e.ra;
        }
        __syncthreads();
    }
}

extern "C" int annchor_enemies_graph(annchor_ctx *c, int32_t nn, int64_t *idx, double *dist)
{
    if (!c || !idx || !dist) return ANNCHOR_EINVAL;
    EnemyState *s = en_state(c, false);
    ANN_REQUIRE(c, s && s->have_candidates && s->have_prediction, ANNCHOR_ESTATE, "annchor_enemies_candidates / _predict first");
    ANN_REQUIRE(c, nn >= 1, ANNCHOR_EINVAL, "nn must be >= 1");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    const int64_t nx = c->nx;
    ANN_TRY(en_reserve(c, s->out_i, sizeof(int64_t) * (size_t)nx * nn));
    ANN_TRY(en_reserve(c, s->out_d, sizeof(double) * (size_t)nx * nn));
    {
        ProfScope ps(c, "enemy_row_topk", (double)(c->n + s->n_new) * 2 * 30.0 * (nn + 1));
        k_en_graph<<<(int)nx, ROW_THREADS, 0, c->stream>>>(en_rows(c, s), nx, nn, s->out_i.as<int64_t>(), s->out_d.as<double>());
    }
    ANN_CHECK_HIP(c, hipGetLastError());
    ANN_TRY(ann_d2h(c, idx, s->out_i.p, sizeof(int64_t) * (size_t)nx * nn));
    return ann_d2h(c, dist, s->out_d.p, sizeof(double) * (size_t)nx * nn);
}

__global__ void k_en_pack(int64_t n, const int2 *__restrict__ ij, const double *__restrict__ lb, const double *__restrict__ ub,
                          const double *__restrict__ dad, const uint8_t *__restrict__ anc, int64_t *__restrict__ oij, double *__restrict__ of)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    oij[2 * p] = ij[p].x; oij[2 * p + 1] = ij[p].y;
    of[4 * p] = lb[p]; of[4 * p + 1] = ub[p]; of[4 * p + 2] = dad[p]; of[4 * p + 3] = (double)anc[p];
}

// The new pairs for the host's views (annchor.py:729-740 appends them to IJs / features / RefineApprox /
// not_computed_mask / I): ij int64 [n_new][2], feats float64 [n_new][4], RA float64 [n_new], ncm uint8 [n_new],
// I_ptr int64 [nx + 1] and I_idx int64 [2 n_new] (positions into the NEW pairs, rows ordered by other endpoint).
extern "C" int annchor_enemies_download(annchor_ctx *c, int64_t *ij, double *feats, double *RA, uint8_t *ncm, int64_t *I_ptr, int64_t *I_idx)
{
    if (!c || !I_ptr) return ANNCHOR_EINVAL;
    EnemyState *s = en_state(c, false);
    ANN_REQUIRE(c, s && s->have_candidates, ANNCHOR_ESTATE, "annchor_enemies_candidates first");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    const int64_t n = s->n_new, nx = c->nx;
    ANN_TRY(ann_d2h(c, I_ptr, s->Iptr.p, sizeof(int64_t) * (size_t)(nx + 1)));
    if (n == 0) return ANNCHOR_OK;
    ANN_REQUIRE(c, ij && feats && RA && ncm && I_idx, ANNCHOR_EINVAL, "output arrays missing");
    ANN_TRY(ann_reserve(c, c->stage_out, (sizeof(int64_t) * 2 + sizeof(double) * 4) * (size_t)n));
    int64_t *oij = c->stage_out.as<int64_t>();
    double *of = reinterpret_cast<double *>(oij + 2 * n);
    k_en_pack<<<ann_blocks(n, 256), 256, 0, c->stream>>>(n, s->ij.as<int2>(), s->lb.as<double>(), s->ub.as<double>(), s->dad.as<double>(),
                                                         s->anc.as<uint8_t>(), oij, of);
    ANN_CHECK_HIP(c, hipGetLastError());
    ANN_TRY(ann_d2h(c, ij, oij, sizeof(int64_t) * 2 * (size_t)n));
    ANN_TRY(ann_d2h(c, feats, of, sizeof(double) * 4 * (size_t)n));
    ANN_TRY(ann_d2h(c, RA, s->RA.p, sizeof(double) * (size_t)n));
    ANN_TRY(ann_d2h(c, ncm, s->ncm.p, (size_t)n));
    std::vector<int32_t> h((size_t)2 * n);
    ANN_TRY(ann_d2h(c, h.data(), s->Iidx.p, sizeof(int32_t) * 2 * (size_t)n));
    for (size_t t = 0; t < h.size(); ++t) I_idx[t] = h[t];
    return ANNCHOR_OK;
}
