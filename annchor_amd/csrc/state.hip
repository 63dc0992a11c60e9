// state.hip -- materialise device-resident pipeline state as the NumPy-shaped arrays
// the reference exposes as attributes (self.D, self.IJs, self.I, self.features,
// self.not_computed_mask, self.RefineApprox, ... -- annchor/annchor.py:191-530), so
// that user plugins and tests can read (and, for plugins, write) them.
#include "common.h"

int ann_download_D(annchor_ctx *c, double *dst);

__global__ void k_int2_to_i64(const int2 *__restrict__ in, int64_t n, int64_t *__restrict__ out)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) { out[2 * t] = in[t].x; out[2 * t + 1] = in[t].y; }
}
__global__ void k_i32_to_i64(const int32_t *__restrict__ in, int64_t n, int64_t *__restrict__ out)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = in[t];
}
__global__ void k_u8_to_i64(const uint8_t *__restrict__ in, int64_t n, int64_t *__restrict__ out)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = in[t] == 255 ? -1 : (int64_t)in[t];
}
__global__ void k_pack_features(const double *__restrict__ lb, const double *__restrict__ ub, const double *__restrict__ dad,
                                const uint8_t *__restrict__ anc, int64_t n, double *__restrict__ out)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) { out[4 * t] = lb[t]; out[4 * t + 1] = ub[t]; out[4 * t + 2] = dad[t]; out[4 * t + 3] = (double)anc[t]; }
}
__global__ void k_unpack_features(const double *__restrict__ in, int64_t n, double *__restrict__ lb, double *__restrict__ ub,
                                  double *__restrict__ dad, uint8_t *__restrict__ anc)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) { lb[t] = in[4 * t]; ub[t] = in[4 * t + 1]; dad[t] = in[4 * t + 2]; anc[t] = in[4 * t + 3] >= 1.0; }
}

static int field_elems(annchor_ctx *c, int32_t f, int64_t *n)
{
    switch (f) {
    case ANNCHOR_F_D: *n = c->nx * c->na; return ANNCHOR_OK;
    case ANNCHOR_F_A: *n = c->nA; return ANNCHOR_OK;
    case ANNCHOR_F_SID: *n = c->n > 0 ? c->nx * c->sid_nw : 0; return ANNCHOR_OK;   // (sid_nw words per point: 1 up to 64 anchors)
    case ANNCHOR_F_IJS: *n = 2 * c->n; return ANNCHOR_OK;
    case ANNCHOR_F_I_PTR: *n = c->n > 0 ? c->nx + 1 : 0; return ANNCHOR_OK;
    case ANNCHOR_F_I_IDX: *n = 2 * c->n; return ANNCHOR_OK;
    case ANNCHOR_F_FEATURES: *n = c->have_features ? 4 * c->n : 0; return ANNCHOR_OK;
    case ANNCHOR_F_NCM: *n = c->have_features ? c->n : 0; return ANNCHOR_OK;
    case ANNCHOR_F_RA: *n = c->have_RA ? c->n : 0; return ANNCHOR_OK;
    case ANNCHOR_F_LABELS: *n = c->have_RA ? c->n : 0; return ANNCHOR_OK;
    case ANNCHOR_F_THRESH: *n = c->thresh.p ? c->nx : 0; return ANNCHOR_OK;
    case ANNCHOR_F_PROB: *n = c->have_RA ? c->n : 0; return ANNCHOR_OK;
    case ANNCHOR_F_CAND: *n = c->ncand; return ANNCHOR_OK;
    case ANNCHOR_F_NEXT: *n = c->nnext; return ANNCHOR_OK;
    case ANNCHOR_F_DAD: *n = c->have_features ? c->n : 0; return ANNCHOR_OK;
    default: ann_set_err(c, "unknown field %d", f); return ANNCHOR_EINVAL;
    }
}

extern "C" int annchor_field_size(annchor_ctx *c, int32_t field, int64_t *n_elems)
{
    if (!c || !n_elems) return ANNCHOR_EINVAL;
    return field_elems(c, field, n_elems);
}

extern "C" int annchor_download(annchor_ctx *c, int32_t field, void *dst, int64_t n_elems)
{
    if (!c || !dst) return ANNCHOR_EINVAL;
    int64_t n = 0;
    ANN_TRY(field_elems(c, field, &n));
    ANN_REQUIRE(c, n == n_elems, ANNCHOR_EINVAL, "field %d has %lld elements, caller expects %lld", field, (long long)n,
                (long long)n_elems);
    if (n == 0) return ANNCHOR_OK;
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    const int B = 256;
    auto conv_i32 = [&](const int32_t *src) -> int {
        ANN_TRY(ann_reserve(c, c->stage_out, 8 * (size_t)n));
        k_i32_to_i64<<<ann_blocks(n, B), B, 0, c->stream>>>(src, n, c->stage_out.as<int64_t>());
        ANN_CHECK_HIP(c, hipGetLastError());
        return ann_d2h(c, dst, c->stage_out.p, 8 * (size_t)n);
    };
    switch (field) {
    case ANNCHOR_F_D: return ann_download_D(c, (double *)dst);
    case ANNCHOR_F_A: return conv_i32(c->A.as<int32_t>());
    case ANNCHOR_F_SID: return ann_d2h(c, dst, c->sid.p, 8 * (size_t)n);
    case ANNCHOR_F_IJS:
        ANN_TRY(ann_reserve(c, c->stage_out, 8 * (size_t)n));
        k_int2_to_i64<<<ann_blocks(c->n, B), B, 0, c->stream>>>(c->ij.as<int2>(), c->n, c->stage_out.as<int64_t>());
        ANN_CHECK_HIP(c, hipGetLastError());
        return ann_d2h(c, dst, c->stage_out.p, 8 * (size_t)n);
    case ANNCHOR_F_I_PTR: return ann_d2h(c, dst, c->Iptr.p, 8 * (size_t)n);
    case ANNCHOR_F_I_IDX: return conv_i32(c->Iidx.as<int32_t>());
    case ANNCHOR_F_FEATURES:
        ANN_TRY(ann_reserve(c, c->stage_out, 8 * (size_t)n));
        k_pack_features<<<ann_blocks(c->n, B), B, 0, c->stream>>>(c->lb.as<double>(), c->ub.as<double>(), c->dad.as<double>(),
                                                                 c->anc.as<uint8_t>(), c->n, c->stage_out.as<double>());
        ANN_CHECK_HIP(c, hipGetLastError());
        return ann_d2h(c, dst, c->stage_out.p, 8 * (size_t)n);
    case ANNCHOR_F_NCM: return ann_d2h(c, dst, c->ncm.p, (size_t)n);
    case ANNCHOR_F_RA: return ann_d2h(c, dst, c->RA.p, 8 * (size_t)n);
    case ANNCHOR_F_LABELS:
        ANN_TRY(ann_reserve(c, c->stage_out, 8 * (size_t)n));
        k_u8_to_i64<<<ann_blocks(n, B), B, 0, c->stream>>>(c->label.as<uint8_t>(), n, c->stage_out.as<int64_t>());
        ANN_CHECK_HIP(c, hipGetLastError());
        return ann_d2h(c, dst, c->stage_out.p, 8 * (size_t)n);
    case ANNCHOR_F_THRESH: return ann_d2h(c, dst, c->thresh.p, 8 * (size_t)n);
    case ANNCHOR_F_PROB: return ann_d2h(c, dst, c->prob.p, 8 * (size_t)n);
    case ANNCHOR_F_CAND: return conv_i32(c->cand.as<int32_t>());
    case ANNCHOR_F_NEXT: return conv_i32(c->next.as<int32_t>());
    case ANNCHOR_F_DAD: return ann_d2h(c, dst, c->dad.p, 8 * (size_t)n);
    }
    return ANNCHOR_EINVAL;
}

// Plugins may hand back modified arrays (custom samplers/regressors read features;
// tests inject reference state).  Supported: FEATURES, NCM, RA.
extern "C" int annchor_upload(annchor_ctx *c, int32_t field, const void *src, int64_t n_elems)
{
    if (!c || !src) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->have_features, ANNCHOR_EINVAL, "features not computed");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    const int64_t n = c->n;
    switch (field) {
    case ANNCHOR_F_FEATURES:
        ANN_REQUIRE(c, n_elems == 4 * n, ANNCHOR_EINVAL, "features: expected %lld elements", (long long)(4 * n));
        ANN_TRY(ann_reserve(c, c->stage_in, 32 * (size_t)n));
        ANN_TRY(ann_h2d(c, c->stage_in.p, src, 32 * (size_t)n));
        k_unpack_features<<<ann_blocks(n, 256), 256, 0, c->stream>>>(c->stage_in.as<double>(), n, c->lb.as<double>(),
                                                                    c->ub.as<double>(), c->dad.as<double>(), c->anc.as<uint8_t>());
        ANN_CHECK_HIP(c, hipGetLastError());
        return ANNCHOR_OK;
    case ANNCHOR_F_NCM:
        ANN_REQUIRE(c, n_elems == n, ANNCHOR_EINVAL, "ncm: expected %lld elements", (long long)n);
        c->n_unc = -1; c->sel_prepared = false;
        return ann_h2d(c, c->ncm.p, src, (size_t)n);
    case ANNCHOR_F_RA:
        ANN_REQUIRE(c, n_elems == n, ANNCHOR_EINVAL, "RA: expected %lld elements", (long long)n);
        ANN_TRY(ann_h2d(c, c->RA.p, src, 8 * (size_t)n));
        c->have_RA = true; c->sel_prepared = false;
        return ANNCHOR_OK;
    default: ann_set_err(c, "field %d is not uploadable", field); return ANNCHOR_EINVAL;
    }
}


// ------------------------------------------------------------------ to_sparse_matrix (f3)
// Annchor.to_sparse_matrix (reference annchor/annchor.py:625-641) fills a DOK matrix cell by cell:
//     for i: for (j, d) in row i of the graph:  D[i, j] = D[j, i] = d + eps
// (eps = nextafter(0, 1): explicit zeros survive).  A later assignment overwrites an earlier one, so
// when i lists j AND j lists i, both cells end up with the value of the listing of the LARGER row.
// Device form: one thread per listing (i, c); a listing is dead iff j > i and row j lists i; live
// listings emit (i, j, v) and (j, i, v) (once when j == i) at offsets from a prefix sum -- the
// symmetric matrix in COO form, each cell exactly once, in (i, c) order.
__global__ void k_sparse_count(const int64_t *__restrict__ idx, int64_t nx, int k, int32_t *__restrict__ cnt)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nx * k) return;
    const int64_t i = t / k;
    const int64_t j = idx[t];
    int out = 0;
    if (j >= 0 && j < nx) {
        bool dead = false;
        if (j > i)
            for (int c = 0; c < k; ++c) dead |= idx[j * k + c] == i;
        // a column listed twice in one row: the later listing wins (the loop order of the reference)
        for (int c = (int)(t - i * k) + 1; c < k; ++c) dead |= idx[i * k + c] == j;
        out = dead ? 0 : (j == i ? 1 : 2);
    }
    cnt[t] = out;
}

__global__ void k_sparse_emit(const int64_t *__restrict__ idx, const double *__restrict__ dist, int64_t nx, int k,
                              const int32_t *__restrict__ cnt, const int64_t *__restrict__ off, double eps,
                              int64_t *__restrict__ rows, int64_t *__restrict__ cols, double *__restrict__ vals)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nx * k) return;
    const int n = cnt[t];
    if (n == 0) return;
    const int64_t i = t / k, j = idx[t], o = off[t];
    const double v = dist[t] + eps;
    rows[o] = i; cols[o] = j; vals[o] = v;
    if (n == 2) { rows[o + 1] = j; cols[o + 1] = i; vals[o + 1] = v; }
}

extern "C" int annchor_graph_to_coo(annchor_ctx *c, const int64_t *ng_idx, const double *ng_dist, int64_t nx, int32_t k,
                                    int64_t *rows, int64_t *cols, double *vals, int64_t *nnz)
{
    if (!c || !ng_idx || !ng_dist || !rows || !cols || !vals || !nnz || nx < 0 || k < 1) return ANNCHOR_EINVAL;
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    const int64_t m = nx * k;
    *nnz = 0;
    if (m == 0) return ANNCHOR_OK;
    ANN_REQUIRE(c, m < (1ll << 31), ANNCHOR_ELIMIT, "graph of %lld cells is too large", (long long)m);
    void *d_idx = nullptr, *d_dist = nullptr, *d_cnt = nullptr, *d_off = nullptr, *d_out = nullptr;
    auto release = [&]() { (void)hipFree(d_idx); (void)hipFree(d_dist); (void)hipFree(d_cnt); (void)hipFree(d_off); (void)hipFree(d_out); };
    int rc = ANNCHOR_OK;
    do {
        if (hipMalloc(&d_idx, 8 * (size_t)m) != hipSuccess || hipMalloc(&d_dist, 8 * (size_t)m) != hipSuccess ||
            hipMalloc(&d_cnt, 4 * (size_t)m) != hipSuccess || hipMalloc(&d_off, 8 * (size_t)(m + 1)) != hipSuccess ||
            hipMalloc(&d_out, 24 * 2 * (size_t)m) != hipSuccess) { ann_set_err(c, "out of device memory"); rc = ANNCHOR_EHIP; break; }
        if ((rc = ann_h2d(c, d_idx, ng_idx, 8 * (size_t)m)) != ANNCHOR_OK) break;
        if ((rc = ann_h2d(c, d_dist, ng_dist, 8 * (size_t)m)) != ANNCHOR_OK) break;
        k_sparse_count<<<ann_blocks(m, 256), 256, 0, c->stream>>>((const int64_t *)d_idx, nx, k, (int32_t *)d_cnt);
        if ((rc = ann_exclusive_scan_i32_to_i64(c, (const int32_t *)d_cnt, (int64_t *)d_off, m)) != ANNCHOR_OK) break;
        int64_t *o_rows = (int64_t *)d_out, *o_cols = o_rows + 2 * m;
        double *o_vals = (double *)(o_cols + 2 * m);
        k_sparse_emit<<<ann_blocks(m, 256), 256, 0, c->stream>>>((const int64_t *)d_idx, (const double *)d_dist, nx, k, (const int32_t *)d_cnt,
                                                                (const int64_t *)d_off, 4.9406564584124654e-324, o_rows, o_cols, o_vals);
        if (hipGetLastError() != hipSuccess) { ann_set_err(c, "sparse emit launch failed"); rc = ANNCHOR_EHIP; break; }
        int64_t total = 0;
        if ((rc = ann_d2h(c, &total, (int64_t *)d_off + m, 8)) != ANNCHOR_OK) break;
        if (total > 0) {
            if ((rc = ann_d2h(c, rows, o_rows, 8 * (size_t)total)) != ANNCHOR_OK) break;
            if ((rc = ann_d2h(c, cols, o_cols, 8 * (size_t)total)) != ANNCHOR_OK) break;
            if ((rc = ann_d2h(c, vals, o_vals, 8 * (size_t)total)) != ANNCHOR_OK) break;
        }
        *nnz = total;
    } while (0);
    release();
    return rc;
}
