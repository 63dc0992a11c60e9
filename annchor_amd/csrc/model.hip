// model.hip -- the per-iteration models of Annchor.fit() fitted ON THE DEVICE, so that an iteration runs from the
// sampling step to the candidate selection without a host round trip:
//   annchor_fit_regression_device  SimpleStratifiedLinearRegression.fit (annchor/regressors.py:39-69) -- per partition
//                                  (lo < dad <= hi) ordinary least squares with intercept on [lb, ub, dad] -> y -- followed
//                                  by the fused predict / clip / merge / label pass (regressors.py:71-103,
//                                  annchor.py:356-380) from the coefficients where they are;
//   annchor_fit_errors_device      SimpleStratifiedErrorRegression.fit (annchor/error_predictors.py:26-53): per partition
//                                  (lo <= dad <= hi, both sides closed) the sorted residuals y - prediction.
// The reference solves each partition with LAPACK's SVD-based dgelsd on the centred samples (sklearn LinearRegression ->
// scipy.linalg.lstsq); here a partition is one workgroup running Householder QR on the same centred matrix in float64
// (backward stable: coefficients agree with dgelsd to ~cond * eps, 1e-13 relative on the bundled data -- tested at 1e-11;
// NOT bit-equal, which no two LAPACK builds are either).  Deterministic: rows are compacted in sample order and every
// reduction runs over a fixed tree, so two runs give the same bits.  A partition that is (numerically) rank deficient, or
// has fewer rows than columns, raises a status flag and the host redoes the step with dgelsd (min-norm solution).
#include "common.h"

#define OLS_T 256
#define ERR_CAP 8192   // residuals per partition the LDS sorter takes

int ann_dev_flags(annchor_ctx *c)
{
    ANN_TRY(ann_reserve(c, c->dev_flags, sizeof(int32_t) * 16));
    if (!c->dev_flags_clean) {
        ANN_CHECK_HIP(c, hipMemsetAsync(c->dev_flags.p, 0, sizeof(int32_t) * 16, c->stream));
        c->dev_flags_clean = true;
    }
    return ANNCHOR_OK;
}

// fixed-tree block sum (the same association for the same block size, whatever the data)
__device__ __forceinline__ double block_sum(double v, double *sh /*[OLS_T / 64]*/)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = sh[0];
    for (int w = 1; w < OLS_T / 64; ++w) s += sh[w];
    return s;
}

#define OLS_RCOND 1e-13   // relative singular value below which a direction of a rank-deficient partition is dropped
// One workgroup per partition.  A: this partition's scratch, double [4][cap] (columns lb, ub, dad, y of its rows).
struct ModelEdges { double e[MAXBINS + 1]; };
// (the model's header -- edges, partition count, statuses -- is written here too: workgroup 0 for the shared part, every
// workgroup for its own partition; k_model_init used to be a launch of its own in front)
__global__ __launch_bounds__(OLS_T) void k_ols_bins(const double *__restrict__ sfeat, const double *__restrict__ sy, int64_t m,
                                                   DeviceModel *__restrict__ dm, double *__restrict__ scratch, int64_t cap,
                                                   int32_t *__restrict__ flags, ModelEdges ed, int nb)
{
    __shared__ double sh[OLS_T / 64];
    __shared__ int wcnt[OLS_T / 64];
    __shared__ int base_sh;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double lo = ed.e[b], hi = ed.e[b + 1];
    if (tid == 0) dm->status[b] = 0;
    if (b == 0) {
        if (tid >= nb && tid < MAXBINS) { dm->status[tid] = 0; dm->rows[tid] = 0; }
        if (tid <= nb) dm->reg.e[tid] = ed.e[tid];
        if (tid == 0) { dm->reg.nb = nb; dm->err_status = 0; }
    }
    double *A = scratch + (size_t)b * 4 * cap;
    double *col[4] = {A, A + cap, A + 2 * cap, A + 3 * cap};
    // ---- rows of the partition, compacted in sample order (deterministic)
    if (tid == 0) base_sh = 0;
    __syncthreads();
    for (int64_t t0 = 0; t0 < m; t0 += OLS_T) {
        const int64_t t = t0 + tid;
        const double d = t < m ? sfeat[4 * t + 2] : 0.0;
        const bool in = t < m && d > lo && d <= hi;
        const unsigned long long bal = __ballot(in);
        if (lane == 0) wcnt[wave] = __popcll(bal);
        __syncthreads();
        int off = base_sh;
        for (int w = 0; w < wave; ++w) off += wcnt[w];
        if (in) {
            const int r = off + __popcll(bal & ((1ull << lane) - 1ull));
            col[0][r] = sfeat[4 * t]; col[1][r] = sfeat[4 * t + 1]; col[2][r] = d; col[3][r] = sy[t];
        }
        __syncthreads();
        if (tid == 0) base_sh += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        __syncthreads();
    }
    const int n = base_sh;
    if (tid == 0) dm->rows[b] = n;
    if (n < 3) {   // fewer rows than columns: the host decides (dgelsd's minimum-norm solution / the reference's error)
        if (tid == 0) {
            dm->status[b] = 2; dm->reg.w[b][0] = dm->reg.w[b][1] = dm->reg.w[b][2] = 0.0; dm->reg.c[b] = 0.0;
            atomicMax(&flags[1], 2);   // (sticky: a partition the device solver does not take)
        }
        return;
    }
    // ---- centre (sklearn LinearRegression(fit_intercept=True))
    double mean[4];
    for (int k = 0; k < 4; ++k) {
        double s = 0.0;
        for (int r = tid; r < n; r += OLS_T) s += col[k][r];
        mean[k] = block_sum(s, sh) / (double)n;
    }
    for (int k = 0; k < 4; ++k)
        for (int r = tid; r < n; r += OLS_T) col[k][r] -= mean[k];
    __syncthreads();
    // ---- Householder QR of [Xc | yc]: after step j column j is (.., R_jj, 0, ..) and rows j.. of the later columns hold Q^T
    double Rd[3], R01 = 0, R02 = 0, R12 = 0, qy[3];
    double cmax = 0.0;
    for (int j = 0; j < 3; ++j) {
        double s = 0.0;
        for (int r = j + tid; r < n; r += OLS_T) s += col[j][r] * col[j][r];
        const double sigma = block_sum(s, sh);
        const double ajj = col[j][j];
        const double alpha = ajj > 0.0 ? -sqrt(sigma) : sqrt(sigma);
        Rd[j] = alpha;
        cmax = fmax(cmax, fabs(alpha));
        // v = x; v[0] -= alpha; v^T v = sigma - 2 alpha a_jj + alpha^2 = 2 (sigma - alpha a_jj)
        const double vtv = 2.0 * (sigma - alpha * ajj);
        __syncthreads();
        if (tid == 0) col[j][j] = ajj - alpha;   // v[0]
        __syncthreads();
        for (int k = j + 1; k < 4; ++k) {
            double dsum = 0.0;
            for (int r = j + tid; r < n; r += OLS_T) dsum += col[j][r] * col[k][r];
            const double dot = block_sum(dsum, sh);
            const double tau = vtv > 0.0 ? 2.0 * dot / vtv : 0.0;
            for (int r = j + tid; r < n; r += OLS_T) col[k][r] -= tau * col[j][r];
            __syncthreads();
        }
        if (j == 0) { R01 = col[1][0]; R02 = col[2][0]; }
        if (j == 1) R12 = col[2][1];
        qy[j] = col[3][j];
    }
    if (tid == 0) {
        // numerically rank deficient (a constant or collinear feature inside the partition): leave it to dgelsd
        const double tol = 1e-10 * cmax;
        int st = 0;
        for (int j = 0; j < 3; ++j)
            if (!(fabs(Rd[j]) > tol)) st = 1;
        double w2 = 0, w1 = 0, w0 = 0;
        if (!st) {
            w2 = qy[2] / Rd[2];
            w1 = (qy[1] - R12 * w2) / Rd[1];
            w0 = ((qy[0] - R01 * w1) - R02 * w2) / Rd[0];
        } else {
            // Rank deficient inside the partition -- typically EXACT: among the closest pairs of Euclidean data both points
            // share their nearest anchor a*, which also gives the tightest upper bound, so ub = D[a*][i] + D[a*][j] = 2 dad
            // bit for bit.  What scipy's lstsq (dgelsd) returns there is the minimum-norm solution: the singular value
            // decomposition of the 3 x 3 factor R by one-sided Jacobi rotations, singular values below OLS_RCOND of the largest
            // dropped.  (dgelsd drops below machine epsilon; an exactly dependent column arrives at ~1e-16 of the largest in
            // either code, on one side or the other of that line by rounding luck, and when it lands above, the reference's
            // coefficients are ~1e15 and its predictions whatever the clip to [lb, ub] leaves.  The wider margin takes the
            // dependency for what it is.)  Restarting the whole fit with the host solver, as round 3 first did, doubled the
            // fit time of every Euclidean data set of this kind.
            double M[3][3] = {{Rd[0], R01, R02}, {0.0, Rd[1], R12}, {0.0, 0.0, Rd[2]}};
            double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
            bool conv = false;
            for (int sweep = 0; sweep < 40 && !conv; ++sweep) {
                conv = true;
                for (int p = 0; p < 2; ++p)
                    for (int q = p + 1; q < 3; ++q) {
                        double al = 0, be = 0, ga = 0;
                        for (int r = 0; r < 3; ++r) { al += M[r][p] * M[r][p]; be += M[r][q] * M[r][q]; ga += M[r][p] * M[r][q]; }
                        // (a column 1e-20 of the other is dropped by OLS_RCOND whatever it is orthogonal to; rotating against
                        // it would chase denormals)
                        if (ga == 0.0 || al <= 1e-40 * be || be <= 1e-40 * al || fabs(ga) <= 4.5e-16 * sqrt(al * be)) continue;
                        conv = false;
                        const double zeta = (be - al) / (2.0 * ga);
                        const double tt = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                        const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
                        for (int r = 0; r < 3; ++r) {
                            const double mp = M[r][p], mq = M[r][q];
                            M[r][p] = cs * mp - sn * mq; M[r][q] = sn * mp + cs * mq;
                            const double vp = V[r][p], vq = V[r][q];
                            V[r][p] = cs * vp - sn * vq; V[r][q] = sn * vp + cs * vq;
                        }
                    }
            }
            double sg2[3], smax2 = 0.0;
            for (int i = 0; i < 3; ++i) {
                sg2[i] = (M[0][i] * M[0][i] + M[1][i] * M[1][i]) + M[2][i] * M[2][i];
                smax2 = fmax(smax2, sg2[i]);
            }
            if (conv && smax2 > 0.0) {
                double w[3] = {0, 0, 0};
                for (int i = 0; i < 3; ++i)
                    if (sg2[i] > (OLS_RCOND * OLS_RCOND) * smax2) {
                        const double coef = ((M[0][i] * qy[0] + M[1][i] * qy[1]) + M[2][i] * qy[2]) / sg2[i];
                        for (int r = 0; r < 3; ++r) w[r] += coef * V[r][i];
                    }
                w0 = w[0]; w1 = w[1]; w2 = w[2];
                st = 0;
            } else if (smax2 == 0.0) st = 0;   // every feature constant inside the partition: the intercept alone (w = 0)
        }
        dm->status[b] = st;
        if (st) atomicMax(&flags[1], st);
        dm->reg.w[b][0] = w0; dm->reg.w[b][1] = w1; dm->reg.w[b][2] = w2;
        dm->reg.c[b] = mean[3] - ((mean[0] * w0 + mean[1] * w1) + mean[2] * w2);
    }
}

// bins: the partition edges (HOST, float64 [nb + 1]); the samples are the ones annchor_sample_pairs_device left on the
// device.  Enqueues the fit and the fused predict / clip / merge / label pass; no host wait.  A partition the device
// solver does not take (status != 0) raises dev_flags[1]; the caller reads it with annchor_model_status.
extern "C" int annchor_fit_regression_device(annchor_ctx *c, const double *bins, int32_t nb, int32_t first_iteration, int32_t is_metric)
{
    if (!c || !bins) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->have_features, ANNCHOR_EINVAL, "features not computed");
    ANN_REQUIRE(c, nb >= 1 && nb <= MAXBINS, ANNCHOR_ELIMIT, "1..%d partitions supported", MAXBINS);
    ANN_REQUIRE(c, first_iteration || c->have_RA, ANNCHOR_EINVAL, "RefineApprox not initialised");
    ANN_REQUIRE(c, c->nsamp > 0 && c->sfeat.p && c->sy.p, ANNCHOR_ESTATE, "no device-resident sample (annchor_sample_pairs_device)");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    const int64_t m = c->nsamp;
    ANN_TRY(ann_reserve(c, c->model, sizeof(DeviceModel)));
    ANN_TRY(ann_reserve(c, c->ols_scratch, sizeof(double) * 4 * (size_t)m * (size_t)nb));
    ANN_TRY(ann_dev_flags(c));
    DeviceModel *dm = c->model.as<DeviceModel>();
    ModelEdges ed;
    for (int k = 0; k <= MAXBINS; ++k) ed.e[k] = k <= nb ? bins[k] : 0.0;
    static_assert(MAXBINS + 1 <= OLS_T, "one thread per edge");
    {
        ProfScope ps(c, "ols_partitions", (double)m * 40.0 * nb);
        k_ols_bins<<<nb, OLS_T, 0, c->stream>>>(c->sfeat.as<double>(), c->sy.as<double>(), m, dm, c->ols_scratch.as<double>(), m,
                                                c->dev_flags.as<int32_t>(), ed, nb);
    }
    ANN_CHECK_HIP(c, hipGetLastError());
    c->model_fitted = true; c->model_nb = nb; c->errs_on_device = false; c->model_cache_valid = false;
    return ann_predict_merge_device(c, &dm->reg, first_iteration, is_metric);
}

// ---- residual lists: one workgroup per partition counts, compacts and sorts
// ascending order-preserving key of a double (NaNs last)
__device__ __forceinline__ unsigned long long err_key(double v) { return ann_key_asc(v); }

__global__ __launch_bounds__(1024) void k_err_sort(const double *__restrict__ sfeat, const double *__restrict__ sy,
                                                   const double *__restrict__ spred, int64_t m, DeviceModel *__restrict__ dm,
                                                   double *__restrict__ errs, int nb, int32_t *__restrict__ flags,
                                                   int64_t *__restrict__ errptr_out)
{
    extern __shared__ unsigned long long keys[];   // [P2]
    __shared__ int wcnt[16];
    __shared__ int base_sh;
    __shared__ int pcnt[MAXBINS];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double lo = dm->reg.e[b], hi = dm->reg.e[b + 1];
    // every partition's population (closed intervals: a sample on an inner edge counts for both neighbours), counted by every
    // workgroup for itself -- m samples x nb edges, a few microseconds; k_err_count used to be a launch of its own in front
    if (tid < MAXBINS) pcnt[tid] = 0;
    __syncthreads();
    for (int64_t t = tid; t < m; t += 1024) {
        const double d = sfeat[4 * t + 2];
        for (int q = 0; q < nb; ++q)
            if (d >= dm->reg.e[q] && d <= dm->reg.e[q + 1]) atomicAdd(&pcnt[q], 1);
    }
    __syncthreads();
    const int n = pcnt[b];
    if (tid == 0) dm->rows[b] = n;   // (rows is reused: the regression is done with it)
    // where this partition's list starts: the counts of the partitions before it; partition 0's workgroup also publishes the
    // offsets and the status (what a launch of its own, k_err_ptr, used to do)
    int64_t my_at = 0;
    for (int q = 0; q < b; ++q) my_at += pcnt[q];
    if (b == 0 && tid == 0) {
        int64_t at = 0;
        int st = 0;
        for (int q = 0; q < nb; ++q) {
            dm->errptr[q] = at;
            errptr_out[q] = at;
            const int64_t r = pcnt[q];
            if (r == 0) st = max(st, 1);
            if (r > ERR_CAP) st = max(st, 2);
            at += r;
        }
        dm->errptr[nb] = at;
        errptr_out[nb] = at;
        dm->err_status = st;
        if (st) flags[2] = st;
    }
    if (n == 0 || n > ERR_CAP) return;
    int P2 = 1;
    while (P2 < n) P2 <<= 1;
    for (int t = tid; t < P2; t += 1024) keys[t] = ~0ull;
    if (tid == 0) base_sh = 0;
    __syncthreads();
    for (int64_t t0 = 0; t0 < m; t0 += 1024) {
        const int64_t t = t0 + tid;
        const double d = t < m ? sfeat[4 * t + 2] : 0.0;
        const bool in = t < m && d >= lo && d <= hi;
        const unsigned long long bal = __ballot(in);
        if (lane == 0) wcnt[wave] = __popcll(bal);
        __syncthreads();
        int off = base_sh;
        for (int w = 0; w < wave; ++w) off += wcnt[w];
        if (in) keys[off + __popcll(bal & ((1ull << lane) - 1ull))] = err_key(sy[t] - spred[t]);
        __syncthreads();
        if (tid == 0) {
            int s = 0;
            for (int w = 0; w < 16; ++w) s += wcnt[w];
            base_sh += s;
        }
        __syncthreads();
    }
    for (int k = 2; k <= P2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < P2; t += 1024) {
                const int p = t ^ j;
                if (p > t) {
                    const unsigned long long a = keys[t], bb = keys[p];
                    const bool up = (t & k) == 0;
                    if ((a > bb) == up) { keys[t] = bb; keys[p] = a; }
                }
            }
            __syncthreads();
        }
    double *out = errs + my_at;
    for (int t = tid; t < n; t += 1024) out[t] = ann_key_asc_inv(keys[t]);
}

// After annchor_fit_regression_device: the residual lists of the same samples / partition edges, sorted, into the
// context's `errs` / `errptr` (what annchor_select_candidates reads when it is handed errs == NULL).  No host wait; an
// empty partition or one longer than the sorter takes raises dev_flags[2].
extern "C" int annchor_fit_errors_device(annchor_ctx *c)
{
    if (!c) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->model_fitted && c->nsamp > 0 && c->spred.p, ANNCHOR_ESTATE, "annchor_fit_regression_device first");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    const int nb = c->model_nb;
    const int64_t m = c->nsamp;
    DeviceModel *dm = c->model.as<DeviceModel>();
    // a sample on an inner edge counts for both neighbours: at most 2 m entries
    ANN_TRY(ann_reserve(c, c->errs, sizeof(double) * 2 * (size_t)m));
    ANN_TRY(ann_reserve(c, c->errptr, sizeof(int64_t) * (size_t)(MAXBINS + 1)));
    {
        ProfScope ps(c, "error_residual_lists", (double)m * 32.0 * nb);
        const size_t lds = sizeof(unsigned long long) * ERR_CAP;
        ANN_CHECK_HIP(c, hipFuncSetAttribute((const void *)k_err_sort, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_err_sort<<<nb, 1024, lds, c->stream>>>(c->sfeat.as<double>(), c->sy.as<double>(), c->spred.as<double>(), m, dm, c->errs.as<double>(),
                                                 nb, c->dev_flags.as<int32_t>(), c->errptr.as<int64_t>());
    }
    ANN_CHECK_HIP(c, hipGetLastError());
    c->errs_on_device = true; c->model_cache_valid = false;
    return ANNCHOR_OK;
}

// The fitted model for the host (waits): coefficients W float64 [nb][3], intercepts float64 [nb], per-partition solver
// status int32 [nb] (0 = solved on the device), err_ptr int64 [nb + 1] (NULL: skip; valid after annchor_fit_errors_device),
// flags int32 [3] = the sticky flags (sample step, regression, residual lists), cleared by this call.
extern "C" int annchor_model_download(annchor_ctx *c, double *W, double *cc, int32_t *status, int64_t *err_ptr, int32_t *flags)
{
    if (!c || !W || !cc || !status || !flags) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->model.p && c->model_nb > 0, ANNCHOR_ESTATE, "no model on this context");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    DeviceModel h;
    int32_t f[16];
    ANN_TRY(ann_dev_flags(c));
    ANN_TRY(ann_d2h2(c, &h, c->model.p, sizeof h, f, c->dev_flags.p, sizeof f));
    for (int b = 0; b < c->model_nb; ++b) {
        W[3 * b] = h.reg.w[b][0]; W[3 * b + 1] = h.reg.w[b][1]; W[3 * b + 2] = h.reg.w[b][2];
        cc[b] = h.reg.c[b];
        status[b] = h.status[b];
    }
    if (err_ptr)
        for (int b = 0; b <= c->model_nb; ++b) err_ptr[b] = h.errptr[b];
    flags[0] = f[0]; flags[1] = f[1]; flags[2] = f[2];
    if (f[0] | f[1] | f[2]) {
        ANN_CHECK_HIP(c, hipMemsetAsync(c->dev_flags.p, 0, sizeof(int32_t) * 16, c->stream));
    }
    return ANNCHOR_OK;
}

// The same three copies queued early -- annchor_neighbor_graph issues them before its kernel, into the tail of the pinned region
// its graph arrives in -- so that the end of a fit has ONE host wait (graph + model) instead of two.
int ann_model_prefetch_begin(annchor_ctx *c, unsigned char *at, size_t room, size_t *used)
{
    *used = 0;
    c->model_cache_valid = false;
    if (!(c->model_fitted && c->errs_on_device && c->model.p && c->errs.p && c->model_nb > 0)) return ANNCHOR_OK;
    const size_t eb = sizeof(double) * 2 * (size_t)c->nsamp;
    const size_t off_f = (sizeof(DeviceModel) + 63) & ~(size_t)63, off_e = off_f + 64;
    if (off_e + eb > room) return ANNCHOR_OK;
    ANN_TRY(ann_dev_flags(c));
    ANN_CHECK_HIP(c, hipMemcpyAsync(at, c->model.p, sizeof(DeviceModel), hipMemcpyDeviceToHost, c->stream));
    ANN_CHECK_HIP(c, hipMemcpyAsync(at + off_f, c->dev_flags.p, sizeof(int32_t) * 16, hipMemcpyDeviceToHost, c->stream));
    if (eb) ANN_CHECK_HIP(c, hipMemcpyAsync(at + off_e, c->errs.p, eb, hipMemcpyDeviceToHost, c->stream));
    c->model_cache_errs = eb;
    *used = off_e + eb;
    return ANNCHOR_OK;
}
void ann_model_prefetch_end(annchor_ctx *c, const unsigned char *at, size_t used)
{
    if (!used) return;
    c->model_cache.assign(at, at + used);
    c->model_cache_valid = true;
}

// annchor_model_download and annchor_errors_download behind ONE wait: the residual lists (at most 2 x n_samples doubles:
// a sample on a bin edge belongs to two partitions) are copied whole, *n_errs = err_ptr[nb] of them are meaningful.
extern "C" int annchor_model_download_with_errors(annchor_ctx *c, double *W, double *cc, int32_t *status, int64_t *err_ptr, int32_t *flags,
                                                  double *errs, int64_t errs_cap, int64_t *n_errs)
{
    if (!c || !W || !cc || !status || !flags || !err_ptr || !errs || !n_errs) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->model.p && c->model_nb > 0, ANNCHOR_ESTATE, "no model on this context");
    ANN_REQUIRE(c, c->errs_on_device && c->errs.p, ANNCHOR_ESTATE, "no device-resident residual lists");
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    size_t eb = sizeof(double) * (size_t)std::min<int64_t>(errs_cap, 2 * c->nsamp);
    const size_t off_f = (sizeof(DeviceModel) + 63) & ~(size_t)63, off_e = off_f + 64;
    const bool cached = c->model_cache_valid && c->model_cache.size() >= off_e + c->model_cache_errs;
    c->model_cache_valid = false;
    if (!cached && (!c->pin || off_e + eb > annchor_ctx::PIN_DL_BYTES)) {   // (large sample sets: the two-wait way)
        ANN_TRY(annchor_model_download(c, W, cc, status, err_ptr, flags));
        *n_errs = err_ptr[c->model_nb];
        ANN_REQUIRE(c, *n_errs <= errs_cap, ANNCHOR_EINVAL, "residual lists hold %lld entries, room for %lld", (long long)*n_errs, (long long)errs_cap);
        return annchor_errors_download(c, errs, *n_errs);
    }
    ANN_TRY(ann_dev_flags(c));
    const unsigned char *slot = c->pin + (size_t)annchor_ctx::PIN_SLOTS * annchor_ctx::PIN_SLOT_BYTES;
    if (cached) {   // fetched with the graph (ann_model_prefetch_*): no wait here
        slot = c->model_cache.data();
        eb = c->model_cache_errs;
    } else {
        unsigned char *dst = c->pin + (size_t)annchor_ctx::PIN_SLOTS * annchor_ctx::PIN_SLOT_BYTES;
        ANN_CHECK_HIP(c, hipMemcpyAsync(dst, c->model.p, sizeof(DeviceModel), hipMemcpyDeviceToHost, c->stream));
        ANN_CHECK_HIP(c, hipMemcpyAsync(dst + off_f, c->dev_flags.p, sizeof(int32_t) * 16, hipMemcpyDeviceToHost, c->stream));
        if (eb) ANN_CHECK_HIP(c, hipMemcpyAsync(dst + off_e, c->errs.p, eb, hipMemcpyDeviceToHost, c->stream));
        ANN_CHECK_HIP(c, ann_sync(c, __func__));
    }
    const DeviceModel &h = *reinterpret_cast<const DeviceModel *>(slot);
    const int32_t *f = reinterpret_cast<const int32_t *>(slot + off_f);
    for (int b = 0; b < c->model_nb; ++b) {
        W[3 * b] = h.reg.w[b][0]; W[3 * b + 1] = h.reg.w[b][1]; W[3 * b + 2] = h.reg.w[b][2];
        cc[b] = h.reg.c[b];
        status[b] = h.status[b];
    }
    for (int b = 0; b <= c->model_nb; ++b) err_ptr[b] = h.errptr[b];
    flags[0] = f[0]; flags[1] = f[1]; flags[2] = f[2];
    *n_errs = err_ptr[c->model_nb];
    if (*n_errs < 0 || (size_t)*n_errs * sizeof(double) > eb || *n_errs > errs_cap) *n_errs = -1;   // (a failed step: the flags say so; no list to hand out)
    else memcpy(errs, slot + off_e, sizeof(double) * (size_t)*n_errs);
    if (f[0] | f[1] | f[2]) ANN_CHECK_HIP(c, hipMemsetAsync(c->dev_flags.p, 0, sizeof(int32_t) * 16, c->stream));
    return ANNCHOR_OK;
}

// the sorted residuals of annchor_fit_errors_device (float64 [n_errs], n_errs = err_ptr[nb])
extern "C" int annchor_errors_download(annchor_ctx *c, double *errs, int64_t n_errs)
{
    if (!c || (n_errs > 0 && !errs)) return ANNCHOR_EINVAL;
    ANN_REQUIRE(c, c->errs_on_device && c->errs.p, ANNCHOR_ESTATE, "no device-resident residual lists");
    if (n_errs == 0) return ANNCHOR_OK;
    ANN_CHECK_HIP(c, hipSetDevice(c->device));
    return ann_d2h(c, errs, c->errs.p, sizeof(double) * (size_t)n_errs);
}
