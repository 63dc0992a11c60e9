// hostols.hip -- host-side (no device code): the per-bin ordinary least squares of
// SimpleStratifiedLinearRegression.fit (reference annchor/regressors.py:60-84: one sklearn
// LinearRegression per partition = centre, LAPACK dgelsd, intercept), batched.
//
// The Python restatement (annchor_amd/regressors.py) spends ~45 us per bin in interpreter and
// wrapper overhead around a ~8 us LAPACK call; seven bins, twice per fit, with the GPU idle.
// This file does the centring and calls the SAME dgelsd (scipy's LAPACK, handed in as a function
// pointer taken from scipy.linalg.cython_lapack) with the same workspace query, so that the
// coefficients are the bits the Python path produces:
//   * column means / the target mean are NumPy's pairwise sums (numpy/_core/src/umath/
//     loops_utils.h.src: 8 accumulators up to 128 elements, halves rounded down to a multiple of
//     8 above) divided by the count -- what ndarray.mean does along a contiguous axis;
//   * A = X - mean, b = y - mean elementwise; rcond = max(m, n) * eps; x = first n rows of b.
// tests/test_host_logic.py compares it with the Python path bit for bit.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/annchor_hip.h"

namespace {
typedef void (*dgelsd_fn)(int *m, int *n, int *nrhs, double *a, int *lda, double *b, int *ldb, double *s, double *rcond, int *rank,
                          double *work, int *lwork, int *iwork, int *info);

double pairwise_sum(const double *a, int64_t n)
{
    if (n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
}
}  // namespace

// X: nf columns of `ld` doubles each (column k at X + k * ld), y: ld doubles; bin b = rows
// cuts[b] .. cuts[b + 1].  Out per bin: coef[b * nf ..], xmean[b * nf ..], ymean[b], status[b]
// (0 = solved; 1 = fewer rows than features: the caller's general path; > 1 = LAPACK info).
extern "C" int annchor_ols_bins(void *dgelsd_ptr, const double *X, const double *y, int64_t ld, int32_t nf, const int64_t *cuts,
                                int32_t nbins, double *coef, double *xmean, double *ymean, int32_t *status)
{
    if (!dgelsd_ptr || !X || !y || !cuts || !coef || !xmean || !ymean || !status || nf < 1 || nbins < 0) return ANNCHOR_EINVAL;
    dgelsd_fn gelsd = reinterpret_cast<dgelsd_fn>(dgelsd_ptr);
    std::vector<double> A, B, S((size_t)nf), work;
    std::vector<int> iwork;
    for (int b = 0; b < nbins; ++b) {
        const int64_t lo = cuts[b], hi = cuts[b + 1];
        if (lo < 0 || hi < lo || hi > ld) return ANNCHOR_EINVAL;
        const int64_t m64 = hi - lo;
        status[b] = 1;
        for (int k = 0; k < nf; ++k) { coef[(size_t)b * nf + k] = 0.0; xmean[(size_t)b * nf + k] = 0.0; }
        ymean[b] = 0.0;
        if (m64 < nf || m64 > 0x7fffffff) continue;
        int m = (int)m64, n = nf, nrhs = 1, lda = m, ldb = m > n ? m : n, rank = 0, info = 0;
        A.resize((size_t)m * n);
        B.assign((size_t)ldb, 0.0);
        for (int k = 0; k < n; ++k) {
            const double *col = X + (size_t)k * ld + lo;
            const double mu = pairwise_sum(col, m) / (double)m;
            xmean[(size_t)b * nf + k] = mu;
            for (int i = 0; i < m; ++i) A[(size_t)k * m + i] = col[i] - mu;
        }
        const double ym = pairwise_sum(y + lo, m) / (double)m;
        ymean[b] = ym;
        for (int i = 0; i < m; ++i) B[(size_t)i] = y[lo + i] - ym;
        double rcond = (double)(m > n ? m : n) * 2.220446049250313e-16;
        // workspace query, as scipy.linalg.lapack.dgelsd_lwork does it
        double wq = 0.0;
        int iwq = 0, lwork = -1;
        gelsd(&m, &n, &nrhs, A.data(), &lda, B.data(), &ldb, S.data(), &rcond, &rank, &wq, &lwork, &iwq, &info);
        if (info != 0) { status[b] = 2; continue; }
        lwork = (int)wq;
        if ((int)work.size() < lwork + 1) work.resize((size_t)lwork + 1);
        if ((int)iwork.size() < iwq + 1) iwork.resize((size_t)(iwq > 0 ? iwq : 1) + 1);
        gelsd(&m, &n, &nrhs, A.data(), &lda, B.data(), &ldb, S.data(), &rcond, &rank, work.data(), &lwork, iwork.data(), &info);
        if (info != 0) { status[b] = 2 + (info > 0 ? info : 0); continue; }
        for (int k = 0; k < n; ++k) coef[(size_t)b * nf + k] = B[(size_t)k];
        status[b] = 0;
    }
    return ANNCHOR_OK;
}
