"""
Regressors (reference annchor/regressors.py:18-103).  Protocol, unchanged:
    regression.fit(sample_features, feature_names, sample_y, sample_bins=None)
    regression.predict(features, feature_names) -> float64[n]

`SimpleStratifiedLinearRegression` fits 7 tiny OLS models on the host (7 x ~700 x 3)
and exposes them as `coefficients()`; inside `Annchor` its predict over all pairs is
the fused HIP kernel (predict + clip + merge + label).  `predict()` on a NumPy array
is kept for plugin compatibility and evaluates the same expression, in the same
floating-point order, with NumPy.
"""
import os

import numpy as np
import scipy.linalg


try:  # the LAPACK driver scipy.linalg.lstsq dispatches to, minus ~20 us of argument checking per call
    from scipy.linalg.lapack import dgelsd as _gelsd, dgelsd_lwork as _gelsd_lwork
except ImportError:  # pragma: no cover
    _gelsd = None


def _lstsq(A, b, cond):
    if _gelsd is None:
        return scipy.linalg.lstsq(A, b, cond=cond)[0]
    m, n = A.shape
    work, iwork, info = _gelsd_lwork(m, n, 1, cond)
    if info != 0:
        return scipy.linalg.lstsq(A, b, cond=cond)[0]
    x, _, _, info = _gelsd(A, b.reshape(-1, 1), int(work.real), int(iwork), cond, False, False)
    if info != 0:
        return scipy.linalg.lstsq(A, b, cond=cond)[0]
    return x[:n, 0]


def _ols(X, y):
    # sklearn LinearRegression(fit_intercept=True): centre, lstsq (gelsd), intercept
    xm, ym = X.mean(axis=0), y.mean()
    Xc, yc = X - xm, y - ym
    cond = max(Xc.shape) * np.finfo(np.float64).eps
    coef = _lstsq(Xc, yc, cond) if Xc.shape[0] >= Xc.shape[1] else scipy.linalg.lstsq(Xc, yc, cond=cond)[0]
    return coef, ym - xm @ coef


class SimpleStratifiedLinearRegression:
    def __init__(self, reg_feature_names=["lower bound", "upper bound", "double anchor distance"],
                 partition_feature_name="double anchor distance", n_partitions=7):
        self.n_partitions = n_partitions
        self.partition_feature_name = partition_feature_name
        self.reg_feature_names = reg_feature_names
        self.coef_ = None
        self.intercept_ = None

    def _columns(self, feature_names):
        return (feature_names.index(self.partition_feature_name),
                [i for i, name in enumerate(feature_names) if name in self.reg_feature_names])

    def fit(self, sample_features, feature_names, sample_y, sample_bins=None):
        i_part, i_features = self._columns(feature_names)
        F = sample_features[:, i_part]
        if sample_bins is None:
            n = F.shape[0]
            iq1, iq3 = int(n / 100), int(99 * n / 100)
            q1, q3 = np.partition(F, iq1)[iq1], np.partition(F, iq3)[iq3]
            self.sample_bins = np.hstack([-np.inf, np.linspace(q1, q3, self.n_partitions - 1), np.inf])
        else:
            self.n_partitions = sample_bins.shape[0] - 1
            self.sample_bins = sample_bins
        self.coef_ = np.zeros((self.n_partitions, len(i_features)))
        self.intercept_ = np.zeros(self.n_partitions)
        # bin of a sample: bins[b] < F <= bins[b + 1]  ==  number of interior edges below F.  One
        # stable sort groups the samples (original order kept inside a bin, so every sum runs
        # over the same numbers in the same order as with a boolean mask per bin)
        b = np.searchsorted(self.sample_bins[1:-1], F, side="left")
        if b.size < 2 or bool(np.all(b[1:] >= b[:-1])):
            # the stratified sampler hands its samples over bin by bin: already grouped (a stable
            # sort of a sorted key is the identity), unless a value sits exactly on an edge
            Xs, ys, bs = sample_features[:, i_features], sample_y, b
        else:
            order = np.argsort(b, kind="stable")
            Xs, ys, bs = sample_features[order][:, i_features], sample_y[order], b[order]
        cuts = np.searchsorted(bs, np.arange(self.n_partitions + 1), side="left")
        batch = self._batched(Xs, ys, cuts)
        for nbin in range(self.n_partitions):
            lo, hi = cuts[nbin], cuts[nbin + 1]
            if batch is not None and batch[3][nbin] == 0:
                coef, xm, ym = batch[0][nbin], batch[1][nbin], batch[2][nbin]
                self.coef_[nbin], self.intercept_[nbin] = coef, ym - xm @ coef
            else:
                self.coef_[nbin], self.intercept_[nbin] = _ols(Xs[lo:hi], ys[lo:hi])

    @staticmethod
    def _batched(Xs, ys, cuts):
        """All partitions' centring + dgelsd in one native call (annchor_ols_bins: the same LAPACK
        routine and the same summation orders as _ols, bit for bit -- tests/test_host_logic.py);
        None when the library or scipy's LAPACK pointer is not there (the loop above then runs
        _ols per partition)."""
        if _gelsd is None or os.environ.get("ANNCHOR_OLS_PYTHON"):
            return None
        try:
            from . import _native
            return _native.ols_bins(Xs, ys, cuts)
        except Exception:  # noqa: BLE001 -- e.g. the shared library has not been built
            return None

    def coefficients(self):
        """(bins [nb+1], W [nb,3], c [nb]) for the fused device predict; None if the
        feature selection is not the default (lb, ub, dad)."""
        if list(self.reg_feature_names) != ["lower bound", "upper bound", "double anchor distance"] or \
                self.partition_feature_name != "double anchor distance":
            return None
        return self.sample_bins, self.coef_, self.intercept_

    def predict(self, features, feature_names):
        i_part, i_features = self._columns(feature_names)
        X, F = features[:, i_features], features[:, i_part]
        y = np.zeros(X.shape[0])
        for nbin in range(self.n_partitions):
            mask = (F > self.sample_bins[nbin]) * (F <= self.sample_bins[nbin + 1])
            acc = np.zeros(int(mask.sum()))
            for k in range(X.shape[1]):
                acc = acc + self.coef_[nbin, k] * X[mask, k]
            y[mask] = acc + self.intercept_[nbin]
        return y
