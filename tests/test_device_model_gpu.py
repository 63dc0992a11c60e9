"""The iteration's models fitted on the device (csrc/model.hip: per-partition OLS by Householder QR, residual lists)
against the host path (scipy's dgelsd, the reference's solver: annchor/regressors.py:39-69,
annchor/error_predictors.py:26-53) and against the oracle."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(__file__))

from oracle import metrics as om  # noqa: E402


def _strings():
    return np.array(om.load_strings()[0])


def _check_same_model(a, b):
    """a: ols='device', b: ols='lapack' after fit(): same samples, coefficients to 1e-11 relative, same residual lists
    to 1e-9 absolute, same graph."""
    assert np.array_equal(a.sample_ixs, b.sample_ixs)
    assert np.array_equal(a.sample_features, b.sample_features) and np.array_equal(a.sample_y, b.sample_y)
    assert np.array_equal(a.regression.sample_bins, b.regression.sample_bins)
    np.testing.assert_allclose(a.regression.coef_, b.regression.coef_, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(a.regression.intercept_, b.regression.intercept_, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(a.sample_predict, b.sample_predict, rtol=0, atol=1e-9)
    assert list(a.error_predictor.labels) == list(b.error_predictor.labels)
    for lab in b.error_predictor.labels:
        np.testing.assert_allclose(a.error_predictor.errs[lab], b.error_predictor.errs[lab], rtol=0, atol=1e-9)
    assert a.evals == b.evals


@pytest.mark.parametrize("cfg", [dict(n_anchors=8, n_neighbors=10, n_samples=700, p_work=0.3, niters=2),
                                 dict(n_anchors=12, n_neighbors=6, n_samples=300, p_work=0.2, niters=3)])
def test_device_model_equals_lapack_model_strings_small(cfg):
    from annchor_amd import Annchor

    X = _strings()[::5]
    a = Annchor(X, "levenshtein", ols="device", **cfg).fit()
    b = Annchor(X, "levenshtein", ols="lapack", **cfg).fit()
    # (with these few samples a partition can be collinear: the device then refuses and fit() starts over on dgelsd --
    # either way the models must agree)
    assert not b.__dict__.get("_model_on_device")
    _check_same_model(a, b)
    assert np.array_equal(a.neighbor_graph[1], b.neighbor_graph[1]) and np.array_equal(a.neighbor_graph[0], b.neighbor_graph[0])


def test_device_model_c2_graph_equals_lapack_graph_and_oracle():
    """BASELINE configs[1]: the default (device-fitted models) gives the graph of the LAPACK path -- which is the
    oracle's, bit for bit."""
    from annchor_amd import Annchor

    X = _strings()
    cfg = dict(n_anchors=15, n_neighbors=25, p_work=0.12, random_seed=42)
    a = Annchor(X, "levenshtein", **cfg).fit()
    b = Annchor(X, "levenshtein", ols="lapack", **cfg).fit()
    assert a.ols == "device" and a._model_on_device
    _check_same_model(a, b)
    assert np.array_equal(a.neighbor_graph[1], b.neighbor_graph[1]) and np.array_equal(a.neighbor_graph[0], b.neighbor_graph[0])
    Go = np.load(os.path.join(os.path.dirname(__file__), "golden", "strings_full_oracle.npz"))
    assert np.array_equal(a.neighbor_graph[1], Go["c1_ng_dist"].astype(np.float64))
    assert np.array_equal(a.neighbor_graph[0], Go["c1_ng_idx"].astype(np.int64))


def test_device_model_euclid_and_digits():
    from annchor_amd import Annchor
    from annchor_amd.datasets import load_digits

    rng = np.random.default_rng(3)
    X = (rng.standard_normal((1500, 5)) @ rng.standard_normal((5, 24))).astype(np.float64)
    cfg = dict(n_anchors=10, n_neighbors=8, n_samples=1000, p_work=0.15)
    a = Annchor(X, "euclidean", **cfg).fit()
    b = Annchor(X, "euclidean", ols="lapack", **cfg).fit()
    _check_same_model(a, b)
    np.testing.assert_allclose(a.neighbor_graph[1], b.neighbor_graph[1], rtol=0, atol=0)
    D = load_digits()
    Xd = D["X"][:600]
    cfg = dict(func_kwargs={"cost_matrix": D["cost_matrix"]}, n_anchors=10, n_neighbors=10, n_samples=1500, p_work=0.2)
    a = Annchor(Xd, "wasserstein", **cfg).fit()
    b = Annchor(Xd, "wasserstein", ols="lapack", **cfg).fit()
    _check_same_model(a, b)
    np.testing.assert_allclose(a.neighbor_graph[1], b.neighbor_graph[1], rtol=0, atol=0)


def test_rank_deficient_partition_minimum_norm_on_the_device():
    """Integer grid data: inside a partition every sample can have the same double anchor distance (a constant
    column after centring), or ub = 2 dad exactly -- the QR sees the dependency and the partition gets dgelsd's answer, the
    minimum-norm solution, from the singular value decomposition of the 3 x 3 factor, on the device: no restart (one pass
    through the stages), coefficients equal to the LAPACK path's, the same graph."""
    from annchor_amd import Annchor

    g = np.arange(12, dtype=np.float64)
    X = np.stack(np.meshgrid(g, g), axis=-1).reshape(-1, 2)     # 144 grid points, massive ties
    X = np.concatenate([X, X[:40]])                             # + duplicates
    cfg = dict(n_anchors=4, n_neighbors=5, n_samples=300, p_work=0.5, locality=3)
    a = Annchor(X, "euclidean", **cfg).fit()
    b = Annchor(X, "euclidean", ols="lapack", **cfg).fit()
    assert getattr(a, "_device_model_refused", None) is None and a._model_on_device
    np.testing.assert_allclose(a.neighbor_graph[1], b.neighbor_graph[1], rtol=0, atol=0)
    np.testing.assert_allclose(a.regression.coef_, b.regression.coef_, rtol=0, atol=1e-9)
    np.testing.assert_allclose(a.regression.intercept_, b.regression.intercept_, rtol=0, atol=1e-9)


def test_exactly_dependent_bounds_do_not_restart_the_fit():
    """Continuous Euclidean data: among the closest pairs both points share their nearest anchor, which also gives the
    tightest upper bound -- ub = 2 dad bit for bit inside the first partition.  The fit runs its stages once and agrees with
    the LAPACK path on (nearly) every neighbour."""
    from annchor_amd import Annchor, compare_neighbor_graphs

    rng = np.random.default_rng(5)
    Z = rng.standard_normal((4000, 6))
    X = (Z @ rng.standard_normal((6, 48)) + 0.05 * rng.standard_normal((4000, 48))).astype(np.float64)
    cfg = dict(n_anchors=24, n_neighbors=15, p_work=0.1, n_samples=5000)
    a = Annchor(X, "euclidean", **cfg).fit()
    assert getattr(a, "_device_model_refused", None) is None and a._model_on_device
    b = Annchor(X, "euclidean", ols="lapack", **cfg).fit()
    assert compare_neighbor_graphs(a.neighbor_graph, b.neighbor_graph, 15) <= 0.001 * 4000 * 15


def test_host_waits_per_c2_fit():
    """The host waits for the stream 7 times in a C2 fit (was 13 before the sampling statistics were chained on the device,
    the cut values stayed there and the not-computed count rode with the locality download; 8 before the fitted model rode with the
    graph): locality sizes, sampling statistics x 2, guarantee_nmin flags, selection state x 2, the graph + the fitted model."""
    import os, subprocess, sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "wait_census.py")], cwd=root, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stderr.splitlines()
    fit = lines[lines.index("=== fit") + 1:lines.index("=== end")]
    waits = [ln for ln in fit if ln.startswith("sync ")]
    assert len(waits) <= 7, fit


def test_many_samples_take_the_host_models_before_the_iteration(monkeypatch):
    """More samples per partition than the device's residual sorter holds (ERR_CAP = 8192): decided from the sampling step's
    own numbers BEFORE the iteration -- the models of that fit are fitted on the host -- instead of running the whole fit
    on unsorted residual lists and starting over.  Same graph as ols='lapack'; and with the limit lowered so that the
    C2 configuration trips it, no restart happens either."""
    import annchor_amd.annchor as A
    from annchor_amd import Annchor

    X = _strings()
    cfg = dict(n_anchors=10, n_neighbors=15, n_samples=60000, p_work=0.2, random_seed=42)
    a = Annchor(X, "levenshtein", **cfg)
    restarts = []
    orig = A.Annchor._fit_stages
    monkeypatch.setattr(A.Annchor, "_fit_stages", lambda self, *args: (restarts.append(1), orig(self, *args))[1])
    a.fit()
    b = Annchor(X, "levenshtein", ols="lapack", **cfg).fit()
    assert a._device_models_off and not a.__dict__.get("_model_on_device")
    assert len(restarts) == 2   # one pass through the stages per fit: no restart
    assert np.array_equal(a.neighbor_graph[0], b.neighbor_graph[0]) and np.array_equal(a.neighbor_graph[1], b.neighbor_graph[1])
    monkeypatch.setattr(A, "DEVICE_MODEL_MAX_PER_BIN", 500)
    del restarts[:]
    c = Annchor(X, "levenshtein", n_anchors=15, n_neighbors=25, p_work=0.12, random_seed=42).fit()   # 5000 samples: 715 per bin
    d = Annchor(X, "levenshtein", n_anchors=15, n_neighbors=25, p_work=0.12, random_seed=42, ols="lapack").fit()
    assert c._device_models_off and len(restarts) == 2
    assert np.array_equal(c.neighbor_graph[1], d.neighbor_graph[1])
