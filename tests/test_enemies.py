"""Nearest-enemy graph / selective subset / alpha-RSS (SURVEY.md section 8, row f4;
reference annchor/annchor.py:685-927, tests/test_examples.py:61-85).

tests/golden/enemies_state.npz holds the reference's complete post-fit state on a 400-point
two-moons set and its outputs from there; tests/golden/enemies.npz the reference's end-to-end
outputs on the two data sets of its own test (make_golden.py: gen_enemies_state, gen_enemies)."""
import os

import numpy as np
import pytest

from oracle import annchor_oracle as O
from oracle import metrics as om

GOLD = os.path.join(os.path.dirname(__file__), "golden")
NAMES = ["lower bound", "upper bound", "double anchor distance", "is anchor"]


@pytest.fixture(scope="module")
def state():
    return np.load(os.path.join(GOLD, "enemies_state.npz"))


def oracle_from_state(g):
    X = g["X"]
    o = O.OracleAnnchor.__new__(O.OracleAnnchor)
    o.nx = len(X)
    o.metric_pairs = lambda IJ: om.euclidean_pairs(X, np.asarray(IJ, dtype=np.int64))
    for k in ("D", "A", "sid", "IJs", "I_ptr", "I_idx", "features", "RA", "bins", "W", "c", "ncm"):
        setattr(o, k, g[k].copy())
    o.loc_thresh = int(g["loc_thresh"])
    o.neighbor_graph = (g["ng_idx"].copy(), g["ng_dist"].copy())
    return o


@pytest.mark.parametrize("trunc", [False, True])
def test_oracle_nearest_enemies_from_reference_state(state, trunc):
    g = state
    o = oracle_from_state(g)
    ni, nd = O.nearest_enemies(o, g["y"], nn=3, loc_min=80, reference_truncation=trunc)
    n0 = len(g["IJs"])
    assert np.array_equal(o.IJs[n0:], g["ne_IJs_new"])
    assert np.array_equal(o.ncm, g["ne_ncm"])
    assert np.allclose(o.RA, g["ne_RA"], rtol=0, atol=1e-12)  # OLS predict: sklearn vs restatement, last bits
    assert np.array_equal(ni, g["ne_idx"]) and np.array_equal(nd, g["ne_dist"])
    for a in (0, 0.2):
        assert np.array_equal(O.selective_subset(o, g["y"], alpha=a), g["ss_a%g" % a])
    # the scan order among equal nearest-enemy distances (mutual enemies) is an unstable argsort
    # in the reference; membership does not depend on it
    assert set(O.alpha_rss(o, g["y"])) == set(g["alpha_rss"])
    assert set(O.alpha_rss(o, g["y"], alpha=0.2)) == set(g["alpha_rss_a0.2"])


@pytest.mark.gpu
def test_device_nearest_enemies_from_reference_state(state):
    """The reference's complete post-fit state (tests/golden/enemies_state.npz: anchors, anchor distances, RefineApprox,
    not_computed_mask, regression coefficients) injected into an engine; the device's nearest-enemy stages
    (csrc/enemies.hip) must then give the REFERENCE's own outputs: the new pairs, the updated mask / distances, the
    graph, and -- through the host set covers -- the selective subsets."""
    from annchor_amd import Annchor
    from annchor_amd import _native
    from annchor_amd.regressors import SimpleStratifiedLinearRegression

    g = state
    X, y = g["X"], g["y"]
    ann = Annchor(X, "euclidean", n_anchors=20, n_neighbors=6, n_samples=1000, p_work=0.2, locality=3)
    eng = ann._engine
    eng.set_anchor_distances(g["D"], g["A"])
    n, _ = eng.build_locality(3, int(g["loc_thresh"]), int(ann.loc_min))
    ann.n_pairs = n
    eng.compute_features()
    assert np.array_equal(eng.download(_native.F_IJS).reshape(-1, 2), g["IJs"])        # same candidate list as the reference
    assert np.array_equal(eng.download(_native.F_FEATURES).reshape(-1, 4)[:, 2:], g["features"][:, 2:])   # dad, anchor flag
    eng.upload(_native.F_FEATURES, g["features"])     # (bounds as update_anchor_points left them at the end of the reference's fit)
    eng.upload(_native.F_RA, g["RA"])
    eng.upload(_native.F_NCM, g["ncm"].astype(np.uint8))
    r = SimpleStratifiedLinearRegression()
    r.sample_bins, r.coef_, r.intercept_, r.n_partitions = g["bins"], g["W"], g["c"], len(g["c"])
    ann.regression, ann.neighbor_graph = r, (g["ng_idx"], g["ng_dist"])
    ann._cache["A"] = g["A"]
    ann.get_nearest_enemies(y, nn=3, loc_min=80)
    ni, nd = ann.nearest_enemy_graph
    assert np.array_equal(ann.IJs[len(g["IJs"]):], g["ne_IJs_new"])
    assert np.array_equal(ann.not_computed_mask, g["ne_ncm"])
    assert np.allclose(ann.RefineApprox, g["ne_RA"], rtol=0, atol=1e-12)   # OLS predict: sklearn vs this build, last bits
    assert np.array_equal(ni, g["ne_idx"]) and np.array_equal(nd, g["ne_dist"])
    for al in (0, 0.2):
        assert np.array_equal(ann.annchor_selective_subset(y, alpha=al), g["ss_a%g" % al])
    assert set(ann.alpha_rss(y)) == set(g["alpha_rss"])
    assert set(ann.alpha_rss(y, alpha=0.2)) == set(g["alpha_rss_a0.2"])
    with pytest.raises(AssertionError):
        ann.get_nearest_enemies(np.zeros(len(X)), nn=3)  # one label only
    with pytest.raises(Exception, match="distance zero"):
        ann.annchor_selective_subset(y, dne=np.zeros(len(X)))
    with pytest.raises(RuntimeError):
        ann.get_sample()    # the stage methods are closed once the list is extended


@pytest.mark.gpu
def test_device_nearest_enemies_host_metric_and_custom_regression():
    """The same stages with a Python metric (the todo pairs go through get_exact_ijs on the host) and with a custom
    regression object (its predict() sees the new pairs' feature rows): same graph as the device-metric run."""
    from annchor_amd import Annchor
    from annchor_amd.regressors import SimpleStratifiedLinearRegression

    g = np.load(os.path.join(GOLD, "enemies.npz"))
    X, y = g["moons_X"], g["moons_y"]
    cfg = dict(n_neighbors=15, p_work=0.2)
    a = Annchor(X, "euclidean", **cfg).fit()
    a.get_nearest_enemies(y)

    class MyReg(SimpleStratifiedLinearRegression):
        pass

    def py_metric(u, v):
        return float(np.sqrt(((u - v) ** 2).sum()))

    b = Annchor(X, "euclidean", regression=MyReg(), ols="lapack", **cfg).fit()
    b.get_nearest_enemies(y)
    assert np.array_equal(a.nearest_enemy_graph[0], b.nearest_enemy_graph[0])
    np.testing.assert_allclose(a.nearest_enemy_graph[1], b.nearest_enemy_graph[1], rtol=1e-12, atol=0)
    sub = slice(0, 300)
    c = Annchor(X[sub], "euclidean", **cfg).fit()
    d = Annchor(X[sub], py_metric, **cfg).fit()
    c.get_nearest_enemies(y[sub]); d.get_nearest_enemies(y[sub])
    np.testing.assert_allclose(c.nearest_enemy_graph[1], d.nearest_enemy_graph[1], rtol=1e-9, atol=0)


@pytest.mark.gpu
def test_device_nearest_enemies_20000_points():
    """N = 20 000 (2 x 10^8 fitted candidate pairs): the nearest-enemy graph entirely on the device; truth = exact
    nearest enemies of 400 rows by brute force.  (The oracle's dense nx x nx restatement does not reach this size.)"""
    from annchor_amd import Annchor
    from annchor_amd.samplers import DeviceStratifiedSampler

    rng = np.random.default_rng(12)
    n = 20000
    cent = rng.uniform(-6, 6, (12, 16))
    lab = rng.integers(0, 12, n)
    X = (cent[lab] + rng.standard_normal((n, 16))).astype(np.float64)
    y = lab % 3                                        # three labels, four blobs each
    ann = Annchor(X, "euclidean", n_anchors=20, n_neighbors=10, p_work=0.03).fit()   # default arguments
    assert type(ann.sampler) is DeviceStratifiedSampler   # (2 x 10^8 candidate pairs: the automatic choice)
    ni, nd = ann.get_nearest_enemies(y, nn=3) or ann.nearest_enemy_graph
    assert ni.shape == (n, 3) and np.all(y[ni] != y[:, None]) and np.all(np.diff(nd, axis=1) >= 0)
    rows = rng.choice(n, 400, replace=False)
    good = 0
    for r in rows:
        d = np.sqrt(((X - X[r]) ** 2).sum(axis=1))
        d[y == y[r]] = np.inf
        good += abs(nd[r, 0] - d.min()) <= 1e-9 * (1 + d.min())
        dd = np.sqrt(((X[ni[r]] - X[r]) ** 2).sum(axis=1))   # reported distances are the reported pairs' distances
        np.testing.assert_allclose(dd, nd[r], rtol=1e-12, atol=0)
    # (the reference's heuristic -- exact distances only for each row's 50 closest-LOOKING enemies -- finds the true nearest
    # enemy for ~87 % of the rows of this set; the device follows it, it does not improve on it)
    assert good >= 0.8 * len(rows), good
    ann._engine.close()


def test_oracle_end_to_end_close_to_reference():
    """End to end on the reference test's blobs (1000 points): the restated fit() ends in a
    slightly different state than the reference run (whose update_anchor_points hit its 10 s
    cutoff under the stand-in numba), so compare outcomes, not bits."""
    g = np.load(os.path.join(GOLD, "enemies.npz"))
    X, y = g["blobs_X"], g["blobs_y"]
    o = O.OracleAnnchor(len(X), lambda IJ: om.euclidean_pairs(X, np.asarray(IJ, dtype=np.int64)),
                        n_neighbors=15, p_work=0.2).fit()
    assert np.array_equal(o.neighbor_graph[1], g["blobs_ng_dist"])
    ni, nd = O.nearest_enemies(o, y)
    assert (ni == g["blobs_ne_idx"]).all(axis=1).mean() >= 0.995
    assert len(o.IJs) == int(g["blobs_n_pairs_ne"])
    ss = O.selective_subset(o, y, alpha=0)
    # the subset follows the fitted graph: which of the equally probable pairs were refined moves its size by a few
    # points (reference under numba's RNG: 90; under the stand-in: 89; here: 92)
    assert abs(len(ss) - len(g["blobs_ss_a0"])) <= 4 and len(set(ss) & set(g["blobs_ss_a0"])) >= 0.9 * len(ss)
    assert set(O.alpha_rss(o, y)) == set(g["blobs_alpha_rss"])


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["blobs", "moons"])
def test_selective_subset_gpu_matches_oracle(tag):
    """reference tests/test_examples.py:61-85 on the GPU build: same calls, results equal to the
    restatement's (and sizes as the reference produced for the same RNG stream: 89 / 12; the
    reference test's 90 / 16 belong to numba's stream)."""
    from annchor_amd import Annchor

    g = np.load(os.path.join(GOLD, "enemies.npz"))
    X, y = g[tag + "_X"], g[tag + "_y"]
    ann = Annchor(X, "euclidean", n_neighbors=15, p_work=0.2)
    ann.fit()
    o = O.OracleAnnchor(len(X), lambda IJ: om.euclidean_pairs(X, np.asarray(IJ, dtype=np.int64)),
                        n_neighbors=15, p_work=0.2).fit()
    assert np.array_equal(ann.neighbor_graph[0], o.neighbor_graph[0])
    ss = ann.annchor_selective_subset(y=y, alpha=0)  # computes the nearest-enemy graph on demand
    oss = O.selective_subset(o, y, alpha=0)
    assert np.array_equal(ann.nearest_enemy_graph[0], o.nearest_enemy_graph[0])
    assert np.allclose(ann.nearest_enemy_graph[1], o.nearest_enemy_graph[1], rtol=1e-12, atol=0)
    assert np.array_equal(ss, oss)
    # sizes the reference produced: 89 / 12 under the stand-in's NumPy RNG (golden), 90 / 16 under numba's RNG (the
    # values its own test pins, tests/test_examples.py:61-85): the subset follows the fitted graph, so the sample and
    # the order inside tie groups move it by a few points
    ref_sizes = {"blobs": (89, 90), "moons": (12, 16)}[tag]
    assert min(abs(len(ss) - r) for r in ref_sizes) <= 4, len(ss)
    assert len(set(ss) & set(g[tag + "_ss_a0"])) >= 0.9 * len(g[tag + "_ss_a0"])
    assert np.array_equal(ann.annchor_selective_subset(y=y, alpha=0.1), O.selective_subset(o, y, alpha=0.1))
    assert np.array_equal(ann.alpha_rss(y), O.alpha_rss(o, y))
    assert len(ann.IJs) == len(o.IJs) and np.array_equal(ann.not_computed_mask, o.ncm)
