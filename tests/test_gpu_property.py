"""Property-based parity of the three metric kernels against the oracle's C restatements
(hypothesis draws the shapes; every example is one batched launch through the C-ABI)."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import metrics as om

pytestmark = pytest.mark.gpu
SETTINGS = dict(max_examples=25, deadline=None, suppress_health_check=list(HealthCheck))


@settings(**SETTINGS)
@given(seed=st.integers(0, 2 ** 31 - 1), n=st.integers(2, 60), max_len=st.integers(0, 300), alpha=st.integers(1, 40),
       variant=st.sampled_from(["1", "9"]))
def test_levenshtein_kernel_equals_dp(seed, n, max_len, alpha, variant):
    import os

    from annchor_amd import _native
    from annchor_amd.distances import levenshtein

    rng = np.random.default_rng(seed)
    letters = [chr(48 + a) for a in range(alpha)]
    X = ["".join(rng.choice(letters, rng.integers(0, max_len + 1))) for _ in range(n)]
    # related strings as well as unrelated ones
    for t in range(0, n - 1, 3):
        s = list(X[t])
        for _ in range(rng.integers(0, 6)):
            if s and rng.random() < 0.5:
                s.pop(rng.integers(0, len(s)))
            else:
                s.insert(rng.integers(0, len(s) + 1), rng.choice(letters))
        X[t + 1] = "".join(s)
    os.environ["ANNCHOR_LEV_R"] = variant
    try:
        eng = _native.Engine(0)
        levenshtein.bind(eng, X)
        IJ = rng.integers(0, n, (200, 2))
        got = eng.metric_pairs(IJ)
    finally:
        del os.environ["ANNCHOR_LEV_R"]
    want = np.array([om.levenshtein(X[i], X[j], "dp") for i, j in IJ], dtype=np.float64)
    assert np.array_equal(got, want)
    # metric properties: identity, symmetry
    sym = eng.metric_pairs(IJ[:, ::-1].copy())
    assert np.array_equal(sym, got)
    assert np.all(got[IJ[:, 0] == IJ[:, 1]] == 0)


@settings(**SETTINGS)
@given(seed=st.integers(0, 2 ** 31 - 1), n=st.integers(2, 80), dim=st.integers(1, 300), f32=st.booleans(), cos=st.booleans())
def test_point_metrics(seed, n, dim, f32, cos):
    from scipy.spatial.distance import cosine as sp_cosine

    from annchor_amd import _native

    rng = np.random.default_rng(seed)
    X = (rng.standard_normal((n, dim)) * rng.uniform(0.1, 10) + rng.uniform(-1, 1)).astype(np.float32 if f32 else np.float64)
    eng = _native.Engine(0)
    eng.set_points(X, cosine=cos)
    IJ = rng.integers(0, n, (300, 2))
    got = eng.metric_pairs(IJ)
    if cos:
        want = np.array([sp_cosine(X[i], X[j]) if i != j or True else 0.0 for i, j in IJ])
        np.testing.assert_allclose(got, want, rtol=0, atol=4e-6 if f32 else 1e-12)
    else:
        want = om.euclidean_pairs(X, IJ)
        # one rounding of the input precision (reference: np.linalg.norm in X's dtype)
        np.testing.assert_allclose(got, want, rtol=2e-6 if f32 else 1e-14, atol=0)
        assert np.all(got[IJ[:, 0] == IJ[:, 1]] == 0)


@settings(**dict(SETTINGS, max_examples=15))
@given(seed=st.integers(0, 2 ** 31 - 1), n=st.integers(2, 40), bins=st.integers(2, 64), integral=st.booleans())
def test_wasserstein_kernel_equals_exact_ot(seed, n, bins, integral):
    from annchor_amd import _native

    rng = np.random.default_rng(seed)
    H = rng.integers(0, 17, (n, bins)).astype(np.float64) if integral else rng.random((n, bins)) * (rng.random((n, bins)) < 0.6)
    H[H.sum(axis=1) == 0, 0] = 1.0
    side = int(np.ceil(np.sqrt(bins)))
    P = np.array([(i // side, i % side) for i in range(bins)], dtype=np.float64)
    M = np.sqrt(((P[:, None, :] - P[None, :, :]) ** 2).sum(-1))
    eng = _native.Engine(0)
    eng.set_histograms(H, M)
    IJ = rng.integers(0, n, (120, 2))
    got = eng.metric_pairs(IJ)
    want = om.Histograms(H, M).pairs(IJ)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)


def test_guarantee_nmin_rounds_equal_the_sequential_sweep(monkeypatch):
    """guarantee_nmin (utils.py:600-621) as parallel fixed-point rounds == the row-by-row walk: same graph,
    same evaluation count, same refined state, on inputs with many ties (integer distances) and without."""
    from annchor_amd import Annchor

    rng = np.random.default_rng(23)
    cases = []
    Z = rng.standard_normal((1800, 4))
    cases.append((np.round(Z @ rng.standard_normal((4, 12)), 1).astype(np.float64), "euclidean", dict(n_anchors=12, n_neighbors=20, p_work=0.12)))
    X = ["".join(rng.choice(list("ab"), rng.integers(20, 60))) for _ in range(900)]
    cases.append((X, "levenshtein", dict(n_anchors=10, n_neighbors=30, p_work=0.2, n_samples=1500)))
    for data, metric, kw in cases:
        out = {}
        for mode in ("rounds", "sequential"):
            monkeypatch.setenv("ANNCHOR_GN_SWEEP", mode)
            # (the second setting also takes the wave-per-pair form of update_bounds instead of the row-grouped one)
            monkeypatch.setenv("ANNCHOR_UPDATE_BOUNDS", "bits16" if mode == "rounds" else "pairs")
            ann = Annchor(data, metric, random_seed=3, **kw).fit()
            out[mode] = (ann.neighbor_graph[0].copy(), ann.neighbor_graph[1].copy(), ann.evals, ann.RefineApprox.copy(),
                         ann.not_computed_mask.copy())
        for a, b in zip(out["rounds"], out["sequential"]):
            assert np.array_equal(a, b)
        # the row-grouped form reading 4-byte keys (point sets beyond 131 072 take it; forced here)
        monkeypatch.setenv("ANNCHOR_GN_SWEEP", "rounds")
        for form in ("bits32", "pairs32", "pairs16"):   # ("pairs32" / "pairs16": two / four pairs per wave)
            monkeypatch.setenv("ANNCHOR_UPDATE_BOUNDS", form)
            ann = Annchor(data, metric, random_seed=3, **kw).fit()
            bm = (ann.neighbor_graph[0], ann.neighbor_graph[1], ann.evals, ann.RefineApprox, ann.not_computed_mask)
            for a, b in zip(out["rounds"], bm):
                assert np.array_equal(a, b), form


def test_select_prepare_is_equivalent_and_voided_by_state_changes():
    """annchor_select_prepare (thresholds + guarantee_nmin launched ahead of the selection) must not change the
    result: used as fit() uses it, with mismatching parameters (ignored), and followed by a call that rewrites
    RefineApprox (voided) -- all equal to the plain staged run."""
    from annchor_amd import Annchor

    from annchor_amd.datasets import load_strings

    Xs = np.array(load_strings()["X"][::4])
    cfg = dict(n_anchors=9, n_neighbors=12, n_samples=800, p_work=0.25, random_seed=7, niters=2)

    def run(mode):
        ann = Annchor(Xs, "levenshtein", **cfg)
        ann.get_anchors(); ann.get_locality(); ann.get_features()
        for it in range(ann.niters):
            ann.get_sample()
            ann.fit_predict_regression()
            nn = ann.n_neighbors
            nmin = 3 * nn // 2 if it == 0 else 0
            if mode == "prepared":
                ann._engine.select_prepare(nn, nmin)
            elif mode == "mismatch":
                ann._engine.select_prepare(nn + 1, 0)
            elif mode == "voided":
                ann._engine.select_prepare(nn, nmin)
                ann._first_merge = it == 0           # rewrite RefineApprox: the preparation no longer applies
                ann.fit_predict_regression()
            ann.fit_predict_errors()
            ann.select_refine_candidate_pairs(w=1 / ann.niters, it=it)
            if it < ann.niters - 1:
                ann.update_anchor_points()
        ann.get_ann()
        return ann.neighbor_graph[0].copy(), ann.neighbor_graph[1].copy(), ann.evals, ann.RefineApprox.copy()

    ref = run("plain")
    for mode in ("prepared", "mismatch", "voided"):
        got = run(mode)
        for a, b in zip(ref, got):
            assert np.array_equal(a, b), mode
