"""Pins the oracle's metric restatements (oracle/lev.c, oracle/emd.c) to the reference's
own known answers and stored golden distances."""
import numpy as np

from oracle import metrics as om


def test_levenshtein_known_answers():
    """reference annchor/tests/test_distances.py:9-12."""
    for algo in ("dp", "myers"):
        assert om.levenshtein("cat", "cart", algo) == 1   # insertion
        assert om.levenshtein("cat", "cap", algo) == 1    # substitution
        assert om.levenshtein("cat", "at", algo) == 1     # deletion
        assert om.levenshtein("123456789", "92346781", algo) == 3
        assert om.levenshtein("", "abc", algo) == 3 and om.levenshtein("abc", "", algo) == 3
        assert om.levenshtein("", "", algo) == 0


def test_strings_dataset_known_answer():
    """reference annchor/tests/test_datasets.py:205-235: shapes, X[10] prefix, (10,165) -> 299."""
    X, y = om.load_strings()
    assert len(X) == 1600 and y.shape == (1600,)
    assert X[10].startswith("uofsjurgdrwshktxprvojrluttjiakqesuhdlkymvrjl") and X[10].endswith("enyeyhawhoqzgkwmu")
    assert y[10] == 0
    assert om.levenshtein(X[10], X[165]) == 299
    assert om.levenshtein(X[10], X[165], "dp") == 299


def test_myers_equals_dp_on_dataset_sample():
    X, _ = om.load_strings()
    P = om.PackedStrings(X)
    IJ = np.random.default_rng(5).integers(0, 1600, (400, 2))
    assert np.array_equal(P.pairs(IJ, algo=1), P.pairs(IJ, algo=0))


def test_emd_known_answer_and_stored_graph():
    """reference annchor/tests/test_datasets.py:17-108 and the 179 700 stored exact-EMD
    distances of annchor/data/digits_data.npz (sampled)."""
    d = om.load_digits()
    X, M, (ngi, ngd) = d["X"], d["cost_matrix"], d["neighbor_graph"]
    assert X.shape == (1797, 64) and ngi.shape == (1797, 100)
    assert ngi[10, 15] == 676 and np.isclose(ngd[10, 15], 0.305587260000565)
    assert X[10][:8].tolist() == [0, 0, 1, 9, 15, 11, 0, 0]
    H = om.Histograms(X, M)
    assert np.isclose(H.pairs(np.array([[10, 676]]))[0], 0.305587260000565, rtol=0, atol=1e-12)
    rng = np.random.default_rng(0)
    rows = rng.integers(0, 1797, 4000)
    cols = rng.integers(0, 100, 4000)
    IJ = np.stack([rows, ngi[rows, cols]], axis=1)
    np.testing.assert_allclose(H.pairs(IJ), ngd[rows, cols], rtol=0, atol=1e-12)


def test_emd_against_independent_lp():
    """Cross-check with scipy's HiGHS LP on far pairs (not in the stored graph)."""
    from scipy.optimize import linprog

    d = om.load_digits()
    X, M = d["X"], d["cost_matrix"]
    H = om.Histograms(X, M)
    rng = np.random.default_rng(1)
    IJ = rng.integers(0, 1797, (12, 2))
    got = H.pairs(IJ)
    for (i, j), g in zip(IJ, got):
        r, c = np.nonzero(X[i])[0], np.nonzero(X[j])[0]
        a, b = X[i][r] / X[i][r].sum(), X[j][c] / X[j][c].sum()
        C = M[np.ix_(r, c)]
        n, m = len(r), len(c)
        Aeq = np.zeros((n + m, n * m))
        for k in range(n):
            Aeq[k, k * m:(k + 1) * m] = 1
        for k in range(m):
            Aeq[n + k, k::m] = 1
        res = linprog(C.ravel(), A_eq=Aeq[:-1], b_eq=np.concatenate([a, b])[:-1], bounds=(0, None), method="highs")
        assert res.status == 0
        assert abs(res.fun - g) < 1e-9


def test_euclidean_matches_numpy_norm():
    """reference annchor/tests/test_distances.py:15-19."""
    rng = np.random.default_rng(3)
    X = rng.random((10, 100))
    IJ = np.array([[0, 1], [2, 3], [4, 4]])
    got = om.euclidean_pairs(X, IJ)
    for (i, j), g in zip(IJ, got):
        assert np.isclose(g, np.linalg.norm(X[i] - X[j]))
