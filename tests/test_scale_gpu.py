"""Pair-list form at sizes well above the BASELINE configs (size-independent properties:
exactness of every reported distance, recall against brute force, sortedness, budget)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_pairlist_euclidean_8000_points():
    from annchor_amd import Annchor, BruteForce, compare_neighbor_graphs

    rng = np.random.default_rng(5)
    n, k = 8000, 15
    Z = rng.standard_normal((n, 6))
    X = (Z @ rng.standard_normal((6, 48)) + 0.05 * rng.standard_normal((n, 48))).astype(np.float64)
    ann = Annchor(X, "euclidean", n_anchors=24, n_neighbors=k, p_work=0.1, n_samples=5000)
    ann.fit()
    idx, dist = ann.neighbor_graph
    assert idx.shape == (n, k) and np.array_equal(idx[:, 0], np.arange(n)) and np.all(dist[:, 0] == 0)
    assert np.all(np.diff(dist, axis=1) >= 0)
    # the work budget is respected (annchor.py:117-148)
    assert ann.evals <= 0.1 * n * (n - 1) / 2 + n
    # every reported distance is the true distance of the reported pair
    rows = rng.choice(n, 300, replace=False)
    for r in rows:
        d = np.sqrt(((X[idx[r]] - X[r]) ** 2).sum(axis=1))
        np.testing.assert_allclose(d, dist[r], rtol=1e-12, atol=1e-12)
    bf = BruteForce(X, "euclidean").fit(n_neighbors=k)
    err = compare_neighbor_graphs(bf.neighbor_graph, ann.neighbor_graph, k)
    assert err <= 0.02 * n * k, err   # recall >= 0.98 at 10 % of the work


def test_pairlist_strings_6000():
    """Random strings in 30 families (mutated copies): Levenshtein, ragged lengths 40..200."""
    from annchor_amd import Annchor, BruteForce, compare_neighbor_graphs

    rng = np.random.default_rng(6)
    alphabet = np.array(list("ACGT"))
    seeds = ["".join(rng.choice(alphabet, rng.integers(60, 180))) for _ in range(30)]
    X = []
    for _ in range(6000):
        s = list(seeds[rng.integers(0, 30)])
        for _ in range(rng.integers(0, 25)):
            p = rng.integers(0, len(s))
            op = rng.integers(0, 3)
            if op == 0 and len(s) > 40:
                s.pop(p)
            elif op == 1:
                s.insert(p, rng.choice(alphabet))
            else:
                s[p] = rng.choice(alphabet)
        X.append("".join(s))
    n, k = len(X), 10
    ann = Annchor(X, "levenshtein", n_anchors=20, n_neighbors=k, p_work=0.08)
    ann.fit()
    bf = BruteForce(X, "levenshtein").fit(n_neighbors=k)
    err = compare_neighbor_graphs(bf.neighbor_graph, ann.neighbor_graph, k)
    assert err <= 0.05 * n * k, err
    # reported distances are exact edit distances
    idx, dist = ann.neighbor_graph
    assert np.all(dist == np.round(dist))
    from annchor_amd.distances import levenshtein

    r = rng.choice(n, 40, replace=False)
    xs = [X[i] for i in r for _ in range(k)]
    ys = [X[j] for i in r for j in idx[i]]
    assert np.array_equal(levenshtein.many(xs, ys), dist[r].ravel())
