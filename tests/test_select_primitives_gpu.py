"""The two order-statistics primitives of the sampler / candidate selection, driven through
the C-ABI with injected feature columns so that every branch is reached:

* annchor_kth_uncomputed_dad (filter-then-finish radix selection, scan.hip) against
  np.partition: smooth values (few candidates -> LDS finish), heavy ties (a bucket that is one
  repeated value -> resolved from the OR/AND of the candidates), a mixed bucket larger than
  the LDS capacity (byte-pass fallback), masks, ranks at both ends, negative and tiny values;
* annchor_select_by_rank (rank-in-bin selection) against a NumPy restatement.

The reference computes these with np.sort / np.searchsorted inside
annchor/utils.py:536-575 (the sampler's bins and its choice of pairs by rank).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine_with_pairs():
    from annchor_amd import Annchor, _native
    from annchor_amd.datasets import load_strings
    X = load_strings()["X"][:700]
    ann = Annchor(X, "levenshtein", n_anchors=10, n_neighbors=10, p_work=0.3, niters=1, random_seed=3)
    ann.get_anchors()
    ann.get_locality()
    ann.get_features()
    eng = ann._engine
    n = eng.field_size(_native.F_NCM)
    assert n > 100_000
    return eng, n, _native


def _inject(eng, nat, dad, ncm):
    n = dad.size
    feats = np.zeros((n, 4))
    feats[:, 2] = dad
    eng.upload(nat.F_FEATURES, feats)
    eng.upload(nat.F_NCM, ncm.astype(np.uint8))


def _check(eng, dad, ncm, ks):
    got = eng.kth_uncomputed_dad(np.asarray(ks, dtype=np.int64))
    pool = np.sort(dad[ncm.astype(bool)])
    want = pool[np.asarray(ks)]
    assert np.array_equal(got, want), (ks, got, want)


@pytest.mark.parametrize("sampled", [False, True])
@pytest.mark.parametrize("levels", [2, 3])
@pytest.mark.parametrize("kind", ["uniform", "normal", "halfint", "tied_one", "mixed_big_bucket", "tiny", "few_values", "sorted",
                                  "sample_misleads"])
def test_kth_smallest_vs_numpy(engine_with_pairs, kind, levels, sampled, monkeypatch):
    eng, n, nat = engine_with_pairs
    # lists of 4 M keys and more filter three times before the finishing workgroup; force that here
    monkeypatch.setenv("ANNCHOR_SEL_LEVEL3_MIN", "1" if levels == 3 else str(1 << 40))
    # ... and lists of 8 M keys and more bracket the wanted ranks from a strided sample first (one read of the list)
    monkeypatch.setenv("ANNCHOR_SEL_SAMPLE_MIN", "1" if sampled else str(1 << 40))
    rng = np.random.RandomState(11)
    if kind == "uniform":
        dad = rng.rand(n)
    elif kind == "normal":
        dad = rng.randn(n) * 50.0            # negative keys too
    elif kind == "halfint":
        dad = rng.randint(0, 1200, n) / 2.0  # what Levenshtein produces: massive ties
    elif kind == "tied_one":
        dad = np.full(n, 153.5)
        dad[:100] = rng.rand(100)
    elif kind == "mixed_big_bucket":
        # > 4096 candidates share the top 32 key bits but differ below: the byte-pass fallback
        base = np.float64(1.5)
        ulps = rng.randint(0, 1 << 20, n).astype(np.uint64)
        dad = (np.full(n, base).view(np.uint64) + ulps).view(np.float64)
    elif kind == "tiny":
        dad = rng.rand(n) * 1e-300
    elif kind == "sorted":
        dad = np.sort(rng.randn(n))
    elif kind == "sample_misleads":
        # the strided sample (positions s * n // 8192) sees only large values, everything else is small: the brackets miss
        # the wanted ranks and the selection is repeated the plain way
        dad = rng.rand(n)
        dad[(np.arange(8192, dtype=np.int64) * n) // 8192] += 10.0
    else:
        dad = rng.choice(np.array([0.0, 0.25, 1.0, 1e-9, 7.0]), n)
    ncm = rng.rand(n) < 0.7
    _inject(eng, nat, dad, ncm)
    m = int(ncm.sum())
    for ks in ([0], [m - 1], [m // 2], [m // 3, m // 2], [1, m // 7, m // 2, m - 2], [m - 25, m - 125]):
        _check(eng, dad, ncm, ks)
    # back-to-back calls reuse the tables the previous call left zeroed
    _check(eng, dad, ncm, [m // 5])
    _check(eng, dad, ncm, [m // 5, m // 4])


def test_kth_smallest_sampled_odd_length(monkeypatch):
    """The bracket pass reads two consecutive keys per lane: a list of odd length ends in a lone key (246 051 pairs)."""
    from annchor_amd import Annchor, _native
    from annchor_amd.datasets import load_strings
    ann = Annchor(load_strings()["X"][:702], "levenshtein", n_anchors=10, n_neighbors=10, p_work=0.3, niters=1, random_seed=3)
    ann.get_anchors()
    ann.get_locality()
    ann.get_features()
    eng = ann._engine
    n = eng.field_size(_native.F_NCM)
    assert n % 2 == 1
    monkeypatch.setenv("ANNCHOR_SEL_SAMPLE_MIN", "1")
    rng = np.random.RandomState(2)
    dad = rng.rand(n)
    dad[-1] = 5.0      # the lone last key is the maximum ...
    for ncm in (np.ones(n, dtype=bool), rng.rand(n) < 0.5):
        ncm[-1] = True
        _inject(eng, _native, dad, ncm)
        m = int(ncm.sum())
        _check(eng, dad, ncm, [m - 1])
        _check(eng, dad, ncm, [0, m // 2, m - 1])
    dad[-1] = -1.0     # ... or the minimum
    _inject(eng, _native, dad, ncm)
    _check(eng, dad, ncm, [0, 1, int(ncm.sum()) - 1])
    eng.close()


def test_kth_smallest_all_flagged_and_single(engine_with_pairs):
    eng, n, nat = engine_with_pairs
    rng = np.random.RandomState(5)
    dad = rng.rand(n)
    ncm = np.ones(n, dtype=bool)
    _inject(eng, nat, dad, ncm)
    _check(eng, dad, ncm, [0, n - 1])
    ncm = np.zeros(n, dtype=bool)
    ncm[n // 2] = True
    _inject(eng, nat, dad, ncm)
    _check(eng, dad, ncm, [0])


@pytest.mark.parametrize("nbins", [1, 7, 64])
def test_select_by_rank_vs_numpy(engine_with_pairs, nbins):
    eng, n, nat = engine_with_pairs
    rng = np.random.RandomState(nbins)
    dad = rng.randint(0, 1200, n) / 2.0
    ncm = rng.rand(n) < 0.6
    _inject(eng, nat, dad, ncm)
    edges = np.quantile(dad[ncm], np.linspace(0, 1, nbins + 1))
    edges[0] -= 1.0
    edges[-1] += 1.0
    edges = np.unique(edges)
    nb = edges.size - 1
    counts = eng.bin_counts(edges)
    member = [np.flatnonzero(ncm & (dad >= edges[b]) & (dad < edges[b + 1])) for b in range(nb)]
    assert np.array_equal(counts, [m.size for m in member])
    bin_of, ranks = [], []
    for b in range(nb):
        if counts[b] == 0:
            continue
        r = np.unique(np.r_[0, counts[b] - 1, rng.randint(0, counts[b], 200)])
        rng.shuffle(r)
        bin_of += [b] * r.size
        ranks += list(r)
    pos = eng.select_by_rank(edges, np.asarray(bin_of, dtype=np.int32), np.asarray(ranks, dtype=np.int64))
    want = np.array([member[b][r] for b, r in zip(bin_of, ranks)])
    assert np.array_equal(pos, want)


@pytest.mark.parametrize("kind", ["uniform", "halfint", "constant"])
def test_sampler_stats_equals_the_three_separate_calls(engine_with_pairs, kind):
    """annchor_sampler_stats (quantiles -> np.linspace edges -> partition populations chained on the device, one host wait)
    against kth_uncomputed_dad + np.linspace + bin_counts, for 2..64 partitions; the device's edges must be NumPy's bit for bit."""
    eng, n, nat = engine_with_pairs
    rng = np.random.RandomState(4)
    dad = rng.rand(n) * 37.0 if kind == "uniform" else rng.randint(0, 1200, n) / 2.0 if kind == "halfint" else np.full(n, 2.5)
    ncm = rng.rand(n) < 0.8
    _inject(eng, nat, dad, ncm)
    m = int(ncm.sum())
    for P in (2, 3, 5, 7, 12, 33, 64):
        for iq1, iq3 in ((m // 100, 99 * m // 100), (m // 10, 9 * m // 10), (0, m - 1)):
            q1, q3, edges, counts = eng.sampler_stats(iq1, iq3, P)
            want_q = eng.kth_uncomputed_dad(np.array([iq1, iq3], dtype=np.int64))
            assert q1 == want_q[0] and q3 == want_q[1]
            bins = np.hstack([-np.inf, np.linspace(q1, q3, P - 1), np.inf])
            assert edges is not None and np.array_equal(edges, bins), (P, edges, bins)
            assert np.array_equal(counts, eng.bin_counts(bins)), (P, kind)
            assert counts.sum() == m
