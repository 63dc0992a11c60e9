"""CPU tests of the host layer: constructor arithmetic, metric resolution, samplers /
regressors / error predictors (NumPy protocol forms) against the oracle and the golden
vectors, and the RNG identity the device sampler relies on."""
import os

import numpy as np
import pytest

from oracle import annchor_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
NAMES = ["lower bound", "upper bound", "double anchor distance", "is anchor"]


def test_budget_matches_reference_arithmetic():
    """reference annchor/tests/test_annchor.py:148-160 (test_bad_pwork)."""
    from annchor_amd.annchor import budget

    b = budget(1600, 20, 15, 5000, 1.1)
    assert b["p_work"] == 1.0
    b = budget(1600, 20, 15, 5000, 0.0)
    assert b["p_work"] == (2 * (b["na"] + 5000) + 1) / b["N"]
    assert b["N"] == 1600 * 1599 // 2 and b["na"] == sum(1600 - j for j in range(1, 21))
    for nx, na, k, ns, pw in [(1600, 15, 25, 5000, 0.12), (300, 8, 10, 700, 0.3), (50, 30, 40, 100, 0.5)]:
        o = O.budget(nx, na, ns, pw, k)
        h = budget(nx, na, k, ns, pw)
        assert (h["N"], h["na"], h["p_work"], h["loc_min"]) == (o["N"], o["na"], o["p_work"], o["loc_min"])


def test_metric_resolution():
    """reference annchor/utils.py:62-107."""
    from annchor_amd import distances
    from annchor_amd.utils import get_function_from_input

    assert get_function_from_input("levenshtein", None) is distances.levenshtein
    assert get_function_from_input("euclidean", None) is distances.euclidean
    w = get_function_from_input("wasserstein", {"cost_matrix": np.eye(4)})
    assert isinstance(w, distances.Wasserstein) and w.cost_matrix.shape == (4, 4)
    with pytest.raises(AssertionError):
        get_function_from_input("wasserstein", {})
    with pytest.raises(AssertionError):
        get_function_from_input("manhattan", None)
    f = get_function_from_input(lambda x, y, p=1: abs(x - y) ** p, {"p": 2})
    assert f(1, 4) == 9
    g = lambda x, y: 7  # noqa: E731
    assert get_function_from_input(g, None) is g


def test_host_evaluator_serial_and_parallel():
    from annchor_amd.utils import get_exact_ijs_

    X = np.arange(10.0)
    IJ = np.array([[0, 3], [2, 2], [9, 1]])
    f = lambda a, b: abs(a - b)  # noqa: E731
    assert np.array_equal(get_exact_ijs_(f, parallel=False)(f, X, IJ), [3, 0, 8])
    assert np.array_equal(get_exact_ijs_(f)(f, X, IJ), [3, 0, 8])


@pytest.mark.parametrize("backend", ["loky", "multiprocessing", "threading"])
def test_host_evaluator_chunks_equal_the_per_pair_form(backend):
    """The chunked evaluator (annchor_amd/utils.py) returns np.array([f(X[i], X[j]) for i, j in IJ]) exactly as the
    reference's one-task-per-pair form (utils.py:152-175): arrays of rows, arrays of strings and plain lists, list
    lengths around the chunk boundaries, order preserved; the pool never shrinks between calls."""
    from annchor_amd import utils

    rng = np.random.default_rng(5)
    Xf = rng.normal(size=(200, 7))
    Xs = np.array(["".join(rng.choice(list("abcd"), size=int(rng.integers(1, 30)))) for _ in range(200)])
    ge = utils.get_exact_ijs_(abs_sum, backend=backend)
    workers = 0
    for n in (1, 2, 5, utils.MIN_CHUNK, utils.MIN_CHUNK + 1, 777, 3000):
        IJ = rng.integers(0, 200, size=(n, 2))
        want = np.array([abs_sum(Xf[i], Xf[j]) for i, j in IJ])
        got = ge(abs_sum, Xf, IJ)
        assert got.dtype == np.float64 and np.array_equal(got, want)
        assert ge.state["workers"] >= workers
        workers = ge.state["workers"]
    gs = utils.get_exact_ijs_(len_diff, backend=backend)
    IJ = rng.integers(0, 200, size=(500, 2))
    want = np.array([len_diff(Xs[i], Xs[j]) for i, j in IJ])
    assert np.array_equal(gs(len_diff, Xs, IJ), want)
    assert np.array_equal(gs(len_diff, list(Xs), IJ), want)
    assert gs(len_diff, Xs, np.zeros((0, 2), dtype=np.int64)).shape == (0,)


def slow_abs(a, b):
    import time

    time.sleep(0.12)
    return float(abs(a - b))


def test_host_evaluator_slow_metric_gets_the_reference_allowance(monkeypatch):
    """A metric that costs seconds per pair finishes: the per-task timeout is the reference's 30 s per PAIR
    (utils.py:152-175) times the pairs of the task, and slow metrics are submitted in small chunks.  (a
    chunk of >= 32 pairs with timeout max(30, 0.25 chunk) raised TimeoutError mid-fit for a 1 s / pair metric.)  Run
    here with the allowance scaled to 0.2 s per pair: 70 pairs of a 0.12 s metric pass, and every chunk stays small."""
    from annchor_amd import utils

    monkeypatch.setattr(utils, "PAIR_TIMEOUT_SECONDS", 0.2 * 30)   # (x30: joblib adds its own start-up; the point is the scaling)
    X = np.arange(50.0)
    rng = np.random.default_rng(2)
    IJ = rng.integers(0, 50, size=(70, 2))
    ge = utils.get_exact_ijs_(slow_abs, backend="threading")
    got = ge(slow_abs, X, IJ)
    assert np.array_equal(got, np.abs(X[IJ[:, 0]] - X[IJ[:, 1]]))
    assert ge.state["last_chunk"] <= 8 and ge.state["last_timeout"] >= 0.2 * 30 * ge.state["last_chunk"]


def abs_sum(a, b):
    return float(np.abs(a - b).sum())


def len_diff(a, b):
    return float(abs(len(a) - len(b)) + (a[0] != b[0]))


def test_string_encoding_roundtrip():
    from annchor_amd.distances import encode_strings

    codes, offs, lens, A = encode_strings(["abc", "", "cab", "zz"])
    assert A == 4 and list(lens) == [3, 0, 3, 2] and list(offs) == [0, 3, 3, 6]
    assert list(codes) == [0, 1, 2, 2, 0, 1, 3, 3]
    # more than 256 distinct symbols: 16-bit dense codes (annchor_set_strings_u16)
    wide = ["".join(chr(300 + k) for k in range(300)), "\u4e00\u0141"]
    codes, offs, lens, A = encode_strings(wide)
    assert codes.dtype == np.uint16 and A == 301 and list(lens) == [300, 2]
    assert list(codes[:300]) == list(range(300))          # dense codes in code-point order
    assert codes[300] == 300 and codes[301] == 0x141 - 300   # U+4E00 is the largest code point; U+0141 = chr(321) is one of the 300


def test_legacy_choice_is_permutation_prefix():
    """The device sampler draws `permutation(c)[:want]`; NumPy's legacy
    `choice(a, size, replace=False)` is exactly `a[permutation(len(a))[:size]]`."""
    a = np.arange(1000) * 3 + 1
    for seed in (0, 42, 43):
        np.random.seed(seed)
        x1 = np.random.choice(a, size=37, replace=False)
        x2 = np.random.choice(a[:500], size=11, replace=False)
        np.random.seed(seed)
        y1 = a[np.random.permutation(1000)[:37]]
        y2 = a[:500][np.random.permutation(500)[:11]]
        assert np.array_equal(x1, y1) and np.array_equal(x2, y2)


@pytest.mark.parametrize("name", ["strings_small", "euclid_small"])
def test_numpy_protocol_plugins_match_golden(name):
    """The NumPy forms of the built-in plugins (what a user's code sees) reproduce the
    reference's captured stage outputs."""
    from annchor_amd.error_predictors import SimpleStratifiedErrorRegression
    from annchor_amd.regressors import SimpleStratifiedLinearRegression
    from annchor_amd.samplers import SimpleStratifiedSampler

    G = np.load(os.path.join(GOLD, name + ".npz"))
    feats, ncm = G["features0"], G["ncm0"]
    s = SimpleStratifiedSampler()
    ixs, n, bins = s.sample(feats, NAMES, int(G["cfg_n_samples"]), ncm, int(G["cfg_random_seed"]))
    assert np.array_equal(ixs, G["it0_sample_ixs"]) and np.array_equal(bins, G["it0_bins"])
    r = SimpleStratifiedLinearRegression()
    r.fit(feats[ixs], NAMES, G["it0_sample_y"], sample_bins=bins)
    np.testing.assert_allclose(r.coef_, G["it0_coef"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(r.intercept_, G["it0_intercept"], rtol=1e-9, atol=1e-9)
    pred = r.predict(feats, NAMES)
    np.testing.assert_allclose(pred[ixs], G["it0_sample_predict"], rtol=1e-9, atol=1e-9)
    e = SimpleStratifiedErrorRegression()
    e.fit(feats[ixs], NAMES, G["it0_sample_y"] - G["it0_sample_predict"], sample_bins=bins)
    for b in range(7):
        assert np.array_equal(e.errs[b], G["it0_errs%d" % b])
    assert np.array_equal(e.predict(feats, NAMES), G["it0_labels"])


def test_compare_neighbor_graphs_counts_injected_errors():
    """reference annchor/tests/test_annchor.py:15-32."""
    from annchor_amd import compare_neighbor_graphs
    from annchor_amd.datasets import load_digits

    ng = load_digits()["neighbor_graph"]
    assert compare_neighbor_graphs(ng, ng, 30) == 0
    rs = np.random.RandomState(42)
    ixs, ds = ng[0].copy(), ng[1].copy()
    for i in range(ds.shape[0]):
        ds[i, rs.randint(20, 100)] += rs.random_sample() + 0.01
    assert compare_neighbor_graphs(ng, (ixs, ds), 100) == ds.shape[0]
    assert compare_neighbor_graphs(ng, (ixs, ds), 20) == 0


def test_datasets_known_values():
    """reference annchor/tests/test_datasets.py:17-108,205-235."""
    from annchor_amd.datasets import load_digits, load_strings

    d = load_digits()
    assert d["X"].shape == (1797, 64) and d["y"].shape == (1797,) and d["neighbor_graph"].shape == (2, 1797, 100)
    assert d["y"][10] == 0 and int(d["neighbor_graph"][0][10, 15]) == 676
    s = load_strings()
    assert s["X"].shape == (1600,) and s["y"].shape == (1600,) and len(s["X"][10]) == 501


def test_library_rng_matches_numpy_legacy_stream():
    """annchor_legacy_choice_ranks (host code of the library) == NumPy's legacy stream."""
    import __graft_entry__ as g

    g.build()
    from annchor_amd import _native

    cases = [([667, 37493, 60, 5, 4, 18000, 873, 1, 0, 715, 2, 3], [715, 715, 714, 714, 714, 714, 714, 0, 0, 715, 2, 2]),
             ([1700, 18800, 17100, 17600, 18300, 33000, 18500], [715, 715, 714, 714, 714, 714, 714]),
             ([1, 2, 3], [1, 1, 3])]
    for seed in (0, 42, 43, 2 ** 32 - 1):
        for counts, want in cases:
            if seed in (42, 43):   # with and without a background-prefetched stream (short one: forces inline extension)
                _native.legacy_prefetch(seed, 5000 if seed == 42 else 200000)
            got = _native.legacy_choice_ranks(seed, counts, want)
            np.random.seed(seed)
            ref = [np.arange(c) if c < w else np.random.permutation(int(c))[:w] for c, w in zip(counts, want)]
            assert all(np.array_equal(a, b) for a, b in zip(got, ref))


def test_library_rng_async_worker_matches_numpy():
    """annchor_legacy_choice_begin / _end (the draw on the library's persistent worker thread, several
    tickets outstanding, interleaved with synchronous calls) == NumPy's legacy stream; large bins so
    that the lagged-threshold AVX-512 scan and the pooled backward traces are the code that runs."""
    import __graft_entry__ as g

    g.build()
    from annchor_amd import _native

    rng = np.random.RandomState(7)
    jobs = []
    for seed in (5, 6, 7, 8):
        counts = rng.randint(50, 400000, 7)
        want = np.array([715, 715, 714, 714, 714, 714, 714])
        jobs.append((seed, counts, want, _native.legacy_choice_begin(seed, counts, want)))
    sync = _native.legacy_choice_ranks(9, [300000, 2000, 70], [500, 500, 100])
    np.random.seed(9)
    ref = [np.arange(c) if c < w else np.random.permutation(int(c))[:w] for c, w in zip([300000, 2000, 70], [500, 500, 100])]
    assert all(np.array_equal(a, b) for a, b in zip(sync, ref))
    for seed, counts, want, ticket in jobs:
        got = _native.legacy_choice_end(ticket)
        np.random.seed(seed)
        ref = [np.arange(c) if c < w else np.random.permutation(int(c))[:w] for c, w in zip(counts, want)]
        assert len(got) == len(ref) and all(np.array_equal(a, b) for a, b in zip(got, ref))


def test_native_batched_ols_matches_python_path_bit_for_bit():
    """annchor_ols_bins (centring with NumPy's pairwise sums + scipy's own dgelsd through its C
    pointer) against regressors._ols, the Python restatement of the reference's per-partition
    sklearn LinearRegression (annchor/regressors.py:60-84): identical coefficients and intercepts,
    including rank-deficient partitions; partitions with fewer rows than features are left to the
    Python path (status 1)."""
    import __graft_entry__ as g

    g.build()
    from annchor_amd import _native
    from annchor_amd.regressors import SimpleStratifiedLinearRegression, _ols

    rng = np.random.RandomState(3)
    compared = 0
    for trial in range(60):
        n = rng.randint(30, 4000)
        F = rng.randn(n, 4) * rng.choice([1, 50, 1e-2])
        yv = F[:, 0] * 0.3 + F[:, 1] * 0.5 + rng.randn(n) * 0.1
        if trial % 7 == 0:
            F[:, 1] = F[:, 0]   # rank deficient
        order = np.argsort(rng.randint(0, 7, n), kind="stable")
        Xs, ys = F[order][:, [0, 1, 2]], yv[order]
        cuts = np.sort(np.r_[0, rng.randint(0, n + 1, 6), n])
        out = _native.ols_bins(Xs, ys, cuts)
        if out is None:
            pytest.skip("scipy.linalg.cython_lapack does not expose dgelsd here")
        coef, xm, ym, st = out
        for b in range(7):
            lo, hi = cuts[b], cuts[b + 1]
            if hi - lo < 3:
                assert st[b] == 1
                continue
            c, i = _ols(Xs[lo:hi], ys[lo:hi])
            assert st[b] == 0 and np.array_equal(c, coef[b]) and i == ym[b] - xm[b] @ coef[b]
            compared += 1
    assert compared > 300
    # and through the regressor: batched and per-partition fits agree exactly
    names = ["lower bound", "upper bound", "double anchor distance", "is anchor"]
    F = rng.rand(5000, 4) * 300
    y = F[:, 2] * 0.8 + rng.randn(5000)
    bins = np.hstack([-np.inf, np.linspace(30, 270, 6), np.inf])
    a, b = SimpleStratifiedLinearRegression(), SimpleStratifiedLinearRegression()
    a.fit(F, names, y, bins)
    os.environ["ANNCHOR_OLS_PYTHON"] = "1"
    try:
        b.fit(F, names, y, bins)
    finally:
        del os.environ["ANNCHOR_OLS_PYTHON"]
    assert np.array_equal(a.coef_, b.coef_) and np.array_equal(a.intercept_, b.intercept_)


def test_library_rng_workers_survive_fork():
    """The draw worker and the trace helpers are persistent threads; a forked child has none of them
    and must start its own (pthread_atfork handler) instead of waiting for the parent's."""
    import __graft_entry__ as g

    g.build()
    from annchor_amd import _native

    counts, want = np.array([12000, 150000, 250000]), np.array([700, 700, 700])
    a = _native.legacy_choice_end(_native.legacy_choice_begin(5, counts, want))
    pid = os.fork()
    if pid == 0:
        try:
            b = _native.legacy_choice_end(_native.legacy_choice_begin(5, counts, want))
            ok = all(np.array_equal(x, y) for x, y in zip(a, b))
        except BaseException:  # noqa: BLE001
            ok = False
        os._exit(0 if ok else 3)
    import time
    deadline = time.time() + 60
    while time.time() < deadline:
        done, status = os.waitpid(pid, os.WNOHANG)
        if done:
            assert os.WEXITSTATUS(status) == 0
            return
        time.sleep(0.05)
    os.kill(pid, 9)
    raise AssertionError("forked child hung in the draw")


def test_library_rng_portable_path_matches_numpy():
    """The same check with the AVX-512 scan disabled (ANNCHOR_RNG_SCALAR=1 is read once per
    process, hence the subprocess)."""
    import subprocess
    import sys

    code = (
        "import numpy as np, __graft_entry__ as g\n"
        "g.build()\n"
        "from annchor_amd import _native\n"
        "counts, want = [1700, 18800, 17100, 17600, 18300, 33000, 18500, 3, 0], [715, 715, 714, 714, 714, 714, 714, 5, 0]\n"
        "for seed in (0, 42):\n"
        "    got = _native.legacy_choice_ranks(seed, counts, want)\n"
        "    np.random.seed(seed)\n"
        "    ref = [np.arange(c) if c < w else np.random.permutation(int(c))[:w] for c, w in zip(counts, want)]\n"
        "    assert all(np.array_equal(a, b) for a, b in zip(got, ref))\n"
        "print('ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ANNCHOR_RNG_SCALAR="1", PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_library_rng_long_bins_match_numpy():
    """Bins of a million pairs and more trace through the hashed tracked set (default route: one sequential scan, the
    traces on the pinned helpers): same stream, same result."""
    import __graft_entry__ as g

    g.build()
    from annchor_amd import _native

    counts = [300000, 1500000, 900000, 40, 1200000, 3, 700000]
    want = [715, 715, 714, 714, 714, 5, 714]
    for seed in (7, 42):
        got = _native.legacy_choice_ranks(seed, counts, want)
        np.random.seed(seed)
        ref = [np.arange(c) if c < w else np.random.permutation(int(c))[:w] for c, w in zip(counts, want)]
        assert all(np.array_equal(a, b) for a, b in zip(got, ref))


@pytest.mark.parametrize("env", [{"ANNCHOR_RNG_PAR_MIN": "4194304"}, {"ANNCHOR_RNG_PAR_MIN": "4194304", "ANNCHOR_RNG_PAR_FRESH": "1"},
                                 {"ANNCHOR_RNG_HASHED_MIN": "1000"}, {"ANNCHOR_RNG_HASHED_MIN": "1000000000000"},
                                 {"ANNCHOR_RNG_PIN": "0", "ANNCHOR_RNG_HELPERS": "1"}])
def test_library_rng_routes_match_numpy(env):
    """The routes behind the environment switches (read once per process, hence the subprocess): the count pass + parallel
    bins of round 2 (pooled helpers or fresh threads), the hashed trace forced on short bins / switched off on long ones,
    free-running helpers."""
    import subprocess
    import sys

    code = (
        "import numpy as np, __graft_entry__ as g\n"
        "g.build()\n"
        "from annchor_amd import _native\n"
        "counts, want = [300000, 1500000, 900000, 40, 1200000, 3, 700000], [715, 715, 714, 714, 714, 5, 714]\n"
        "for seed in (7, 42):\n"
        "    got = _native.legacy_choice_ranks(seed, counts, want)\n"
        "    np.random.seed(seed)\n"
        "    ref = [np.arange(c) if c < w else np.random.permutation(int(c))[:w] for c, w in zip(counts, want)]\n"
        "    assert all(np.array_equal(a, b) for a, b in zip(got, ref))\n"
        "print('ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, PYTHONPATH=root, **env), capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_to_sparse_matrix_equals_reference_loop():
    """annchor.py:625-641: vectorised construction == the reference's cell-by-cell DOK fill
    (symmetric, eps on every stored entry, later assignment wins)."""
    from scipy.sparse import dok_matrix

    from annchor_amd.annchor import Annchor

    rng = np.random.default_rng(0)
    nx, k = 200, 6
    idx = np.stack([np.r_[i, rng.choice(np.delete(np.arange(nx), i), k - 1, replace=False)] for i in range(nx)])
    dist = np.sort(rng.random((nx, k)), axis=1)
    dist[:, 0] = 0

    class Holder:
        pass

    a = Holder()
    a.neighbor_graph, a.nx = (idx, dist), nx
    got = Annchor._sparse_from_graph_host(idx, dist)
    want = dok_matrix((nx, nx), dtype=np.float64)
    eps = np.nextafter(0, 1)
    for i, (js, ds) in enumerate(zip(idx, dist)):
        for j, d in zip(js, ds):
            want[i, j] = want[j, i] = d + eps
    assert isinstance(got, dok_matrix) and (got != want).nnz == 0 and got.nnz == want.nnz
    assert got[5, 5] == eps   # explicit zero survives


def test_device_stratified_sampler_numpy_protocol_equals_restatement():
    """DeviceStratifiedSampler through the plain sample() protocol == the oracle's hashed_stratified_sample
    (same partitions, same keys, same ties), including loop_num progression and sparse partitions."""
    from annchor_amd.samplers import DeviceStratifiedSampler
    from oracle import annchor_oracle as O

    names = ["lower bound", "upper bound", "double anchor distance", "is anchor"]
    rng = np.random.default_rng(3)
    for trial in range(12):
        n = int(rng.integers(4000, 30000))
        f = np.round(rng.normal(10, 4, (n, 4)), int(rng.integers(0, 3)))
        m = rng.random(n) < 0.85
        s = DeviceStratifiedSampler()
        for loop in range(2):
            ns = int(rng.integers(60, 900))
            got = s.sample(f, names, ns, m, 42)
            want = O.hashed_stratified_sample(f, m, ns, 42, loop)
            assert np.array_equal(got[0], want[0]) and got[1] == want[1] and np.array_equal(got[2], want[2])
            assert len(np.unique(got[0])) == len(got[0]) and m[got[0]].all()
            m = m.copy()
            m[got[0]] = False
