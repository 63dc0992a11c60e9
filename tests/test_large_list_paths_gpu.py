"""The kernels that take over on long pair lists (tiled feature kernel, column-half transposes + streamed
row kernels, second candidate cut, ECDF bucket index, sampled-bracket selection, run-based / super-tiled pair emission) must give bit-identical state to the small-list
kernels.  The thresholds are lowered through the environment so that a 3 M-pair list exercises them; each
setting runs in its own process (the library reads the thresholds once)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LARGE = {"ANNCHOR_TRANSPOSE_MIN": "0", "ANNCHOR_FEATURES_TILED_MIN": "0", "ANNCHOR_ROWC_SHRINK_MIN": "16", "ANNCHOR_ECDF_INDEX_MIN": "0",
         "ANNCHOR_SEL_SAMPLE_MIN": "1", "ANNCHOR_EMIT_RUN_MIN": "0", "ANNCHOR_LOC_THRESH_HIST": "1", "ANNCHOR_KEEP_COLS_MIN": "0", "ANNCHOR_EMIT_SUPER_MIN": "1"}


def _run(tmp_path, tag, metric, extra):
    out = str(tmp_path / (tag + ".npz"))
    env = dict(os.environ, **extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "large_paths_worker.py"), out, metric], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


@pytest.mark.parametrize("metric", ["euclidean", "levenshtein"])
def test_large_list_kernels_equal_small_list_kernels(tmp_path, metric):
    a = _run(tmp_path, "small", metric, {})
    b = _run(tmp_path, "large", metric, dict(LARGE, ANNCHOR_FEATURES_FORM="tiled"))
    # (thinned lists take the anchor-outer form of the tiled feature kernel; forced here on the complete list)
    d = _run(tmp_path, "large_dense", metric, dict(LARGE, ANNCHOR_FEATURES_FORM="dense"))
    for key in ("A", "D", "evals", "n_pairs", "features", "ncm", "RA", "idx", "dist"):
        assert np.array_equal(a[key], b[key]), key
        assert np.array_equal(a[key], d[key]), key


OLDER = {"ANNCHOR_FEATURES_FORM": "tiled", "ANNCHOR_EMIT_RUN_MIN": str(1 << 60), "ANNCHOR_KEEP_COLS_MIN": str(1 << 60),
         "ANNCHOR_EMIT_SUPER_MIN": str(1 << 30), "ANNCHOR_LOC_THRESH_HIST": "1", "ANNCHOR_UPDATE_BOUNDS": "pairs", "ANNCHOR_COMP_GENERIC": "1"}


def test_thinned_list_kernels_equal_older_forms_at_their_own_size(tmp_path):
    """20 000 points, candidates thinned by the locality filter (2 of the 4 nearest anchors in common): the kernels that take over
    by default there -- anchor-outer features, run-based / super-tiled emission, column-stationary keep bitmap, register-counter
    locality threshold, bit-table update_bounds on 2-byte keys, vectorised computed-neighbour CSR -- against the forms they
    replaced, whole state bit-identical."""
    a = _run(tmp_path, "new", "euclidean_thinned", {})
    b = _run(tmp_path, "older", "euclidean_thinned", OLDER)
    assert 20_000_000 < int(a["n_pairs"]) < 95_000_000   # (under half of all pairs: the anchor-outer feature kernel)
    for key in ("A", "D", "evals", "n_pairs", "features", "ncm", "RA", "idx", "dist"):
        assert np.array_equal(a[key], b[key]), key


def test_no_device_memory_leak_across_contexts():
    """Every buffer a context allocated outside its slab is released with it: pair lists past the slab's 32 M pairs (here 45 M:
    float64 Euclidean, 9500 points) allocate ~40 buffers of their own; the device's free memory after six create / fit / close
    cycles is what it was after the first (round 3 found 18 buffers missing from the release list: 1.3 GB per context at
    127 M pairs, and gigabyte allocations answered in seconds once the device filled up)."""
    from annchor_amd import Annchor, _native
    from annchor_amd.samplers import DeviceStratifiedSampler

    rng = np.random.default_rng(3)
    X = (rng.standard_normal((9500, 5)) @ rng.standard_normal((5, 24))).astype(np.float64)
    free = []
    for rep in range(6):
        ann = Annchor(X, "euclidean", n_anchors=16, n_neighbors=10, p_work=0.05, sampler=DeviceStratifiedSampler()).fit()
        assert ann.n_pairs > 40_000_000
        ann._engine.close()
        del ann
        _native.load_library().annchor_release_parked()
        f, t = _native._i64(), _native._i64()
        assert _native.load_library().annchor_device_mem_info(0, __import__("ctypes").byref(f), __import__("ctypes").byref(t)) == 0
        free.append(f.value)
    assert free[-1] >= free[0] - (64 << 20), free


def test_update_bounds_forms_agree_beyond_65536_points(monkeypatch):
    """The three forms of update_bounds on a thinned list of 70 000 points (2.4 x 10^8 candidates, 4 x 10^7 lookahead pairs): the
    wave-per-pair binary-search form, the row-grouped form on 2-byte keys (the lists' keys >= 65 536 carry their 17th bit through
    the per-list split position) and on 4-byte keys must leave bit-identical bounds."""
    from annchor_amd import Annchor
    from annchor_amd.samplers import DeviceStratifiedSampler

    n = 70000
    rng = np.random.default_rng(11)
    cent = rng.standard_normal((40, 6)) * 4
    X = np.round(cent[rng.integers(0, 40, n)] + rng.standard_normal((n, 6)), 2)
    cfg = dict(n_anchors=40, n_neighbors=15, p_work=0.01, n_samples=5000, locality=5, loc_thresh=3, random_seed=2)
    ref = None
    for form in ("pairs", "bits16", "bits32"):
        monkeypatch.setenv("ANNCHOR_UPDATE_BOUNDS", form)
        ann = Annchor(X, "euclidean", sampler=DeviceStratifiedSampler(), **cfg)
        ann.get_anchors(); ann.get_locality(); ann.get_features(); ann.get_sample(); ann.fit_predict_regression(); ann.fit_predict_errors()
        ann.select_refine_candidate_pairs(w=0.5, it=0)
        before = ann.features[:, :2].copy()
        ann._invalidate("features")
        ann.update_anchor_points()
        cur = ann.features[:, :2].copy()
        ann._engine.close()
        assert (cur[:, 0] >= before[:, 0]).all() and (cur[:, 1] <= before[:, 1]).all()
        assert (cur != before).any()
        if ref is None:
            ref = cur
        else:
            assert np.array_equal(ref, cur), form
