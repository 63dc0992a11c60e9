"""The kernels that take over on long pair lists (tiled feature kernel, column-half transposes + streamed
row kernels, second candidate cut, ECDF bucket index, sampled-bracket selection) must give bit-identical state to the small-list
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
         "ANNCHOR_SEL_SAMPLE_MIN": "1"}


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
    b = _run(tmp_path, "large", metric, LARGE)
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
