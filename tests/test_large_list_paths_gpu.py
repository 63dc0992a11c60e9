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
