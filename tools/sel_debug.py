"""kth_uncomputed_dad at N = 16000 (127 M pairs) against NumPy on the downloaded column."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from annchor_amd import Annchor, _native
rng = np.random.default_rng(5)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
Z = rng.standard_normal((n, 6))
X = (Z @ rng.standard_normal((6, 48)) + 0.05 * rng.standard_normal((n, 48))).astype(np.float64)
ann = Annchor(X, "euclidean", n_anchors=24, n_neighbors=15, p_work=0.05, n_samples=5000)
ann.get_anchors(); ann.get_locality(); ann.get_features()
eng = ann._engine
F = eng.download(_native.F_FEATURES).reshape(-1, 4)
ncm = eng.download(_native.F_NCM).astype(bool)
dad = F[:, 2]
pool = np.sort(dad[ncm])
m = pool.size
for ks in ([m // 100, m - m // 100], [m // 2], [5, m - 5]):
    got = eng.kth_uncomputed_dad(np.asarray(ks, dtype=np.int64))
    print(ks, got, pool[ks], np.array_equal(got, pool[ks]))
