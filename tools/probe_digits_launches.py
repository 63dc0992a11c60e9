import sys, time, numpy as np
sys.path.insert(0, '.')
from annchor_amd import Annchor
from annchor_amd.datasets import load_digits
d = load_digits(); X, M = d["X"], d["cost_matrix"]
ann = Annchor(X, "wasserstein", func_kwargs={"cost_matrix": M}, n_anchors=20, n_neighbors=25, n_samples=5000, p_work=0.16)
ann.fit()
ann = Annchor(X, "wasserstein", func_kwargs={"cost_matrix": M}, n_anchors=20, n_neighbors=25, n_samples=5000, p_work=0.16)
ann.fit()
print({k: round(v * 1e3, 2) for k, v in ann.timings.items()})
eng = ann._engine
eng.prof_enable(True)
rng = np.random.default_rng(0)
nx = len(X)
for label, IJ in (("one-to-all 1797", np.stack([np.full(nx, 7), np.arange(nx)], 1)),
                  ("sample 5000", rng.integers(0, nx, (5000, 2))),
                  ("refine 108000", rng.integers(0, nx, (108000, 2)))):
    IJ = np.ascontiguousarray(IJ, np.int64)
    eng.metric_pairs(IJ); eng.prof_reset()
    for _ in range(5): eng.metric_pairs(IJ)
    p = eng.prof_get()["wasserstein_pairs"]
    print("   %s: %.1f us / launch  (%.3f us/pair)" % (label, p["ms"] / p["launches"] * 1e3, p["ms"] / p["launches"] * 1e3 / len(IJ)))
