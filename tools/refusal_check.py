"""Does any of the usual configurations still make fit() start over with the host solver (a device-fitted partition refused)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from annchor_amd import Annchor
from annchor_amd.datasets import load_strings, load_digits
rng = np.random.default_rng(1)
cases = []
S = load_strings()["X"]
cases.append(("C2 strings", S, "levenshtein", {}, dict(n_anchors=15, n_neighbors=25, p_work=0.12)))
cases.append(("README strings", S, "levenshtein", {}, dict(n_anchors=20, n_neighbors=15, p_work=0.1)))
D = load_digits()
cases.append(("C4 digits", D["X"], "wasserstein", {"cost_matrix": D["cost_matrix"]}, dict(n_anchors=20, n_neighbors=25, n_samples=5000, p_work=0.16)))
Z = rng.standard_normal((5000, 8))
cases.append(("euclid f64 5000", (Z @ rng.standard_normal((8, 32))).astype(np.float64), "euclidean", {}, dict(n_anchors=20, n_neighbors=15, p_work=0.1)))
cases.append(("euclid f32 5000", (Z @ rng.standard_normal((8, 32))).astype(np.float32), "euclidean", {}, dict(n_anchors=20, n_neighbors=15, p_work=0.1)))
cases.append(("cosine f64 5000", (Z @ rng.standard_normal((8, 32))).astype(np.float64) + 3.0, "cosine", {}, dict(n_anchors=20, n_neighbors=15, p_work=0.1)))
cases.append(("int grid 3000", rng.integers(0, 12, (3000, 3)).astype(np.float64), "euclidean", {}, dict(n_anchors=10, n_neighbors=10, p_work=0.2)))
for name, X, metric, fk, cfg in cases:
    for niters in (2, 4):
        a = Annchor(X, metric, func_kwargs=fk or None, niters=niters, **cfg)
        t = time.perf_counter(); a.fit(); dt = time.perf_counter() - t
        print("%-18s niters=%d fit %.4f s  on_device=%s refused=%s" % (name, niters, dt, a.__dict__.get("_model_on_device"), getattr(a, "_device_model_refused", None)))
