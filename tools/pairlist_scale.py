"""Pair-list kernels at a size where their HBM traffic dominates launch latency (N = 16000
Euclidean points: 128 M candidate pairs would not fit the locality budget, so loc_min keeps ~20 M)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from annchor_amd import Annchor
rng = np.random.default_rng(5)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
Z = rng.standard_normal((n, 6))
X = (Z @ rng.standard_normal((6, 48)) + 0.05 * rng.standard_normal((n, 48))).astype(np.float64)
for rep in range(2):
    ann = Annchor(X, "euclidean", n_anchors=24, n_neighbors=15, p_work=0.05, n_samples=5000)
    ann._engine.prof_enable(1)
    t = time.perf_counter(); ann.fit(); dt = time.perf_counter() - t
print("N=%d pairs=%d fit %.1f ms evals %d" % (n, ann.n_pairs, dt * 1e3, ann.evals))
for name, e in sorted(ann._engine.prof_get().items(), key=lambda kv: -kv[1]["ms"]):
    if e["launches"]:
        us = e["ms"] / e["launches"] * 1e3
        print("  %-28s %8.1f us/launch x %3d   %7.1f GB/s algorithmic (%.1f %% of 8 TB/s)" % (
            name, us, e["launches"], e["alg_bytes"] / e["launches"] / us / 1e3, e["alg_bytes"] / e["launches"] / us / 1e3 / 80))
print("host stage ms:", {k: round(v * 1e3, 1) for k, v in ann.timings.items()})
