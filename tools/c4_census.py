"""BASELINE configs[3] (digits, Wasserstein): stage times and kernel families of one fit."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from annchor_amd import Annchor
from annchor_amd.datasets import load_digits
d = load_digits()
X, M = d["X"], d["cost_matrix"]
cfg = dict(n_anchors=20, n_neighbors=25, p_work=0.16)
for _ in range(2):
    Annchor(X, "wasserstein", func_kwargs={"cost_matrix": M}, **cfg).fit()
ann = Annchor(X, "wasserstein", func_kwargs={"cost_matrix": M}, **cfg)
ann._engine.prof_enable(1)
t = time.perf_counter(); ann.fit(); dt = time.perf_counter() - t
print("fit %.1f ms evals %d" % (dt * 1e3, ann.evals))
for name, e in sorted(ann._engine.prof_get().items(), key=lambda kv: -kv[1]["ms"])[:8]:
    if e["launches"]:
        print("  %-28s %8.1f us/launch x %3d = %.2f ms" % (name, e["ms"] / e["launches"] * 1e3, e["launches"], e["ms"]))
print("host stage ms:", {k: round(v * 1e3, 2) for k, v in ann.timings.items()})
