"""A mid-size slow-metric fit with the DEFAULT plugins (where does the time go?): usage mid_profile.py N [metric]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from annchor_amd import Annchor
from annchor_amd.datasets import synthetic_string_clusters
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
X = synthetic_string_clusters(n)
cfg = dict(n_anchors=24, n_neighbors=15, p_work=0.05)
Annchor(X, "levenshtein", **cfg).fit()
for rep in range(2):
    ann = Annchor(X, "levenshtein", **cfg)
    ann._engine.prof_enable(1)
    t = time.perf_counter(); ann.fit(); dt = time.perf_counter() - t
print("N=%d pairs=%d fit %.1f ms evals %d refused=%s" % (n, ann.n_pairs, dt * 1e3, ann.evals, getattr(ann, "_device_model_refused", None)))
print("host stage ms:", {k: round(v * 1e3, 2) for k, v in ann.timings.items()})
tot = 0
for name, e in sorted(ann._engine.prof_get().items(), key=lambda kv: -kv[1]["ms"])[:14]:
    print("  %-28s %9.3f ms total  x %d" % (name, e["ms"], e["launches"])); tot += e["ms"]
print("  (top kernels sum %.2f ms)" % tot)
