"""Where the N = 100 000 Levenshtein fit (bench.py: levenshtein_100k_thinned_pairlist) spends its time: usage lev100k_profile.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from annchor_amd import Annchor
from annchor_amd.datasets import synthetic_string_clusters
from annchor_amd.samplers import DeviceStratifiedSampler
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
X = synthetic_string_clusters(n)
cfg = dict(n_anchors=60, n_neighbors=15, p_work=0.02, n_samples=5000, locality=5, loc_thresh=3)
reps = int(os.environ.get('LEV100K_REPS', '2'))
for rep in range(reps):
    ann = Annchor(X, "levenshtein", sampler=DeviceStratifiedSampler(), **cfg)
    if rep == reps - 1:
        ann._engine.prof_enable(1)
    t = time.perf_counter(); ann.fit(); dt = time.perf_counter() - t
    print("rep %d: N=%d pairs=%d fit %.1f ms evals %d" % (rep, n, ann.n_pairs, dt * 1e3, ann.evals))
    if dt > 1.0:
        print("   slow rep, host stage ms:", {k: round(v * 1e3, 1) for k, v in ann.timings.items()})
    if rep < reps - 1:
        ann._engine.close()
print("host stage ms:", {k: round(v * 1e3, 2) for k, v in ann.timings.items()})
tot = 0
for name, e in sorted(ann._engine.prof_get().items(), key=lambda kv: -kv[1]["ms"])[:24]:
    print("  %-28s %9.3f ms total  x %d" % (name, e["ms"], e["launches"])); tot += e["ms"]
print("  (top kernels sum %.2f ms)" % tot)
