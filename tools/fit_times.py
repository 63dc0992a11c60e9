"""Per-fit wall times of the C2 workload (distribution, not just the mean)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ.setdefault("ANNCHOR_RNG_NO_CACHE", "1")
from annchor_amd import Annchor, _native
from annchor_amd.datasets import load_strings
_native.bind_to_device_numa(0)
X = load_strings()["X"]
cfg = dict(n_anchors=15, n_neighbors=25, p_work=0.12, random_seed=42)
anns = [Annchor(X, "levenshtein", **cfg) for _ in range(43)]
ts = []
for a in anns:
    t = time.perf_counter(); a.fit(); ts.append((time.perf_counter() - t) * 1e3)
ts = np.array(ts[3:])
print("fits %d: mean %.3f median %.3f min %.3f max %.3f ms; p90 %.3f" % (len(ts), ts.mean(), np.median(ts), ts.min(), ts.max(), np.quantile(ts, 0.9)))
print("stages of the last fit:", {k: round(v * 1e3, 3) for k, v in anns[-1].timings.items()})
