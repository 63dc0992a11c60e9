"""Per-fit wall times of the headline workload, the way bench.py runs it (one Annchor object per fit, constructed before
the timed region): spread across fits, with and without the per-kernel HIP events of the timed region.
usage: fit_times.py [n_fits]"""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ANNCHOR_RNG_NO_CACHE", "1")
import numpy as np
from annchor_amd import Annchor, _native
from annchor_amd.datasets import load_strings
n = int(sys.argv[1]) if len(sys.argv) > 1 else 23
try:
    print("affinity:", _native.bind_to_device_numa(0))
except Exception as e:
    print("no affinity", e)
X = load_strings()["X"]
cfg = dict(n_anchors=15, n_neighbors=25, p_work=0.12, random_seed=42)
for events in (0, 2, 0):
    anns = [Annchor(X, "levenshtein", **cfg) for _ in range(n)]
    for a in anns[:3]:
        a.fit()
    for a in anns[3:]:
        a._engine.prof_enable(events)
    gc.collect(); gc.disable()
    ts, tot = [], []
    t0 = time.perf_counter()
    for a in anns[3:]:
        s = time.perf_counter(); a.fit(); ts.append(time.perf_counter() - s); tot.append(a.timings["total"])
    el = time.perf_counter() - t0
    gc.enable()
    ts = np.array(ts) * 1e3
    print("events=%d: mean %.3f ms (loop %.3f)  median %.3f  min %.3f  max %.3f" % (events, ts.mean(), el / len(ts) * 1e3, np.median(ts), ts.min(), ts.max()))
    print("   ", " ".join("%.2f" % v for v in ts))
    st = {}
    for a in anns[3:]:
        for k, v in a.timings.items():
            st.setdefault(k, []).append(v * 1e3)
    print("    stage medians:", {k: round(float(np.median(v)), 3) for k, v in st.items()})
    print("    stage means  :", {k: round(float(np.mean(v)), 3) for k, v in st.items()})
    for a in anns:
        a._engine.close()
