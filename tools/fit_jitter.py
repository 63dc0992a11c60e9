"""Distribution of C2 fit() wall times (is the mean dragged by outliers?)."""
import sys, time, gc, numpy as np
sys.path.insert(0, '.')
from annchor_amd import Annchor
from annchor_amd.datasets import load_strings
X = load_strings()["X"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
anns = [Annchor(X, "levenshtein", n_anchors=15, n_neighbors=25, p_work=0.12) for _ in range(n + 3)]
for a in anns[:3]: a.fit()
gc.collect(); gc.disable()
ts = []
for a in anns[3:]:
    t = time.perf_counter(); a.fit(); a._engine.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
ts = np.array(ts)
print("fits %d  mean %.2f  median %.2f  p10 %.2f  p90 %.2f  min %.2f  max %.2f ms" % (len(ts), ts.mean(), np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90), ts.min(), ts.max()))
st = {}
for a in anns[3:]:
    for k, v in a.timings.items(): st.setdefault(k, []).append(v * 1e3)
print({k: (round(float(np.median(v)), 2), round(float(np.max(v)), 2)) for k, v in st.items()})
