"""Per-kernel-family device time of one C2 fit (HIP events), plus host stage times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from annchor_amd import Annchor
from annchor_amd.datasets import load_strings
X = load_strings()["X"]
cfg = dict(n_anchors=15, n_neighbors=25, p_work=0.12)
kw = dict(ols=sys.argv[1]) if len(sys.argv) > 1 else {}
for _ in range(3):
    Annchor(X, "levenshtein", **cfg, **kw).fit()
a = Annchor(X, "levenshtein", **cfg, **kw)
a._engine.prof_enable(1)
a.fit()
tot = 0
for name, e in sorted(a._engine.prof_get().items(), key=lambda kv: -kv[1]["ms"]):
    if e["launches"]:
        print("%-28s %7.1f us  %3d launches" % (name, e["ms"] * 1e3, e["launches"]))
        tot += e["ms"]
print("device total %.3f ms; refused: %s" % (tot, a.__dict__.get("_device_model_refused")))
ts = []
for _ in range(10):
    b = Annchor(X, "levenshtein", **cfg, **kw)
    t = time.perf_counter(); b.fit(); ts.append(time.perf_counter() - t)
print("fit median %.3f ms  min %.3f" % (sorted(ts)[5] * 1e3, min(ts) * 1e3), {k: round(v * 1e3, 3) for k, v in b.timings.items()})
