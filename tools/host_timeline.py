"""Host-side timeline of one C2 fit without a profiler attached: every stage of fit() (enter / exit) and every host wait of the
library (ANNCHOR_SYNC_TIMING=1: when the wait began, how long it blocked), on one monotonic clock.  A wait that returns at once
means the GPU was idle waiting for the host before it; a long one means the host was ahead.
usage: host_timeline.py [digits] [ENV=VALUE ...]     (digits: BASELINE configs[3], exact-OT Wasserstein, instead of the strings)"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ANNCHOR_RNG_NO_CACHE", "1")
os.environ["ANNCHOR_SYNC_TIMING"] = "1"
digits = "digits" in sys.argv[1:]
for kv in sys.argv[1:]:
    if "=" in kv:
        k, v = kv.split("=", 1)
        os.environ[k] = v
log = tempfile.NamedTemporaryFile(prefix="timeline", suffix=".log", delete=False)
os.dup2(log.fileno(), 2)   # the library's stderr lines
import numpy as np
from annchor_amd import Annchor, _native
from annchor_amd.datasets import load_strings, load_digits
_native.bind_to_device_numa(0)
if digits:
    d = load_digits()
    X, metric, kw = d["X"], "wasserstein", {"func_kwargs": {"cost_matrix": d["cost_matrix"]}}
    cfg = dict(n_anchors=20, n_neighbors=25, n_samples=5000, p_work=0.16, random_seed=42)
else:
    X, metric, kw = load_strings()["X"], "levenshtein", {}
    cfg = dict(n_anchors=15, n_neighbors=25, p_work=0.12, random_seed=42)
STAGES = ["get_anchors", "get_locality", "get_features", "get_sample", "fit_predict_regression", "fit_predict_errors",
          "select_refine_candidate_pairs", "update_anchor_points", "get_ann"]
marks = []
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.monotonic_ns()
        try:
            return f(*a, **k)
        finally:
            marks.append((name, t0, time.monotonic_ns()))
    setattr(obj, name, g)
anns = [Annchor(X, metric, **kw, **cfg) for _ in range(12)]
spans = []
for a in anns:
    for s in STAGES:
        wrap(a, s)
    t0 = time.monotonic_ns()
    a.fit()
    spans.append((t0, time.monotonic_ns()))
sys.stderr.flush()
lines = [l.split() for l in open(log.name) if l.startswith("T ")]
ev = [(int(l[3]), "wait %s" % l[2] if l[1] == "wait" else "scan %s" % l[2], int(l[4]), int(l[5])) for l in lines]
f0, f1 = spans[-1]
print("fit: %.3f ms (all: %s)" % ((f1 - f0) / 1e6, " ".join("%.2f" % ((b - a) / 1e6) for a, b in spans)))
rows = [(t0, "stage " + n, t0, t1) for n, t0, t1 in marks if f0 <= t0 <= f1] + [e for e in ev if f0 <= e[0] <= f1]
blocked = 0
for t_in, what, t_w, t_out in sorted(rows):
    if what.startswith("stage"):
        print("%9.1f us  %-46s %8.1f us" % ((t_in - f0) / 1e3, what, (t_out - t_in) / 1e3))
    else:
        print("%9.1f us      %-42s parked work %6.1f us, blocked %7.1f us" % ((t_in - f0) / 1e3, what, (t_w - t_in) / 1e3, (t_out - t_w) / 1e3))
        if what.startswith("wait"):
            blocked += t_out - t_w
print("host blocked in waits: %.3f ms of %.3f" % (blocked / 1e6, (f1 - f0) / 1e6))
