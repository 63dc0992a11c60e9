"""Host waits of one C2 fit: which engine call (and which stage) each stream synchronisation belongs to.
ANNCHOR_SYNC_TRACE=1 makes the library name every wait on stderr; the Engine methods and the stages are logged in between."""
import os, sys, collections
os.environ["ANNCHOR_SYNC_TRACE"] = "1"
os.environ.setdefault("ANNCHOR_RNG_NO_CACHE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import annchor_amd._native as N
from annchor_amd import Annchor
from annchor_amd.datasets import load_strings

def say(s):
    os.write(2, (s + "\n").encode())

for name in dir(N.Engine):
    f = getattr(N.Engine, name)
    if callable(f) and not name.startswith("_"):
        def wrap(f=f, name=name):
            def g(self, *a, **k):
                say("call " + name)
                return f(self, *a, **k)
            return g
        setattr(N.Engine, name, wrap())
X = load_strings()["X"]
cfg = dict(n_anchors=15, n_neighbors=25, p_work=0.12, random_seed=42)
Annchor(X, "levenshtein", **cfg).fit()
ann = Annchor(X, "levenshtein", **cfg)
for st in ("get_anchors", "get_locality", "get_features", "get_sample", "fit_predict_regression", "fit_predict_errors",
           "select_refine_candidate_pairs", "update_anchor_points", "get_ann"):
    f = getattr(Annchor, st)
    def wrap(f=f, st=st):
        def g(self, *a, **k):
            say("stage " + st)
            return f(self, *a, **k)
        return g
    setattr(Annchor, st, wrap())
say("=== fit")
ann.fit()
say("=== end")
