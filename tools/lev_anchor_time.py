"""Time of the picker's anchor rounds (annchor_pick_anchors_maxmin, C2 strings): all rounds in one launch (k_lev_ap) against
the round-by-round launches of k_lev_a2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from annchor_amd import _native
from annchor_amd.distances import levenshtein
from annchor_amd.datasets import load_strings

X = list(load_strings()["X"])
ref = None
for persist in ("1", "0", "1", "0"):
    os.environ["ANNCHOR_LEV_PERSIST"] = persist
    eng = _native.Engine(0)
    levenshtein.bind(eng, X)
    eng.pick_anchors_maxmin(15, 1126)
    eng.synchronize()
    eng.prof_enable(True)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        eng.pick_anchors_maxmin(15, 1126)
        eng.synchronize()
        ts.append(time.perf_counter() - t0)
    p = eng.prof_get()["levenshtein_pairs"]
    A = eng.download(_native.F_A)
    D = eng.download(_native.F_D)
    if ref is None:
        ref = (A, D)
    same = np.array_equal(A, ref[0]) and np.array_equal(D, ref[1])
    print("ANNCHOR_LEV_PERSIST=%s: events %.1f us per call of 15 rounds (%d event pairs), wall %.1f us (min %.1f), A[:5]=%s same=%s" % (
        persist, p["ms"] / 21 * 1e3, p["launches"], np.median(ts) * 1e6, min(ts) * 1e6, A[:5], same))
    eng.close()
