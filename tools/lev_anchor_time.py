"""Time of the picker's anchor rounds (annchor_pick_anchors_maxmin, C2 strings): k_lev_a vs the pair-list kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from annchor_amd import _native
from annchor_amd.distances import levenshtein
from annchor_amd.datasets import load_strings

X = list(load_strings()["X"])
for mode, ldsreq in (("2", "0"), ("2", "40960"), ("2", "53248"), ("2", "0"), ("2", "40960"), ("0", "0")):
    os.environ["ANNCHOR_LEV_ANCHOR"] = mode
    os.environ["ANNCHOR_LEV_A2_LDS"] = ldsreq
    eng = _native.Engine(0)
    levenshtein.bind(eng, X)
    eng.pick_anchors_maxmin(15, 1126)
    eng.synchronize()
    eng.prof_enable(True)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        eng.pick_anchors_maxmin(15, 1126)
        eng.synchronize()
        ts.append(time.perf_counter() - t0)
    p = eng.prof_get()["levenshtein_pairs"]
    A = eng.download(_native.F_A)
    print("LDS request %s ANNCHOR_LEV_ANCHOR=%s: %.1f us per launch (events), 15 rounds wall %.1f us (min %.1f), A[:5]=%s" % (
        ldsreq, mode, p["ms"] / p["launches"] * 1e3, np.median(ts) * 1e6, min(ts) * 1e6, A[:5]))
    eng.close()
