"""C2 fits with the anchor rounds as one persistent launch (k_lev_ap) and as 15 launches (k_lev_a2), alternating inside one
process: Levenshtein time by the library's events and the fit's wall time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from annchor_amd import Annchor
from annchor_amd.datasets import load_strings
X = np.array(list(load_strings()["X"]))
cfg = dict(n_anchors=15, n_neighbors=25, p_work=0.12, random_seed=42)
res = {"1": [], "0": []}
for rep in range(40):
    mode = "1" if rep % 2 == 0 else "0"
    os.environ["ANNCHOR_LEV_PERSIST"] = mode
    a = Annchor(X, "levenshtein", **cfg)
    a._engine.prof_enable(True)
    t = time.perf_counter(); a.fit(); dt = time.perf_counter() - t
    p = a._engine.prof_get()["levenshtein_pairs"]
    if rep >= 4:
        res[mode].append((dt * 1e3, p["ms"], a.timings["get_anchors"] * 1e3, a.timings["get_locality"] * 1e3))
    a._engine.close()
for mode in ("1", "0"):
    r = np.array(res[mode])
    print("ANNCHOR_LEV_PERSIST=%s: fit %.3f ms (median), levenshtein events %.3f ms, get_anchors %.3f + get_locality %.3f ms host" % (
        mode, np.median(r[:, 0]), np.median(r[:, 1]), np.median(r[:, 2]), np.median(r[:, 3])))
