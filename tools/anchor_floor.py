"""Fixed cost of an anchor round (k_lev_a2): 15 rounds over 1600 strings of a given length, kernel time per round."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from annchor_amd import _native
from annchor_amd.distances import levenshtein
rng = np.random.default_rng(0)
for L in (8, 40, 128, 256, 500):
    X = ["".join(rng.choice(list("abcdefghijklmnopqrstuvwxyz"), L + int(rng.integers(0, 5)))) for _ in range(1600)]
    eng = _native.Engine(0)
    levenshtein.bind(eng, X)
    eng.pick_anchors_maxmin(15, 0)
    eng.prof_enable(2)
    eng.prof_reset()
    for _ in range(5):
        eng.pick_anchors_maxmin(15, 0)
    p = eng.prof_get()["levenshtein_pairs"]
    print("len %4d: %.1f us per round (%d launches)" % (L, p["ms"] / p["launches"] * 1e3, p["launches"]))
    eng.close()
