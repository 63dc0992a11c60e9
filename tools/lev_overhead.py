import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from annchor_amd import _native
from annchor_amd.distances import levenshtein
rng = np.random.default_rng(0)
for L in (8, 64, 128, 256, 500):
    X = ["".join(rng.choice(list("abcdefghijklmnopqrstuvwxyz"), L)) for _ in range(1600)]
    X[0] = "".join(rng.choice(list("abcdefghijklmnopqrstuvwxyz"), 594))  # same slot width as the fixture
    eng = _native.Engine(0)
    levenshtein.bind(eng, X)
    one = np.ascontiguousarray(np.stack([np.full(1600, 7), np.arange(1600)], 1), np.int64)
    eng.metric_pairs(one)
    eng.prof_enable(True)
    for _ in range(20):
        eng.metric_pairs(one)
    p = eng.prof_get()["levenshtein_pairs"]
    print("len %d: %.1f us per one-to-all launch" % (L, p["ms"] / p["launches"] * 1e3))
    eng.close()
