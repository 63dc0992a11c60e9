"""Where does the packed latency kernel (k_lev_p2) stop winning over the throughput kernel (k_lev_f)?  Kernel time (HIP events) of
pair-list launches of n random pairs of the C2 strings, both ways."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from annchor_amd import _native
from annchor_amd.distances import levenshtein
from annchor_amd.datasets import load_strings
X = load_strings()["X"]
eng = _native.Engine(0)
levenshtein.bind(eng, X)
rng = np.random.default_rng(1)
eng.prof_enable(2)
for n in (1000, 2500, 5000, 8000, 12000, 16000, 24000, 32000):
    IJ = rng.integers(0, len(X), (n, 2))
    res = {}
    for mode in ("0", "1000000"):
        os.environ["ANNCHOR_LEV_P2_MAX"] = mode
        eng.metric_pairs(IJ)
        eng.prof_reset()
        for _ in range(5):
            d = eng.metric_pairs(IJ)
        p = eng.prof_get()["levenshtein_pairs"]
        res[mode] = (p["ms"] / p["launches"] * 1e3, d)
    assert np.array_equal(res["0"][1], res["1000000"][1])
    print("n=%6d  k_lev_f %7.1f us   k_lev_p2 %7.1f us" % (n, res["0"][0], res["1000000"][0]))
