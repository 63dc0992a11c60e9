"""Refine-like and anchor-like Levenshtein launches only (for rocprofv3 --pmc passes, tools/pmc_lev2.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from annchor_amd import _native
from annchor_amd.datasets import load_strings
from annchor_amd.distances import levenshtein
X = load_strings()["X"]
eng = _native.Engine(0)
levenshtein.bind(eng, X)
rng = np.random.default_rng(0)
nx = len(X)
big = np.ascontiguousarray(rng.integers(0, nx, (65536, 2)), np.int64)
one = np.ascontiguousarray(np.stack([np.full(nx, 7), np.arange(nx)], 1), np.int64)
eng.prof_enable(True)
for _ in range(5):
    eng.metric_pairs(big)
for _ in range(5):
    eng.metric_pairs(one)
# the picker's own one-to-all launches (k_lev_a2: two pairs per wave, forward / backward half-chains)
eng.pick_anchors_selected([7, 1126, 543, 2, 640])
# the max-min picker's 15 rounds as ONE persistent launch (k_lev_ap; ANNCHOR_LEV_R set = a forced variant: the per-round path)
for _ in range(3):
    eng.pick_anchors_maxmin(15, 1126)
p = eng.prof_get()["levenshtein_pairs"]
print("R=%s: %.1f us avg over %d launches" % (os.environ.get("ANNCHOR_LEV_R", "auto"), p["ms"] / p["launches"] * 1e3, p["launches"]))
