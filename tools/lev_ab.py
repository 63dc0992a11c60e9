import os, sys, time, numpy as np
sys.path.insert(0, '.')
from annchor_amd import Annchor
from annchor_amd.datasets import load_strings
X = load_strings()["X"]
ann = Annchor(X, "levenshtein", n_anchors=15, n_neighbors=25, p_work=0.12)
for rep in range(3):
    ann2 = Annchor(X, "levenshtein", n_anchors=15, n_neighbors=25, p_work=0.12); ann2._engine.prof_enable(True)
    t = time.perf_counter(); ann2.fit(); dt = time.perf_counter() - t
    p = ann2._engine.prof_get()["levenshtein_pairs"]
    if rep == 2: print("   stages (ms):", {k: round(v * 1e3, 2) for k, v in ann2.timings.items()})
    print("R=%s ILP=%s fit %.2f ms lev %.3f ms / %d launches" % (os.environ.get("ANNCHOR_LEV_R", "auto"), "1", dt * 1e3, p["ms"], p["launches"]))

# launch-type split: an anchor-like launch (one string against all) and a refine-like launch
from annchor_amd.distances import DeviceMetric
import torch
rng = np.random.default_rng(0)
eng = ann2._engine
nx = len(X)
for label, IJ in (("anchor-like 1600", np.stack([np.full(nx, 7), np.arange(nx)], 1)),
                  ("refine-like 65536", rng.integers(0, nx, (65536, 2)))):
    IJ = np.ascontiguousarray(IJ, np.int64)
    eng.metric_pairs(IJ)
    eng.prof_reset()
    for _ in range(10):
        eng.metric_pairs(IJ)
    p = eng.prof_get()["levenshtein_pairs"]
    print("   %s: %.1f us / launch" % (label, p["ms"] / p["launches"] * 1e3))
