"""C2 fit and the N=16000 pair-list fit with the DeviceStratifiedSampler plugin vs the default sampler."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ.setdefault("ANNCHOR_RNG_NO_CACHE", "1")
from annchor_amd import Annchor, _native, compare_neighbor_graphs
from annchor_amd.datasets import load_strings
from annchor_amd.samplers import DeviceStratifiedSampler
_native.bind_to_device_numa(0)
X = load_strings()["X"]
cfg = dict(n_anchors=15, n_neighbors=25, p_work=0.12, random_seed=42)
G = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "strings_full.npz"))
truth = (G["truth_idx"].astype(np.int64), G["truth_dist"].astype(np.float64))
for name, mk in (("default", lambda: None), ("device", lambda: DeviceStratifiedSampler())):
    anns = [Annchor(X, "levenshtein", sampler=mk(), **cfg) for _ in range(33)]
    ts = []
    for a in anns:
        t = time.perf_counter(); a.fit(); ts.append((time.perf_counter() - t) * 1e3)
    ts = np.array(ts[3:])
    print("C2 %s sampler: median %.3f ms  errors %d  stages %s" % (name, np.median(ts), compare_neighbor_graphs(truth, anns[-1].neighbor_graph, 25),
          {k: round(v * 1e3, 2) for k, v in anns[-1].timings.items()}), flush=True)
    for a in anns: a._engine.close()
rng = np.random.default_rng(5); n = 16000
Z = rng.standard_normal((n, 6)); Xe = (Z @ rng.standard_normal((6, 48)) + 0.05 * rng.standard_normal((n, 48))).astype(np.float64)
for name, mk in (("default", lambda: None), ("device", lambda: DeviceStratifiedSampler())):
    Annchor(Xe, "euclidean", n_anchors=24, n_neighbors=15, p_work=0.05, sampler=mk()).fit()
    a = Annchor(Xe, "euclidean", n_anchors=24, n_neighbors=15, p_work=0.05, sampler=mk())
    t = time.perf_counter(); a.fit(); dt = time.perf_counter() - t
    print("N=16000 %s sampler: fit %.1f ms  stages %s" % (name, dt * 1e3, {k: round(v * 1e3, 1) for k, v in a.timings.items()}), flush=True)
    a._engine.close()
