"""A/B inside one process: the second sampling step's draw on the library's worker thread (overlapping the refinement kernel
from another core) or deferred to finish_device on the calling thread (after refinement and update_bounds are enqueued)."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ANNCHOR_RNG_NO_CACHE", "1")
import numpy as np
import annchor_amd.annchor as A
from annchor_amd import Annchor, _native
from annchor_amd.datasets import load_strings
print("affinity:", _native.bind_to_device_numa(0))
X = load_strings()["X"]
cfg = dict(n_anchors=15, n_neighbors=25, p_work=0.12, random_seed=42)
for a in [Annchor(X, "levenshtein", **cfg) for _ in range(5)]:
    a.fit()
for rep in range(4):
    for mode in (True, "defer"):
        A._DRAW_OVERLAP = mode
        anns = [Annchor(X, "levenshtein", **cfg) for _ in range(20)]
        gc.collect(); gc.disable()
        t0 = time.perf_counter()
        for a in anns:
            a.fit()
        el = (time.perf_counter() - t0) / len(anns) * 1e3
        gc.enable()
        print("rep %d draw=%-6s %.3f ms/fit   get_sample %.3f select %.3f" % (rep, "worker" if mode is True else "defer", el,
              np.median([a.timings["get_sample"] for a in anns]) * 1e3, np.median([a.timings["select_refine_candidate_pairs"] for a in anns]) * 1e3))
        for a in anns:
            a._engine.close()
