"""A/B inside one process (the switch is read per launch): short pair lists through k_lev_p2 (packed forward / backward
kernel) or through k_lev_f."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ANNCHOR_RNG_NO_CACHE", "1")
import numpy as np
from annchor_amd import Annchor, _native
from annchor_amd.datasets import load_strings
print("affinity:", _native.bind_to_device_numa(0))
X = load_strings()["X"]
cfg = dict(n_anchors=15, n_neighbors=25, p_work=0.12, random_seed=42)
for a in [Annchor(X, "levenshtein", **cfg) for _ in range(5)]:
    a.fit()
for rep in range(6):
    for mode in ("0", "auto"):
        os.environ.pop("ANNCHOR_LEV_P2_MAX", None)
        if mode != "auto":
            os.environ["ANNCHOR_LEV_P2_MAX"] = mode
        anns = [Annchor(X, "levenshtein", **cfg) for _ in range(20)]
        for a in anns:
            a._engine.prof_enable(2)
        gc.collect(); gc.disable()
        t0 = time.perf_counter()
        for a in anns:
            a.fit()
        el = (time.perf_counter() - t0) / len(anns) * 1e3
        gc.enable()
        lev = np.mean([a._engine.prof_get()["levenshtein_pairs"]["ms"] for a in anns])
        print("rep %d p2_max=%-6s %.3f ms/fit   levenshtein %.3f ms/fit   get_sample %.3f" % (rep, mode, el, lev, np.median([a.timings["get_sample"] for a in anns]) * 1e3))
        for a in anns:
            a._engine.close()
