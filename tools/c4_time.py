"""C4 (digits, exact-OT Wasserstein) fit time; ANNCHOR_EMD_WAVES limits the waves per CU of the OT kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from annchor_amd import Annchor
from annchor_amd.datasets import load_digits
d = load_digits()
cfg = dict(n_anchors=20, n_neighbors=25, n_samples=5000, p_work=0.16, random_seed=42)
ts = []
for r in range(4):
    a = Annchor(d["X"], "wasserstein", func_kwargs={"cost_matrix": d["cost_matrix"]}, **cfg)
    t = time.perf_counter(); a.fit(); ts.append(time.perf_counter() - t)
print("EMD_WAVES=%s: fit %.1f ms (min of 4)" % (os.environ.get("ANNCHOR_EMD_WAVES", "default"), min(ts) * 1e3))
