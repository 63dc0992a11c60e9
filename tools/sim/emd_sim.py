"""Writes the C4-like pair sample for tools/sim/emd_sim.cpp: digits histograms, cost matrix, near pairs (from the stored exact
100-NN graph, with their stored distances) and random pairs."""
import os, sys, struct
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from annchor_amd.datasets import load_digits
d = load_digits()
X, M, ng = d["X"], d["cost_matrix"], d["neighbor_graph"]
rng = np.random.default_rng(0)
near_i = rng.integers(0, X.shape[0], 8000)
near_k = rng.integers(1, 60, 8000)
pairs = [(i, int(ng[0][i, k]), ng[1][i, k]) for i, k in zip(near_i, near_k)]
pairs += [(int(i), int(j), np.nan) for i, j in rng.integers(0, X.shape[0], (2000, 2)) if i != j]
with open(sys.argv[1], "wb") as f:
    f.write(struct.pack("iii", X.shape[0], X.shape[1], len(pairs)))
    f.write(np.ascontiguousarray(X, dtype=np.float64).tobytes())
    f.write(np.ascontiguousarray(M, dtype=np.float64).tobytes())
    f.write(np.asarray([(p[0], p[1]) for p in pairs], dtype=np.int32).tobytes())
    f.write(np.asarray([p[2] for p in pairs], dtype=np.float64).tobytes())
