"""Device-fitted models (minimum-norm solution for rank-deficient partitions) against the host / LAPACK route: how many graph
entries differ, and is either graph better against brute force?  Continuous Euclidean data (exactly dependent bounds in the
first partition) and an integer lattice (mathematically tied predictions)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from annchor_amd import Annchor, BruteForce, compare_neighbor_graphs
rng = np.random.default_rng(5)
cases = []
Z = rng.standard_normal((4000, 6))
cases.append(("continuous 4000 x 48", (Z @ rng.standard_normal((6, 48)) + 0.05 * rng.standard_normal((4000, 48))).astype(np.float64),
              dict(n_anchors=24, n_neighbors=15, p_work=0.1, n_samples=5000)))
cases.append(("lattice 639 x 3 (values 0..4)", rng.integers(0, 5, (639, 3)).astype(np.float64), dict(n_anchors=11, n_neighbors=16, p_work=0.2, n_samples=1332, niters=1)))
cases.append(("lattice 3000 x 4 (values 0..11)", rng.integers(0, 12, (3000, 4)).astype(np.float64), dict(n_anchors=12, n_neighbors=10, p_work=0.15)))
for name, X, cfg in cases:
    k = cfg["n_neighbors"]
    bf = BruteForce(X, "euclidean").fit().neighbor_graph if hasattr(BruteForce(X, "euclidean").fit(), "neighbor_graph") else None
    a = Annchor(X, "euclidean", ols="device", **cfg).fit()
    b = Annchor(X, "euclidean", ols="lapack", **cfg).fit()
    nd = int((a.neighbor_graph[1] != b.neighbor_graph[1]).sum())
    bfk = (bf[0][:, :k], bf[1][:, :k]) if bf is not None and bf[0].shape[1] >= k else None
    ea = compare_neighbor_graphs(bfk, a.neighbor_graph, k) if bfk else -1
    eb = compare_neighbor_graphs(bfk, b.neighbor_graph, k) if bfk else -1
    print("%-32s evals %d / %d   differing entries %d of %d   errors vs brute force: device %d, lapack %d" % (
        name, a.evals, b.evals, nd, a.neighbor_graph[1].size, ea, eb))
