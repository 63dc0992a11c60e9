import sys, time, numpy as np
sys.path.insert(0, '.')
from annchor_amd import Annchor, BruteForce, compare_neighbor_graphs
from annchor_amd.datasets import load_digits
d = load_digits(); X, M = d["X"], d["cost_matrix"]; ng = d["neighbor_graph"]
for rep in range(3):
    ann = Annchor(X, "wasserstein", func_kwargs={"cost_matrix": M}, n_anchors=20, n_neighbors=25, n_samples=5000, p_work=0.16)
    ann._engine.prof_enable(True)
    t = time.perf_counter(); ann.fit(); dt = time.perf_counter() - t
    err = compare_neighbor_graphs(ng, ann.neighbor_graph, 25)
    p = ann._engine.prof_get()
    print("C4 digits fit %.1f ms evals %d errors %d  emd: %.2f ms over %d launches (%.2f us/pair)" % (
        dt * 1e3, ann.evals, err, p["wasserstein_pairs"]["ms"], p["wasserstein_pairs"]["launches"], p["wasserstein_pairs"]["ms"] * 1e3 / ann.evals))
t = time.perf_counter(); bf = BruteForce(X, "wasserstein", func_kwargs={"cost_matrix": M}).fit(n_neighbors=100); dt = time.perf_counter() - t
print("BruteForce digits (1 613 706 EMDs) %.1f ms, errors vs stored graph %d" % (dt * 1e3, compare_neighbor_graphs(ng, bf.neighbor_graph, 100)))
from annchor_amd.datasets import load_strings
Xs = load_strings()["X"]
t = time.perf_counter(); bf = BruteForce(Xs, "levenshtein").fit(n_neighbors=100); dt = time.perf_counter() - t
print("BruteForce strings (1 279 200 Levenshtein) %.1f ms" % (dt * 1e3))
