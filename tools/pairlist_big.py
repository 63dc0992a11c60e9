"""The pair-list form beyond 30 000 points: how far does the materialised candidate list go on one 288 GB GPU?
usage: pairlist_big.py N [metric] [p_work] [locality] [loc_thresh] [n_anchors]   (metric: euclidean | levenshtein)
Beyond the size whose complete pair list fits (46 341 points) the locality filter has to thin the candidates: e.g.
  pairlist_big.py 100000 levenshtein 0.005 5 3 24"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import annchor_amd.annchor as A
from annchor_amd import Annchor, BruteForce, compare_neighbor_graphs
from annchor_amd.samplers import DeviceStratifiedSampler
n = int(sys.argv[1])
metric = sys.argv[2] if len(sys.argv) > 2 else "euclidean"
rng = np.random.default_rng(5)
if metric == "euclidean":
    Z = rng.standard_normal((n, 6))
    X = (Z @ rng.standard_normal((6, 48)) + 0.05 * rng.standard_normal((n, 48))).astype(np.float64)
else:
    from annchor_amd.datasets import synthetic_string_clusters
    X = synthetic_string_clusters(n, cluster=int(os.environ.get("CLUSTER", 2000)))
k = 15
p_work = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
locality = int(sys.argv[4]) if len(sys.argv) > 4 else 5
loc_thresh = int(sys.argv[5]) if len(sys.argv) > 5 else 1
n_anchors = int(sys.argv[6]) if len(sys.argv) > 6 else 24
t = time.perf_counter()
ann = Annchor(X, metric, n_anchors=n_anchors, n_neighbors=k, p_work=p_work, n_samples=5000, locality=locality, loc_thresh=loc_thresh,
              sampler=DeviceStratifiedSampler())
tc = time.perf_counter() - t
ann._engine.prof_enable(1)
t = time.perf_counter(); ann.fit(); dt = time.perf_counter() - t
print("N=%d %s pairs=%d ctor %.2f s fit %.3f s evals %d (%.2f %% of all pairs)" % (n, metric, ann.n_pairs, tc, dt, ann.evals, 100.0 * ann.evals / (n * (n - 1) / 2)))
print("host stage s:", {kk: round(v, 3) for kk, v in ann.timings.items()})
for name, e in sorted(ann._engine.prof_get().items(), key=lambda kv: -kv[1]["ms"])[:int(os.environ.get("TOPK", 8))]:
    print("  %-28s %9.2f ms x %d" % (name, e["ms"] / max(1, e["launches"]), e["launches"]))
# recall on a row subset against exact rows (metric_pairs one-to-all)
rows = rng.choice(n, 200, replace=False)
err = 0
eng = ann._engine
for r in rows:
    IJ = np.stack([np.full(n, r), np.arange(n)], axis=1)
    d = eng.metric_pairs(IJ)
    d[r] = -1
    want = np.sort(d)[:k]; want[0] = 0
    got = ann.neighbor_graph[1][r]
    err += compare_neighbor_graphs((np.zeros((1, k), dtype=np.int64), want[None, :]), (np.zeros((1, k), dtype=np.int64), got[None, :]), k)
print("recall@%d on %d rows: %.4f" % (k, len(rows), 1 - err / (len(rows) * k)))
