"""Randomised check of the streamed form: with the full budget (p_work = 1) the graph must equal the exact k-NN
(distances within f32 rounding), for random N / dim / k / anchors, single rank and the sharded code path
(force_exchange).  python tools/stress_streamed.py [n_cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from annchor_amd.streamed import StreamedAnnchor
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
bad = 0
for case in range(ncases):
    n = int(rng.integers(300, 9000)); dim = int(rng.choice([3, 7, 32, 33, 64, 100, 128, 200, 256])); k = int(rng.integers(2, 34))
    na = int(rng.integers(2, 40)); lat = int(rng.integers(2, 9))
    X = (rng.standard_normal((n, lat)) @ rng.standard_normal((lat, dim)) + 0.05 * rng.standard_normal((n, dim))).astype(np.float32)
    if case % 3 == 0:
        X[rng.integers(0, n, n // 10)] = X[rng.integers(0, n, n // 10)]   # duplicated rows
    sa = StreamedAnnchor(X, n_anchors=na, n_neighbors=k, p_work=1.0, force_exchange=bool(case % 2)).fit()
    idx, dist = sa.neighbor_graph
    Xd = X.astype(np.float64)
    rows = rng.choice(n, min(n, 400), replace=False)
    d2 = np.maximum((Xd[rows] ** 2).sum(1)[:, None] + (Xd ** 2).sum(1)[None, :] - 2.0 * Xd[rows] @ Xd.T, 0)
    d2[np.arange(len(rows)), rows] = -1
    truth = np.sqrt(np.maximum(np.sort(d2, axis=1)[:, :k], 0))
    ok = np.allclose(dist[rows], truth, rtol=2e-4, atol=2e-4) and np.array_equal(idx[rows, 0], rows)
    # reported pairs carry their true distance
    rep = np.sqrt(((Xd[idx[rows]] - Xd[rows][:, None, :]) ** 2).sum(-1))
    ok = ok and np.allclose(rep, dist[rows], rtol=1e-4, atol=1e-4)
    bad += not ok
    print("case %2d n=%5d dim=%3d k=%2d na=%2d exchange=%d -> %s" % (case, n, dim, k, na, case % 2, "OK" if ok else "MISMATCH"), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
