"""Streamed query at N = 1M: throughput and recall against the work budget."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from annchor_amd.streamed import StreamedAnnchor
rng = np.random.default_rng(1234); n = 1000000; nq = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
W = rng.standard_normal((8, 128))
X = (rng.standard_normal((n, 8)) @ W + 0.05 * rng.standard_normal((n, 128))).astype(np.float32)
Q = (rng.standard_normal((nq, 8)) @ W + 0.05 * rng.standard_normal((nq, 128))).astype(np.float32)
sa = StreamedAnnchor(X, n_anchors=32, n_neighbors=15, p_work=0.1).fit()
rows = rng.choice(nq, 200, replace=False)
Xd = X.astype(np.float64)
truth = np.array([np.sort(np.sqrt(((Xd - Q[r].astype(np.float64)) ** 2).sum(1)))[:15] for r in rows])
for pw in (0.05, 0.1, 0.2, 0.3, 0.5):
    sa.query(Q[:1000], nn=15, p_work=pw)
    t = time.perf_counter(); idx, dist = sa.query(Q, nn=15, p_work=pw); dt = time.perf_counter() - t
    ok = sum(np.isclose(dist[r], truth[t_], rtol=1e-5).sum() for t_, r in enumerate(rows))
    print("nq=%d p_work=%.2f: %.3f s  %.2f M queries/s  recall@15 %.4f" % (nq, pw, dt, nq / dt / 1e6, ok / (200 * 15)))
