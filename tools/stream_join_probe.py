"""C3-size probe of the streamed form: fit time / recall (10 000-row subset, exact truth from the
streamed query with the full budget) for a grid of (p_work, join_passes)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from annchor_amd.streamed import StreamedAnnchor

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
grid = [(0.1, 0), (0.1, 1), (0.1, 2), (0.1, 3), (0.05, 2), (0.05, 3), (0.03, 3)]
rng = np.random.default_rng(1234)
Z = rng.standard_normal((N, 8)); W = rng.standard_normal((8, 128))
X = (Z @ W + 0.05 * rng.standard_normal((N, 128))).astype(np.float32)
k = 15
rows = np.sort(np.random.default_rng(99).choice(N, 10000, replace=False))
truth = None
for pw, jp in grid:
    sa = StreamedAnnchor(X, n_anchors=32, n_neighbors=k, p_work=pw, join_passes=jp)
    sa._engine.prof_enable(True)
    t = time.perf_counter(); sa.fit(); dt = time.perf_counter() - t
    if truth is None:
        ti, td = sa.query(X[rows], nn=k + 1, p_work=1.0)
        truth = td[:, 1:]
        assert np.all(td[:, 0] < 1e-3)
    got = sa.neighbor_graph[1][rows][:, 1:]
    # recall by distance multiset (compare_neighbor_graphs semantics, 3 decimals)
    from annchor_amd import compare_neighbor_graphs
    err = compare_neighbor_graphs((np.zeros_like(truth, dtype=np.int64), truth), (np.zeros_like(got, dtype=np.int64), got), k - 1)
    prof = sa._engine.prof_get()
    ks = {n: round(v["ms"], 1) for n, v in prof.items() if v["ms"] > 0.5}
    print("p_work %.2f joins %d: fit %.3f s  recall %.5f  tile_evals %d  %s" % (pw, jp, dt, 1 - err / (10000.0 * (k - 1)), sa.tile_evals, ks), flush=True)
    sa._engine.close()
