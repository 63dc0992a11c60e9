"""C3 parameter sweep of the streamed form: fit time / recall for (p_work, join passes, early-stop window / tau, join budget share)."""
import sys, time, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from annchor_amd.streamed import StreamedAnnchor
from annchor_amd import compare_neighbor_graphs

N = 1_000_000
rng = np.random.default_rng(1234)
Z = rng.standard_normal((N, 8)); W = rng.standard_normal((8, 128))
X = (Z @ W + 0.05 * rng.standard_normal((N, 128))).astype(np.float32)
k = 15
rows = np.sort(np.random.default_rng(99).choice(N, 10000, replace=False))
truth = None
grid = [dict(pw=0.1, jp=2, win=64, tau=None, div=8, cap=24),
        dict(pw=0.1, jp=2, win=32, tau=None, div=8, cap=48), dict(pw=0.1, jp=2, win=32, tau=None, div=8, cap=64),
        dict(pw=0.1, jp=3, win=32, tau=None, div=8, cap=48), dict(pw=0.1, jp=2, win=32, tau=40, div=8, cap=64),
        dict(pw=0.1, jp=3, win=32, tau=40, div=8, cap=64), dict(pw=0.1, jp=4, win=32, tau=40, div=8, cap=48),
        dict(pw=0.1, jp=2, win=48, tau=None, div=8, cap=48), dict(pw=0.1, jp=3, win=48, tau=30, div=8, cap=48)]
for g in grid:
    os.environ["ANNCHOR_ST_EARLY_WINDOW"] = str(g["win"])
    if g["tau"] is None: os.environ.pop("ANNCHOR_ST_EARLY_TAU", None)
    else: os.environ["ANNCHOR_ST_EARLY_TAU"] = str(g["tau"])
    os.environ["ANNCHOR_JOIN_DIV"] = str(g["div"])
    ts = []
    for rep in range(2):
        sa = StreamedAnnchor(X, n_anchors=32, n_neighbors=k, p_work=g["pw"], join_passes=g["jp"])
        t = time.perf_counter(); sa.fit(); ts.append(time.perf_counter() - t)
        if rep == 0: sa._engine.close()
    if truth is None:
        ti, td = sa.query(X[rows], nn=k, p_work=1.0)
        truth = (ti, td)
    err = compare_neighbor_graphs(truth, (sa.neighbor_graph[0][rows], sa.neighbor_graph[1][rows]), k)
    print("%s: fit %.3f s  recall %.5f  tile_evals %d  budget %s" % (g, min(ts), 1 - err / (10000.0 * k), sa.tile_evals, sa._budget(sa.n_tiles_total)), flush=True)
    sa._engine.close()
