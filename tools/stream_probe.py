import sys, time, numpy as np
sys.path.insert(0, '.')
from annchor_amd.streamed import StreamedAnnchor
from annchor_amd import compare_neighbor_graphs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
pws = [float(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0.02, 0.05, 0.1, 0.2, 1.0]
rng = np.random.default_rng(1234); Z = rng.standard_normal((n, 8)); W = rng.standard_normal((8, 128))
X = (Z @ W + 0.05 * rng.standard_normal((n, 128))).astype(np.float32)
rows = np.random.default_rng(1).choice(n, 300, replace=False)
Xd = X.astype(np.float64); bi = []; bd = []
for r in rows:
    d = np.sqrt(((Xd - Xd[r]) ** 2).sum(1)); d[r] = -1; o = np.argsort(d)[:15]; bi.append(o); bd.append(np.maximum(d[o], 0))
bi = np.array(bi); bd = np.array(bd)
for pw in pws:
    sa = StreamedAnnchor(X, n_anchors=32, n_neighbors=15, p_work=pw)
    t = time.time(); sa.fit(); dt = time.time() - t
    err = compare_neighbor_graphs((bi, bd), (sa.neighbor_graph[0][rows], sa.neighbor_graph[1][rows]), 15)
    nt = sa.n_tiles_total
    print("n=%d p_work=%.2f fit %.3fs tiles %d (%.3f of all) recall %.4f timings %s kernel_ms %.1f" % (
        n, pw, dt, sa.tile_evals, sa.tile_evals / nt / nt, 1 - err / (300 * 15), {k: round(v, 3) for k, v in sa.timings.items()}, sa._engine.last_kernel_ms()))
