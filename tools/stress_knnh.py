"""Randomised check of the two-stage tile kernel's shapes (padded dimension 128, <= 14 neighbours kept): with the full budget the graph
must equal the exact k-NN (float64 brute force on sampled rows; the early stop of the sharded path switched off), duplicated rows and
clustered data included; with a small budget two
builds must agree bit for bit.  python tools/stress_knnh.py [n_cases] [seed]"""
import os, sys
os.environ.setdefault("ANNCHOR_ST_EARLY_WINDOW", "0")   # (builds followed by join passes stop a row tile early when its yield dries up: off -- full budget = exact)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from annchor_amd.streamed import StreamedAnnchor
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
bad = 0
for case in range(ncases):
    n = int(rng.integers(1500, 70000)); dim = int(rng.choice([65, 100, 127, 128])); k = int(rng.integers(2, 16))
    na = int(rng.integers(2, 40)); lat = int(rng.integers(2, 10))
    X = (rng.standard_normal((n, lat)) @ rng.standard_normal((lat, dim)) + 0.05 * rng.standard_normal((n, dim))).astype(np.float32)
    if case % 3 == 0:
        X[rng.integers(0, n, n // 10)] = X[rng.integers(0, n, n // 10)]   # duplicated rows
    if case % 4 == 1:
        X += (30.0 * rng.standard_normal((1, dim))).astype(np.float32)    # far from the origin (the kernels centre the data)
    if case % 5 == 2:
        X[: n // 2] = (X[: n // 2] * 1e-2 + X[0]).astype(np.float32)        # a tight cluster beside a wide cloud
    sa = StreamedAnnchor(X, n_anchors=na, n_neighbors=k, p_work=1.0, force_exchange=bool(case % 2)).fit()
    two_stage = sa._engine.stream_last_tile_kernels()[0] if hasattr(sa._engine, "stream_last_tile_kernels") else None
    idx, dist = sa.neighbor_graph
    Xd = X.astype(np.float64)
    rows = rng.choice(n, min(n, 300), replace=False)
    d2 = ((Xd[rows][:, None, :] - Xd[None, :, :]) ** 2).sum(-1) if n <= 6000 else np.maximum((Xd[rows] ** 2).sum(1)[:, None] + (Xd ** 2).sum(1)[None, :] - 2.0 * Xd[rows] @ Xd.T, 0)
    d2[np.arange(len(rows)), rows] = -1
    truth = np.sqrt(np.maximum(np.sort(d2, axis=1)[:, :k], 0))
    scale = max(1.0, float(np.abs(Xd).max()))
    ok = np.allclose(dist[rows], truth, rtol=3e-4, atol=3e-4 * scale) and np.array_equal(idx[rows, 0], rows)
    rep = np.sqrt(((Xd[idx[rows]] - Xd[rows][:, None, :]) ** 2).sum(-1))
    ok = ok and np.allclose(rep, dist[rows], rtol=1e-4, atol=1e-4 * scale)
    if not ok:
        badrows = np.where(~np.isclose(dist[rows], truth, rtol=3e-4, atol=3e-4 * scale).all(1))[0]
        print("   rows off: %d of %d; first: row %d got %s want %s; self first: %s; reported-vs-true ok: %s" % (
            len(badrows), len(rows), rows[badrows[0]] if len(badrows) else -1, dist[rows[badrows[0]]][:6] if len(badrows) else None,
            truth[badrows[0]][:6] if len(badrows) else None, np.array_equal(idx[rows, 0], rows), np.allclose(rep, dist[rows], rtol=1e-4, atol=1e-4 * scale)))
    # small budget: deterministic
    a = StreamedAnnchor(X, n_anchors=na, n_neighbors=k, p_work=0.1).fit().neighbor_graph
    b = StreamedAnnchor(X, n_anchors=na, n_neighbors=k, p_work=0.1).fit().neighbor_graph
    det = np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    bad += (not ok) or (not det)
    print("case %2d n=%5d dim=%3d k=%2d na=%2d exchange=%d two-stage=%s -> %s%s" % (case, n, dim, k, na, case % 2, two_stage, "OK" if ok else "MISMATCH", "" if det else " NOT DETERMINISTIC"), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
