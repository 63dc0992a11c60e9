"""A/B of the streamed tile kernel at C3 (N = 10^6, d = 128): one child process per ANNCHOR_ST_KERNEL setting
(bf4 = knnbf.hip's split-bf16 kernel, the default; 4wave = the exact-f32 k_st_knn); prints fit time, tile-kernel time / TFLOP/s and recall on 10 000 rows."""
import json, os, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(n):
    import numpy as np
    from annchor_amd import compare_neighbor_graphs
    from annchor_amd.streamed import StreamedAnnchor
    from bench import euclid_shard

    X = euclid_shard(0, n)
    k = 15
    res = []
    for rep in range(3):
        sa = StreamedAnnchor(X, n_anchors=32, n_neighbors=k, p_work=0.1)
        sa._engine.prof_enable(True)
        t = time.perf_counter(); sa.fit(); dt = time.perf_counter() - t
        prof = sa._engine.prof_get()
        g = prof["stream_tile_gemm_topk"]["ms"] * 1e-3
        tp, jc = sa._engine.stream_last_counts()
        res.append(dict(fit_s=round(dt, 4), gemm_s=round(g, 4), tile_pairs=int(tp), tflops=round(tp * 128.0 * 128 * 256 / g / 1e12, 1),
                        join_ms=round(prof.get("stream_join_gemm_topk", {"ms": 0})["ms"], 2)))
        if rep < 2:
            sa._engine.close()
    rows = np.sort(np.random.default_rng(99).choice(n, 10000, replace=False))
    ti, td = sa.query(X[rows], nn=k, p_work=1.0)
    err = compare_neighbor_graphs((ti, td), (sa.neighbor_graph[0][rows], sa.neighbor_graph[1][rows]), k)
    print(json.dumps(dict(kernel=os.environ.get("ANNCHOR_ST_KERNEL", "bf4"), n=n, runs=res, recall=1 - err / (10000.0 * k))), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]))
    else:
        n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
        for kern in (sys.argv[2:] or ["bf4", "4wave"]):
            env = dict(os.environ, ANNCHOR_ST_KERNEL=kern)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n)], env=env)
