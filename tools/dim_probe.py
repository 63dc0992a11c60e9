"""Streamed form at rows of more than 128 dimensions (the k-blocked kernel, csrc/knnbk.hip): fit time, tile-phase time and recall@15
against a float64 brute force on 1000 rows.   python tools/dim_probe.py N d [p_work]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

from annchor_amd import compare_neighbor_graphs
from annchor_amd.streamed import StreamedAnnchor
from test_c5_gpu import truth_f64

n, d = int(sys.argv[1]), int(sys.argv[2])
pw = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
g = torch.Generator(device="cuda"); g.manual_seed(7)
W = torch.randn(8, d, generator=g, device="cuda")
X = (torch.randn(n, 8, generator=g, device="cuda") @ W + 0.05 * torch.randn(n, d, generator=g, device="cuda")).cpu().numpy()
torch.cuda.empty_cache()
k = 15
res = []
for rep in range(3):
    sa = StreamedAnnchor(X, n_anchors=32, n_neighbors=k, p_work=pw)
    sa._engine.prof_enable(True)
    t = time.perf_counter(); sa.fit(); dt = time.perf_counter() - t
    prof = {a: round(b["ms"], 2) for a, b in sa._engine.prof_get().items() if b["ms"] > 0.3}
    tp, jc = sa._engine.stream_last_counts()
    g_s = prof.get("stream_tile_gemm_topk", 0) * 1e-3
    res.append(dict(fit_s=round(dt, 4), kernels_ms=prof, tile_pairs=int(tp), tflops_split=round(tp * 128.0 * 128 * 2 * 3 * d / max(g_s, 1e-9) / 1e12, 1)))
    if rep < 2:
        sa._engine.close()
rows = np.sort(np.random.default_rng(3).choice(n, 1000, replace=False))
gi, gd = sa.neighbor_graph
sa._engine.close()
bd = truth_f64(X[rows], [X], k); bd[:, 0] = 0
err = compare_neighbor_graphs((gi[rows], bd), (gi[rows], gd[rows]), k)
print(json.dumps(dict(n=n, d=d, p_work=pw, runs=res, recall_at_15=1 - err / (1000.0 * k))))
