"""Worker of test_large_list_paths_gpu.py: one pair-list fit, state dumped to an .npz (the kernel-path
thresholds are read from the environment once per process, hence a process per setting)."""
import sys
import numpy as np

sys.path.insert(0, ".")
from annchor_amd import Annchor  # noqa: E402

out, metric = sys.argv[1], sys.argv[2]
rng = np.random.default_rng(17)
if metric == "euclidean":
    n, k = 2500, 15
    Z = rng.standard_normal((n, 5))
    X = (Z @ rng.standard_normal((5, 24)) + 0.05 * rng.standard_normal((n, 24))).astype(np.float64)
    ann = Annchor(X, "euclidean", n_anchors=16, n_neighbors=k, p_work=0.1, n_samples=3000)
else:
    n, k = 1500, 12
    seeds = ["".join(rng.choice(list("ACGT"), rng.integers(50, 120))) for _ in range(12)]
    X = []
    for _ in range(n):
        s = list(seeds[rng.integers(0, 12)])
        for _ in range(rng.integers(0, 15)):
            s[rng.integers(0, len(s))] = rng.choice(list("ACGT"))
        X.append("".join(s))
    ann = Annchor(X, "levenshtein", n_anchors=12, n_neighbors=k, p_work=0.15, n_samples=2000)
ann.fit()
idx, dist = ann.neighbor_graph
np.savez(out, idx=idx, dist=dist, evals=ann.evals, A=np.asarray(ann.A), D=ann.D, RA=ann.RefineApprox,
         ncm=ann.not_computed_mask, features=ann.features, n_pairs=ann.n_pairs)
