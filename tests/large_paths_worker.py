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
elif metric == "euclidean_thinned":
    # a locality-thinned list at a size where the thinned-list kernels are the DEFAULT (20 000 points: 6.3 M bitmap words)
    n, k = 20000, 15
    cent = rng.standard_normal((30, 6)) * 4
    X = np.round(cent[rng.integers(0, 30, n)] + rng.standard_normal((n, 6)), 2)
    from annchor_amd.samplers import DeviceStratifiedSampler
    ann = Annchor(X, "euclidean", n_anchors=30, n_neighbors=k, p_work=0.02, n_samples=4000, locality=4, loc_thresh=2,
                  sampler=DeviceStratifiedSampler(), random_seed=4)
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
