import sys, numpy as np, traceback
sys.path.insert(0, '.')
from annchor_amd import Annchor, BruteForce
def run(name, fn):
    try:
        r = fn(); print("OK  ", name, r if r is not None else "")
    except BaseException as e:
        print("EXC ", name, type(e).__name__, str(e)[:140].replace("\n", " | "))
rng = np.random.default_rng(0)
S = ["".join(rng.choice(list("abcd"), rng.integers(5, 30))) for _ in range(60)]
run("tiny strings n=60 k=5", lambda: Annchor(S, "levenshtein", n_anchors=5, n_neighbors=5, n_samples=200, p_work=0.5).fit().neighbor_graph[0].shape)
run("identical strings", lambda: Annchor(["aaaa"] * 50, "levenshtein", n_anchors=4, n_neighbors=5, n_samples=100, p_work=0.9).fit().neighbor_graph[1].max())
run("n=12 < locality needs", lambda: Annchor(S[:12], "levenshtein", n_anchors=4, n_neighbors=3, n_samples=20, p_work=1.0).fit().neighbor_graph[0].shape)
run("n_anchors > nx", lambda: Annchor(S[:8], "levenshtein", n_anchors=12, n_neighbors=3, n_samples=20, p_work=1.0).fit().neighbor_graph[0].shape)
run("k > nx", lambda: Annchor(S[:8], "levenshtein", n_anchors=3, n_neighbors=20, n_samples=20, p_work=1.0).fit().neighbor_graph[0].shape)
run("empty strings present", lambda: Annchor(S + ["", ""], "levenshtein", n_anchors=5, n_neighbors=5, n_samples=200, p_work=0.5).fit().neighbor_graph[1][-1])
X = rng.standard_normal((40, 3))
run("euclid n=40", lambda: Annchor(X, "euclidean", n_anchors=5, n_neighbors=4, n_samples=100, p_work=0.6).fit().neighbor_graph[0].shape)
run("euclid duplicates", lambda: Annchor(np.repeat(X[:10], 5, axis=0), "euclidean", n_anchors=5, n_neighbors=4, n_samples=100, p_work=0.9).fit().neighbor_graph[1][:, 1].max())
run("brute n=1", lambda: BruteForce(X[:1], "euclidean").fit().neighbor_graph[0].shape)
run("brute n=3 k default", lambda: BruteForce(X[:3], "euclidean").fit().neighbor_graph[0].shape)
run("1-d data", lambda: Annchor(rng.standard_normal(80), "euclidean", n_anchors=5, n_neighbors=4, n_samples=200, p_work=0.6).fit().neighbor_graph[0].shape)
run("p_work > 1", lambda: Annchor(S, "levenshtein", n_anchors=5, n_neighbors=5, n_samples=200, p_work=3.0).fit().p_work)
run("bad metric string", lambda: Annchor(S, "hamming"))
run("wasserstein no kwargs", lambda: Annchor(X, "wasserstein"))
