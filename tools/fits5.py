import sys
sys.path.insert(0, '.')
from annchor_amd import Annchor
from annchor_amd.datasets import load_strings
X = load_strings()["X"]
anns = [Annchor(X, "levenshtein", n_anchors=15, n_neighbors=25, p_work=0.12) for _ in range(8)]
for a in anns:
    a.fit()
