"""Repeatability: the same fit many times must give the same graph (bit for bit)."""
import sys, hashlib, numpy as np
sys.path.insert(0, '.')
from annchor_amd import Annchor
from annchor_amd.datasets import load_strings
X = load_strings()["X"]
seen = {}
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 200):
    ann = Annchor(X, "levenshtein", n_anchors=15, n_neighbors=25, p_work=0.12).fit()
    h = hashlib.sha1(ann.neighbor_graph[0].tobytes() + ann.neighbor_graph[1].tobytes()).hexdigest()
    seen[h] = seen.get(h, 0) + 1
print("strings C2: %d fits, %d distinct graphs %s" % (sum(seen.values()), len(seen), seen))
from annchor_amd.streamed import StreamedAnnchor
rng = np.random.default_rng(1234); n = 200000
Xe = (rng.standard_normal((n, 8)) @ rng.standard_normal((8, 128)) + 0.05 * rng.standard_normal((n, 128))).astype(np.float32)
seen = {}
for rep in range(6):
    sa = StreamedAnnchor(Xe, n_anchors=32, n_neighbors=15, p_work=0.1).fit()
    h = hashlib.sha1(sa.neighbor_graph[0].tobytes() + sa.neighbor_graph[1].tobytes()).hexdigest()
    seen[h] = seen.get(h, 0) + 1
print("streamed N=200000 p_work=0.1: %d fits, %d distinct graphs" % (sum(seen.values()), len(seen)))
