"""cProfile of the C2 fit on the host side (where the wall-clock goes besides the device work)."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ANNCHOR_RNG_NO_CACHE", "1")
import numpy as np
from annchor_amd import Annchor
from annchor_amd.datasets import load_strings
X = load_strings()["X"]
cfg = dict(n_anchors=15, n_neighbors=25, p_work=0.12, random_seed=42)
anns = [Annchor(X, "levenshtein", **cfg) for _ in range(45)]
for a in anns[:5]:
    a.fit()
pr = cProfile.Profile()
pr.enable()
for a in anns[5:]:
    a.fit()
pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("tottime")
ps.print_stats(28)
print(s.getvalue()[:6000])
