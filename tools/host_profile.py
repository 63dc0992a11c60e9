import cProfile, pstats, sys, time, io
sys.path.insert(0, '.')
from annchor_amd import Annchor
from annchor_amd.datasets import load_strings
X = load_strings()["X"]
for _ in range(3):
    Annchor(X, "levenshtein", n_anchors=15, n_neighbors=25, p_work=0.12).fit()
anns = [Annchor(X, "levenshtein", n_anchors=15, n_neighbors=25, p_work=0.12) for _ in range(10)]
pr = cProfile.Profile()
pr.enable()
for a in anns:
    a.fit()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
