"""Wall-clock of the whole life of an Annchor object on the C2 workload: constructor (engine +
upload), fit(), release -- what a caller who builds one graph per object pays."""
import sys, time, gc, numpy as np
sys.path.insert(0, '.')
from annchor_amd import Annchor
from annchor_amd.datasets import load_strings
X = load_strings()["X"]
cfg = dict(n_anchors=15, n_neighbors=25, p_work=0.12)
for _ in range(3):
    Annchor(X, "levenshtein", **cfg).fit()
gc.collect(); gc.disable()
tc, tf, td = [], [], []
for _ in range(40):
    t0 = time.perf_counter(); a = Annchor(X, "levenshtein", **cfg); t1 = time.perf_counter()
    a.fit(); t2 = time.perf_counter()
    del a; t3 = time.perf_counter()
    tc.append(t1 - t0); tf.append(t2 - t1); td.append(t3 - t2)
med = lambda v: float(np.median(v)) * 1e3
print("constructor %.2f ms  fit %.2f ms  release %.2f ms  total %.2f ms" % (med(tc), med(tf), med(td), med(tc) + med(tf) + med(td)))
