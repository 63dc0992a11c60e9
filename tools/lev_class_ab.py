"""A/B of the Levenshtein slot classes: 65 536 random pairs with and without the short-pattern class
(ANNCHOR_LEV_CLASS_MIN decides; read once per process -> one process per setting), results compared."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import numpy as np
    from annchor_amd import _native
    from annchor_amd.datasets import load_strings
    from annchor_amd.distances import levenshtein
    X = load_strings()["X"]
    eng = _native.Engine(0)
    levenshtein.bind(eng, X)
    rng = np.random.default_rng(0)
    big = np.ascontiguousarray(rng.integers(0, len(X), (65536, 2)), np.int64)
    eng.metric_pairs(big)
    eng.prof_enable(True)
    for _ in range(10):
        d = eng.metric_pairs(big)
    p = eng.prof_get()["levenshtein_pairs"]
    print("%s: %.1f us per 65536 pairs, checksum %d" % (sys.argv[1], p["ms"] / p["launches"] * 1e3, int(d.sum())))
else:
    for tag, v in (("one class", "2000000000"), ("two classes", "4096")):
        subprocess.run([sys.executable, __file__, tag], env=dict(os.environ, ANNCHOR_LEV_CLASS_MIN=v))
