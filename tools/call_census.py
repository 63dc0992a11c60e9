"""Per-fit census of the C-ABI calls Annchor.fit() makes on the C2 workload: calls and host
wall-clock per entry point (each includes whatever synchronisation it does)."""
import sys, time, collections
sys.path.insert(0, '.')
from annchor_amd import Annchor, _native
from annchor_amd.datasets import load_strings
X = load_strings()["X"]
cfg = dict(n_anchors=15, n_neighbors=25, p_work=0.12)
for _ in range(3):
    Annchor(X, "levenshtein", **cfg).fit()
stat = collections.defaultdict(lambda: [0, 0.0])
lib = _native._load() if hasattr(_native, "_load") else None
orig = {}
for name in _native._SIGNATURES:
    for holder in [getattr(_native.Engine, "_lib_holder", None)]:
        pass
eng_cls = _native.Engine
def wrap(fn, name):
    def w(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        s = stat[name]; s[0] += 1; s[1] += time.perf_counter() - t
        return r
    return w
for name in dir(eng_cls):
    if name.startswith("_"):
        continue
    f = getattr(eng_cls, name)
    if callable(f):
        setattr(eng_cls, name, wrap(f, name))
NF = 10
anns = [Annchor(X, "levenshtein", **cfg) for _ in range(NF)]
stat.clear()
t0 = time.perf_counter()
for a in anns:
    a.fit()
tot = time.perf_counter() - t0
print("fit %.3f ms" % (tot / NF * 1e3))
acc = 0.0
for name, (c, t) in sorted(stat.items(), key=lambda kv: -kv[1][1]):
    acc += t
    print("  %-28s %5.1f calls/fit  %7.1f us/fit  %6.1f us/call" % (name, c / NF, t / NF * 1e6, t / c * 1e6))
print("  in C-ABI calls: %.3f ms/fit; Python/NumPy around them: %.3f ms/fit" % (acc / NF * 1e3, (tot - acc) / NF * 1e3))
print("host stages ms:", {k: round(v * 1e3, 3) for k, v in anns[-1].timings.items()})
