"""Host-only: cost split of the sampler's legacy draw at C2 bin sizes (no GPU needed): stream generation vs
scan + trace.  ANNCHOR_RNG_NO_CACHE=1 so that every seed is generated afresh."""
import os, sys, time
os.environ["ANNCHOR_RNG_NO_CACHE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from annchor_amd import _native
counts = np.array([12500, 310000, 302000, 313000, 312500], dtype=np.int64)   # C2-like: 1.25 M not-computed pairs in 5 bins
want = np.full(5, 1000, dtype=np.int64)
nd = int(counts.sum())
def med(f, reps=15):
    ts = []
    for r in range(reps):
        ts.append(f(1000 + r))
    return np.median(ts) * 1e3
def cold(seed):
    t = time.perf_counter(); _native.legacy_choice_ranks(seed, counts, want); return time.perf_counter() - t
def warm(seed):
    _native.legacy_prefetch(seed, nd); time.sleep(0.01)
    t = time.perf_counter(); _native.legacy_choice_ranks(seed, counts, want); return time.perf_counter() - t
def lagged(lag):
    def f(seed):
        _native.legacy_prefetch(seed, nd); t0 = time.perf_counter()
        while time.perf_counter() - t0 < lag: pass
        t = time.perf_counter(); _native.legacy_choice_ranks(seed, counts, want); return time.perf_counter() - t
    return f
for w in range(5): cold(500 + w)
print("draw with no prefetch (generation on demand): %.2f ms" % med(cold))
print("draw after the stream is complete:            %.2f ms" % med(warm))
for lag in (0.0, 0.3e-3, 0.6e-3, 0.9e-3, 1.2e-3):
    print("draw %.1f ms after the prefetch started:      %.2f ms" % (lag * 1e3, med(lagged(lag))))
