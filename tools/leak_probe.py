"""Device memory left after each create / fit / close cycle of the N = 100 000 thinned-list fit (a leak shows as a falling number)."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from annchor_amd import Annchor, _native
from annchor_amd.datasets import synthetic_string_clusters
from annchor_amd.samplers import DeviceStratifiedSampler
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
X = synthetic_string_clusters(n)
cfg = dict(n_anchors=60, n_neighbors=15, p_work=0.02, n_samples=5000, locality=5, loc_thresh=3)
lib = _native.load_library()
def free_gb():
    f, t = _native._i64(), _native._i64()
    lib.annchor_device_mem_info(0, ctypes.byref(f), ctypes.byref(t))
    return f.value / 2**30
print("start: free %.2f GB" % free_gb())
for rep in range(4):
    ann = Annchor(X, "levenshtein", sampler=DeviceStratifiedSampler(), **cfg)
    t = time.perf_counter(); ann.fit(); dt = time.perf_counter() - t
    f_in = free_gb()
    ann._engine.close(); del ann
    f_pool = free_gb()
    lib.annchor_release_parked()
    print("rep %d: fit %.0f ms; free with the context alive %.2f GB, closed (pooled) %.2f GB, pool released %.2f GB" % (rep, dt * 1e3, f_in, f_pool, free_gb()), flush=True)
