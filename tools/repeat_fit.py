"""Create / fit / close cycles at N = 16000 (127 M pairs): fit times and the device's free memory after every cycle."""
import ctypes, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from annchor_amd import Annchor, _native
from annchor_amd.samplers import DeviceStratifiedSampler
rng = np.random.default_rng(5)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
Z = rng.standard_normal((n, 6))
X = (Z @ rng.standard_normal((6, 48)) + 0.05 * rng.standard_normal((n, 48))).astype(np.float64)
cfg = dict(n_anchors=24, n_neighbors=15, p_work=0.05, n_samples=5000)
lib = _native.load_library()
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    ann = Annchor(X, "euclidean", sampler=DeviceStratifiedSampler(), **cfg)
    t = time.perf_counter(); ann.fit(); dt = time.perf_counter() - t
    ann._engine.close()
    f, tot = _native._i64(), _native._i64()
    lib.annchor_device_mem_info(0, ctypes.byref(f), ctypes.byref(tot))
    print("rep %2d fit %7.1f ms   free %.2f GB" % (rep, dt * 1e3, f.value / 1e9), flush=True)
