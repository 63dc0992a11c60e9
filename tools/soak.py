"""Many objects of every kind in one process: device free memory (after annchor_release_parked) at checkpoints -- a leak shows
as a steady decline."""
import ctypes, gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from annchor_amd import Annchor, BruteForce, _native
from annchor_amd.datasets import load_strings, load_digits
from annchor_amd.streamed import StreamedAnnchor
lib = _native.load_library()
def free_gb():
    gc.collect()
    lib.annchor_release_parked()
    f, t = _native._i64(), _native._i64()
    lib.annchor_device_mem_info(0, ctypes.byref(f), ctypes.byref(t))
    return f.value / 1e9
S = load_strings()["X"]
D = load_digits()
rng = np.random.default_rng(0)
E = (rng.standard_normal((9000, 5)) @ rng.standard_normal((5, 24))).astype(np.float64)
F = (rng.standard_normal((40000, 6)) @ rng.standard_normal((6, 32))).astype(np.float32)
y = np.asarray(D["y"]) if "y" in D else rng.integers(0, 10, len(D["X"]))
print("start            free %.3f GB" % free_gb())
for rnd in range(3):
    for _ in range(40):
        Annchor(S, "levenshtein", n_anchors=15, n_neighbors=25, p_work=0.12).fit()
    print("round %d C2 x40    free %.3f GB" % (rnd, free_gb()))
    for _ in range(5):
        a = Annchor(D["X"][:1400], "wasserstein", func_kwargs={"cost_matrix": D["cost_matrix"]}, n_anchors=20, n_neighbors=15, p_work=0.16).fit()
        a.query(D["X"][1400:1700], nn=10, p_work=0.3)
    print("round %d C4+query  free %.3f GB" % (rnd, free_gb()))
    for _ in range(3):
        a = Annchor(D["X"], "wasserstein", func_kwargs={"cost_matrix": D["cost_matrix"]}, n_anchors=20, n_neighbors=15, p_work=0.16).fit()
        a.get_nearest_enemies(y[:len(D["X"])], nn=3) if hasattr(a, "get_nearest_enemies") else None
    print("round %d enemies   free %.3f GB" % (rnd, free_gb()))
    for _ in range(3):
        Annchor(E, "euclidean", n_anchors=16, n_neighbors=10, p_work=0.05).fit()
        BruteForce(S[:600], "levenshtein").fit()
    print("round %d big+brute free %.3f GB" % (rnd, free_gb()))
    for _ in range(3):
        sa = StreamedAnnchor(F, n_anchors=16, n_neighbors=10, p_work=0.2).fit()
        sa._engine.close()
    print("round %d streamed  free %.3f GB" % (rnd, free_gb()))
