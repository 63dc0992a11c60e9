"""One C3-size streamed fit; with an ST_PROFILE build (ANNCHOR_HIP_LIB) the tile kernel's phase split goes to stderr."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from annchor_amd.streamed import StreamedAnnchor
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(1234)
Z = rng.standard_normal((N, 8)); W = rng.standard_normal((8, 128))
X = (Z @ W + 0.05 * rng.standard_normal((N, 128))).astype(np.float32)
for rep in range(2):
    sa = StreamedAnnchor(X, n_anchors=32, n_neighbors=15, p_work=0.1)
    sa._engine.prof_enable(True)
    t = time.perf_counter(); sa.fit(); dt = time.perf_counter() - t
    prof = sa._engine.prof_get()
    print("fit %.3f s tile_evals %d %s" % (dt, sa.tile_evals, {n: round(v["ms"], 1) for n, v in prof.items() if v["ms"] > 0.5}), flush=True)
    sa._engine.close()
