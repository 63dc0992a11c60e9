"""Where the non-kernel part of a C3 streamed fit goes: stage wall times against the per-stage sums of kernel times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from annchor_amd.streamed import StreamedAnnchor
from bench import euclid_shard
X = euclid_shard(0, int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000)
for rep in range(3):
    sa = StreamedAnnchor(X, n_anchors=32, n_neighbors=15, p_work=0.1)
    sa._engine.prof_enable(True)
    t = time.perf_counter(); sa.fit(); dt = time.perf_counter() - t
    prof = sa._engine.prof_get()
    ksum = sum(v["ms"] for v in prof.values())
    print("fit %.1f ms; stages %s; kernels %.1f ms: %s" % (dt * 1e3, {k: round(v * 1e3, 1) for k, v in sa.timings.items()}, ksum,
          {k: round(v["ms"], 1) for k, v in prof.items()}), flush=True)
    sa._engine.close()
