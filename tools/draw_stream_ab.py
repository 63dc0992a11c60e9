"""A/B inside one process, fit by fit: the draw's trace streamed from pinned memory while the host scans (default) against the
partners uploaded bin by bin and the trace queued after the scan (ANNCHOR_DRAW_STREAM=0; the switch is read per call).
usage: draw_stream_ab.py [fits per mode] [ENV_NAME a b]   (any per-call switch: ENV_NAME set to a / b alternately)"""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ANNCHOR_RNG_NO_CACHE", "1")
import numpy as np
from annchor_amd import Annchor, _native
from annchor_amd.datasets import load_strings
nfit = int(sys.argv[1]) if len(sys.argv) > 1 else 60
name, va, vb = (sys.argv[2], sys.argv[3], sys.argv[4]) if len(sys.argv) > 4 else ("ANNCHOR_DRAW_STREAM", "1", "0")
print("affinity:", _native.bind_to_device_numa(0))
X = load_strings()["X"]
cfg = dict(n_anchors=15, n_neighbors=25, p_work=0.12, random_seed=42)
ref = None
for a in [Annchor(X, "levenshtein", **cfg) for _ in range(5)]:
    a.fit()
    ref = a.neighbor_graph
    a._engine.close()
times = {va: [], vb: []}
stages = {va: {}, vb: {}}
same = True
for block in range(nfit // 10):
    anns = [Annchor(X, "levenshtein", **cfg) for _ in range(20)]
    gc.collect(); gc.disable()
    for q, a in enumerate(anns):
        mode = (va, vb)[(q + block) & 1]
        os.environ[name] = mode
        t0 = time.perf_counter()
        a.fit()
        times[mode].append(time.perf_counter() - t0)
        for k, v in a.timings.items():
            stages[mode].setdefault(k, []).append(v * 1e3)
    gc.enable()
    same = same and all(np.array_equal(a.neighbor_graph[0], ref[0]) and np.array_equal(a.neighbor_graph[1], ref[1]) for a in anns[:4])
    for a in anns:
        a._engine.close()
for mode in (va, vb):
    t = np.array(times[mode]) * 1e3
    print("%s=%s: %d fits  mean %.3f  median %.3f  p10 %.3f  min %.3f ms" % (name, mode, len(t), t.mean(), np.median(t), np.percentile(t, 10), t.min()))
    print("    stage medians:", {k: round(float(np.median(v)), 3) for k, v in stages[mode].items()})
print("same graph as the first fit:", same)
