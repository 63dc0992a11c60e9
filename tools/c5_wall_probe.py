"""Where the wall time of a one-rank C5 fit goes beyond its kernels: per-phase wall clock against the ProfScope sums, and the
host link's rate into pageable / registered memory for a graph-sized download.

  python tools/c5_wall_probe.py [--n 8000000]
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8_000_000)
    ap.add_argument("--fits", type=int, default=3)
    args = ap.parse_args()
    from test_c5_gpu import shard_rows

    from annchor_amd.streamed import StreamedAnnchor

    X = shard_rows(0, args.n)
    rep = {"n": args.n, "fits": []}
    for it in range(args.fits):
        t0 = time.perf_counter()
        sa = StreamedAnnchor(X, n_anchors=32, n_neighbors=15, p_work=0.1)
        t1 = time.perf_counter()
        sa._engine.prof_enable(it == args.fits - 1)
        sa.fit()
        t2 = time.perf_counter()
        prof = sa._engine.prof_get() if it == args.fits - 1 else {}
        rep["fits"].append({"constructor_s": t1 - t0, "fit_s": t2 - t1, "timings": sa.timings,
                            "prof_ms": {k: round(v["ms"], 2) for k, v in prof.items()}, "prof_sum_ms": round(sum(v["ms"] for v in prof.values()), 1)})
        eng = sa._engine
        if it == args.fits - 1:
            hip = ctypes.CDLL("libamdhip64.so")
            nbytes = args.n * 15 * 8
            dev = eng.device_alloc(nbytes)
            host = np.zeros(nbytes // 8, dtype=np.int64)
            rates = {}
            for name in ("pageable", "pageable_again", "registered", "registered_again"):
                if name == "registered":
                    t = time.perf_counter()
                    rc = hip.hipHostRegister(ctypes.c_void_p(host.ctypes.data), ctypes.c_size_t(nbytes), 0)
                    rates["register_s"] = time.perf_counter() - t
                    assert rc == 0, rc
                t = time.perf_counter()
                eng.device_copy(host.ctypes.data, dev, nbytes, "d2h")
                dt = time.perf_counter() - t
                rates[name] = {"s": dt, "GBps": nbytes / dt / 1e9}
            t = time.perf_counter()
            hip.hipHostUnregister(ctypes.c_void_p(host.ctypes.data))
            rates["unregister_s"] = time.perf_counter() - t
            eng.device_free(dev)
            rep["d2h_%d_bytes" % nbytes] = rates
        eng.close()
        print(json.dumps(rep["fits"][-1]), flush=True)
    print(json.dumps(rep))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rep, open("gpurun_out/c5_wall_probe.json", "w"), indent=1)


if __name__ == "__main__":
    main()
