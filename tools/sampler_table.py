"""DeviceStratifiedSampler (the default from 2829 points on) against the legacy NumPy-stream draw (the default below, the one whose
graphs the tests pin): error counts against brute force over seeds, data sets and sizes -- VERDICT r5, item 5.

  python tools/sampler_table.py [--seeds 24] [--out gpurun_out/sampler_table.json]

Data sets (three metrics / geometries) x N in {3000, 16000} x seeds x {legacy, device}; the truth is the library's device brute force
(BruteForce, csrc/brute.hip: pinned against the oracle on strings and the stored digits graph).  Per cell: median / mean / max
error count of each sampler and the one-sided Mann-Whitney p of "device errors are larger".  C2 (N = 1600, below the switch) is
added for reference: there the default IS the legacy draw."""
import argparse, io, json, os, sys, time, contextlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def latent(n, d=32, k=8, seed=5):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((n, k)) @ rng.standard_normal((k, d)) + 0.05 * rng.standard_normal((n, d))).astype(np.float64)


def blobs(n, d=16, centres=40, seed=6):
    rng = np.random.default_rng(seed)
    c = rng.standard_normal((centres, d)) * 6.0
    return (c[rng.integers(0, centres, n)] + rng.standard_normal((n, d)) * rng.uniform(0.3, 1.5, (n, 1))).astype(np.float64)


def mutated_strings(n, seed=7):
    """n strings grown from the bundled 1600 by point mutations / indels (clusters of edit neighbours, like the original set)"""
    from annchor_amd.datasets import load_strings

    base = list(load_strings()["X"])
    rng = np.random.default_rng(seed)
    alpha = sorted(set("".join(base)))
    out = list(base[:min(n, len(base))])
    while len(out) < n:
        s = list(out[rng.integers(0, len(out))])
        for _ in range(int(rng.integers(1, 12))):
            op, pos = rng.integers(0, 3), int(rng.integers(0, max(1, len(s))))
            if op == 0 and s:
                s[pos] = alpha[rng.integers(0, len(alpha))]
            elif op == 1:
                s.insert(pos, alpha[rng.integers(0, len(alpha))])
            elif s and len(s) > 20:
                del s[pos]
        out.append("".join(s))
    return np.array(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=24)
    ap.add_argument("--out", default="gpurun_out/sampler_table.json")
    ap.add_argument("--sizes", default="3000,16000")
    args = ap.parse_args()
    from scipy.stats import mannwhitneyu

    from annchor_amd import Annchor, BruteForce, compare_neighbor_graphs

    sizes = [int(v) for v in args.sizes.split(",")]
    cells = [("strings C2 (load_strings, below the switch)", "levenshtein", lambda n: mutated_strings(1600), 1600, dict(n_anchors=15, n_neighbors=25, p_work=0.12))]
    for n in sizes:
        cells.append(("euclidean f64: 8-d latent in 32-d", "euclidean", latent, n, dict(n_anchors=20, n_neighbors=15, p_work=0.1 if n < 8000 else 0.05)))
        cells.append(("euclidean f64: 40 blobs of unequal spread in 16-d", "euclidean", blobs, n, dict(n_anchors=20, n_neighbors=15, p_work=0.1 if n < 8000 else 0.05)))
        cells.append(("levenshtein: the bundled strings grown by mutations", "levenshtein", mutated_strings, n, dict(n_anchors=20, n_neighbors=15, p_work=0.1 if n < 8000 else 0.05)))
    rows = []
    for name, metric, make, n, cfg in cells:
        X = make(n)
        k = cfg["n_neighbors"]
        t = time.perf_counter()
        truth = BruteForce(X, metric).fit(n_neighbors=k).neighbor_graph
        t_bf = time.perf_counter() - t
        res = {"legacy": [], "device": []}
        tm = {"legacy": [], "device": []}
        for s in range(42, 42 + args.seeds):
            for smp in ("legacy", "device"):
                with contextlib.redirect_stderr(io.StringIO()), contextlib.redirect_stdout(io.StringIO()):
                    ann = Annchor(X, metric, sampler=smp, random_seed=s, **cfg)
                    t = time.perf_counter()
                    ann.fit()
                    tm[smp].append(time.perf_counter() - t)
                res[smp].append(int(compare_neighbor_graphs(truth, ann.neighbor_graph, k)))
                ann._engine.close() if hasattr(ann, "_engine") else None
        p = float(mannwhitneyu(res["device"], res["legacy"], alternative="greater").pvalue)
        row = {"data": name, "metric": metric, "n": int(n), "cfg": cfg, "cells": int(n) * k, "seeds": args.seeds, "brute_force_s": round(t_bf, 2),
               "legacy": {"errors": res["legacy"], "median": float(np.median(res["legacy"])), "mean": float(np.mean(res["legacy"])), "max": int(max(res["legacy"])),
                          "fit_ms_median": round(1e3 * float(np.median(tm["legacy"])), 2)},
               "device": {"errors": res["device"], "median": float(np.median(res["device"])), "mean": float(np.mean(res["device"])), "max": int(max(res["device"])),
                          "fit_ms_median": round(1e3 * float(np.median(tm["device"])), 2)},
               "p_device_worse": p}
        rows.append(row)
        print("%-52s N=%-6d legacy med %7.1f mean %7.1f max %5d (%.1f ms) | device med %7.1f mean %7.1f max %5d (%.1f ms) | p(device worse)=%.3f"
              % (name, n, row["legacy"]["median"], row["legacy"]["mean"], row["legacy"]["max"], row["legacy"]["fit_ms_median"],
                 row["device"]["median"], row["device"]["mean"], row["device"]["max"], row["device"]["fit_ms_median"], p), flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump({"what": __doc__, "rows": rows}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
