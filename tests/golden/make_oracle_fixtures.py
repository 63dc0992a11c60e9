#!/usr/bin/env python3
"""
tests/golden/make_oracle_fixtures.py -- the ORACLE's own end-to-end results at the full-size
strings configurations (strings_full_oracle.npz), so that the GPU tests can compare graphs
bit-exactly without re-running the oracle (~10 s per configuration on 8 cores).

Unlike make_golden.py this needs no reference: it runs oracle/annchor_oracle.py only.  The
oracle itself is pinned against the reference by tests/test_oracle_golden.py; the reference's
own results at the same configurations live in strings_full.npz (make_golden.py).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np  # noqa: E402

from oracle import annchor_oracle as O  # noqa: E402
from oracle import metrics as om  # noqa: E402

CONFIGS = {
    "c1": dict(n_anchors=15, n_neighbors=25, p_work=0.12, random_seed=42),                        # BASELINE configs[0]/[1]
    "readme": dict(n_anchors=20, n_neighbors=25, p_work=0.12, random_seed=42),                    # README.md:102
    "test": dict(n_anchors=23, n_neighbors=15, p_work=0.12, random_seed=42, niters=4, n_samples=5000),   # tests/test_annchor.py:83-96
}


def main():
    X, _ = om.load_strings()
    P = om.PackedStrings(X)
    G = np.load(os.path.join(HERE, "strings_full.npz"))
    truth = (G["truth_idx"].astype(np.int64), G["truth_dist"].astype(np.float64))
    out = {}
    for tag, cfg in CONFIGS.items():
        ora = O.OracleAnnchor(len(X), P.pairs, **cfg).fit()
        k = cfg["n_neighbors"]
        err = O.compare_neighbor_graphs(truth, ora.neighbor_graph, k)
        print(tag, "evals", ora.evals, "pairs", len(ora.IJs), "errors", err, "(reference run:", int(G[tag + "_errors"]), ")")
        out[tag + "_ng_idx"] = ora.neighbor_graph[0].astype(np.int16)
        out[tag + "_ng_dist"] = ora.neighbor_graph[1].astype(np.int16)
        out[tag + "_errors"], out[tag + "_evals"] = np.int64(err), np.int64(ora.evals)
        out[tag + "_npairs"] = np.int64(len(ora.IJs))
    np.savez_compressed(os.path.join(HERE, "strings_full_oracle.npz"), **out)


if __name__ == "__main__":
    main()
