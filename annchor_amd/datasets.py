"""
Bundled data sets (reference annchor/datasets.py:7-183; the data files are the
reference's own: annchor/data/edit_data.npz -> strings_data.npz, digits_data.npz).
"""
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def load_strings():
    """1600 lower-case strings (length 378-594) in 8 clusters; X: np.array of str,
    y: labels.  The reference's pre-computed 100-NN graph is not shipped with it
    (missing upstream blob); tests/golden/strings_full.npz holds a regenerated one."""
    d = np.load(os.path.join(_DATA, "strings_data.npz"))
    lens = d["lens"]
    offs = np.concatenate([[0], np.cumsum(lens)])
    raw = d["chars"].tobytes().decode("ascii")
    X = np.array([raw[offs[i]:offs[i + 1]] for i in range(len(lens))])
    return {"X": X, "y": d["y"].astype(np.int64)}


def load_digits():
    """UCI OCR digits test set: X float64 [1797, 64], y, cost_matrix [64, 64] and the
    reference's stored exact-EMD 100-NN graph (2, 1797, 100)."""
    d = np.load(os.path.join(_DATA, "digits_data.npz"))
    ng = np.stack([d["ng_idx"].astype(np.float64), d["ng_dist"]])
    return {"X": d["X"].astype(np.float64), "y": d["y"].astype(np.int64), "neighbor_graph": ng,
            "cost_matrix": d["cost_matrix"]}
