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


def synthetic_string_clusters(n, cluster=2000, length=120, seed=5, alphabet="abcdefghijklmnopqrstuvwxyz"):
    """Synthetic strings of the shape of the reference's `load_strings` set (clouds and filaments) at any size: clusters of
    `cluster` strings, each new string a few random edits (substitute / delete / insert) away from an EARLIER member of its
    cluster -- mostly a recent one (filaments), sometimes any (clouds) -- so that within a cluster the distances run from 1
    to the diameter of a deep mutation tree; different clusters are unrelated random strings.  No counterpart in the
    reference (its set is a fixed file of 1600 strings); used by the large-N tools and tests."""
    rng = np.random.default_rng(seed)
    letters = list(alphabet)
    X = []
    for c0 in range(0, n, cluster):
        members = [list(rng.choice(letters, length))]
        for s in range(1, min(cluster, n - c0)):
            par = members[int(rng.integers(max(0, s - 40), s))] if rng.random() < 0.7 else members[int(rng.integers(0, s))]
            b = list(par)
            for _ in range(int(rng.integers(1, 7))):
                p = int(rng.integers(0, len(b)))
                r = rng.random()
                if r < 0.4:
                    b[p] = letters[int(rng.integers(len(letters)))]
                elif r < 0.7 and len(b) > length // 2:
                    b.pop(p)
                else:
                    b.insert(p, letters[int(rng.integers(len(letters)))])
            members.append(b)
        X.extend("".join(b) for b in members)
    return np.array(X)
