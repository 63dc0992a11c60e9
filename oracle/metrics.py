"""
oracle/metrics.py -- ctypes front-end for the oracle's C metric restatements
(oracle/lev.c, oracle/emd.c) plus the NumPy Euclidean restatement.

TEST INFRASTRUCTURE ONLY (see oracle/annchor_oracle.py header).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        L.lev_dp.restype = ctypes.c_int
        L.lev_dp.argtypes = [u8p, ctypes.c_int, u8p, ctypes.c_int]
        L.lev_myers.restype = ctypes.c_int
        L.lev_myers.argtypes = [u8p, ctypes.c_int, u8p, ctypes.c_int]
        L.lev_pairs.restype = ctypes.c_int
        L.lev_pairs.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_void_p, ctypes.c_int64,
                                                       ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.lev_all_pairs.restype = ctypes.c_int
        L.lev_all_pairs.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int]
        L.emd_pairs.restype = ctypes.c_int
        L.emd_pairs.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
                                ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int]
        _LIB = L
    return _LIB


def _b(s):
    return s.encode("utf-8") if isinstance(s, str) else bytes(s)


def levenshtein(x, y, algo="myers"):
    """distances.py:16-20 on two Python strings."""
    a, b = _b(x), _b(y)
    A = (ctypes.c_uint8 * max(1, len(a))).from_buffer_copy(a or b"\0")
    B = (ctypes.c_uint8 * max(1, len(b))).from_buffer_copy(b or b"\0")
    f = lib().lev_myers if algo == "myers" else lib().lev_dp
    return int(f(A, len(a), B, len(b)))


class PackedStrings:
    """Strings packed back to back as bytes (one byte per symbol)."""

    def __init__(self, strings):
        enc = [_b(s) for s in strings]
        self.lens = np.array([len(e) for e in enc], dtype=np.int32)
        self.offs = np.zeros(len(enc), dtype=np.int64)
        if len(enc) > 1:
            np.cumsum(self.lens[:-1], out=self.offs[1:])
        self.chars = np.frombuffer(b"".join(enc) or b"\0", dtype=np.uint8).copy()
        self.nx = len(enc)

    def pairs(self, IJ, algo=1, nthreads=0):
        """utils.py:110-177 get_exact(f, X, IJ) for f = levenshtein."""
        IJ = np.ascontiguousarray(IJ, dtype=np.int64)
        out = np.zeros(IJ.shape[0], dtype=np.float64)
        self.threads = lib().lev_pairs(self.chars.ctypes.data, self.offs.ctypes.data,
                                       self.lens.ctypes.data, IJ.ctypes.data, IJ.shape[0],
                                       out.ctypes.data, algo, nthreads)
        return out

    def all_pairs(self, nthreads=0):
        dense = np.zeros((self.nx, self.nx), dtype=np.float32)
        lib().lev_all_pairs(self.chars.ctypes.data, self.offs.ctypes.data, self.lens.ctypes.data,
                            self.nx, dense.ctypes.data, nthreads)
        return dense


def euclidean_pairs(X, IJ):
    """distances.py:8-13 `np.linalg.norm(x - y)` in X's dtype, stored as float64."""
    X = np.asarray(X)
    diff = X[IJ[:, 0]] - X[IJ[:, 1]]
    return np.sqrt((diff * diff).sum(axis=1)).astype(np.float64)


class Histograms:
    """Wasserstein (utils.py:75-86 -> pynndescent.distances.kantorovich)."""

    def __init__(self, X, cost):
        self.X = np.ascontiguousarray(X, dtype=np.float64)
        self.cost = np.ascontiguousarray(cost, dtype=np.float64)
        self.nx, self.nb = self.X.shape

    def pairs(self, IJ, nthreads=0):
        IJ = np.ascontiguousarray(IJ, dtype=np.int64)
        out = np.zeros(IJ.shape[0], dtype=np.float64)
        self.threads = lib().emd_pairs(self.X.ctypes.data, self.nx, self.nb, self.cost.ctypes.data,
                                       IJ.ctypes.data, IJ.shape[0], out.ctypes.data, nthreads)
        return out


def load_strings():
    d = np.load(os.path.join(_HERE, "..", "annchor_amd", "data", "strings_data.npz"))
    chars, lens = d["chars"], d["lens"]
    offs = np.concatenate([[0], np.cumsum(lens)])
    raw = chars.tobytes().decode("ascii")
    return [raw[offs[i]:offs[i + 1]] for i in range(len(lens))], d["y"].astype(np.int64)


def load_digits():
    d = np.load(os.path.join(_HERE, "..", "annchor_amd", "data", "digits_data.npz"))
    return dict(X=d["X"].astype(np.float64), y=d["y"].astype(np.int64), cost_matrix=d["cost_matrix"],
                neighbor_graph=(d["ng_idx"].astype(np.int64), d["ng_dist"]))
