"""
Bundled metrics (reference annchor/distances.py:8-20 and the wasserstein closure of
annchor/utils.py:75-86).

Each bundled metric is a `DeviceMetric`: a callable f(x, y) -- what the reference
hands to plugins as `ann.f` -- whose arithmetic runs in the HIP kernels of
libannchor_hip.so.  Called on two loose objects it evaluates them on the GPU through
a scratch context; inside `Annchor` the data set is uploaded once and whole pair
lists are evaluated per call (`get_exact_ijs`).  There is no CPU implementation.
"""
import os

import numpy as np

from . import _native


def encode_strings(strings):
    """Python strings -> (codes uint8 or uint16, offs int64, lens int32, alphabet size).
    Symbols are mapped to dense codes 0..A-1: one byte per symbol up to 256 distinct symbols, two bytes up to 65 535."""
    strings = list(strings)
    lens = np.fromiter((len(s) for s in strings), dtype=np.int32, count=len(strings))
    joined = "".join(strings)
    try:
        # every code point below 256 (the usual case): one byte per symbol, dense codes through a
        # 256-entry table -- the general path below sorts the whole text to find its alphabet
        # (10 ms for the 0.8 M symbols of the C2 data set, against a 5.5 ms fit)
        raw = np.frombuffer(joined.encode("latin-1"), dtype=np.uint8)
        present = np.bincount(raw, minlength=256) > 0
        symbols = np.flatnonzero(present)
        codes = (np.cumsum(present) - 1).astype(np.uint8)[raw]
    except UnicodeEncodeError:
        raw = np.frombuffer(joined.encode("utf-32-le"), dtype=np.uint32)
        symbols = np.unique(raw)
        if symbols.size > 65535:
            raise ValueError("levenshtein on the GPU supports at most 65 535 distinct symbols, got %d" % symbols.size)
        codes = np.searchsorted(symbols, raw).astype(np.uint8 if symbols.size <= 256 else np.uint16)
    if os.environ.get("ANNCHOR_LEV_WIDE"):   # test hook: the 16-bit path on any alphabet
        codes = codes.astype(np.uint16)
    offs = np.zeros(len(strings), dtype=np.int64)
    if len(strings) > 1:
        np.cumsum(lens[:-1], out=offs[1:])
    if codes.size == 0:
        codes = np.zeros(1, dtype=codes.dtype)
    return codes, offs, lens, max(1, int(symbols.size))


class DeviceMetric:
    """Base class of the GPU-evaluated metrics."""

    name = "device"
    _scratch = None

    def bind(self, engine, X):
        """Upload the data set X into `engine` for pair-list evaluation."""
        raise NotImplementedError

    def _scratch_engine(self):
        if DeviceMetric._scratch is None:
            DeviceMetric._scratch = _native.Engine(0)
        return DeviceMetric._scratch

    def many(self, xs, ys):
        """f(xs[t], ys[t]) for loose objects, evaluated on the GPU."""
        xs, ys = list(xs), list(ys)
        eng = self._scratch_engine()
        self.bind(eng, xs + ys)
        n = len(xs)
        IJ = np.stack([np.arange(n), np.arange(n) + n], axis=1)
        return eng.metric_pairs(IJ)

    def one_to_many(self, x, ys):
        ys = list(ys)
        eng = self._scratch_engine()
        self.bind(eng, [x] + ys)
        IJ = np.stack([np.zeros(len(ys), dtype=np.int64), np.arange(len(ys)) + 1], axis=1)
        return eng.metric_pairs(IJ)

    def __call__(self, x, y):
        return self.many([x], [y])[0]


class _Levenshtein(DeviceMetric):
    """Unit-cost edit distance (distances.py:16-20)."""

    name = "levenshtein"

    def bind(self, engine, X):
        engine.set_strings(*encode_strings(X))

    def __call__(self, x, y):
        return int(self.many([x], [y])[0])


class _Euclidean(DeviceMetric):
    """np.linalg.norm(x - y) in the dtype of X (distances.py:8-13)."""

    name = "euclidean"

    def bind(self, engine, X):
        X = np.asarray(X)
        if X.ndim == 1:
            X = X[:, None]
        if X.dtype != np.float32:
            X = X.astype(np.float64)
        engine.set_points(X)


class _Cosine(DeviceMetric):
    """scipy.spatial.distance.cosine(x, y) = 1 - x.y / (|x| |y|), the dot products in the dtype of
    X, clipped to [0, 2] (utils.py:14,67)."""

    name = "cosine"

    def bind(self, engine, X):
        X = np.asarray(X)
        if X.ndim == 1:
            X = X[:, None]
        if X.dtype != np.float32:
            X = X.astype(np.float64)
        engine.set_points(X, cosine=True)


class Wasserstein(DeviceMetric):
    """kantorovich(x, y, cost=M): exact optimal transport between the normalised
    histograms restricted to their supports (utils.py:75-86)."""

    name = "wasserstein"

    def __init__(self, cost_matrix):
        self.cost_matrix = np.ascontiguousarray(cost_matrix, dtype=np.float64)

    def bind(self, engine, X):
        engine.set_histograms(np.asarray(X, dtype=np.float64), self.cost_matrix)


levenshtein = _Levenshtein()
euclidean = _Euclidean()
cosine = _Cosine()
