"""
Metric boundary of the host layer (reference annchor/utils.py:62-177,248-271).

  get_function_from_input(func, func_kwargs) -> f          (utils.py:62-107)
  get_exact_ijs_(f, ...) -> get_exact(f, X, IJ)            (utils.py:110-177)
  test_parallelisation(...)                                (utils.py:248-271)

Bundled metric names resolve to `DeviceMetric` objects (HIP kernels).  Any other
callable is a host metric: it is evaluated by a joblib pool exactly as the
reference does -- that host path is what the north star calls the CPU baseline for
user-supplied metrics; everything downstream of the metric still runs on the GPU.
"""
import os
from multiprocessing.context import TimeoutError

import numpy as np

from . import distances

CPU_COUNT = os.cpu_count()


def get_function_from_input(func, func_kwargs):
    if isinstance(func, str):
        allowed_strings = {"euclidean", "cosine", "levenshtein", "wasserstein"}
        assert func in allowed_strings, "Error: The string must be one of {}".format(sorted(allowed_strings))
        if func == "wasserstein":
            assert func_kwargs is not None and "cost_matrix" in func_kwargs, \
                "Error: wassetstein metric requires cost_function kwarg"
            return distances.Wasserstein(func_kwargs["cost_matrix"])
        if func == "euclidean":
            return distances.euclidean
        if func == "levenshtein":
            return distances.levenshtein
        return distances.cosine   # scipy.spatial.distance.cosine, utils.py:14,67
    if func_kwargs is None:
        return func

    def f(x, y):
        return func(x, y, **func_kwargs)

    return f


def get_exact_ijs_(f, parallel=True, verbose=False, backend="loky"):
    """Host evaluator for arbitrary Python metrics: np.array([f(X[i], X[j]) for i, j in IJ])."""
    if not parallel:
        def get_exact(f, X, IJ):
            return np.array([f(X[i], X[j]) for i, j in IJ], dtype=np.float64)
        return get_exact

    def get_exact(f, X, IJ):
        from joblib import Parallel, delayed
        if len(IJ) == 0:
            return np.zeros(0)
        return np.array(Parallel(n_jobs=CPU_COUNT, backend=backend, timeout=30)(
            delayed(f)(X[i], X[j]) for i, j in IJ), dtype=np.float64)

    return get_exact


def test_parallelisation(get_exact_ijs, f, X, nx, backend, s=20):
    try:
        get_exact_ijs(f, X, np.random.randint(nx, size=(s, 2)))
    except TimeoutError:
        print("TimeoutError: Parallelisation failed.")
        if backend == "loky":
            print("Current backend is 'loky', try backend='multiprocessing', or specifying custom "
                  "parallelistation with get_exact_ijs keyword argument.")
        elif backend == "multiprocessing":
            print("Current backend is 'multiprocessing', try backend='loky', or specifying custom "
                  "parallelistation with get_exact_ijs keyword argument.")
        raise TimeoutError()


test_parallelisation.__test__ = False  # not a pytest test
