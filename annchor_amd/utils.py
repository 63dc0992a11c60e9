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
import time
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


MIN_CHUNK = 32          # pairs per task at least: below that the pickling of the task outweighs the metric
PAIR_TIMEOUT_SECONDS = 30.0   # the reference's joblib timeout per task = per pair (utils.py:152-175)
CHUNKS_PER_JOB = 4      # tasks per worker and call: slack for pairs of unequal cost


def host_jobs():
    """Worker processes of the host evaluator: every core (utils.py:152-175 uses CPU_COUNT), ANNCHOR_HOST_JOBS overrides."""
    try:
        return max(1, int(os.environ.get("ANNCHOR_HOST_JOBS", CPU_COUNT or 1)))
    except ValueError:
        return CPU_COUNT or 1


def _take(X, idx):
    """X[idx] for an array, a list of the members otherwise (what a task is handed: its own points, never all of X)."""
    if isinstance(X, np.ndarray):
        return X[idx]
    return [X[int(i)] for i in idx]


def _eval_chunk(f, xi, xj):
    t = time.perf_counter()
    out = np.array([f(a, b) for a, b in zip(xi, xj)], dtype=np.float64)
    return out, time.perf_counter() - t


WORKER_SPAWN_SECONDS = 0.05   # what one more loky worker costs the parent (measured: 256 workers ~12.8 s)


def get_exact_ijs_(f, parallel=True, verbose=False, backend="loky"):
    """Host evaluator for arbitrary Python metrics: np.array([f(X[i], X[j]) for i, j in IJ]) (utils.py:110-177).

    Same signature, backends and result as the reference's; what differs is the unit of submission.  The reference hands
    joblib one task per PAIR (utils.py:152-175): every evaluation then pays the pickling of f and two points, a queue
    round trip and a future, and its pool is always one process per core (measured with that form here: 240
    evaluations/s on 256 cores for a Levenshtein callable, ~15 s of it loky starting 256 workers).  This evaluator
    submits CHUNKS -- ceil(len(IJ) / (4 n_jobs)) pairs, at least MIN_CHUNK -- each carrying only the points of its own
    pairs, and sizes the pool by the work: the cost of one evaluation is known from the first call on (its first two
    pairs are evaluated on the calling thread; afterwards every chunk reports its time), and w workers cost
    w x WORKER_SPAWN_SECONDS to start and finish T seconds of metric in T / w, so a call uses sqrt(T / spawn) of them --
    T = the evaluations still expected (`state["expected_pairs"]`: Annchor sets it to its budget p_work N; this call's
    pairs otherwise) -- never fewer than an earlier call of the same evaluator did (loky grows a pool in place;
    shrinking it would restart workers) and never more than the cores / the chunks.  With 256 workers for every call
    the same fit spent 12 of its 14 s starting processes.  The
    per-task timeout is the reference's 30 s per pair times the pairs of a chunk: a metric the reference finishes
    (< 30 s per evaluation) never times out here either, and a pool that does not come up is still reported."""
    if not parallel:
        def get_exact(f, X, IJ):
            return np.array([f(X[i], X[j]) for i, j in IJ], dtype=np.float64)
        return get_exact

    state = {"t_pair": None, "workers": 2, "expected_pairs": 0, "done": 0}

    def get_exact(f, X, IJ):
        from joblib import Parallel, delayed
        IJ = np.asarray(IJ)
        n = len(IJ)
        if n == 0:
            return np.zeros(0)
        IJ = IJ.reshape(n, 2)
        head = []
        if state["t_pair"] is None and n > 4:
            t = time.perf_counter()
            head = [f(X[i], X[j]) for i, j in IJ[:2]]
            state["t_pair"] = (time.perf_counter() - t) / 2
        rest = IJ[len(head):]
        m = len(rest)
        jobs = host_jobs()
        if state["t_pair"] is not None:
            horizon = max(m, state["expected_pairs"] - state["done"]) * state["t_pair"]
            state["workers"] = max(state["workers"], min(jobs, int(np.ceil(np.sqrt(horizon / WORKER_SPAWN_SECONDS)))))
        workers = min(jobs, state["workers"]) if state["t_pair"] is not None else jobs
        # (a slow metric -- a second or more per pair -- is balanced pair by pair: the task overhead no longer matters)
        min_chunk = MIN_CHUNK if not state["t_pair"] else max(1, min(MIN_CHUNK, int(1.0 / state["t_pair"])))
        chunk = max(min_chunk, -(-m // (CHUNKS_PER_JOB * workers)))
        if chunk >= m and m >= 2:
            chunk = -(-m // 2)      # (two tasks at least: the pool itself is exercised, see test_parallelisation)
        cuts = list(range(0, m, chunk))
        state["last_chunk"], state["last_timeout"] = chunk, PAIR_TIMEOUT_SECONDS * max(1, chunk)
        # 30 s per PAIR as in the reference (utils.py:152-175: one task per pair, timeout=30): a chunk of c pairs gets 30 c;
        # n_jobs is the monotonic pool size whatever the number of chunks (a smaller n_jobs would shrink loky's reusable
        # executor and the next large call would start its workers again)
        parts = Parallel(n_jobs=max(1, workers), backend=backend, timeout=PAIR_TIMEOUT_SECONDS * max(1, chunk))(
            delayed(_eval_chunk)(f, _take(X, rest[c:c + chunk, 0]), _take(X, rest[c:c + chunk, 1])) for c in cuts)
        state["done"] += n
        spent = sum(p[1] for p in parts)
        if m:
            state["t_pair"] = spent / m if state["t_pair"] is None else 0.5 * (state["t_pair"] + spent / m)
        return np.concatenate([np.asarray(head, dtype=np.float64)] + [p[0] for p in parts])

    get_exact.state = state
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
