"""Pass-through stand-in for `numba`, used ONLY by tests/golden/make_golden.py in
the build container to import the pure-Python reference (numba is not installed
and cannot be).  Decorators return the undecorated function wrapped in a callable
`CPUDispatcher`; no arithmetic is replaced -- the reference's own NumPy code runs
as plain Python."""
from .core.registry import CPUDispatcher


def _wrap(f):
    # numba's typed List(int64) -> array conversion keeps int64 even when the list
    # is empty; plain np.array([]) would give float64 (utils.py:432-434 list_to_arr).
    if getattr(f, "__name__", "") == "list_to_arr":
        import numpy as np

        return CPUDispatcher(lambda _list: np.array(list(_list), dtype=np.int64))
    return CPUDispatcher(f)


def _decorate(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return _wrap(args[0])
    return _wrap


njit = jit = _decorate
prange = range


class _Type:
    def __getitem__(self, _):
        return self


class types:
    int64 = _Type()
    float64 = _Type()


def typeof(_):
    return _Type()
