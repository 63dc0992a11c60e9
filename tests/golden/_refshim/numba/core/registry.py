class CPUDispatcher:
    def __init__(self, py_func):
        self.py_func = py_func
        self.__name__ = getattr(py_func, "__name__", "f")

    def __call__(self, *a, **k):
        return self.py_func(*a, **k)
