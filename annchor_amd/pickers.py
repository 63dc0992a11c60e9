"""
Anchor pickers (reference annchor/pickers.py:18-128).  Protocol, unchanged:
    picker.get_anchors(ann) -> (A, D[nx, n_anchors], n_evals)
A picker may read any attribute of `ann` (nx, n_anchors, random_seed, X, f,
get_exact_ijs, verbose).  The built-in pickers run on the GPU when `ann` evaluates
its metric on the device (they fill the engine directly and flag it); with a host
metric they follow the reference's loops through `ann.get_exact_ijs`.
"""
import numpy as np


def _on_device(ann):
    return getattr(ann, "_device_metric", False) and getattr(ann, "_engine", None) is not None


class MaxMinAnchorPicker:
    """pickers.py:18-52.  Note the reference's running min excludes anchor 0's row
    after the first round (`np_min(D[1:], 0)`); kept, it is what pins the anchors."""

    def get_anchors(self, ann):
        nx, na = ann.nx, ann.n_anchors
        np.random.seed(ann.random_seed)
        ix = np.random.randint(nx)
        if _on_device(ann):
            ann._engine.pick_anchors_maxmin(na, ix)
            ann._anchors_on_device = True
            return None, None, na * nx
        D = np.zeros((na, nx)) + np.inf
        A = np.zeros(na).astype(int)
        for i in range(na):
            A[i] = ix
            IJs = np.array([[ix, j] for j in range(nx)])
            D[i] = ann.get_exact_ijs(ann.f, ann.X, IJs)
            ix = np.argmax(np.min(D[:1], axis=0)) if i == 0 else np.argmax(np.min(D[1:i + 1], axis=0))
        return A, D.T, na * nx


class ExternalAnchorPicker:
    """pickers.py:55-83: anchors are arbitrary objects, not data-set members."""

    def __init__(self, A):
        self.A = A
        self.is_anchor_safe = False

    def get_anchors(self, ann):
        nx, na = ann.nx, ann.n_anchors
        np.random.seed(ann.random_seed)
        D = np.zeros((na, nx)) + np.inf
        batched = getattr(ann.f, "one_to_many", None)
        for i in range(na):
            if batched is not None:
                D[i] = batched(self.A[i], ann.X)
            else:
                D[i] = np.array([ann.f(x, self.A[i]) for x in ann.X])
        return np.array([]), D.T, na * nx


class SelectedAnchorPicker:
    """pickers.py:86-106."""

    def __init__(self, A):
        self.A = A

    def get_anchors(self, ann):
        nx, na = ann.nx, ann.n_anchors
        np.random.seed(ann.random_seed)
        return _selected(ann, np.asarray(self.A), nx, na)


class RandomAnchorPicker:
    """pickers.py:109-128."""

    def get_anchors(self, ann):
        nx, na = ann.nx, ann.n_anchors
        np.random.seed(ann.random_seed)
        A = np.random.choice(np.arange(nx), na, replace=False)
        return _selected(ann, A, nx, na)


def _selected(ann, A, nx, na):
    if _on_device(ann):
        ann._engine.pick_anchors_selected(A)
        ann._anchors_on_device = True
        return None, None, na * nx
    IJ = np.array([[i, j] for i in A for j in range(nx)])
    D = ann.get_exact_ijs(ann.f, ann.X, IJ).reshape(na, nx)
    return A, D.T, na * nx
