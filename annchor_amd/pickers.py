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
        # host metric: one one-to-all sweep per round through the caller's evaluator.  `spread` is
        # the running minimum the next anchor maximises; from round 1 on it restarts WITHOUT
        # anchor 0's row (the reference quirk described above)
        everyone = np.arange(nx, dtype=np.int64)
        rows, chosen, spread = [], [], None
        for rnd in range(na):
            chosen.append(int(ix))
            row = np.asarray(ann.get_exact_ijs(ann.f, ann.X, np.column_stack((np.full(nx, ix, dtype=np.int64), everyone))),
                             dtype=np.float64)
            rows.append(row)
            if rnd == 0:
                ix = int(np.argmax(row))
            else:
                spread = row.copy() if spread is None else np.minimum(spread, row)
                ix = int(np.argmax(spread))
        return np.array(chosen, dtype=int), np.stack(rows, axis=1), na * nx


class ExternalAnchorPicker:
    """pickers.py:55-83: anchors are arbitrary objects, not data-set members."""

    def __init__(self, A):
        self.A = A
        self.is_anchor_safe = False

    def get_anchors(self, ann):
        nx, na = ann.nx, ann.n_anchors
        np.random.seed(ann.random_seed)
        sweep = getattr(ann.f, "one_to_many", None)
        cols = [np.asarray(sweep(a, ann.X) if sweep is not None else [ann.f(x, a) for x in ann.X], dtype=np.float64)
                for a in (self.A[i] for i in range(na))]
        return np.array([]), np.stack(cols, axis=1), na * nx


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
    A = np.asarray(A)
    IJ = np.column_stack((np.repeat(A.astype(np.int64), nx), np.tile(np.arange(nx, dtype=np.int64), len(A))))
    D = np.asarray(ann.get_exact_ijs(ann.f, ann.X, IJ), dtype=np.float64).reshape(len(A), nx)
    return A, D.T, na * nx
