"""
Nearest-enemy graph, selective subset and alpha-RSS (reference annchor/annchor.py:685-927;
SURVEY.md section 8, row f4).

The nearest-enemy graph itself runs on the device (csrc/enemies.hip: label-masked candidate generation,
features / prediction of the new pairs, exact distances of each row's closest-looking enemies, row top-nn);
the selective subset and alpha-RSS -- greedy set covers on the k-NN graph, host code in the reference too --
are NumPy over the (lazily assembled) extended pair-list views, every metric evaluation through
`ann.get_exact_ijs`, i.e. the HIP metric kernels for the built-in metrics.

Semantics notes (shared with the rest of the build): ties are broken by stable sorts on list
order (the reference's argsorts are unstable); per-point candidate lists are complete (the
reference's `get_IJs_from_check` drops entries of its last groups, utils.py:518,521).
"""
import numpy as np

ROW_BLOCK = 2048
FIRST_ENEMIES = 50  # annchor.py:755-756: exact distances for the 50 closest-looking enemies


class RowLists:
    """Flat view of per-point candidate lists: entry e belongs to point owner[e], refers to
    pair pos[e] whose other endpoint is other[e]; rows are contiguous (ptr)."""

    def __init__(self, ptr, pos, IJs):
        self.ptr, self.pos = ptr, pos
        nx = ptr.shape[0] - 1
        self.owner = np.repeat(np.arange(nx, dtype=np.int64), np.diff(ptr))
        f = IJs[pos]
        self.other = np.where(f[:, 0] == self.owner, f[:, 1], f[:, 0])

    @staticmethod
    def merge(ptr_a, pos_a, ptr_b, pos_b):
        """Row-wise concatenation [a-row, b-row]."""
        nx = ptr_a.shape[0] - 1
        la, lb = np.diff(ptr_a), np.diff(ptr_b)
        ptr = np.zeros(nx + 1, dtype=np.int64)
        np.cumsum(la + lb, out=ptr[1:])
        pos = np.empty(ptr[-1], dtype=np.int64)
        row_a = np.repeat(np.arange(nx), la)
        row_b = np.repeat(np.arange(nx), lb)
        pos[ptr[row_a] + (np.arange(pos_a.shape[0]) - ptr_a[row_a])] = pos_a
        pos[ptr[row_b] + la[row_b] + (np.arange(pos_b.shape[0]) - ptr_b[row_b])] = pos_b
        return ptr, pos


def rows_of_pairs(IJs, nx):
    """CSR of pair positions per point, entries ordered by the other endpoint."""
    n = IJs.shape[0]
    p = np.arange(n, dtype=np.int64)
    owner = np.concatenate([IJs[:, 1], IJs[:, 0]])
    other = np.concatenate([IJs[:, 0], IJs[:, 1]])
    order = np.lexsort((other, owner))
    ptr = np.zeros(nx + 1, dtype=np.int64)
    np.cumsum(np.bincount(owner, minlength=nx), out=ptr[1:])
    return ptr, np.concatenate([p, p])[order]


def _rank_in_row(owner_sorted):
    """0-based rank of each entry inside its (contiguous) row."""
    n = owner_sorted.shape[0]
    start = np.ones(n, dtype=bool)
    start[1:] = owner_sorted[1:] != owner_sorted[:-1]
    first = np.maximum.accumulate(np.where(start, np.arange(n), 0))
    return np.arange(n) - first


def nearest_enemies(ann, y, nn=3, loc_min=100):
    """Annchor.get_nearest_enemies (annchor.py:685-782) on the device (csrc/enemies.hip): label-masked candidate
    generation (a second bitmap next to the fitted one), features / prediction of the new pairs, exact distances of every
    row's closest-looking enemies, row top-nn.  The object's pair-list views (IJs, I, features, not_computed_mask,
    RefineApprox) are extended by the enemy pairs like the reference's attributes -- lazily: they are assembled from the
    device state when somebody reads them.  Returns (idx int64 [nx, nn], dist float64 [nx, nn])."""
    from .regressors import SimpleStratifiedLinearRegression

    nx = ann.nx
    y = np.asarray(y)
    assert len(y) == nx, "Label dimension mismatch: len(y)=%d, len(X)=%d" % (len(y), nx)
    labels, counts = np.unique(y, return_counts=True)
    assert len(labels) > 1, "Data must have more than one label"
    assert np.all(counts >= nn), "At least one label occurs fewer times than specified nn=%d" % nn
    eng = ann._engine
    codes = np.unique(y, return_inverse=True)[1].astype(np.int32)
    eng.enemies_candidates(codes, ann.loc_thresh, loc_min)
    model = ann.regression.coefficients() if type(ann.regression) is SimpleStratifiedLinearRegression else None
    if model is not None:
        eng.enemies_predict(*model)
    else:   # a custom regression sees the reference's arrays
        Fn = eng.enemies_download()[1]
        eng.enemies_predict(pred=np.asarray(ann.regression.predict(Fn, ann.feature_names), dtype=np.float64))
    if ann._device_metric:
        n_eval = eng.enemies_first(FIRST_ENEMIES, nn, True)
    else:
        todo = eng.enemies_first(FIRST_ENEMIES, nn, False)
        n_eval = len(todo)
        if n_eval:
            eng.enemies_set_exact(np.asarray(ann.get_exact_ijs(ann.f, ann.X, todo), dtype=np.float64))
    ann.evals += int(n_eval)
    ngi, ngd = eng.enemies_graph(nn)
    # Like the reference (annchor.py:729-740, 748-761), the object's pair-list views now include the enemy pairs and the
    # exact distances just computed; they are rebuilt from the device state on demand (Annchor._view).  The stage methods
    # and the views of the fitted list alone (thresh, candidates, ...) are closed from here on.
    ann._invalidate("IJs", "RA", "ncm", "features", "I")
    ann._enemy_extended = True
    ann.nearest_enemy_graph = (ngi, ngd)
    return ngi, ngd


def extended_views(ann):
    """IJs / RefineApprox / not_computed_mask / features / I of the fitted pair list followed by the enemy pairs
    (annchor.py:729-740), assembled from the device state."""
    from . import _native
    from .annchor import _IndexCSR

    eng = ann._engine
    IJs0 = eng.download(_native.F_IJS).reshape(-1, 2)
    n0 = IJs0.shape[0]
    IJn, Fn, RAn, ncmn, ptr_n, idx_n = eng.enemies_download()
    ptr0, idx0 = eng.download(_native.F_I_PTR), eng.download(_native.F_I_IDX)
    ptr, pos = RowLists.merge(ptr0, idx0, ptr_n, idx_n + n0)
    return dict(IJs=np.vstack([IJs0, IJn]), RA=np.concatenate([eng.download(_native.F_RA), RAn]),
                ncm=np.concatenate([eng.download(_native.F_NCM).astype(bool), ncmn]),
                features=np.vstack([eng.download(_native.F_FEATURES).reshape(-1, 4), Fn]), I=_IndexCSR(ptr, pos))


def _cover_counts(sorted_d, limit):
    """searchsorted(row, limit - 1e-6) for rows given as a dense [nx, k] sorted matrix."""
    return (sorted_d < (limit - 1e-6)[:, None]).sum(axis=1)


def selective_subset(ann, y, dne=None, alpha=0):
    """Annchor.annchor_selective_subset (annchor.py:784-901)."""
    nx = ann.nx
    if dne is None:
        if not hasattr(ann, "nearest_enemy_graph"):
            nearest_enemies(ann, y)
        dne = ann.nearest_enemy_graph[1][:, 0]
    dne = np.asarray(dne, dtype=np.float64)
    zero = np.nonzero(dne == 0)[0]
    if zero.shape[0] > 0:
        raise Exception("Error: The following indices are distance zero from a point  with a different label:\n"
                        + "".join("\t %d\n" % i for i in zero))
    adne = dne / (1 + alpha)
    ngi, ngd = ann.neighbor_graph
    k = ngi.shape[1]
    eb = _cover_counts(ngd, adne)  # neighbours (self included) closer than the nearest enemy
    inbuf = np.arange(k)[None, :] < eb[:, None]

    # greedy cover on the k-NN graph: a point is done when the FIRST subset member in its
    # neighbour list lies inside its buffer
    in_rss = np.zeros(nx, dtype=bool)
    rss = list(np.nonzero(eb == 1)[0])
    in_rss[rss] = True
    first_hit = np.full(nx, k, dtype=np.int64)  # column of the first subset member per row

    def absorb(members_mask):
        hit = members_mask[ngi]
        col = np.where(hit.any(axis=1), hit.argmax(axis=1), k)
        np.minimum(first_hit, col, out=first_hit)

    absorb(in_rss)
    done = first_hit < eb
    while not done.all():
        votes = np.bincount(ngi[~done][inbuf[~done]], minlength=nx)
        nxt = int(np.argmax(votes))
        rss.append(nxt)
        one = np.zeros(nx, dtype=bool)
        one[nxt] = True
        absorb(one)
        done |= first_hit < eb
    rss = np.array(rss, dtype=np.int64)

    # pruning against the full candidate rows, uncomputed pairs at their upper bound
    RA, ncm = ann.RefineApprox, ann.not_computed_mask
    iub = ann.feature_names.index("upper bound")
    dists = np.where(ncm, ann.features[:, iub], RA)
    I = ann.I
    L = RowLists(I.ptr, I.idx, ann.IJs)
    dl = dists[L.pos]
    near = dl < (adne - 1e-6)[L.owner]
    # sorted-row prefix semantics of searchsorted: entries strictly below the limit
    pos_in_rss = -np.ones(nx, dtype=np.int64)
    pos_in_rss[rss] = np.arange(rss.shape[0])
    member = np.zeros((nx, rss.shape[0]), dtype=bool)
    sel = near & (pos_in_rss[L.other] >= 0)
    member[L.owner[sel], pos_in_rss[L.other[sel]]] = True
    self_in = (pos_in_rss >= 0) & (adne - 1e-6 > 0)  # the point itself sits at distance 0
    member[np.nonzero(self_in)[0], pos_in_rss[self_in]] = True
    cover = member.sum(axis=1)
    keep = np.ones(rss.shape[0], dtype=bool)
    for r in range(rss.shape[0]):
        if np.min(cover - member[:, r]) != 0:
            cover = cover - member[:, r]
            keep[r] = False
    return rss[keep]


def alpha_rss(ann, y, dne=None, alpha=0):
    """Annchor.alpha_rss (annchor.py:903-927)."""
    if dne is None:
        if not hasattr(ann, "nearest_enemy_graph"):
            nearest_enemies(ann, y)
        dne = ann.nearest_enemy_graph[1][:, 0]
    dne = np.asarray(dne, dtype=np.float64)
    order = np.argsort(dne, kind="stable")
    adne = dne / (1 + alpha)
    rss = [int(order[0])]
    ann.rssDs = {}
    for i in order:
        IJ = np.stack([np.full(len(rss), i, dtype=np.int64), np.array(rss, dtype=np.int64)], axis=1)
        ds = np.asarray(ann.get_exact_ijs(ann.f, ann.X, IJ))
        ann.rssDs[int(i)] = ds
        m = ds.min()
        if m > adne[i] or np.isclose(m, adne[i]):
            rss.append(int(i))
    return np.array(rss, dtype=np.int64)
