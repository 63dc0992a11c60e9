"""
Nearest-enemy graph, selective subset and alpha-RSS (reference annchor/annchor.py:685-927;
SURVEY.md section 8, row f4).

Host orchestration in NumPy over the fitted pair-list state (as the reference's is); every
metric evaluation goes through `ann.get_exact_ijs`, i.e. the HIP metric kernels for the
built-in metrics.  All arrays are flat / CSR-shaped; nothing here loops over pairs in Python.

Semantics notes (shared with the rest of the build): ties are broken by stable sorts on list
order (the reference's argsorts are unstable); per-point candidate lists are complete (the
reference's `get_IJs_from_check` drops entries of its last groups, utils.py:518,521).
"""
import numpy as np

ROW_BLOCK = 2048
FIRST_ENEMIES = 50  # annchor.py:755-756: exact distances for the 50 closest-looking enemies


def _pair_keys(IJs, nx):
    return IJs[:, 0].astype(np.int64) * nx + IJs[:, 1].astype(np.int64)


def _anchor_sets_matrix(sid, nx, na):
    Am = np.zeros((nx, na), dtype=np.float32)
    np.put_along_axis(Am, np.asarray(sid, dtype=np.int64), 1.0, axis=1)
    return Am


def enemy_candidate_pairs(sid, n_anchors, y, fit_IJs, loc_thresh, loc_min):
    """New candidate pairs (i < j, sorted by (i, j)): enemies sharing nearest anchors, per
    get_check with the label filter (utils.py:454-491) + adjust_check (utils.py:437-451), minus
    the pairs fit() already holds (annchor.py:717-722)."""
    nx = y.shape[0]
    Am = _anchor_sets_matrix(sid, nx, n_anchors)
    thr = np.empty(nx, dtype=np.int64)
    for r0 in range(0, nx, ROW_BLOCK):  # pass 1: per-row threshold among enemies only
        r1 = min(nx, r0 + ROW_BLOCK)
        C = (Am[r0:r1] @ Am.T).astype(np.int32)
        enemy = y[r0:r1, None] != y[None, :]
        Cm = np.where(enemy, C, -1)
        n_enemy = enemy.sum(axis=1)
        lm = np.minimum(loc_min, n_enemy - 1)
        Cs = -np.sort(-Cm, axis=1)
        kth = np.take_along_axis(Cs, lm[:, None], axis=1)[:, 0]
        thr[r0:r1] = np.minimum(loc_thresh, kth)
    lowered = bool(np.any(thr < loc_thresh))
    fit_keys = np.sort(_pair_keys(fit_IJs, nx))
    out = []
    for r0 in range(0, nx, ROW_BLOCK):  # pass 2: emit (a, b), a < b
        r1 = min(nx, r0 + ROW_BLOCK)
        C = (Am[r0:r1] @ Am.T).astype(np.int32)
        keep = C >= thr[r0:r1, None]
        if lowered:  # the smaller index learns the pair from the larger one's list
            keep |= C >= thr[None, :]
        keep &= y[r0:r1, None] != y[None, :]
        keep &= np.arange(nx)[None, :] > np.arange(r0, r1)[:, None]
        a, b = np.nonzero(keep)
        keys = (a + r0).astype(np.int64) * nx + b
        pos = np.searchsorted(fit_keys, keys)
        known = (pos < fit_keys.shape[0]) & (fit_keys[np.minimum(pos, fit_keys.shape[0] - 1)] == keys)
        out.append(keys[~known])
    keys = np.concatenate(out) if out else np.zeros(0, dtype=np.int64)
    return np.stack([keys // nx, keys % nx], axis=1).astype(np.int64)


def pair_features(IJs, D, A, chunk=1 << 18):
    """[lb, ub, dad, is_anchor] per pair (get_features_IJ, annchor.py:258-303) from the anchor
    distance table, in bounded chunks."""
    n = IJs.shape[0]
    out = np.empty((n, 4))
    cA = np.argmin(D, axis=1)
    is_anchor = np.zeros(D.shape[0], dtype=bool)
    is_anchor[np.asarray(A, dtype=np.int64)] = True
    for s in range(0, n, chunk):
        i, j = IJs[s:s + chunk, 0], IJs[s:s + chunk, 1]
        Di, Dj = D[i], D[j]
        out[s:s + chunk, 0] = np.abs(Di - Dj).max(axis=1)
        out[s:s + chunk, 1] = (Di + Dj).min(axis=1)
        out[s:s + chunk, 2] = (D[i, cA[j]] + D[j, cA[i]]) / 2
        out[s:s + chunk, 3] = is_anchor[i] | is_anchor[j]
    return out


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
    """Annchor.get_nearest_enemies (annchor.py:685-782).  Extends ann's pair-list views
    (IJs, I, features, not_computed_mask, RefineApprox) in place of the reference's appends and
    returns (idx int64 [nx, nn], dist float64 [nx, nn])."""
    from .annchor import _IndexCSR

    nx = ann.nx
    y = np.asarray(y)
    assert len(y) == nx, "Label dimension mismatch: len(y)=%d, len(X)=%d" % (len(y), nx)
    labels, counts = np.unique(y, return_counts=True)
    assert len(labels) > 1, "Data must have more than one label"
    assert np.all(counts >= nn), "At least one label occurs fewer times than specified nn=%d" % nn

    IJs0, RA0, ncm0, F0 = ann.IJs, ann.RefineApprox, ann.not_computed_mask, ann.features
    n0 = IJs0.shape[0]
    IJn = enemy_candidate_pairs(ann.sid, ann.n_anchors, y, IJs0, ann.loc_thresh, loc_min)
    Fn = pair_features(IJn, ann.D, ann.A)
    pred = ann.regression.predict(Fn, ann.feature_names)
    ilb, iub = ann.feature_names.index("lower bound"), ann.feature_names.index("upper bound")
    pred = np.clip(pred, Fn[:, ilb], Fn[:, iub])
    IJs = np.vstack([IJs0, IJn])
    RA = np.concatenate([RA0, pred])
    ncm = np.concatenate([ncm0, Fn[:, 3] < 1])
    feats = np.vstack([F0, Fn])
    ptr_n, pos_n = rows_of_pairs(IJn, nx)
    ptr, pos = RowLists.merge(ann.I.ptr, ann.I.idx, ptr_n, pos_n + n0)
    L = RowLists(ptr, pos, IJs)
    if np.any(np.diff(ptr) <= nn):
        raise ValueError("a point has no more than nn=%d candidates" % nn)
    enemy = y[L.other] != y[L.owner]

    # exact distances for the uncomputed among each row's FIRST_ENEMIES closest-looking enemies
    e = np.nonzero(enemy)[0]
    order = e[np.lexsort((RA[pos[e]], L.owner[e]))]
    head = order[_rank_in_row(L.owner[order]) < FIRST_ENEMIES]
    todo = pos[head]
    todo = todo[ncm[todo]]
    if todo.shape[0] > 0:
        RA[todo] = ann.get_exact_ijs(ann.f, ann.X, IJs[todo])
        ncm[todo] = False

    # nn nearest per row: computed enemies first (uncomputed and same-label entries pushed
    # behind by the row maximum, annchor.py:766-772)
    mx = np.maximum.reduceat(RA[pos], ptr[:-1])[L.owner]
    d = RA[pos] + mx * ncm[pos] + mx * (~enemy)
    order = np.lexsort((d, L.owner))
    top = order[_rank_in_row(L.owner[order]) < nn]
    ngi = L.other[top].reshape(nx, nn)
    ngd = RA[pos[top]].reshape(nx, nn)

    # Like the reference (annchor.py:748-781), the object's pair-list views now include the enemy pairs.
    # The device keeps the fitted (shorter) state, so the views that are loaded lazily from it and the
    # stage methods no longer describe the same pair list: the object says so instead of mixing lengths.
    ann._cache.update(IJs=IJs, RA=RA, ncm=ncm, features=feats, I=_IndexCSR(ptr, pos))
    ann._enemy_extended = True
    ann.nearest_enemy_graph = (ngi, ngd)
    return ngi, ngd


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
