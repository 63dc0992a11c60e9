"""
oracle/annchor_oracle.py -- CPU (NumPy) restatement of the ANNchor `fit()` hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under annchor_amd/ may import this module; only
tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg use it, as the
checker and the timed CPU baseline -- never as the thing shipped or measured as
the product.

Every function cites the reference lines (relative to /root/reference/) it
restates.  The reference is pure Python/NumPy + numba; its own functions are
deterministic except where NumPy's *unstable* selection routines decide between
tied keys (np.argsort default kind, np.argpartition, numba quicksort).  The
restatement fixes ONE tie rule everywhere ("ties resolve to the smaller index /
earlier position") -- the same rule the HIP kernels implement -- so that
oracle == GPU is a bit-exact comparison, while oracle-vs-reference is compared
exactly on every tie-free quantity and set-wise (cut-off value + membership
above/below it) on the tie-affected selections.  See tests/test_oracle_golden.py.

Pinning status:
  * reference's own functions: pinned by golden vectors captured from the imported
    reference (tests/golden/make_golden.py, run in the build container where
    /root/reference exists) and by the reference tests' known answers;
  * RNG stream of the sampler: the reference seeds numba's in-njit RNG
    (utils.py:572), which cannot be run here; this restatement (and the golden
    capture) use NumPy's legacy global RNG with the same seed arithmetic --
    "parity unpinned" for the exact sample set of a real numba run;
  * OLS coefficients: sklearn LinearRegression (regressors.py:33,68) -- pinned by
    captured coefficients only;
  * query (query_functions.py:10-212, annchor.py:643-683): pinned -- gen_query() in
    tests/golden/make_golden.py drives the imported reference's helpers stage by stage and
    Annchor.query end to end (strings split + the digits split of the reference's own test);
  * order inside groups of equal refinement probability (annchor.py:444-457: np.argpartition,
    arbitrary): a fixed scrambled-position order here (select_candidates); compared with the
    reference set-wise (same cut value, same members strictly above it);
  * hashed_stratified_sample: restates the BUILD's DeviceStratifiedSampler plugin, which has
    no reference counterpart (the reference's draw is stratified_sample).
"""
from __future__ import annotations

import numpy as np
import scipy.linalg

FEATURE_NAMES = ["lower bound", "upper bound", "double anchor distance", "is anchor"]


# --------------------------------------------------------------------------- a0
def budget(nx, n_anchors, n_samples, p_work, n_neighbors, loc_min=None):
    """Constructor arithmetic, annchor.py:117-148,167-168."""
    N = (nx * (nx - 1)) // 2
    na = int(sum(nx - j for j in range(1, n_anchors + 1)))
    if p_work > 1:
        p_work = 1.0
    min_p_work = (2 * (na + n_samples) + 1) / N
    min_p_work = 1 if min_p_work > 1 else min_p_work
    if p_work < min_p_work:
        p_work = min_p_work
    lm = 10 * n_neighbors if loc_min is None else loc_min
    lm = int(np.clip(lm, 0, nx - 1))
    return dict(N=N, na=na, p_work=p_work, loc_min=lm)


# --------------------------------------------------------------------------- a6
def maxmin_anchors(one_to_all, nx, n_anchors, seed):
    """MaxMinAnchorPicker.get_anchors, pickers.py:18-52.

    one_to_all(ix) -> float64[nx] with out[j] = f(X[ix], X[j]).
    Note the reference quirk kept here: after the first round the running min
    excludes anchor 0's row (`np_min(D[1:], 0)`, pickers.py:47-50).
    np.argmax returns the first maximal index (defined tie rule).
    """
    np.random.seed(seed)
    D = np.zeros((n_anchors, nx)) + np.inf
    A = np.zeros(n_anchors, dtype=np.int64)
    ix = np.random.randint(nx)
    for i in range(n_anchors):
        A[i] = ix
        D[i] = one_to_all(ix)
        if i == 0:
            ix = int(np.argmax(D[:1].min(axis=0)))
        else:
            ix = int(np.argmax(D[1 : i + 1].min(axis=0)))
    return A, np.ascontiguousarray(D.T)


def maxmin_first_index(nx, seed):
    """The only RNG draw of the picker (pickers.py:21,30)."""
    np.random.seed(seed)
    return int(np.random.randint(nx))


# --------------------------------------------------------------------------- a7
def nearest_anchor_sets(D, locality):
    """sid = argsort(D, axis=1)[:, :locality], annchor.py:235 (stable tie rule)."""
    return np.argsort(D, axis=1, kind="stable")[:, :locality]


def locality_pairs(D, locality, loc_thresh, loc_min):
    """get_locality / get_check / adjust_check / get_IJs_from_check,
    annchor.py:208-256, utils.py:437-540.

    Returns sid, IJs (int64 [n,2], i<j, sorted by (i,j)), I_ptr (int64 [nx+1]),
    I_idx (int64 [2n]) with I_idx[I_ptr[i]:I_ptr[i+1]] = positions in IJs that
    contain i, ordered by the other endpoint ascending (the reference's order
    inside a group comes from an unstable argsort, utils.py:512, i.e. is
    arbitrary; every consumer is order-free up to ties).

    Deviation (documented): the reference's trailing-boundary arithmetic
    (utils.py:518,521) truncates I[.] for the last groups when the final i-group
    has more than one pair; the restatement keeps the full groups.
    """
    nx, na = D.shape
    sid = nearest_anchor_sets(D, locality)
    Am = np.zeros((nx, na), dtype=np.int32)
    np.put_along_axis(Am, sid, 1, axis=1)
    C = Am @ Am.T  # C[i,j] = |sid[i] & sid[j]|  == sum(A[sid[i], :], axis=0)[j]
    _loc_min = min(loc_min, nx - 1)
    # (loc_min+1)-th largest count in each row (utils.py:472-473)
    kth = -np.partition(-C, _loc_min, axis=1)[:, _loc_min]
    thr = np.minimum(loc_thresh, kth)  # utils.py:475-480
    keep = (C >= thr[:, None]) | (C >= thr[None, :])  # adjust_check symmetrisation
    iu = np.triu(keep, k=1)
    I0, J0 = np.nonzero(iu)  # row-major => sorted by (i, j)
    IJs = np.stack([I0, J0], axis=1).astype(np.int64)
    I_ptr, I_idx = build_I(IJs, nx)
    return sid, IJs, I_ptr, I_idx


def build_I(IJs, nx, reference_truncation=False):
    """CSR of pair positions per point (get_IJs_from_check, utils.py:502-540).
    reference_truncation=True reproduces the reference's trailing-boundary arithmetic
    (utils.py:518,521: both group-boundary arrays are closed with `start of the last i-group
    + 1` resp. `+ 2`), which drops entries of the last i-group and of the last j-group; it is
    used only to show that the remaining differences from the reference's nearest-enemy
    outputs are that latent bug and nothing else."""
    if reference_truncation:
        return _build_I_truncated(IJs, nx)
    n = IJs.shape[0]
    pos = np.arange(n, dtype=np.int64)
    owner = np.concatenate([IJs[:, 1], IJs[:, 0]])
    other = np.concatenate([IJs[:, 0], IJs[:, 1]])
    p2 = np.concatenate([pos, pos])
    order = np.lexsort((other, owner))
    I_idx = p2[order]
    counts = np.bincount(owner, minlength=nx)
    I_ptr = np.zeros(nx + 1, dtype=np.int64)
    np.cumsum(counts, out=I_ptr[1:])
    return I_ptr, I_idx


def _build_I_truncated(IJs, nx):
    n = IJs.shape[0]
    fi = IJs[:, 0]
    jsort = np.argsort(IJs[:, 1])  # NumPy default kind, as the reference calls it
    fj = IJs[jsort, 1]
    i_start = np.concatenate([[0], np.nonzero(np.diff(fi))[0] + 1])
    j_start = np.concatenate([[0], np.nonzero(np.diff(fj))[0] + 1])
    i_end = np.concatenate([i_start[1:], [i_start[-1] + 1]])   # last i-group: one entry
    j_end = np.concatenate([j_start[1:], [i_start[-1] + 2]])   # last j-group: closed with the i bound
    rows = [[np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)] for _ in range(nx)]
    for s0, e0 in zip(j_start, j_end):
        rows[fj[s0]][0] = jsort[s0:max(e0, s0)]
    for s0, e0 in zip(i_start, i_end):
        rows[fi[s0]][1] = np.arange(s0, e0, dtype=np.int64)
    rows = [np.concatenate(r) for r in rows]
    I_ptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    return I_ptr, (np.concatenate(rows).astype(np.int64) if n else np.zeros(0, dtype=np.int64))


def check_locality_size(I_ptr, n_neighbors):
    """utils.py:592-597 / annchor.py:252-256."""
    return bool(np.any(np.diff(I_ptr) < n_neighbors))


# ------------------------------------------------------------------- a8, a9, a10
def bounds(IJs, D):
    """get_bounds_njit_ijs, utils.py:274-301."""
    Di, Dj = D[IJs[:, 0]], D[IJs[:, 1]]
    return np.abs(Di - Dj).max(axis=1), (Di + Dj).min(axis=1)


def dad(IJs, D):
    """get_dad_ijs, utils.py:355-380 (argmin = first minimal index)."""
    cA = np.argmin(D, axis=1)
    i, j = IJs[:, 0], IJs[:, 1]
    return (D[i, cA[j]] + D[j, cA[i]]) / 2


def features(IJs, D, A, I_ptr, I_idx):
    """get_features_IJ, annchor.py:258-303."""
    lb, ub = bounds(IJs, D)
    dd = dad(IJs, D)
    anchors = np.zeros(IJs.shape[0])
    for a in np.asarray(A, dtype=np.int64):
        anchors[I_idx[I_ptr[a] : I_ptr[a + 1]]] = 1
    feats = np.vstack([lb, ub, dd, anchors]).T
    return feats, feats[:, 3] < 1


# -------------------------------------------------------------------------- a11
def stratified_partition(sample_feature, n_samples, n_partitions=7):
    """SimpleStratifiedSampler.get_partition, samplers.py:119-140."""
    n = sample_feature.shape[0]
    iq1, iq3 = int(n / 100), int(99 * n / 100)
    if iq1 * n_partitions < n_samples:
        iq1, iq3 = int(n / 10), int(9 * n / 10)
    if iq1 * n_partitions < n_samples:
        n_samples = iq1 * n_partitions
    q1 = np.partition(sample_feature, iq1)[iq1]
    q3 = np.partition(sample_feature, iq3)[iq3]
    b = np.linspace(q1, q3, n_partitions - 1)
    return np.hstack([-np.inf, b, np.inf]), n_samples


class NothingToSample(Exception):
    pass


def stratified_sample(feats, ncm, n_samples, seed, loop_num, n_partitions=7):
    """Sampler.sample + sample_partition + loop_partitions,
    samplers.py:44-110, utils.py:543-578 (NumPy legacy RNG stands in for numba's)."""
    if not ncm.any():
        raise NothingToSample()
    sf = feats[ncm][:, 2]
    indices = np.arange(ncm.shape[0])[ncm]
    bins, n_samples = stratified_partition(sf, n_samples, n_partitions)
    if n_samples == 0:
        raise NothingToSample()
    bin_size, remainder = n_samples // n_partitions, n_samples % n_partitions
    np.random.seed(seed + loop_num)
    out = []
    for b in range(n_partitions):
        mask = (sf >= bins[b]) & (sf < bins[b + 1])
        ixmask = indices[mask]
        want = bin_size + (b < remainder)
        if ixmask.shape[0] < want:
            got = ixmask
        else:
            got = np.random.choice(ixmask, size=want, replace=False)
        if len(got) < 2:
            raise Exception("Some sampler bins contain too few samples")
        out.append(got)
    ixs = np.hstack(out)
    return ixs, ixs.shape[0], bins


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def hashed_stratified_sample(feats, ncm, n_samples, seed, loop_num, n_partitions=7):
    """The build's DeviceStratifiedSampler (no reference counterpart: the reference's choice inside a
    partition is np.random.choice, samplers.py:44-73): same partitions and quotas as
    stratified_sample, members chosen by the smallest splitmix64(splitmix64(seed + loop_num) ^ position), ties to
    the smaller position; output partition by partition, ascending position inside a partition."""
    idx = np.nonzero(ncm)[0]
    if idx.shape[0] == 0:
        raise NothingToSample()
    f = feats[idx, 2]
    bins, n_samples = stratified_partition(f, n_samples, n_partitions)
    if n_samples == 0:
        raise NothingToSample()
    step_key = splitmix64(np.uint64((int(seed) + int(loop_num)) & ((1 << 64) - 1)))
    keys = splitmix64(step_key ^ idx.astype(np.uint64))
    out = []
    for b in range(n_partitions):
        m = np.nonzero((f >= bins[b]) & (f < bins[b + 1]))[0]
        want = n_samples // n_partitions + (1 if b < n_samples % n_partitions else 0)
        if len(m) > want:
            m = m[np.lexsort((idx[m], keys[m]))[:want]]
        out.append(np.sort(idx[m]))
    if min(len(o) for o in out) < 2:
        raise Exception("Some sampler bins contain too few samples")
    ixs = np.concatenate(out)
    return ixs, ixs.shape[0], bins


# -------------------------------------------------------------------------- a12
def ols(X, y):
    """sklearn LinearRegression(fit_intercept=True).fit: centre, lstsq (gelsd)."""
    xm, ym = X.mean(axis=0), y.mean()
    Xc, yc = X - xm, y - ym
    cond = max(Xc.shape) * np.finfo(Xc.dtype).eps
    coef = scipy.linalg.lstsq(Xc, yc, cond=cond)[0]
    return coef, ym - xm @ coef


def regression_fit(sample_feats, sample_y, bins):
    """SimpleStratifiedLinearRegression.fit, regressors.py:39-69 (bin edges
    `lo < F <= hi`)."""
    F = sample_feats[:, 2]
    nb = bins.shape[0] - 1
    W = np.zeros((nb, 3))
    c = np.zeros(nb)
    for b in range(nb):
        m = (F > bins[b]) & (F <= bins[b + 1])
        W[b], c[b] = ols(sample_feats[m][:, :3], sample_y[m])
    return W, c


def regression_bin(F, bins):
    """Index b with bins[b] < F <= bins[b+1] (regressors.py:84-87,97-101)."""
    return np.clip(np.searchsorted(bins, F, side="left") - 1, 0, bins.shape[0] - 2)


def regression_predict(feats, bins, W, c):
    """SimpleStratifiedLinearRegression.predict, regressors.py:71-103.
    Arithmetic order fixed as ((w0*lb + w1*ub) + w2*dad) + c, no FMA."""
    b = regression_bin(feats[:, 2], bins)
    w = W[b]
    return ((w[:, 0] * feats[:, 0] + w[:, 1] * feats[:, 1]) + w[:, 2] * feats[:, 2]) + c[b]


def merge_prediction(RA, pred, feats, ncm, sample_ixs, sample_y):
    """annchor.py:359-380: clip to [lb, ub]; first call initialises RefineApprox,
    later calls overwrite only not-computed entries; samples get exact values."""
    pred = np.minimum(np.maximum(pred, feats[:, 0]), feats[:, 1])
    if RA is None:
        RA = pred.copy()
    else:
        RA[ncm] = pred[ncm]
    RA[sample_ixs] = sample_y
    return RA


# -------------------------------------------------------------------------- a13
def error_fit(sample_feats, sample_err, bins):
    """SimpleStratifiedErrorRegression.fit, error_predictors.py:26-54
    (closed on both sides)."""
    sf = sample_feats[:, 2]
    errs = []
    for b in range(bins.shape[0] - 1):
        m = (sf >= bins[b]) & (sf <= bins[b + 1])
        errs.append(np.sort(sample_err[m]))
    return errs


def error_labels(F, bins):
    """SimpleStratifiedErrorRegression.predict, error_predictors.py:56-67: later
    bins overwrite earlier ones at shared edges => largest b with lo_b <= F."""
    return np.clip(np.searchsorted(bins, F, side="right") - 1, 0, bins.shape[0] - 2)


# -------------------------------------------------------------------------- a14
def row_kth(RA, I_ptr, I_idx, k):
    """thresh[i] = np.partition(RA[I[i]], k)[k], annchor.py:399-404."""
    nx = I_ptr.shape[0] - 1
    out = np.empty(nx)
    for i in range(nx):
        v = RA[I_idx[I_ptr[i] : I_ptr[i + 1]]]
        out[i] = np.partition(v, k)[k]
    return out


def guarantee_nmin(RA, ncm, I_ptr, I_idx, nmin):
    """guarantee_nmin + argpartition, utils.py:600-621 (sequential in i, in place)."""
    nx = I_ptr.shape[0] - 1
    for i in range(nx):
        Ii = I_idx[I_ptr[i] : I_ptr[i + 1]]
        mask = ncm[Ii]
        n_todo = nmin - int(np.sum(~mask))
        if n_todo > 0:
            a = RA[Ii][mask]
            dxs = np.partition(a, n_todo)[n_todo]
            RA[Ii[mask][a < dxs]] = -1
    return RA


def ecdf_prob(p, labels, errs):
    """get_probs, utils.py:581-589: searchsorted(side='left') / len."""
    prob = np.empty(p.shape)
    for b, e in enumerate(errs):
        m = labels == b
        prob[m] = np.searchsorted(e, p[m])
        prob[m] /= len(e)
    return prob


def n_refine_budget(p_work, N, na, n_samples, w):
    """annchor.py:438-442."""
    n = int((p_work * N - na - n_samples) * w) + 1
    return 0 if n < 0 else n


TIE_SCRAMBLE = np.uint64(0x9E3779B97F4A7C15)


def tie_scramble(positions):
    """Order of equally probable pairs: ascending (position * 0x9E3779B97F4A7C15 mod 2^64) >> 11 --
    a fixed multiplicative hash of the pair's position in the pair list."""
    with np.errstate(over="ignore"):
        return (np.asarray(positions, dtype=np.uint64) * TIE_SCRAMBLE) >> np.uint64(11)


def select_candidates(prob, n_refine, lookahead, positions=None):
    """annchor.py:444-457.  The reference takes the top of np.argpartition(-prob): inside a group of
    equal probabilities (the ECDF has a few thousand distinct values for ~10^6 pairs, so the group on
    the cut holds hundreds to thousands of pairs) its choice is arbitrary.  Resolving such a group by
    position would hand the whole remainder of the budget to the first rows of the pair list;
    resolving it by predicted distance biases the population the next iteration's model is fitted on
    (measured: query recall 0.96-0.98 instead of 1.0 on the reference's digits test).  Tie rule here
    and in the kernels: (prob descending, tie_scramble(position) ascending, position ascending) -- a
    fixed pseudo-random order, like the reference's in effect, but reproducible.  `positions` = the
    pair-list positions of the entries of `prob` (default: 0..n-1).  Returns (candidates, next) as
    indices into `prob`, each sorted ascending."""
    n = prob.shape[0]
    if n_refine >= n:
        return np.arange(n), np.arange(n)
    pos = np.arange(n) if positions is None else np.asarray(positions)
    order = np.lexsort((pos, tie_scramble(pos), -prob))
    big = order if n_refine * lookahead >= n else order[: n_refine * lookahead]
    return np.sort(big[:n_refine]), np.sort(big[n_refine:])


def refine_probabilities(RA, ncm, IJs, thresh, labels_all, errs):
    """p and prob of annchor.py:416-436 on the compacted not-computed array."""
    p0 = (thresh[IJs[:, 0]] - RA)[ncm]
    p1 = (thresh[IJs[:, 1]] - RA)[ncm]
    p = np.maximum(p0, p1)
    return ecdf_prob(p, labels_all[ncm], errs)


# -------------------------------------------------------------------------- a15
def update_bounds(IJs, RA, ncm, I_ptr, I_idx, nextback, lb, ub):
    """update_anchor_points + update_bounds/get_bounds_alt, annchor.py:475-512,
    utils.py:304-352, without the wall-clock cut-off (all chunks processed)."""
    nx = I_ptr.shape[0] - 1
    dis, ds = [], []
    for y in range(nx):
        Iy = I_idx[I_ptr[y] : I_ptr[y + 1]]
        m = Iy[~ncm[Iy]]
        pr = IJs[m]
        other = np.where(pr[:, 0] == y, pr[:, 1], pr[:, 0])
        o = np.argsort(other, kind="stable")
        dis.append(other[o])
        ds.append(RA[m][o])
    lb, ub = lb.copy(), ub.copy()
    for t in nextback:
        i, j = IJs[t]
        _, ia, ja = np.intersect1d(dis[i], dis[j], assume_unique=True, return_indices=True)
        if ia.size:
            a = ds[i][ia] + ds[j][ja]
            b = np.abs(ds[i][ia] - ds[j][ja])
            ub[t] = min(ub[t], a.min())
            lb[t] = max(lb[t], b.max())
    return lb, ub


# -------------------------------------------------------------------------- a16
def get_nn(RA, ncm, IJs, I_ptr, I_idx, nn):
    """get_nn + get_ann, utils.py:383-429, annchor.py:514-530.  Tie rule: equal
    distances come out in I[i] order (= other endpoint ascending)."""
    nx = I_ptr.shape[0] - 1
    ngi = np.zeros((nx, nn - 1), dtype=np.int64)
    ngd = np.zeros((nx, nn - 1))
    for i in range(nx):
        Ii = I_idx[I_ptr[i] : I_ptr[i + 1]]
        d = RA[Ii].copy()
        mx = d.max()
        d[ncm[Ii]] += mx
        t = np.partition(d, nn - 1)[nn - 1]
        m = d <= t
        iy = Ii[m][np.argsort(d[m], kind="stable")][: nn - 1]
        ngd[i] = RA[iy]
        f = IJs[iy]
        ngi[i] = np.where(f[:, 0] == i, f[:, 1], f[:, 0])
    return (
        np.hstack([np.arange(nx)[:, None], ngi]),
        np.hstack([np.zeros((nx, 1)), ngd]),
    )


# -------------------------------------------------------------------------- a17
def compare_neighbor_graphs(nng_1, nng_2, n_neighbors):
    """annchor.py:1026-1066."""
    from collections import Counter

    err = 0
    for ix in range(nng_1[0].shape[0]):
        a = Counter(np.round(nng_1[1][ix][:n_neighbors], 3).astype(np.float32))
        b = Counter(np.round(nng_2[1][ix][:n_neighbors], 3).astype(np.float32))
        err += len(a - b)
    return err


def brute_force(metric_pairs, nx):
    """BruteForce.fit, annchor.py:1004-1023 (stable sort)."""
    iu = np.triu_indices(nx, k=1)
    IJ = np.stack(iu, axis=1).astype(np.int64)
    d = metric_pairs(IJ)
    D = np.zeros((nx, nx))
    D[iu] = d
    D = D + D.T
    idx = np.argsort(D, axis=1, kind="stable")
    return idx, np.take_along_axis(D, idx, axis=1), D


# ----------------------------------------------------------------------- driver
class OracleAnnchor:
    """Annchor.fit(), annchor.py:532-623, on top of the stage functions above.

    metric_pairs(IJ int64[n,2]) -> float64[n] is the a2 boundary
    (utils.py:110-177).  Default plugins only (MaxMin picker, stratified
    sampler / linear regression / error regression)."""

    def __init__(self, nx, metric_pairs, n_anchors=20, n_neighbors=15, n_samples=5000,
                 p_work=0.1, random_seed=42, locality=5, loc_thresh=1, loc_min=None,
                 niters=2, lookahead=5, anchors=None, trace=None, sampler="legacy"):
        self.nx, self.metric_pairs = nx, metric_pairs
        b = budget(nx, n_anchors, n_samples, p_work, n_neighbors, loc_min)
        self.N, self.na, self.p_work, self.loc_min = b["N"], b["na"], b["p_work"], b["loc_min"]
        self.n_anchors, self.n_neighbors, self.n_samples = n_anchors, n_neighbors, n_samples
        self.random_seed, self.locality, self.loc_thresh = random_seed, locality, loc_thresh
        self.niters, self.lookahead = niters, lookahead
        self.evals = 0
        self.anchors = anchors
        self.sampler = sampler   # "legacy": NumPy-stream choice (the reference's); "hashed": DeviceStratifiedSampler
        self.trace = trace  # optional dict collecting per-stage snapshots

    def _snap(self, key, **kw):
        if self.trace is not None:
            self.trace[key] = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in kw.items()}

    def one_to_all(self, ix):
        IJ = np.stack([np.full(self.nx, ix, dtype=np.int64), np.arange(self.nx, dtype=np.int64)], axis=1)
        return self.metric_pairs(IJ)

    def fit(self):
        nx, k = self.nx, self.n_neighbors
        if self.anchors is None:
            self.A, self.D = maxmin_anchors(self.one_to_all, nx, self.n_anchors, self.random_seed)
        else:  # SelectedAnchorPicker, pickers.py:86-106
            self.A = np.asarray(self.anchors, dtype=np.int64)
            self.D = np.stack([self.one_to_all(int(a)) for a in self.A], axis=1)
        self.evals += self.n_anchors * nx
        self.sid, self.IJs, self.I_ptr, self.I_idx = locality_pairs(
            self.D, self.locality, self.loc_thresh, self.loc_min)
        if check_locality_size(self.I_ptr, k):
            raise Exception("Error: Not enough candidates in pool for all indices.\n"
                            "Try again with higher locality.")
        self.features, self.ncm = features(self.IJs, self.D, self.A, self.I_ptr, self.I_idx)
        self._snap("features", features=self.features, ncm=self.ncm)
        self.RA = None
        for it in range(self.niters):
            try:
                draw = hashed_stratified_sample if self.sampler == "hashed" else stratified_sample
                self.sample_ixs, self.n_samples, self.bins = draw(self.features, self.ncm, self.n_samples, self.random_seed, it)
            except NothingToSample:
                if it == 0:
                    raise ValueError("Sampler raised NothingToSample on first iteration.")
                break
            sf = self.features[self.sample_ixs]
            self.sample_y = self.metric_pairs(self.IJs[self.sample_ixs])
            self.ncm[self.sample_ixs] = False
            self.evals += self.sample_y.shape[0]
            self.W, self.c = regression_fit(sf, self.sample_y, self.bins)
            pred = regression_predict(self.features, self.bins, self.W, self.c)
            sample_predict = pred[self.sample_ixs]
            self.RA = merge_prediction(self.RA, pred, self.features, self.ncm,
                                       self.sample_ixs, self.sample_y)
            self.errs = error_fit(sf, self.sample_y - sample_predict, self.bins)
            self.labels = error_labels(self.features[:, 2], self.bins)
            self._snap("regress%d" % it, sample_ixs=self.sample_ixs, bins=self.bins, W=self.W,
                       c=self.c, RA=self.RA, labels=self.labels, sample_y=self.sample_y)
            # select_refine_candidate_pairs
            self.thresh = row_kth(self.RA, self.I_ptr, self.I_idx, k)
            if it == 0:
                self.RA = guarantee_nmin(self.RA, self.ncm, self.I_ptr, self.I_idx, 3 * k // 2)
            prob = refine_probabilities(self.RA, self.ncm, self.IJs, self.thresh,
                                        self.labels, self.errs)
            n_refine = n_refine_budget(self.p_work, self.N, self.na, self.n_samples, 1 / self.niters)
            cand, nxt = select_candidates(prob, n_refine, self.lookahead, positions=np.flatnonzero(self.ncm))
            unc = np.arange(self.ncm.shape[0])[self.ncm]
            self.nextback, mapback = unc[nxt], unc[cand]
            self._snap("select%d" % it, thresh=self.thresh, RA=self.RA, prob=prob,
                       mapback=mapback, nextback=self.nextback, n_refine=n_refine)
            exact = self.metric_pairs(self.IJs[mapback])
            self.evals += exact.shape[0]
            self.RA[mapback] = exact
            self.ncm[mapback] = False
            if it < self.niters - 1:
                lb, ub = update_bounds(self.IJs, self.RA, self.ncm, self.I_ptr, self.I_idx,
                                       self.nextback, self.features[:, 0], self.features[:, 1])
                self.features[:, 0], self.features[:, 1] = lb, ub
                self._snap("update%d" % it, lb=lb, ub=ub)
        self.neighbor_graph = get_nn(self.RA, self.ncm, self.IJs, self.I_ptr, self.I_idx, k)
        return self


# ------------------------------------------------------------------- query (f2)
def query_anchor_dists(query_pairs, A, nq):
    """get_query_anchor_dists, query_functions.py:10-15: QD[j, a] = f(X[A[a]], Q[j])."""
    A = np.asarray(A, dtype=np.int64)
    IJa = np.stack([np.repeat(A, nq), np.tile(np.arange(nq), len(A))], axis=1)
    return query_pairs(IJa).reshape(len(A), nq).T


def query_locality(sid_x, QD, locality, loc_thresh, na):
    """get_query_locality, query_functions.py:18-37, + the pair list of get_query_features (:42-47):
    data points sharing >= loc_thresh of the `locality` nearest anchors with the query, NO loc_min
    widening; pairs (i in X, j in Q) sorted by (j, i).  sid_x = the fitted nearest-anchor sets
    (ann.Amatrix is their one-hot form, annchor.py:237-241)."""
    nx, nq = len(sid_x), len(QD)
    sid_q = nearest_anchor_sets(QD, locality)
    Ax = np.zeros((nx, na), dtype=np.int32)
    np.put_along_axis(Ax, np.asarray(sid_x, dtype=np.int64), 1, axis=1)
    Aq = np.zeros((nq, na), dtype=np.int32)
    np.put_along_axis(Aq, sid_q, 1, axis=1)
    C = Aq @ Ax.T                                  # [nq, nx]
    J, I = np.nonzero(C >= loc_thresh)             # row-major: sorted by (j, i)
    IJs = np.stack([I, J], axis=1).astype(np.int64)
    QI_ptr = np.concatenate([[0], np.cumsum(np.bincount(J, minlength=nq))]).astype(np.int64)
    return sid_q, IJs, QI_ptr


def query_features(IJs, D, QD, A):
    """get_query_features / get_query_dad_ijs / get_query_bounds_njit_ijs, query_functions.py:40-129."""
    Di, Qj = D[IJs[:, 0]], QD[IJs[:, 1]]
    lb, ub = np.abs(Di - Qj).max(axis=1), (Di + Qj).min(axis=1)
    cA, cQA = np.argmin(D, axis=1), np.argmin(QD, axis=1)
    dd = (D[IJs[:, 0], cQA[IJs[:, 1]]] + QD[IJs[:, 1], cA[IJs[:, 0]]]) / 2
    anchors = np.isin(IJs[:, 0], np.asarray(A)).astype(np.float64)
    feats = np.vstack([lb, ub, dd, anchors]).T
    return feats, feats[:, 3] < 1


def query_n_refine(p_work, nq, nx, n_anchors):
    """query_functions.py:163-168."""
    return int((p_work * nq * nx - n_anchors * nq)) + 1


def query_select(QRA, ncm, IJs, QI_ptr, labels, errs, nn, n_refine):
    """select_refine_candidate_query_pairs up to the metric call, query_functions.py:132-176:
    thresholds on the query side only, guarantee_nmin with 3nn//2, one global top-n_refine."""
    QI_idx = np.arange(len(IJs), dtype=np.int64)
    thresh = row_kth(QRA, QI_ptr, QI_idx, nn)
    QRA = guarantee_nmin(QRA, ncm, QI_ptr, QI_idx, 3 * nn // 2)
    p = (thresh[IJs[:, 1]] - QRA)[ncm]
    prob = ecdf_prob(p, labels[ncm], errs)
    cand, _ = select_candidates(prob, max(n_refine, 0), 1, positions=np.flatnonzero(ncm))
    mapback = np.arange(ncm.shape[0])[ncm][cand]
    return thresh, QRA, prob, mapback


def query_get_nn(QRA, ncm, IJs, QI_ptr, nn):
    """get_nn(nq, nn + 1, ...) of query_ (query_functions.py:210): raw output, the neighbour is
    the X endpoint; no self column."""
    QI_idx = np.arange(len(IJs), dtype=np.int64)
    return get_nn(QRA, ncm, np.stack([IJs[:, 0], IJs[:, 0]], axis=1), QI_ptr, QI_idx, nn + 1)


def query_p_work(p_work, nq, nx, n_anchors, nn):
    """The floor Annchor.query applies first (annchor.py:664-675)."""
    limit = ((nq * nn * 3) // 2 - 1 + n_anchors * nq) / (nq * nx)
    return max(p_work, limit)


def query(fitted, query_pairs, nq, nn=15, p_work=0.3, sid_x=None, apply_floor=False):
    """query_ + helpers, annchor/query_functions.py:10-212, on a fitted OracleAnnchor (or any object
    with nx, n_anchors, A, D, locality, loc_thresh, bins, W, c, errs).

    query_pairs(IJ int64 [n,2]) -> float64[n] with IJ[:,0] indexing X and IJ[:,1] indexing Q
    (get_exact_query_ijs, utils.py:180-245).  Tie rules as everywhere in this module."""
    o = fitted
    nx, na = o.nx, o.n_anchors
    if apply_floor:
        p_work = query_p_work(p_work, nq, nx, na, nn)
    QD = query_anchor_dists(query_pairs, o.A, nq)
    if sid_x is None:
        sid_x = nearest_anchor_sets(o.D, o.locality)
    _, IJs, QI_ptr = query_locality(sid_x, QD, o.locality, o.loc_thresh, na)
    feats, ncm = query_features(IJs, o.D, QD, o.A)
    # predict + clip (:198-203)
    pred = regression_predict(feats, o.bins, o.W, o.c)
    QRA = np.minimum(np.maximum(pred, feats[:, 0]), feats[:, 1])
    labels = error_labels(feats[:, 2], o.bins)
    n_refine = query_n_refine(p_work, nq, nx, na)
    _, QRA, _, mapback = query_select(QRA, ncm, IJs, QI_ptr, labels, o.errs, nn, n_refine)
    QRA[mapback] = query_pairs(IJs[mapback])
    ncm[mapback] = False
    idx, dist = query_get_nn(QRA, ncm, IJs, QI_ptr, nn)
    return idx[:, 1:], dist[:, 1:], dict(QD=QD, IJs=IJs, n_refine=n_refine, evals=na * nq + len(mapback))


# ------------------------------------------------ nearest enemies / selective subset (f4)
def nearest_enemies(fitted, y, nn=3, loc_min=100, first=50, reference_truncation=False):
    """Annchor.get_nearest_enemies, annchor.py:685-782, on a fitted OracleAnnchor.

    Extends the fitted pair list by enemy pairs (different labels) chosen by shared nearest
    anchors among enemies only (get_check with the label filter, utils.py:454-491, minus the
    pairs fit() already holds), predicts them with the fitted regression, evaluates exactly
    each row's uncomputed entries among its `first` closest-looking enemies, and reads the nn
    nearest enemies per row.  The fitted object's pair-list state is extended in place, as in
    the reference.  Returns (idx int64 [nx, nn], dist float64 [nx, nn]).
    Tie rule as everywhere in this restatement: stable sorts (the reference's are unstable)."""
    o = fitted
    nx = o.nx
    y = np.asarray(y)
    assert y.shape[0] == nx, "Label dimension mismatch: len(y)=%d, len(X)=%d" % (y.shape[0], nx)
    labels, counts = np.unique(y, return_counts=True)
    assert labels.shape[0] > 1, "Data must have more than one label"
    assert np.all(counts >= nn), "At least one label occurs fewer times than specified nn=%d" % nn

    na = o.D.shape[1]
    Am = np.zeros((nx, na), dtype=np.int32)
    np.put_along_axis(Am, o.sid, 1, axis=1)
    C = Am @ Am.T
    # candidates fit() already holds (symmetric): keep_fit[i, j]
    keep_fit = np.zeros((nx, nx), dtype=bool)
    keep_fit[o.IJs[:, 0], o.IJs[:, 1]] = True
    keep_fit |= keep_fit.T
    np.fill_diagonal(keep_fit, True)
    new = np.zeros((nx, nx), dtype=bool)
    lowered = False
    for i in range(nx):
        enemy = y != y[i]
        c = C[i][enemy]
        lm = min(loc_min, c.shape[0] - 1)
        kth = -np.partition(-c, lm)[lm]
        thr = o.loc_thresh
        if kth < o.loc_thresh:
            thr, lowered = kth, True
        new[i] = enemy & (C[i] >= thr)
    if lowered:  # adjust_check: the smaller index of a pair learns about it from the larger
        new |= np.tril(new, -1).T
    new &= ~keep_fit
    I0, J0 = np.nonzero(np.triu(new, 1))
    IJn = np.stack([I0, J0], axis=1).astype(np.int64)
    n0 = o.IJs.shape[0]
    ptr_n, idx_n = build_I(IJn, nx, reference_truncation)
    fn, ncm_n = features(IJn, o.D, o.A, ptr_n, idx_n)
    pred = np.clip(regression_predict(fn, o.bins, o.W, o.c), fn[:, 0], fn[:, 1])
    # append (annchor.py:729-740)
    o.IJs = np.vstack([o.IJs, IJn])
    o.ncm = np.concatenate([o.ncm, ncm_n])
    o.RA = np.concatenate([o.RA, pred])
    o.features = np.vstack([o.features, fn])
    rows = [np.concatenate([o.I_idx[o.I_ptr[i]:o.I_ptr[i + 1]], idx_n[ptr_n[i]:ptr_n[i + 1]] + n0]) for i in range(nx)]
    o.I_ptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    o.I_idx = np.concatenate(rows).astype(np.int64)

    def other(i, Ii):
        f = o.IJs[Ii]
        return np.where(f[:, 0] == i, f[:, 1], f[:, 0])

    todo = []
    for i in range(nx):
        Ii = rows[i]
        fi = other(i, Ii)
        lm = y[fi] != y[i]
        order = np.argsort(o.RA[Ii][lm], kind="stable")[:first]
        sel = Ii[lm][order]
        todo.append(sel[o.ncm[sel]])
    todo = np.concatenate(todo)
    if todo.shape[0] > 0:
        o.RA[todo] = o.metric_pairs(o.IJs[todo])
        o.ncm[todo] = False
    ngi = np.zeros((nx, nn), dtype=np.int64)
    ngd = np.zeros((nx, nn))
    for i in range(nx):
        Ii = rows[i]
        fi = other(i, Ii)
        d = o.RA[Ii].copy()
        mx = d.max()
        d[o.ncm[Ii]] += mx
        d[y[fi] == y[i]] += mx
        pick = np.argsort(d, kind="stable")[:nn]
        ngd[i] = o.RA[Ii[pick]]
        ngi[i] = fi[pick]
    o.nearest_enemy_graph = (ngi, ngd)
    return ngi, ngd


def _cover_count(ngd_rows, alpha_dne):
    """ebuffer: how many leading entries of each sorted row lie closer than the (scaled)
    nearest-enemy distance, annchor.py:826-831."""
    return np.array([np.searchsorted(r, t - 1e-6) for r, t in zip(ngd_rows, alpha_dne)], dtype=np.int64)


def selective_subset(fitted, y, dne=None, alpha=0):
    """Annchor.annchor_selective_subset, annchor.py:784-901: greedy cover on the k-NN graph
    (a point is covered when a subset member is among its neighbours closer than its nearest
    enemy), then pruning against the full candidate rows (uncomputed pairs at their upper
    bound)."""
    o = fitted
    nx = o.nx
    if dne is None:
        if not hasattr(o, "nearest_enemy_graph"):
            nearest_enemies(o, y)
        dne = o.nearest_enemy_graph[1][:, 0]
    dne = np.asarray(dne, dtype=np.float64)
    if np.any(dne == 0):
        raise Exception("Error: The following indices are distance zero from a point  with a different label:\n"
                        + "".join("\t %d\n" % i for i in np.nonzero(dne == 0)[0]))
    adne = dne / (1 + alpha)
    ngi, ngd = o.neighbor_graph
    eb = _cover_count(ngd, adne)
    rss = [int(i) for i in np.nonzero(eb == 1)[0]]
    in_rss = np.zeros(nx, dtype=bool)
    in_rss[rss] = True

    def covered(i):  # first subset member in row i's neighbour list sits inside its buffer
        hit = np.nonzero(in_rss[ngi[i]])[0]
        return hit.shape[0] > 0 and hit[0] < eb[i]

    done = np.array([covered(i) for i in range(nx)])
    while not done.all():
        votes = np.zeros(nx, dtype=np.int64)
        for i in np.nonzero(~done)[0]:
            np.add.at(votes, ngi[i][:eb[i]], 1)
        nxt = int(np.argmax(votes))  # most frequent, smallest index on ties (np.unique order)
        rss.append(nxt)
        for i in np.nonzero(~done)[0]:
            hit = np.nonzero(ngi[i] == nxt)[0]
            if hit.shape[0] > 0 and hit[0] < eb[i]:
                done[i] = True
    rss = np.array(rss, dtype=np.int64)
    # pruning phase (annchor.py:869-901)
    dists = o.RA.copy()
    dists[o.ncm] = o.features[o.ncm, 1]
    member = np.zeros((nx, rss.shape[0]), dtype=bool)
    pos_in_rss = -np.ones(nx, dtype=np.int64)
    pos_in_rss[rss] = np.arange(rss.shape[0])
    for i in range(nx):
        Ii = o.I_idx[o.I_ptr[i]:o.I_ptr[i + 1]]
        order = np.argsort(dists[Ii], kind="stable")
        f = o.IJs[Ii[order]]
        nbr = np.concatenate([[i], f.sum(axis=1) - i])
        nd = np.concatenate([[0.0], dists[Ii][order]])
        buf = nbr[: np.searchsorted(nd, adne[i] - 1e-6)]
        p = pos_in_rss[buf]
        member[i, p[p >= 0]] = True
    cover = member.sum(axis=1)
    keep = np.ones(rss.shape[0], dtype=bool)
    for r in range(rss.shape[0]):
        if np.min(cover - member[:, r]) != 0:
            cover = cover - member[:, r]
            keep[r] = False
    return rss[keep]


def alpha_rss(fitted, y, dne=None, alpha=0):
    """Annchor.alpha_rss, annchor.py:903-927: scan points by increasing nearest-enemy
    distance; a point joins when no member is closer than its (scaled) nearest enemy."""
    o = fitted
    if dne is None:
        if not hasattr(o, "nearest_enemy_graph"):
            nearest_enemies(o, y)
        dne = o.nearest_enemy_graph[1][:, 0]
    dne = np.asarray(dne, dtype=np.float64)
    order = np.argsort(dne, kind="stable")
    rss = [int(order[0])]
    adne = dne / (1 + alpha)
    for i in order:
        ds = o.metric_pairs(np.array([[i, r] for r in rss], dtype=np.int64))
        m = ds.min()
        if m > adne[i] or np.isclose(m, adne[i]):
            rss.append(int(i))
    return np.array(rss, dtype=np.int64)
