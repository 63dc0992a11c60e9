#!/usr/bin/env python3
"""
tests/golden/make_golden.py -- generates the golden vectors under tests/golden/.

Runs ONLY in the build container, where the read-only reference lives at
/root/reference.  It imports the reference's own Python package and drives its
own `Annchor` stage methods in the order `Annchor.fit()` does
(annchor/annchor.py:532-623), snapshotting the object's state between stages.
Nothing from the reference is copied: the outputs are data (inputs + expected
outputs).

The reference needs three third-party modules that are not installed here
(numba, Levenshtein, pynndescent).  tests/golden/_refshim/ provides:
  * `numba`: decorators become identity wrappers -- the reference's own NumPy
    code then runs unchanged as plain Python (no arithmetic replaced);
  * `Levenshtein.distance`, `pynndescent.distances.kantorovich`: delegate to the
    oracle's C restatements, which are pinned independently (reference test
    known-answers; the reference's stored exact-EMD graph).
So: vectors for the reference's OWN functions (a0, a6-a17) are genuine reference
outputs; metric values inside them come from the oracle's metric restatements.

Usage:  python tests/golden/make_golden.py [small|blobs|strings_full|digits_full|enemies|query|graph_sp|all]
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(HERE, "_refshim"), "/root/reference", ROOT]

import numpy as np  # noqa: E402

import annchor as ref  # noqa: E402  (the reference)
from annchor.samplers import NothingToSample  # noqa: E402
from oracle import metrics as om  # noqa: E402


def I_to_csr(I, nx):
    ptr = np.zeros(nx + 1, dtype=np.int64)
    for i in range(nx):
        ptr[i + 1] = ptr[i] + len(I[i])
    idx = np.concatenate([np.asarray(I[i], dtype=np.int64) for i in range(nx)])
    return ptr, idx


def staged_fit(ann, snaps):
    """Annchor.fit() body (annchor.py:545-623) with snapshots between stages."""
    ann.get_anchors()
    snaps["A"], snaps["D"] = np.asarray(ann.A, dtype=np.int64), np.ascontiguousarray(ann.D)
    ann.get_locality()
    snaps["sid"], snaps["IJs"] = ann.sid.copy(), ann.IJs.copy()
    snaps["I_ptr"], snaps["I_idx"] = I_to_csr(ann.I, ann.nx)
    ann.get_features()
    snaps["features0"], snaps["ncm0"] = ann.features.copy(), ann.not_computed_mask.copy()
    niters = ann.niters
    for it in range(niters):
        p = "it%d_" % it
        try:
            ann.get_sample()
        except NothingToSample:
            break
        snaps[p + "sample_ixs"], snaps[p + "bins"] = ann.sample_ixs.copy(), ann.sample_bins.copy()
        snaps[p + "sample_y"] = ann.sample_y.copy()
        snaps[p + "ncm_after_sample"] = ann.not_computed_mask.copy()
        ann.fit_predict_regression()
        snaps[p + "coef"] = np.array([lr.coef_ for lr in ann.regression.LRs])
        snaps[p + "intercept"] = np.array([lr.intercept_ for lr in ann.regression.LRs])
        snaps[p + "sample_predict"] = ann.sample_predict.copy()
        snaps[p + "RA_after_regression"] = ann.RefineApprox.copy()
        ann.fit_predict_errors()
        for b, e in ann.error_predictor.errs.items():
            snaps[p + "errs%d" % b] = e.copy()
        snaps[p + "labels"] = ann.errors.astype(np.int8)
        ncm_before = ann.not_computed_mask.copy()
        ann.select_refine_candidate_pairs(w=1 / niters, it=it)
        snaps[p + "thresh"] = ann.thresh.copy()
        unc = np.arange(ncm_before.shape[0])[ncm_before]
        snaps[p + "mapback"] = np.sort(unc[ann.candidates])
        snaps[p + "nextback"] = np.sort(ann.nextback)
        snaps[p + "RA_after_refine"] = ann.RefineApprox.copy()
        snaps[p + "ncm_after_refine"] = ann.not_computed_mask.copy()
        if it < niters - 1:
            ann.update_anchor_points(timeout=1e9)
            snaps[p + "lbub_after_update"] = ann.features[:, :2].copy()
    ann.get_ann()
    snaps["ng_idx"], snaps["ng_dist"] = ann.neighbor_graph[0].copy(), ann.neighbor_graph[1].copy()
    snaps["evals"] = np.int64(ann.evals)
    snaps["n_samples_final"] = np.int64(ann.n_samples)
    snaps["p_work"] = np.float64(ann.p_work)
    snaps["na_budget"] = np.int64(ann.na)
    return snaps


def save(name, snaps):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **snaps)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


def strings_evaluator(P):
    return lambda f, X, IJ: P.pairs(np.asarray(IJ, dtype=np.int64))


def gen_small():
    X, _ = om.load_strings()
    # (a) strings, dense locality, integer metric, 2 iterations
    Xs = X[::5]  # 320 strings covering all 8 clusters (a prefix gives empty sampler bins)
    n = len(Xs)
    P = om.PackedStrings(Xs)
    cfg = dict(n_anchors=8, n_neighbors=10, n_samples=700, p_work=0.3, random_seed=42, niters=2)
    ann = ref.Annchor(np.array(Xs), "levenshtein", get_exact_ijs=strings_evaluator(P), **cfg)
    snaps = staged_fit(ann, {"cfg_" + k: np.float64(v) for k, v in cfg.items()})
    snaps["n"] = np.int64(n)
    save("strings_small", snaps)

    # (b) Euclidean float64, clustered => sparse locality, loc_min widening, 3 iterations
    rng = np.random.default_rng(7)
    cent = rng.uniform(-10, 10, (12, 3))
    Xe = (cent[rng.integers(0, 12, 260)] + rng.standard_normal((260, 3))).astype(np.float64)
    cfg = dict(n_anchors=12, n_neighbors=8, n_samples=400, p_work=0.25, random_seed=3, niters=3,
               locality=3)
    ev = lambda f, X, IJ: om.euclidean_pairs(X, np.asarray(IJ, dtype=np.int64))  # noqa: E731
    ann = ref.Annchor(Xe, "euclidean", get_exact_ijs=ev, **cfg)
    snaps = staged_fit(ann, {"cfg_" + k: np.float64(v) for k, v in cfg.items()})
    snaps["X"] = Xe
    save("euclid_small", snaps)


def gen_blobs():
    """reference tests/test_examples.py:88-230: pinned anchor vector and error counts."""
    from sklearn.datasets import make_blobs

    X, _, centers = make_blobs(centers=10, n_samples=1000, random_state=42, return_centers=True)
    ev = lambda f, X, IJ: om.euclidean_pairs(X, np.asarray(IJ, dtype=np.int64))  # noqa: E731
    bf = ref.BruteForce(X, "euclidean", get_exact_ijs=ev)
    bf.fit()
    ann = ref.Annchor(X, "euclidean", n_anchors=10, p_work=0.05, get_exact_ijs=ev)
    ann.fit()
    err = ref.compare_neighbor_graphs(bf.neighbor_graph, ann.neighbor_graph, 15)
    pinned = np.array([102, 674, 347, 586, 214, 963, 365, 348, 430, 429])
    assert np.array_equal(ann.A, pinned), ann.A
    print("blobs: A matches pinned vector, errors =", err)
    save("blobs", dict(X=X, centers=centers, A=np.asarray(ann.A, dtype=np.int64), D=np.ascontiguousarray(ann.D),
                       errors=np.int64(err), evals=np.int64(ann.evals),
                       bf_dist=bf.neighbor_graph[1][:, :16].copy(),
                       ng_dist=ann.neighbor_graph[1].copy()))


def gen_strings_full():
    X, _ = om.load_strings()
    P = om.PackedStrings(X)
    t = time.time()
    dense = P.all_pairs()
    print("brute force %.1fs" % (time.time() - t))
    idx = np.argsort(dense, axis=1, kind="stable")[:, :100]
    truth_d = np.take_along_axis(dense, idx, axis=1)
    out = dict(truth_idx=idx.astype(np.int16), truth_dist=truth_d.astype(np.int16))
    assert dense[10, 165] == 299  # reference tests/test_datasets.py:234-235
    for tag, cfg in {
        "c1": dict(n_anchors=15, n_neighbors=25, p_work=0.12, random_seed=42),          # BASELINE cfg 1/2
        "readme": dict(n_anchors=20, n_neighbors=25, p_work=0.12, random_seed=42),      # README.md:102
        "test": dict(n_anchors=23, n_neighbors=15, p_work=0.12, random_seed=42, niters=4,
                     n_samples=5000),                                                   # tests/test_annchor.py:83-96
    }.items():
        t = time.time()
        ann = ref.Annchor(np.array(X), "levenshtein", get_exact_ijs=strings_evaluator(P), **cfg)
        ann.fit()
        k = cfg["n_neighbors"]
        err = ref.compare_neighbor_graphs((idx, truth_d), ann.neighbor_graph, k)
        print(tag, "evals", ann.evals, "pairs", ann.IJs.shape[0], "errors", err, "%.0fs" % (time.time() - t))
        out[tag + "_A"] = np.asarray(ann.A, dtype=np.int64)
        out[tag + "_D"] = ann.D.astype(np.int16)
        out[tag + "_evals"] = np.int64(ann.evals)
        out[tag + "_npairs"] = np.int64(ann.IJs.shape[0])
        out[tag + "_errors"] = np.int64(err)
        out[tag + "_ng_dist"] = ann.neighbor_graph[1].astype(np.int16)
    save("strings_full", out)


def gen_digits_full():
    d = om.load_digits()
    H = om.Histograms(d["X"], d["cost_matrix"])
    ev = lambda f, X, IJ: H.pairs(np.asarray(IJ, dtype=np.int64))  # noqa: E731
    out = {}
    for tag, cfg in {
        "c4": dict(n_anchors=20, n_neighbors=25, n_samples=5000, p_work=0.16, random_seed=42),
        "test": dict(n_anchors=25, n_neighbors=25, n_samples=5000, p_work=0.16, random_seed=42),
    }.items():
        t = time.time()
        ann = ref.Annchor(d["X"], "wasserstein", func_kwargs={"cost_matrix": d["cost_matrix"]},
                          get_exact_ijs=ev, **cfg)
        ann.fit()
        err = ref.compare_neighbor_graphs(d["neighbor_graph"], ann.neighbor_graph, 25)
        print(tag, "evals", ann.evals, "pairs", ann.IJs.shape[0], "errors", err, "%.0fs" % (time.time() - t))
        out[tag + "_A"] = np.asarray(ann.A, dtype=np.int64)
        out[tag + "_D"] = np.ascontiguousarray(ann.D)
        out[tag + "_evals"] = np.int64(ann.evals)
        out[tag + "_npairs"] = np.int64(ann.IJs.shape[0])
        out[tag + "_errors"] = np.int64(err)
        out[tag + "_ng_dist"] = ann.neighbor_graph[1].copy()
    save("digits_full", out)


def gen_enemies():
    """reference tests/test_examples.py:61-85 (annchor_selective_subset on blobs / moons; the
    pinned sizes there, 90 / 16, belong to numba's RNG stream -- under the pass-through numba
    stand-in the reference draws its samples from NumPy's legacy stream, so the sizes recorded
    here are the reference's own results for THAT stream) plus get_nearest_enemies and
    alpha_rss.  State is snapshotted after fit() so that the restatement can be checked from
    the same starting point."""
    from sklearn.datasets import make_blobs, make_moons

    np.random.seed(1)
    X, y = make_blobs(n_samples=1000, centers=5)
    U, v = make_moons(n_samples=1000, noise=0.1)
    U = np.fliplr(U)
    ev = lambda f, X, IJ: om.euclidean_pairs(X, np.asarray(IJ, dtype=np.int64))  # noqa: E731
    out = {}
    for tag, (Z, lab) in (("blobs", (X, y)), ("moons", (np.ascontiguousarray(U), v))):
        ann = ref.Annchor(Z, "euclidean", n_neighbors=15, p_work=0.2, get_exact_ijs=ev)
        ann.fit()
        out[tag + "_X"], out[tag + "_y"] = Z, np.asarray(lab, dtype=np.int64)
        out[tag + "_ng_idx"], out[tag + "_ng_dist"] = ann.neighbor_graph[0].copy(), ann.neighbor_graph[1].copy()
        out[tag + "_n_pairs_fit"] = np.int64(len(ann.IJs))
        ann.get_nearest_enemies(lab)
        out[tag + "_ne_idx"], out[tag + "_ne_dist"] = (ann.nearest_enemy_graph[0].copy(),
                                                       ann.nearest_enemy_graph[1].copy())
        out[tag + "_n_pairs_ne"] = np.int64(len(ann.IJs))
        out[tag + "_n_computed_ne"] = np.int64((~ann.not_computed_mask).sum())
        for alpha in (0, 0.1):
            ss = ann.annchor_selective_subset(y=lab, alpha=alpha)
            out[tag + "_ss_a%g" % alpha] = np.asarray(ss, dtype=np.int64)
            print(tag, "alpha", alpha, "selective subset size", len(ss))
        rss = ann.alpha_rss(lab, alpha=0)
        out[tag + "_alpha_rss"] = np.asarray(rss, dtype=np.int64)
        print(tag, "alpha_rss size", len(rss))
    save("enemies", out)


def gen_enemies_state():
    """State-pinned vectors for get_nearest_enemies / annchor_selective_subset / alpha_rss:
    the reference's complete post-fit state on a 400-point set (so a restatement can start
    from exactly the same state; under the slow numba stand-in the reference's
    update_anchor_points stops at its 10 s cutoff, so end-to-end states are not comparable)
    and the reference's outputs from there."""
    from sklearn.datasets import make_moons

    np.random.seed(3)
    U, v = make_moons(n_samples=400, noise=0.15)
    U = np.ascontiguousarray(np.fliplr(U))
    ev = lambda f, X, IJ: om.euclidean_pairs(X, np.asarray(IJ, dtype=np.int64))  # noqa: E731
    ann = ref.Annchor(U, "euclidean", n_anchors=20, n_neighbors=6, n_samples=1000, p_work=0.2, locality=3,
                      get_exact_ijs=ev)
    ann.fit()
    nx = ann.nx
    ptr, idx = I_to_csr(ann.I, nx)
    out = dict(X=U, y=np.asarray(v, dtype=np.int64), D=np.ascontiguousarray(ann.D), A=np.asarray(ann.A, dtype=np.int64),
               sid=np.asarray(ann.sid, dtype=np.int64), IJs=ann.IJs.copy(), I_ptr=ptr, I_idx=idx,
               features=ann.features.copy(), ncm=ann.not_computed_mask.copy(), RA=ann.RefineApprox.copy(),
               bins=ann.regression.sample_bins.copy(),
               W=np.array([lr.coef_ for lr in ann.regression.LRs]), c=np.array([lr.intercept_ for lr in ann.regression.LRs]),
               ng_idx=ann.neighbor_graph[0].copy(), ng_dist=ann.neighbor_graph[1].copy(),
               loc_thresh=np.int64(ann.loc_thresh))
    ann.get_nearest_enemies(v, nn=3, loc_min=80)
    ptr2, idx2 = I_to_csr(ann.I, nx)
    out.update(ne_idx=ann.nearest_enemy_graph[0].copy(), ne_dist=ann.nearest_enemy_graph[1].copy(),
               ne_IJs_new=ann.IJs[len(out["IJs"]):].copy(), ne_ncm=ann.not_computed_mask.copy(),
               ne_RA=ann.RefineApprox.copy(), ne_I_ptr=ptr2, ne_I_idx=idx2)
    for alpha in (0, 0.2):
        out["ss_a%g" % alpha] = np.asarray(ann.annchor_selective_subset(y=v, alpha=alpha), dtype=np.int64)
        print("alpha", alpha, "selective subset size", len(out["ss_a%g" % alpha]))
    out["alpha_rss"] = np.asarray(ann.alpha_rss(v, alpha=0), dtype=np.int64)
    out["alpha_rss_a0.2"] = np.asarray(ann.alpha_rss(v, alpha=0.2), dtype=np.int64)
    print("alpha_rss", len(out["alpha_rss"]), len(out["alpha_rss_a0.2"]), "new pairs", len(out["ne_IJs_new"]))
    save("enemies_state", out)


def staged_query(ann, Q, nn, p_work, ev_q, snaps, tag, slim=False):
    """query_ body (query_functions.py:183-212) stage by stage with the reference's own helpers,
    snapshotting between stages; `Annchor.query` itself (annchor.py:643-683) is then run end to
    end on the same inputs and must give the same result (both outputs are stored)."""
    import annchor.query_functions as qf
    from annchor.utils import get_nn

    nq = len(Q)
    # the p_work floor of Annchor.query (annchor.py:668-675)
    na_q, nbf = ann.n_anchors * nq, nq * ann.nx
    limit = ((nq * nn * 3) // 2 - 1 + na_q) / nbf
    p_eff = max(p_work, limit)
    ann.get_exact_query_ijs = ev_q
    QD = qf.get_query_anchor_dists(ann, Q)
    check = qf.get_query_locality(ann, QD)
    IJs, QI, Qf, Qncm = qf.get_query_features(ann, QD, check)
    Qpred = ann.regression.predict(Qf, ann.feature_names)
    Qclip = np.clip(Qpred, Qf[:, 0], Qf[:, 1])
    Qerr = ann.error_predictor.predict(Qf, ann.feature_names[:-1])
    ncm0 = Qncm.copy()
    thresh = np.array([np.partition(Qclip[QI[i]], nn)[nn] for i in range(nq)])
    QRA, Qncm2 = qf.select_refine_candidate_query_pairs(ann, IJs, Q, QI, Qclip.copy(), Qncm.copy(), Qerr, p_eff, nn)
    ngi, ngd = get_nn(nq, nn + 1, QRA, IJs, QI, Qncm2)
    P = tag + "_"
    snaps.update({
        P + "QD": QD, P + "IJs": IJs.astype(np.int64), P + "features": Qf, P + "ncm0": ncm0,
        P + "pred": Qpred, P + "labels": np.asarray(Qerr).astype(np.int8), P + "thresh": thresh,
        P + "mapback": np.sort(np.arange(len(ncm0))[ncm0 & ~Qncm2]), P + "RA_after": QRA, P + "ncm_after": Qncm2,
        P + "ngi_raw": ngi, P + "ngd_raw": ngd, P + "nn": np.int64(nn), P + "p_work": np.float64(p_work),
        P + "p_work_effective": np.float64(p_eff),
    })
    if slim:   # large case: per-pair arrays kept for a fixed random sample of pairs only
        rows = np.sort(np.random.default_rng(0).choice(len(ncm0), 20000, replace=False))
        snaps[P + "rows"] = rows
        for key in ("features", "pred", "labels"):
            snaps[P + key] = snaps[P + key][rows]
        snaps[P + "RA_after_at_mapback"] = QRA[snaps[P + "mapback"]]
        snaps[P + "IJs"] = snaps[P + "IJs"].astype(np.int16)
        snaps[P + "mapback"] = snaps[P + "mapback"].astype(np.int32)
        for key in ("ncm0", "RA_after", "ncm_after"):
            del snaps[P + key]
    e2e = ann.query(Q, nn=nn, p_work=p_work, get_exact_query_ijs=ev_q)
    snaps[P + "e2e_idx"], snaps[P + "e2e_dist"] = e2e[0].copy(), e2e[1].copy()
    return snaps


def fitted_state(ann, snaps):
    """What query() reads from a fitted Annchor (annchor.py:643-683, query_functions.py)."""
    snaps.update(A=np.asarray(ann.A, dtype=np.int64), D=np.ascontiguousarray(ann.D), sid=np.asarray(ann.sid, dtype=np.int64),
                 bins=ann.regression.sample_bins.copy(),
                 W=np.array([lr.coef_ for lr in ann.regression.LRs]), c=np.array([lr.intercept_ for lr in ann.regression.LRs]),
                 err_bins=np.asarray(ann.error_predictor.partition_bins, dtype=np.float64),
                 locality=np.int64(ann.locality), loc_thresh=np.int64(ann.loc_thresh), nx=np.int64(ann.nx))
    for b, e in ann.error_predictor.errs.items():
        snaps["errs%d" % b] = e.copy()
    return snaps


def gen_query():
    """Annchor.query (annchor.py:643-683 -> query_functions.py:10-212; reference test
    tests/test_examples.py:12-58) driven on (a) a strings split and (b) the digits split of the
    reference's own test (train_test_split(random_state=0), n_anchors=25, k=25, n_samples=5000,
    p_work=0.16; query p_work=0.2)."""
    # (a) strings: 400 strings, every 5th held out as a query
    X, _ = om.load_strings()
    sub = X[::4]
    tr = [s for t, s in enumerate(sub) if t % 5]
    qs = [s for t, s in enumerate(sub) if t % 5 == 0]
    P = om.PackedStrings(tr + qs)
    ntr = len(tr)
    cfg = dict(n_anchors=8, n_neighbors=10, n_samples=700, p_work=0.3, random_seed=42, niters=2)
    ann = ref.Annchor(np.array(tr), "levenshtein", get_exact_ijs=strings_evaluator(P), **cfg)
    ann.fit()
    A = np.asarray(ann.A, dtype=np.int64)

    def ev_q(f, Xa, Z, IJ):   # pairs index (Xa[i], Z[j]); Xa is X[A] for the anchor call, X otherwise
        IJ = np.asarray(IJ, dtype=np.int64)
        i = A[IJ[:, 0]] if len(Xa) == len(A) else IJ[:, 0]
        return P.pairs(np.stack([i, ntr + IJ[:, 1]], axis=1))

    snaps = fitted_state(ann, {"cfg_" + k: np.float64(v) for k, v in cfg.items()})
    snaps["n_train"], snaps["n_query"] = np.int64(ntr), np.int64(len(qs))
    staged_query(ann, np.array(qs), 10, 0.3, ev_q, snaps, "q")
    staged_query(ann, np.array(qs[:7]), 5, 0.01, ev_q, snaps, "qlow")   # p_work below the floor: raised (annchor.py:668-675)
    save("query_strings", snaps)

    # (b) digits, the reference test's own split
    from sklearn.model_selection import train_test_split

    d = om.load_digits()
    Xd, yd, M = d["X"], d["y"], d["cost_matrix"]
    idx_tr, idx_te = train_test_split(np.arange(len(Xd)), random_state=0)
    X_train, X_test = train_test_split(Xd, random_state=0)
    assert np.array_equal(X_train, Xd[idx_tr]) and np.array_equal(X_test, Xd[idx_te])
    H = om.Histograms(np.concatenate([X_train, X_test]), M)
    ntr = len(X_train)
    ev = lambda f, X, IJ: H.pairs(np.asarray(IJ, dtype=np.int64))  # noqa: E731
    t = time.time()
    ann = ref.Annchor(X_train, "wasserstein", func_kwargs={"cost_matrix": M}, n_anchors=25, n_neighbors=25,
                      n_samples=5000, p_work=0.16, get_exact_ijs=ev)
    ann.fit()
    A = np.asarray(ann.A, dtype=np.int64)
    print("digits train fit %.0fs" % (time.time() - t))
    snaps = fitted_state(ann, dict(idx_train=idx_tr.astype(np.int64), idx_test=idx_te.astype(np.int64)))
    staged_query(ann, X_test, 15, 0.2, ev_q_factory(H, A, ntr), snaps, "q", slim=True)
    # truth for the reference test's own recall criterion (exact 15-NN of every query)
    IJall = np.stack([np.repeat(np.arange(ntr), len(X_test)), ntr + np.tile(np.arange(len(X_test)), ntr)], axis=1)
    dd = H.pairs(IJall).reshape(ntr, len(X_test)).T
    order = np.argsort(dd, axis=1, kind="stable")[:, :15]
    snaps["truth_idx"], snaps["truth_dist"] = order.astype(np.int64), np.take_along_axis(dd, order, axis=1)
    got = snaps["q_e2e_idx"]
    errs = sum(len(np.setdiff1d(order[i], got[i])) for i in range(len(X_test)))
    snaps["ref_recall"] = np.float64(1 - errs / (15.0 * len(X_test)))
    print("digits query: reference recall@15 = %.5f  (%.0fs)" % (snaps["ref_recall"], time.time() - t))
    save("query_digits", snaps)


def ev_q_factory(H, A, ntr):
    def ev_q(f, Xa, Z, IJ):
        IJ = np.asarray(IJ, dtype=np.int64)
        i = A[IJ[:, 0]] if len(Xa) == len(A) else IJ[:, 0]
        return H.pairs(np.stack([i, ntr + IJ[:, 1]], axis=1))
    return ev_q


def gen_graph_sp():
    """The reference's shortest-path example (tests/test_annchor.py:105-145): X = node ids, the metric an
    arbitrary Python callable.  The fixture holds the reference's data files' arrays (edge list, node sample,
    its stored exact 15-NN graph) and the reference's fit on them; the all-pairs table the callable reads is
    checked against networkx's Dijkstra on the reference test's three known answers."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import dijkstra
    import networkx as nkx
    from annchor.datasets import load_graph_sp

    data = load_graph_sp()
    X, ng, G = data["X"], data["neighbor_graph"], data["G"]
    g = np.load(os.path.join(os.path.dirname(ref.__file__), "data", "graph.npz"))
    edges, weights = g["edges"], g["weights"]
    n = int(edges.max()) + 1
    W = coo_matrix((weights, (edges[:, 0], edges[:, 1])), shape=(n, n)).tocsr()
    SP = dijkstra(W, directed=False)
    for i, j in ((0, 0), (2, 5), (300, 701), (10, 4)):
        assert np.isclose(SP[i, j], nkx.dijkstra_path_length(G, i, j, weight="w"), rtol=0, atol=1e-12), (i, j)

    def sp_dist(i, j):
        return SP[i, j]

    out = dict(edges=edges.astype(np.int32), weights=weights, X=X.astype(np.int32),
               ng_idx=ng[0][:, :16].astype(np.int32), ng_dist=ng[1][:, :16].copy())
    cfg = dict(n_anchors=20, n_neighbors=15, random_seed=42, n_samples=5000, p_work=0.15)
    t = time.time()
    ann = ref.Annchor(X, sp_dist, **cfg)
    ann.fit()
    err = ref.compare_neighbor_graphs(ng, ann.neighbor_graph, 15)
    print("graph_sp evals", ann.evals, "pairs", ann.IJs.shape[0], "errors", err, "%.0fs" % (time.time() - t))
    out["A"] = np.asarray(ann.A, dtype=np.int64)
    out["D"] = np.ascontiguousarray(ann.D)
    out["evals"] = np.int64(ann.evals)
    out["npairs"] = np.int64(ann.IJs.shape[0])
    out["errors"] = np.int64(err)
    out["fit_ng_idx"] = ann.neighbor_graph[0].astype(np.int32)
    out["fit_ng_dist"] = ann.neighbor_graph[1].copy()
    save("graph_sp", out)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("graph_sp", "all"):
        gen_graph_sp()
    if what in ("small", "all"):
        gen_small()
    if what in ("blobs", "all"):
        gen_blobs()
    if what in ("strings_full", "all"):
        gen_strings_full()
    if what in ("digits_full", "all"):
        gen_digits_full()
    if what in ("enemies", "all"):
        gen_enemies()
        gen_enemies_state()
    if what in ("query", "all"):
        gen_query()
