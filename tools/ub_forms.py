"""update_bounds forms against each other (bounds after update_anchor_points must be bit-identical) + their times:
usage ub_forms.py [N] [forms,...]   (N points of low-dimensional float64 data, candidate list thinned by the locality filter)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from annchor_amd import Annchor
from annchor_amd.samplers import DeviceStratifiedSampler
n = int(sys.argv[1]) if len(sys.argv) > 1 else 70000
forms = sys.argv[2].split(",") if len(sys.argv) > 2 else ["pairs", "bitmap", "bits16", "bits32"]
rng = np.random.default_rng(11)
cent = rng.standard_normal((40, 6)) * 4
X = np.round(cent[rng.integers(0, 40, n)] + rng.standard_normal((n, 6)), 2)
cfg = dict(n_anchors=40, n_neighbors=15, p_work=float(os.environ.get("UB_PWORK", "0.01")), n_samples=5000, locality=5, loc_thresh=3, random_seed=2)
ref = None
for form in forms:
    os.environ["ANNCHOR_UPDATE_BOUNDS"] = form
    ann = Annchor(X, "euclidean", sampler=DeviceStratifiedSampler(), **cfg)
    ann._engine.prof_enable(1)
    ann.get_anchors(); ann.get_locality(); ann.get_features(); ann.get_sample(); ann.fit_predict_regression(); ann.fit_predict_errors()
    ann.select_refine_candidate_pairs(w=0.5, it=0)
    ann.update_anchor_points()
    F = ann.features
    pr = ann._engine.prof_get()
    print("%-7s N=%d pairs=%d nnext=%d update_bounds %.3f ms (csr %.3f ms)" % (form, n, ann.n_pairs, len(ann.nextback),
          pr["update_bounds_intersect"]["ms"], pr["computed_neighbour_csr"]["ms"]), flush=True)
    cur = F[:, :2].copy()
    if ref is None:
        ref = cur
    else:
        same = np.array_equal(ref, cur)
        print("        identical to %s: %s" % (forms[0], same))
        if not same:
            bad = np.nonzero((ref != cur).any(axis=1))[0]
            print("        %d rows differ, first %s: %s vs %s" % (len(bad), bad[:5], ref[bad[:3]], cur[bad[:3]]))
    ann._engine.close()
