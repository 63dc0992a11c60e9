import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import annchor_oracle as O
from oracle import metrics as om
from annchor_amd import Annchor, _native
G = np.load("tests/golden/query_digits.npz")
d = om.load_digits(); X, M = d["X"], d["cost_matrix"]
tr, te = G["idx_train"], G["idx_test"]
ann = Annchor(X[tr], "wasserstein", func_kwargs={"cost_matrix": M}, n_anchors=25, n_neighbors=25, n_samples=5000, p_work=0.16).fit()
gi, gd = ann.query(X[te], nn=15, p_work=0.2)
H = om.Histograms(np.concatenate([X[tr], X[te]]), M)
qp = lambda IJ: H.pairs(np.stack([IJ[:, 0], IJ[:, 1] + len(tr)], axis=1))
class F: pass
o = F(); o.nx = len(tr); o.n_anchors = 25; o.A = ann.A; o.D = ann.D; o.locality = 5; o.loc_thresh = 1
o.bins, o.W, o.c = ann.regression.coefficients()
o.errs = [np.asarray(ann.error_predictor.errs[l]) for l in ann.error_predictor.labels]
oi, od, info = O.query(o, qp, len(te), nn=15, p_work=0.2, apply_floor=True)
print("gpu vs oracle diff", O.compare_neighbor_graphs((oi, od), (gi, gd), 15), "evals", ann.query_evals, info["evals"])
print("gpu vs ref", O.compare_neighbor_graphs((G["q_e2e_idx"], G["q_e2e_dist"]), (gi, gd), 15), "oracle vs ref", O.compare_neighbor_graphs((G["q_e2e_idx"], G["q_e2e_dist"]), (oi, od), 15))
