"""How much of the Levenshtein DP would a band of half-width ub (the anchors' upper bound, a guaranteed d <= ub) skip, over the
pairs a C2 fit evaluates?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from annchor_amd import Annchor
from annchor_amd.datasets import load_strings
X = load_strings()["X"]
ann = Annchor(X, "levenshtein", n_anchors=15, n_neighbors=25, p_work=0.12, random_seed=42, niters=2)
ann.get_anchors(); ann.get_locality(); ann.get_features()
F0 = ann.features.copy()          # bounds before any refinement: what a band could rely on at the time
ann2 = Annchor(X, "levenshtein", n_anchors=15, n_neighbors=25, p_work=0.12, random_seed=42, niters=2).fit()
ev = (~ann2.not_computed_mask) & (ann2.features[:, 3] == 0)
IJ = ann2.IJs[ev]
d = ann2.RefineApprox[ev]
ub = F0[ev, 1]
lens = np.array([len(s) for s in X])
la, lb = lens[IJ[:, 0]], lens[IJ[:, 1]]
m, n = np.minimum(la, lb), np.maximum(la, lb)
full = ((m + 31) // 32) * n
band_bits = np.minimum(2 * ub + 1 + 32, m)        # + a word of slack for alignment
banded = ((band_bits + 31) // 32) * n
print("evaluated non-anchor pairs:", ev.sum())
print("d quantiles:", np.quantile(d, [0.1, 0.5, 0.9]), " ub quantiles:", np.quantile(ub, [0.1, 0.5, 0.9]), " min len quantiles:", np.quantile(m, [0.1, 0.5, 0.9]))
print("word-steps full %.3e banded(ub) %.3e ratio %.3f" % (full.sum(), banded.sum(), banded.sum() / full.sum()))
band2 = np.minimum(2 * d + 1 + 32, m); b2 = ((band2 + 31) // 32) * n
print("with a perfect bound (d itself): ratio %.3f" % (b2.sum() / full.sum()))
