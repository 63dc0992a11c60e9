"""One C4 (digits, exact-OT Wasserstein) fit, for rocprofv3 passes (tools/pmc_emd.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from annchor_amd import Annchor
from annchor_amd.datasets import load_digits
d = load_digits()
a = Annchor(d["X"], "wasserstein", func_kwargs={"cost_matrix": d["cost_matrix"]}, n_anchors=20, n_neighbors=25, n_samples=5000,
            p_work=0.16, random_seed=42).fit()
print("evals", a.evals)
