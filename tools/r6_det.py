import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from test_streamed_gpu import latent
from annchor_amd import _native
from annchor_amd.streamed import StreamedAnnchor, RcclComm
X = latent(6000, 128)
b1 = StreamedAnnchor(X, n_anchors=8, n_neighbors=10, p_work=0.5).fit()
b2 = StreamedAnnchor(X, n_anchors=8, n_neighbors=10, p_work=0.5).fit()
print("single vs single: idx", np.array_equal(b1.neighbor_graph[0], b2.neighbor_graph[0]), "dist", np.array_equal(b1.neighbor_graph[1], b2.neighbor_graph[1]), b1.tile_evals, b2.tile_evals, b1._engine.stream_last_kernel(with_guard=True))
a = StreamedAnnchor(X, n_anchors=8, n_neighbors=10, p_work=0.5, force_exchange=True).fit()
print("sharded vs single: idx", np.array_equal(a.neighbor_graph[0], b1.neighbor_graph[0]), "dist", np.array_equal(a.neighbor_graph[1], b1.neighbor_graph[1]), a.tile_evals)
d = np.abs(a.neighbor_graph[1] - b1.neighbor_graph[1]); print("max abs diff", d.max(), "cells differing", (d > 0).sum())
