"""CPU simulation of the streamed form's tile ranking (design probe, not product): which rank key
finds the true neighbours' column tiles within a budget of p_work * #tiles per row tile?"""
import sys, time
import numpy as np, torch
torch.set_num_threads(8)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
d, na, k, T = 128, 32, 15, 128
rng = np.random.default_rng(1234)
Z = rng.standard_normal((N, 8)); W = rng.standard_normal((8, d))
X = (Z @ W + 0.05 * rng.standard_normal((N, d))).astype(np.float32)
Xt = torch.from_numpy(X); r2 = (Xt * Xt).sum(1)
def dist_to(v):
    return torch.sqrt(torch.clamp(r2 + (v * v).sum() - 2 * (Xt @ v), min=0))
# max-min anchors (reference quirk irrelevant here)
np.random.seed(42); ix = np.random.randint(N)
D = torch.empty((na, N)); run = None
for a in range(na):
    D[a] = dist_to(Xt[ix])
    run = D[a].clone() if a <= 1 else torch.minimum(run, D[a])
    ix = int(torch.argmax(D[0] if a == 0 else run))
D = D.numpy()
# k-d split in anchor space on the coordinate of largest variance
def kd(order, coords):
    if len(order) <= T: return [order]
    sub = coords[:, order]
    a = int(np.argmax(sub.var(axis=1)))
    o = order[np.argsort(sub[a], kind="stable")]
    h = len(o) // 2
    return kd(o[:h], coords) + kd(o[h:], coords)
t0 = time.time()
tiles = kd(np.arange(N), D)
nt = len(tiles); order = np.concatenate(tiles); tile_of = np.empty(N, np.int64)
for t, m in enumerate(tiles): tile_of[m] = t
print("tiles", nt, "%.1fs" % (time.time() - t0))
# truth
t0 = time.time(); B = 2048; truth = np.empty((N, k), np.int64)
for b in range(0, N, B):
    d2 = r2[None, :] + r2[b:b + B, None] - 2 * (Xt[b:b + B] @ Xt.T)
    d2[torch.arange(min(B, N - b)), torch.arange(b, min(b + B, N))] = float("inf")
    truth[b:b + B] = torch.topk(d2, k, dim=1, largest=False).indices.numpy()
print("truth %.1fs" % (time.time() - t0))
cnt = np.zeros((nt, nt), np.int64)
np.add.at(cnt, (np.repeat(tile_of, k), tile_of[truth.ravel()]), 1)
total = N * k
def recall(key, frac):   # key [nt, nt], smaller = better
    m = max(1, int(np.ceil(frac * nt)))
    sel = np.argpartition(key, m - 1, axis=1)[:, :m]
    return np.take_along_axis(cnt, sel, axis=1).sum() / total
Dm = np.stack([D[:, m].mean(1) for m in tiles])            # mean anchor vectors [nt, na]
lo = np.stack([D[:, m].min(1) for m in tiles]); hi = np.stack([D[:, m].max(1) for m in tiles])
C = np.stack([X[m].mean(0) for m in tiles])                # true-space centroids
keys = {}
keys["oracle(best possible)"] = -cnt.astype(np.float64)
keys["mean-anchor-vector L2 (current)"] = ((Dm[:, None, :] - Dm[None, :, :]) ** 2).sum(-1)
keys["centroid L2 (true space)"] = ((C[:, None, :] - C[None, :, :]) ** 2).sum(-1)
keys["mean-anchor Linf"] = np.abs(Dm[:, None, :] - Dm[None, :, :]).max(-1)
# min over rows of the row tile to the column centroid (true space)
Ct = torch.from_numpy(C); c2 = (Ct * Ct).sum(1)
dRC = (r2[:, None] + c2[None, :] - 2 * Xt @ Ct.T).numpy()      # [N, nt]
mn = np.stack([dRC[m].min(0) for m in tiles]); keys["min_i |x_i - c_J|"] = mn
q25 = np.stack([np.partition(dRC[m], len(m) // 4, axis=0)[len(m) // 4] for m in tiles]); keys["q25_i |x_i - c_J|"] = q25
# symmetric: min over both
keys["min(min_i|x_i-c_J|, min_j|x_j-c_I|)"] = np.minimum(mn, mn.T)
# point-to-box lower bound in anchor space, min over rows
lbmin = np.empty((nt, nt)); lbq = np.empty((nt, nt))
for I, m in enumerate(tiles):
    Di = D[:, m].T                                      # [128, na]
    lb = np.maximum(lo[None, :, :] - Di[:, None, :], Di[:, None, :] - hi[None, :, :]).max(-1).clip(min=0)   # [128, nt]
    lbmin[I] = lb.min(0); lbq[I] = np.partition(lb, len(m) // 4, axis=0)[len(m) // 4]
keys["min_i lb(i, box_J)"] = lbmin + 1e-6 * keys["mean-anchor-vector L2 (current)"]
keys["q25_i lb(i, box_J)"] = lbq + 1e-6 * keys["mean-anchor-vector L2 (current)"]
for frac in (0.02, 0.05, 0.1, 0.2):
    print("budget %.2f:" % frac, "  ".join("%s=%.4f" % (n, recall(kk, frac)) for n, kk in keys.items()))
np.savez("/tmp/tile_sim_%d.npz" % N, cnt=cnt)
