"""CPU simulation (design probe): tile phase at a budget, then neighbour-of-neighbour join passes
(the streamed analogue of update_anchor_points: common computed neighbours), recall after each."""
import sys, time
import numpy as np, torch
torch.set_num_threads(8)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
fracs = [float(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0.05, 0.1]
d, na, k, T = 128, 32, 15, 128
rng = np.random.default_rng(1234)
Z = rng.standard_normal((N, 8)); W = rng.standard_normal((8, d))
X = (Z @ W + 0.05 * rng.standard_normal((N, d))).astype(np.float32)
Xt = torch.from_numpy(X); r2 = (Xt * Xt).sum(1)
np.random.seed(42); ix = np.random.randint(N)
D = torch.empty((na, N)); run = None
for a in range(na):
    v = Xt[ix]; D[a] = torch.sqrt(torch.clamp(r2 + (v * v).sum() - 2 * (Xt @ v), min=0))
    run = D[a].clone() if a <= 1 else torch.minimum(run, D[a])
    ix = int(torch.argmax(D[0] if a == 0 else run))
D = D.numpy()
def kd(order):
    if len(order) <= T: return [order]
    sub = D[:, order]; a = int(np.argmax(sub.var(axis=1)))
    o = order[np.argsort(sub[a], kind="stable")]; h = len(o) // 2
    return kd(o[:h]) + kd(o[h:])
tiles = kd(np.arange(N)); nt = len(tiles)
import os
tf = "/tmp/truth_%d.npy" % N
if os.path.exists(tf): truth = np.load(tf)
else:
    B = 2048; truth = np.empty((N, k), np.int64)
    for b in range(0, N, B):
        d2 = r2[None, :] + r2[b:b + B, None] - 2 * (Xt[b:b + B] @ Xt.T)
        d2[torch.arange(min(B, N - b)), torch.arange(b, min(b + B, N))] = float("inf")
        truth[b:b + B] = torch.topk(d2, k, dim=1, largest=False).indices.numpy()
    np.save(tf, truth)
tset = np.sort(truth, axis=1)
def rec(G):
    g = np.sort(G, axis=1); hit = 0
    for c in range(k): hit += (tset == g[:, c:c + 1]).any(1).sum() if False else 0
    # vectorised membership
    return sum(np.isin(G[:, c], truth[:, :]) if False else 0 for c in range(0)) or np.mean([(truth == G[:, c:c + 1]).any(1).mean() for c in range(k)])
Dm = np.stack([D[:, m].mean(1) for m in tiles])
key = ((Dm[:, None, :] - Dm[None, :, :]) ** 2).sum(-1)
def tile_phase(frac):
    m = max(1, int(np.ceil(frac * nt)))
    sel = np.argsort(key, axis=1)[:, :m]
    G = np.empty((N, k), np.int64); Gd = np.empty((N, k), np.float32)
    for I, rows in enumerate(tiles):
        cols = np.concatenate([tiles[j] for j in sel[I]])
        d2 = r2[rows][:, None] + r2[cols][None, :] - 2 * (Xt[rows] @ Xt[cols].T)
        d2[torch.from_numpy(rows)[:, None] == torch.from_numpy(cols)[None, :]] = float("inf")
        v, ixx = torch.topk(d2, k, dim=1, largest=False)
        G[rows] = cols[ixx.numpy()]; Gd[rows] = v.numpy()
    return G, Gd
def join(G, Gd, reverse=False, rk=15):
    """candidates(i) = neighbours of i's neighbours (+ optionally of its reverse neighbours)"""
    if reverse:
        # reverse lists capped at rk (closest first)
        src = np.repeat(np.arange(N), k); dst = G.ravel(); dd = Gd.ravel()
        o = np.lexsort((dd, dst)); dst_s, src_s = dst[o], src[o]
        start = np.searchsorted(dst_s, np.arange(N)); cnt = np.diff(np.append(start, len(dst_s)))
        R = np.full((N, rk), -1, np.int64)
        pos = np.arange(len(dst_s)) - start[dst_s]
        ok = pos < rk; R[dst_s[ok], pos[ok]] = src_s[ok]
        base = np.concatenate([G, np.where(R >= 0, R, G[:, :rk])], axis=1)    # [N, k+rk]
    else:
        base = G
    newG, newD = G.copy(), Gd.copy(); evals = 0
    B = 2048
    for b in range(0, N, B):
        rows = np.arange(b, min(b + B, N))
        cand = base[base[rows]].reshape(len(rows), -1)           # [B, |base|^2]
        if reverse: cand = np.concatenate([cand, base[rows]], axis=1)
        cand = np.sort(cand, axis=1)
        dup = np.zeros_like(cand, bool); dup[:, 1:] = cand[:, 1:] == cand[:, :-1]
        dup |= cand == rows[:, None]
        evals += (~dup).sum()
        Xc = Xt[torch.from_numpy(cand)]                           # [B, C, d]
        d2 = ((Xc - Xt[rows][:, None, :]) ** 2).sum(-1)
        d2[torch.from_numpy(dup)] = float("inf")
        alld = torch.cat([torch.from_numpy(newD[rows]), d2], 1); alli = np.concatenate([newG[rows], cand], 1)
        # dedupe against existing list: mark candidates already in G
        inG = (torch.from_numpy(cand)[:, :, None] == torch.from_numpy(G[rows])[:, None, :]).any(-1)
        alld[:, k:][inG] = float("inf")
        v, ixx = torch.topk(alld, k, dim=1, largest=False)
        newG[rows] = np.take_along_axis(alli, ixx.numpy(), 1); newD[rows] = v.numpy()
    return newG, newD, evals
for frac in fracs:
    t0 = time.time(); G, Gd = tile_phase(frac)
    print("budget %.3f: tile phase recall %.4f (%.0fs)" % (frac, rec(G), time.time() - t0), flush=True)
    for rev in (False, True):
        g, gd = G, Gd
        for it in range(3):
            g, gd, ev = join(g, gd, reverse=rev)
            print("   join%s pass %d: recall %.4f  evals/row %.0f" % (" +reverse" if rev else "", it + 1, rec(g), ev / N), flush=True)
