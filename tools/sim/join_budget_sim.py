"""Design probe: strict total budget (tile phase + joins <= ceil(p_work * nt) tile-equivalents per row
tile); join candidates ranked by multiplicity (number of 2-hop paths from the tile's rows)."""
import sys, time, os
import numpy as np, torch
torch.set_num_threads(8)
exec(open(os.path.join(os.path.dirname(__file__), "join_sim.py")).read().split("for frac in fracs:")[0])
def revlists(G, rk=15):
    src = np.repeat(np.arange(N), k); dst = G.ravel(); pos = np.tile(np.arange(k), N)
    o = np.lexsort((src, pos, dst)); dst_s, src_s = dst[o], src[o]
    start = np.searchsorted(dst_s, np.arange(N))
    R = np.full((N, rk), -1, np.int64); p = np.arange(len(dst_s)) - start[dst_s]
    ok = p < rk; R[dst_s[ok], p[ok]] = src_s[ok]
    return R
def tile_phase_m(m):
    sel = np.argsort(key, axis=1)[:, :m]
    G = np.empty((N, k), np.int64); Gd = np.empty((N, k), np.float32)
    ev = np.zeros((nt, nt), bool)
    for I, rows in enumerate(tiles):
        ev[I, sel[I]] = True
        cols = np.concatenate([tiles[j] for j in sel[I]])
        d2 = r2[rows][:, None] + r2[cols][None, :] - 2 * (Xt[rows] @ Xt[cols].T)
        d2[torch.from_numpy(rows)[:, None] == torch.from_numpy(cols)[None, :]] = float("inf")
        v, ixx = torch.topk(d2, k, dim=1, largest=False)
        G[rows] = cols[ixx.numpy()]; Gd[rows] = v.numpy()
    return G, Gd, ev
tile_of = np.empty(N, np.int64)
for t, m in enumerate(tiles): tile_of[m] = t
def join_budget(G, Gd, ev, chunks):
    R = revlists(G); base = np.concatenate([G, R], axis=1)
    newG, newD = G.copy(), Gd.copy(); used = 0
    for I, rows in enumerate(tiles):
        b1 = base[rows].ravel(); b1 = np.unique(b1[b1 >= 0])
        u = np.concatenate([base[b1].ravel(), b1]); u = u[u >= 0]
        u = u[~ev[I, tile_of[u]]]
        ids, cnt = np.unique(u, return_counts=True)
        if len(ids) > chunks * 128:
            o = np.lexsort((ids, -cnt))[:chunks * 128]; ids = ids[o]
        used += (len(ids) + 127) // 128
        if len(ids) == 0: continue
        cols = np.concatenate([ids, G[rows].ravel()])
        d2 = r2[rows][:, None] + r2[cols][None, :] - 2 * (Xt[rows] @ Xt[cols].T)
        d2[torch.from_numpy(rows)[:, None] == torch.from_numpy(cols)[None, :]] = float("inf")
        # dedupe columns: take unique
        cu, first = np.unique(cols, return_index=True)
        d2 = d2[:, first]
        v, ixx = torch.topk(d2, k, dim=1, largest=False)
        newG[rows] = cu[ixx.numpy()]; newD[rows] = v.numpy()
    return newG, newD, used / nt
T = int(np.ceil(0.1 * nt))
print("N", N, "tiles", nt, "budget", T)
for tphase, plan in ((T, []), (T - 2, [1, 1]), (T - 4, [2, 2]), (T - 6, [3, 3]), (T - 8, [4, 4]), (T - 6, [2, 2, 2]), (T // 2, [T // 4, T // 4])):
    G, Gd, ev = tile_phase_m(tphase)
    s = "tile phase %d: %.4f" % (tphase, rec(G))
    for c in plan:
        G, Gd, used = join_budget(G, Gd, ev, c)
        s += "  join(<=%d): %.4f (%.1f used)" % (c, rec(G), used)
    print(s, flush=True)
