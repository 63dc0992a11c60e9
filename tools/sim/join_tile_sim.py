"""Design probe: join at ROW-TILE granularity -- every row of a 128-row tile is compared with the
union of its tile mates' neighbour-of-neighbour candidates (one gathered tile GEMM)."""
import sys, time, os
import numpy as np, torch
torch.set_num_threads(8)
exec(open(os.path.join(os.path.dirname(__file__), "join_sim.py")).read().split("for frac in fracs:")[0])
def revlists(G, Gd, rk):
    src = np.repeat(np.arange(N), k); dst = G.ravel(); dd = Gd.ravel()
    o = np.lexsort((src, dd, dst)); dst_s, src_s = dst[o], src[o]
    start = np.searchsorted(dst_s, np.arange(N))
    R = np.full((N, rk), -1, np.int64); pos = np.arange(len(dst_s)) - start[dst_s]
    ok = pos < rk; R[dst_s[ok], pos[ok]] = src_s[ok]
    return R
def join_tile(G, Gd, reverse, rk=15, hops2=True):
    base = G
    if reverse:
        R = revlists(G, Gd, rk); base = np.concatenate([G, R], axis=1)
    newG, newD = G.copy(), Gd.copy(); usz = []
    for rows in tiles:
        b1 = base[rows].ravel(); b1 = np.unique(b1[b1 >= 0])
        if hops2:
            u = base[b1].ravel(); u = np.unique(np.concatenate([u[u >= 0], b1]))
        else:
            u = b1
        usz.append(len(u))
        d2 = r2[rows][:, None] + r2[u][None, :] - 2 * (Xt[rows] @ Xt[u].T)
        d2[torch.from_numpy(rows)[:, None] == torch.from_numpy(u)[None, :]] = float("inf")
        v, ixx = torch.topk(d2, k, dim=1, largest=False)   # U contains the current neighbours => superset
        newG[rows] = u[ixx.numpy()]; newD[rows] = v.numpy()
    return newG, newD, np.mean(usz), np.max(usz)
for frac in fracs:
    G, Gd = tile_phase(frac)
    print("budget %.3f: tile phase recall %.4f" % (frac, rec(G)), flush=True)
    for rev, h2 in ((False, True), (True, True), (True, False)):
        g, gd = G, Gd
        for it in range(2):
            g, gd, um, ux = join_tile(g, gd, rev, hops2=h2)
            print("   tile-join rev=%d hops2=%d pass %d: recall %.4f  |U| mean %.0f max %d" % (rev, h2, it + 1, rec(g), um, ux), flush=True)
