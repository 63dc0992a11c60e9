"""NumPy stand-in for the stream_* methods of annchor_amd._native.Engine: lets the CPU
tests run the multi-rank orchestration of annchor_amd.streamed (collectives, shard
bookkeeping, padding, result assembly) under gloo without a GPU.  "Device pointers" are
plain host addresses."""
import ctypes

import numpy as np

TILE = 128


def _view(addr, shape, dtype):
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    buf = (ctypes.c_char * n).from_address(addr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


class FakeStreamEngine:
    device = 0

    def __init__(self):
        self._keep = {}

    def stream_bind(self, X, global_base=0, **_):
        self.X = np.ascontiguousarray(X, dtype=np.float32)
        self.base = int(global_base)
        self.n, self.dim = self.X.shape

    def stream_get_row(self, i):
        return self.X[i].copy()

    def _sweep(self, vec, rnd):
        d = np.sqrt(((self.X - vec[None, :]) ** 2).sum(axis=1, dtype=np.float32)).astype(np.float32)
        self.D[rnd] = d
        self.runmin = d.copy() if rnd <= 1 else np.minimum(self.runmin, d)
        arg = int(np.argmax(self.runmin))
        return float(self.runmin[arg]), arg

    def stream_anchor_round(self, vec, rnd, na):
        if rnd == 0:
            self.D = np.zeros((na, self.n), dtype=np.float32)
            self.na = na
        return self._sweep(np.asarray(vec, dtype=np.float32), rnd)

    # ---- device-resident multi-rank protocol (csrc/sharded.hip), on host memory
    def hip_stream(self):
        return 0

    def stream_anchor_begin(self, na, first_global, world=1):
        self.na, self.D = na, np.zeros((na, self.n), dtype=np.float32)
        self._A, self._avecs = np.zeros(na, dtype=np.int64), np.zeros((na, self.dim), dtype=np.float32)
        self._cand = np.zeros(2 + self.dim)
        mine = self.base <= first_global < self.base + self.n
        self._cand[1] = first_global if mine else -1
        if mine:
            self._cand[2:] = self.X[first_global - self.base]
        self._cand_all = np.zeros((world, 2 + self.dim))
        return self._cand.ctypes.data, self._cand_all.ctypes.data, self._cand.nbytes

    def stream_anchor_step(self, gathered, world, rnd):
        G = _view(gathered, (world, 2 + self.dim), np.float64)
        best = None
        for r in range(world):
            v, i = G[r, 0], int(G[r, 1])
            if i >= 0 and (best is None or v > best[0] or (v == best[0] and i < best[1])):
                best = (v, i, r)
        self._A[rnd] = best[1]
        self._avecs[rnd] = G[best[2], 2:].astype(np.float32)
        v, arg = self._sweep(self._avecs[rnd], rnd)
        self._cand[0], self._cand[1] = v, self.base + arg
        self._cand[2:] = self.X[arg]

    def stream_anchor_end(self, na):
        return self._A.copy(), self._avecs.copy()

    def stream_rows_begin(self, counts):
        most = int(max(counts))
        self._send = np.zeros((most, self.dim), dtype=np.float32)
        self._send[:self.n] = self.X
        self._recv = np.zeros((len(counts), most, self.dim), dtype=np.float32)
        return self._send.ctypes.data, self._recv.ctypes.data, self._send.nbytes

    def stream_anchor_dists_begin(self, counts):
        most = int(max(counts))
        self._dsend = np.zeros((self.na, most), dtype=np.float32)
        self._dsend[:, :self.n] = self.D
        self._drecv = np.zeros((len(counts), self.na, most), dtype=np.float32)
        self._d_gathered = True
        return self._dsend.ctypes.data, self._drecv.ctypes.data, self._dsend.nbytes

    def stream_rows_end(self, counts):
        self.own_base, self.own_n = self.base, self.n
        self.X = np.concatenate([self._recv[r, :int(c)] for r, c in enumerate(counts)])
        self.n, self.base = len(self.X), 0
        D = np.stack([np.sqrt(((self.X - v[None, :]) ** 2).sum(axis=1, dtype=np.float32)).astype(np.float32) for v in self._avecs])
        if getattr(self, "_d_gathered", False):    # the ranks' own anchor distances, all-gathered: what a recomputation gives
            self.D = np.concatenate([self._drecv[r][:, :int(c)] for r, c in enumerate(counts)], axis=1)
            assert np.array_equal(self.D, D), "gathered anchor distances are not in rank order"
            self._d_gathered = False
        else:
            self.D = D

    def stream_lists_all(self, world, nbytes):
        return self._alloc(np.zeros(world * nbytes, dtype=np.uint8))

    def stream_route_begin(self, starts, bases):
        rid, idx, dist, ev = self.stream_knn(*self._run)
        starts, bases = np.asarray(starts), np.asarray(bases)
        k = idx.shape[1]

        def glob(pos):
            r = np.searchsorted(starts, pos, side="right") - 1
            return pos - starts[r] + bases[r]

        real = rid >= 0
        pos = rid[real]
        dest = np.searchsorted(starts, pos, side="right") - 1
        order = np.argsort(dest, kind="stable")[::-1]          # any order inside a destination is allowed
        order = order[np.argsort(dest[order], kind="stable")]
        rec = np.zeros((len(pos), 1 + 2 * (k - 1)), dtype=np.int64)
        rec[:, 0] = glob(pos)
        rec[:, 1:k] = glob(idx[real][:, 1:])
        rec[:, k:] = dist[real][:, 1:].astype(np.float64).view(np.int64)
        self._route_send = np.ascontiguousarray(rec[order])
        self._k = k
        return self._route_send.ctypes.data, np.bincount(dest, minlength=len(bases)).astype(np.int64), rec.shape[1], ev

    def stream_route_recv(self, n_recv):
        self._route_recv = np.zeros((max(n_recv, 1), 2 * self._k - 1), dtype=np.int64)
        return self._route_recv.ctypes.data

    def stream_route_end(self, n_recv, rows_padded, n_own, k):
        assert n_recv == n_own == self.own_n
        rec = self._route_recv[:n_recv]
        self._gi = np.full((rows_padded, k), -1, dtype=np.int64)
        self._gd = np.full((rows_padded, k), np.inf)
        loc = rec[:, 0] - self.own_base
        assert np.all((loc >= 0) & (loc < n_own)) and len(np.unique(loc)) == n_own
        self._gi[loc, 0], self._gd[loc, 0] = rec[:, 0], 0.0
        self._gi[loc, 1:], self._gd[loc, 1:] = rec[:, 1:k], rec[:, k:].view(np.float64)
        return self._gi[:n_own].copy(), self._gd[:n_own].copy()

    def stream_graph_device(self):
        return self._gi.ctypes.data, self._gd.ctypes.data, self._gi.shape[0], self._gi.shape[1]

    def _alloc(self, arr):
        arr = np.ascontiguousarray(arr)
        self._keep[arr.ctypes.data] = arr
        return arr.ctypes.data

    def _full_order(self):
        cA = np.argmin(self.D, axis=0)
        rad = self.D[cA, np.arange(self.n)]
        return np.lexsort((np.arange(self.n), rad, cA))

    def stream_order_begin(self, min_tiles, tile_begin, tile_count):
        """The rank's slice of the order goes through the all-gather; stream_order_end builds everything from the gathered
        slices only (so a slice in the wrong place shows up as a wrong graph)."""
        nt = max((self.n + TILE - 1) // TILE, min_tiles)
        full = np.full(nt * TILE, 0xFFFFFFFF, dtype=np.uint32)
        full[:self.n] = self._full_order()
        self._oslice = np.ascontiguousarray(full[tile_begin * TILE:(tile_begin + tile_count) * TILE])
        self._oall = np.full(nt * TILE, 0xFFFFFFFF, dtype=np.uint32)
        self._omin = min_tiles
        return self._oslice.ctypes.data, self._oall.ctypes.data, self._oslice.nbytes

    def stream_order_end(self):
        order = self._oall[:self.n].astype(np.int64)
        assert np.array_equal(np.sort(order), np.arange(self.n)), "gathered order is not a permutation"
        return self.stream_order(self._omin, order=order)

    def stream_join_rev_begin(self, lists_all):
        ptrs, n_all, nt_all, na, dimp, tile_begin, tile_count, k, p_work = self._run
        self._rev_all = np.full((n_all, 15), -1, dtype=np.int32)
        self._rev_all[tile_begin * TILE:(tile_begin + tile_count) * TILE] = tile_begin
        sl = self._rev_all[tile_begin * TILE:(tile_begin + tile_count) * TILE]
        return sl.ctypes.data, self._rev_all.ctypes.data, sl.nbytes

    def stream_order(self, min_tiles=0, order=None):
        if order is None:
            order = self._full_order()
        nt = max((self.n + TILE - 1) // TILE, min_tiles)
        n_pad = nt * TILE
        Xs = np.zeros((n_pad, self.dim), dtype=np.float32)
        Xs[: self.n] = self.X[order]
        rs = np.full(n_pad, np.inf, dtype=np.float32)
        rs[: self.n] = (Xs[: self.n] ** 2).sum(axis=1)
        perm = np.full(n_pad, -1, dtype=np.int64)
        perm[: self.n] = self.base + order
        lo = np.full((self.na, nt), np.inf, dtype=np.float32)
        hi = np.full((self.na, nt), -np.inf, dtype=np.float32)
        mid = np.full((self.na, nt), np.inf, dtype=np.float32)
        Ds = self.D[:, order]
        for t in range(nt):
            seg = Ds[:, t * TILE:min((t + 1) * TILE, self.n)]
            if seg.shape[1]:
                lo[:, t], hi[:, t], mid[:, t] = seg.min(axis=1), seg.max(axis=1), seg.mean(axis=1)
        ptrs = {k: self._alloc(v) for k, v in dict(Xs=Xs, rs=rs, perm=perm, lo=lo, hi=hi, mid=mid).items()}
        return ptrs, n_pad, nt, self.dim

    def synchronize(self):
        pass

    def stream_join_tables(self, gathered, world, na, nt, joined):
        src = _view(gathered, (world, na, nt), np.float32)
        _view(joined, (na, world * nt), np.float32)[:] = np.concatenate(list(src), axis=1)

    # row-sharded build in steps: the stand-in is exact in its "tile phase", so the join passes only
    # have to move plausible buffers through the collectives
    def stream_budget(self, nt_all, p_work, join_passes):
        total = max(1, min(nt_all, int(np.ceil(p_work * nt_all))))
        return total, max(1, total - join_passes), 1   # (the stand-in's "tile phase" is exact whatever it is given)

    def stream_knn_begin(self, ptrs, n_all, nt_all, na, dimp, tile_begin, tile_count, k, tile_budget):
        self._run = (ptrs, n_all, nt_all, na, dimp, tile_begin, tile_count, k, 1.0)
        nbytes = tile_count * TILE * (k - 1) * 4
        self._lists = self._alloc(np.full(nbytes // 4, tile_begin, dtype=np.int32))
        return self._lists, nbytes

    def stream_knn_join(self, lists_all, per_pass):
        ptrs, n_all, nt_all, na, dimp, tile_begin, tile_count, k, p_work = self._run
        every = _view(lists_all, (n_all, k - 1), np.int32)
        mine = every[tile_begin * TILE:(tile_begin + tile_count) * TILE]
        assert np.all(mine == tile_begin), "all-gathered lists are not in rank order"
        tiles = np.arange(n_all) // TILE // tile_count * tile_count      # first tile of the rank that owns each row
        assert np.array_equal(self._rev_all[:, 0], tiles) and np.array_equal(self._rev_all[:, -1], tiles), \
            "all-gathered reverse lists are not in rank order"
        self.joins = getattr(self, "joins", 0) + 1
        return self._lists, 0

    def stream_knn_end(self, n_local=None):
        return self.stream_knn(*self._run, n_local=n_local)

    def stream_knn(self, ptrs, n_all, nt_all, na, dimp, tile_begin, tile_count, k, p_work, n_local=None, join_passes=0, join_extra=0):
        if n_local is not None:   # shard-order output: rows = the bound shard's own rows
            rid, idx, dist, ev = self.stream_knn(ptrs, n_all, nt_all, na, dimp, tile_begin, tile_count, k, p_work)
            oi, od = np.zeros((n_local, k), dtype=np.int64), np.zeros((n_local, k))
            real = rid >= 0
            oi[rid[real] - self.base], od[rid[real] - self.base] = idx[real], dist[real]
            return None, oi, od, ev
        Xs = _view(ptrs["Xs"], (n_all, dimp), np.float32)
        perm = _view(ptrs["perm"], (n_all,), np.int64)
        rows = np.arange(tile_begin * TILE, (tile_begin + tile_count) * TILE)
        row_ids = perm[rows].copy()
        idx = np.zeros((len(rows), k), dtype=np.int64)
        dist = np.zeros((len(rows), k))
        real_cols = np.nonzero(perm >= 0)[0]
        for o, r in enumerate(rows):
            if row_ids[o] < 0:
                continue
            d = np.sqrt(((Xs[real_cols] - Xs[r][None, :]) ** 2).sum(axis=1, dtype=np.float64)).astype(np.float32)
            d[real_cols == r] = -1  # self first
            best = np.lexsort((perm[real_cols], d))[:k]
            idx[o] = perm[real_cols[best]]
            dist[o] = np.maximum(d[best], 0)
        return row_ids, idx, dist, tile_count * nt_all

    def device_alloc(self, nbytes):
        return self._alloc(np.zeros(nbytes, dtype=np.uint8))

    def device_free(self, ptr):
        self._keep.pop(ptr, None)

    def device_copy(self, dst, src, nbytes, kind):
        ctypes.memmove(dst, src, nbytes)
