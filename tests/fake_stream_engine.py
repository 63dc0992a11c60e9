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

    def stream_anchor_round(self, vec, rnd, na):
        if rnd == 0:
            self.D = np.zeros((na, self.n), dtype=np.float32)
            self.na = na
        d = np.sqrt(((self.X - vec[None, :]) ** 2).sum(axis=1, dtype=np.float32)).astype(np.float32)
        self.D[rnd] = d
        self.runmin = d.copy() if rnd <= 1 else np.minimum(self.runmin, d)
        arg = int(np.argmax(self.runmin))
        return float(self.runmin[arg]), arg

    def _alloc(self, arr):
        arr = np.ascontiguousarray(arr)
        self._keep[arr.ctypes.data] = arr
        return arr.ctypes.data

    def stream_order(self, min_tiles=0):
        cA = np.argmin(self.D, axis=0)
        rad = self.D[cA, np.arange(self.n)]
        order = np.lexsort((np.arange(self.n), rad, cA))
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
