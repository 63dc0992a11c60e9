"""
annchor_amd.streamed -- host orchestration of the streamed (tile-granular) form of the
k-NN graph build for float32 Euclidean data at N >> 10^4, on one or several GPUs.

One process per GPU, rows sharded contiguously: rank r owns rows
[base_r, base_r + n_r).  Per stage (reference annchor/annchor.py:532-623 -> here):

  get_anchors   max-min rounds (pickers.py:18-52).  Every round: every rank sweeps its rows on
                the GPU (one-to-all distances, running min, local arg-max), then ONE all-gather
                of (value, index, that row's coordinates) -- (2 + dim) doubles per rank: the
                arg-max with the first-index tie rule is taken locally and the next anchor's
                vector is already there (no broadcast from its owner).
  exchange      ONE all-gather of the raw rows: every rank needs every row as a column anyway.
                (RCCL over xGMI when the process group is `nccl`; staged through host memory otherwise.)
  locality      every rank recomputes the anchor distances of all rows (it knows the anchors: no
                collective) and orders ALL rows into 128-row tiles of a k-d order in anchor space with
                per-anchor distance intervals -- the same tile structure whatever the number of ranks;
                rank r owns a contiguous range of the global tile order.
  refine+top-k  each rank evaluates its own row tiles against the column tiles that its
                triangle bound cannot exclude (MFMA tile GEMM + in-LDS top-k), within the
                p_work tile budget; then the join passes, each after an all-gather of the ranks'
                current neighbour lists.
  result        the finished rows go back to the ranks that own them (all-to-all); each rank holds
                the graph rows of its shard; `gather_graph()` assembles the full graph on every rank.

With one rank the collectives are no-ops and no torch import happens.
"""
import numpy as np

TILE = 128
MIN_TILE_BUDGET = 64   # tile evaluations per row tile below which p_work is raised (see StreamedAnnchor.__init__)
JOIN_YIELD = 0.01   # extra join passes run while a pass still replaces more than this share of all list entries


# ----------------------------------------------------------------------- comms
# Everything that crosses ranks on the fit path is a numeric buffer: 16 bytes per rank and anchor round
# (value, index) with the anchor's coordinates, the raw rows, the neighbour lists (device buffers), the
# finished graph rows.  No pickled objects.
class SingleComm:
    rank, world = 0, 1

    def allgather_f64(self, values):
        return np.asarray(values, dtype=np.float64)[None, :]

    def allgather_device(self, engine, dptr, nbytes):
        return dptr, None

    def allgather_host(self, arr):
        return [arr]

    def allgather_rows(self, X, counts):
        return "host", X, None

    def exchange_rows(self, dest, arrays):
        return [np.ascontiguousarray(a) for a in arrays]


class TorchComm:
    """torch.distributed adapter.  `nccl` groups (RCCL over xGMI) move device buffers directly;
    any other backend stages through host memory."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.backend = dist.get_backend(group)

    def _dev(self):
        return "cuda" if self.backend == "nccl" else "cpu"

    def allgather_f64(self, values):
        """[world, len(values)] float64: every rank's small vector (arg-max candidates, shard extents)."""
        import torch

        t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=self._dev())
        parts = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t, group=self.group)
        return torch.stack(parts).cpu().numpy()

    def allgather_device(self, engine, dptr, nbytes):
        """All-gather `nbytes` bytes at device pointer `dptr` from every rank; returns the
        device pointer of the rank-ordered concatenation (+ an owner object to keep alive)."""
        import torch

        if self.backend == "nccl":
            inp = device_tensor_u8(dptr, nbytes, engine.device)
            out = torch.empty(self.world * nbytes, dtype=torch.uint8, device=inp.device)
            engine.synchronize()                       # the engine's stream produced the buffer
            self.dist.all_gather_into_tensor(out, inp, group=self.group)
            torch.cuda.synchronize(inp.device)
            return out.data_ptr(), out
        host = np.empty(nbytes, dtype=np.uint8)
        engine.device_copy(host.ctypes.data, dptr, nbytes, "d2h")
        parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.world)]
        self.dist.all_gather(parts, torch.from_numpy(host), group=self.group)
        allh = torch.cat(parts).numpy()
        out = engine.device_alloc(allh.nbytes)
        engine.device_copy(out, allh.ctypes.data, allh.nbytes, "h2d")
        return out, _DeviceOwner(engine, out)

    def allgather_host(self, arr):
        """Every rank's equally shaped NumPy array, as a list in rank order (tensor all-gather; the
        arrays ride through device memory when the group is `nccl`)."""
        import torch

        t = torch.from_numpy(np.ascontiguousarray(arr)).to(self._dev())
        parts = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t, group=self.group)
        return [p.cpu().numpy() for p in parts]


    def allgather_rows(self, X, counts):
        """Every rank's rows (float32 [counts[r], dim]) concatenated in rank order, on every rank:
        ("device", pointer, owner tensor) when the group is `nccl`, ("host", array, None) otherwise."""
        import torch

        most, dim = int(max(counts)), X.shape[1]
        pad = X if X.shape[0] == most else np.concatenate([X, np.zeros((most - X.shape[0], dim), dtype=X.dtype)])
        if self.backend == "nccl":
            t = torch.from_numpy(np.ascontiguousarray(pad)).to("cuda")
            out = torch.empty((self.world * most, dim), dtype=t.dtype, device=t.device)
            self.dist.all_gather_into_tensor(out, t, group=self.group)
            if any(int(c) != most for c in counts):
                out = torch.cat([out[r * most:r * most + int(counts[r])] for r in range(self.world)])
            torch.cuda.synchronize(out.device)
            return "device", out.data_ptr(), out
        parts = [torch.empty((most, dim), dtype=torch.float32) for _ in range(self.world)]
        self.dist.all_gather(parts, torch.from_numpy(np.ascontiguousarray(pad)), group=self.group)
        return "host", np.concatenate([parts[r][:int(counts[r])].numpy() for r in range(self.world)]), None

    def exchange_rows(self, dest, arrays):
        """Row r of every array goes to rank dest[r]; returns the rows this rank receives (source-rank order).
        `nccl`: all_to_all_single with split sizes on device tensors; other backends have no all-to-all:
        padded all-gather, every rank keeps its slice."""
        import torch

        order = np.argsort(dest, kind="stable")
        send = np.bincount(dest, minlength=self.world).astype(np.int64)
        C = self.allgather_f64(send).astype(np.int64)          # C[src, dst]
        recv = C[:, self.rank]
        out = []
        if self.backend == "nccl":
            for a in arrays:
                t = torch.from_numpy(np.ascontiguousarray(a[order])).to("cuda")
                r = torch.empty((int(recv.sum()),) + tuple(a.shape[1:]), dtype=t.dtype, device=t.device)
                self.dist.all_to_all_single(r, t, output_split_sizes=[int(v) for v in recv],
                                            input_split_sizes=[int(v) for v in send], group=self.group)
                out.append(r.cpu().numpy())
            return out
        most = int(C.sum(axis=1).max())
        for a in arrays:
            pad = np.zeros((most,) + tuple(a.shape[1:]), dtype=a.dtype)
            pad[:len(order)] = a[order]
            parts = self.allgather_host(pad)
            out.append(np.concatenate([parts[src][int(C[src, :self.rank].sum()):int(C[src, :self.rank + 1].sum())]
                                       for src in range(self.world)]))
        return out


class _DeviceOwner:
    def __init__(self, engine, ptr):
        self.engine, self.ptr = engine, ptr

    def __del__(self):
        try:
            self.engine.device_free(self.ptr)
        except Exception:
            pass


class _CAI:
    """Minimal __cuda_array_interface__ carrier for a raw device pointer."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def device_tensor_u8(ptr, nbytes, device=0):
    """View `nbytes` bytes of device memory at `ptr` as a torch uint8 tensor (no copy)."""
    import torch

    return torch.as_tensor(_CAI(ptr, nbytes), device=torch.device("cuda", device))


# ------------------------------------------------------------------ helpers
def owner_of(ix, shards):
    """Rank whose shard [base, base+n) contains global row ix."""
    for r, (base, n) in enumerate(shards):
        if base <= ix < base + n:
            return r
    raise ValueError("row %d is in no shard" % ix)


def combine_argmax(cands):
    """Global arg-max of per-rank (value, global index): largest value, then smallest index
    (np.argmax's first-index rule, pickers.py:47-50)."""
    best_v, best_i = -np.inf, None
    for v, i in cands:
        if best_i is None or v > best_v or (v == best_v and i < best_i):
            best_v, best_i = v, i
    return int(best_i)


class StreamedAnnchor:
    """k-NN graph of float32 points under the Euclidean metric, streamed form.

    X is THIS rank's shard (float32 [n_local, dim]); `base` its first global row id.
    `engine` is an `annchor_amd._native.Engine` (or any object with the same stream_*
    methods -- the CPU tests use a NumPy stand-in to exercise the multi-rank protocol)."""

    def __init__(self, X, n_anchors=32, n_neighbors=15, p_work=0.1, random_seed=42, base=0, comm=None, engine=None,
                 device=0, force_exchange=False, join_passes=2, join_extra=4):
        self.X = np.ascontiguousarray(X, dtype=np.float32)
        self.n_local, self.dim = self.X.shape
        self.n_anchors, self.n_neighbors, self.p_work = n_anchors, n_neighbors, p_work
        self.random_seed, self.base = random_seed, int(base)
        self.comm = comm if comm is not None else SingleComm()
        if engine is None:
            from . import _native

            engine = _native.Engine(device)
        self._engine = engine
        self._engine.stream_bind(self.X, self.base)
        self.join_passes, self.join_extra = int(join_passes), int(join_extra)
        ext = self.comm.allgather_f64((self.base, self.n_local))
        self.shards = [(int(b), int(n)) for b, n in ext]
        self.n_total = int(sum(n for _, n in self.shards))
        # The budget is spent in whole 128 x 128 tile evaluations: below MIN_TILE_BUDGET of them per row
        # tile it is too coarse to mean anything, so p_work has a floor -- the same treatment the
        # reference gives a p_work too small for its anchors and samples (annchor.py:136-142)
        nt_total = sum((n + TILE - 1) // TILE for _, n in self.shards)
        floor = min(1.0, MIN_TILE_BUDGET / float(max(nt_total, 1)))
        if self.p_work < floor:
            if self.comm.rank == 0:
                print("Warning: p_work too low for %d tiles of 128 points.\nIncreasing p_work to %5.3f." % (nt_total, floor))
            self.p_work = floor
        self.evals = 0
        self.timings = {}
        self.force_exchange = force_exchange   # run the all-gather path even with one rank (tests)

    def _budget(self, nt_all):
        """(total, tile phase, per join pass) tile evaluations per row tile: annchor_stream_budget."""
        if hasattr(self._engine, "stream_budget"):
            return self._engine.stream_budget(nt_all, self.p_work, self.join_passes)
        from . import _native

        return _native.stream_budget(nt_all, self.p_work, self.join_passes)

    def get_anchors(self):
        """Max-min rounds (pickers.py:18-52) over the sharded rows.  One collective per round: every
        rank contributes its local arg-max (value, global index) TOGETHER with that row's coordinates
        -- (2 + dim) doubles -- so that after the all-gather every rank knows the winner and already
        holds the next anchor's vector (no separate broadcast from the owner)."""
        eng, comm, na = self._engine, self.comm, self.n_anchors
        np.random.seed(self.random_seed)
        ix = int(np.random.randint(self.n_total))  # identical on every rank
        A = np.zeros(na, dtype=np.int64)
        self.anchor_vectors = np.zeros((na, self.dim), dtype=np.float32)   # kept for query()

        def exchange(value, index, local_row):
            mine = np.empty(2 + self.dim, dtype=np.float64)
            mine[0], mine[1] = value, index
            mine[2:] = eng.stream_get_row(local_row) if local_row >= 0 else 0.0
            G = comm.allgather_f64(mine)
            win = combine_argmax([(G[r, 0], int(G[r, 1])) for r in range(G.shape[0]) if G[r, 1] >= 0])
            r = [int(G[q, 1]) for q in range(G.shape[0])].index(win)
            return win, G[r, 2:].astype(np.float32)

        # the first anchor: only its owner has a candidate
        mine = self.base <= ix < self.base + self.n_local
        ix, vec = exchange(0.0, ix if mine else -1, ix - self.base if mine else -1)
        for r in range(na):
            A[r] = ix
            self.anchor_vectors[r] = vec
            lmax, larg = eng.stream_anchor_round(vec, r, na)
            if r + 1 < na:
                ix, vec = exchange(float(lmax), int(self.base + larg), int(larg))
        self.A = A
        self.evals += na * self.n_total

    def fit(self):
        import time

        t0 = time.perf_counter()
        eng, comm = self._engine, self.comm
        self.get_anchors()
        t1 = time.perf_counter()
        sharded = comm.world > 1 or self.force_exchange
        if sharded:
            # ONE tile structure for the whole data set, whatever the number of ranks: every rank gets all rows
            # (one all-gather; it needs them as columns anyway), recomputes their anchor distances from the anchors
            # it already knows (no collective: 32 streaming passes) and orders them itself; the ranks then own
            # contiguous ranges of the GLOBAL tile order.  Ordering each shard separately (round 1) made a tile's
            # cell G times larger -- recall at N = 400 000 fell from 0.990 (1 rank) to 0.968 (2) and 0.942 (4).
            counts = [n for _, n in self.shards]
            kind, Xall, owner = comm.allgather_rows(self.X, counts)
            if kind == "device":
                eng.stream_bind(None, 0, device_ptr=Xall, shape=(self.n_total, self.dim))
            else:
                eng.stream_bind(Xall, 0)
            del owner, Xall
            for r in range(self.n_anchors):
                eng.stream_anchor_round(self.anchor_vectors[r], r, self.n_anchors)
            tiles_per_rank = -(-((self.n_total + TILE - 1) // TILE) // comm.world)
            ptrs, n_pad, nt, dimp = eng.stream_order(tiles_per_rank * comm.world)
            tile_begin, tile_count = comm.rank * tiles_per_rank, tiles_per_rank
        else:
            ptrs, n_pad, nt, dimp = eng.stream_order(0)
            tile_begin, tile_count = 0, nt
        t2 = t3 = time.perf_counter()
        n_all, nt_all = n_pad, nt
        if not sharded:
            row_ids, idx, dist, tile_evals = eng.stream_knn(ptrs, n_all, nt_all, self.n_anchors, dimp, 0, nt, self.n_neighbors,
                                                            self.p_work, n_local=self.n_local, join_passes=self.join_passes,
                                                            join_extra=self.join_extra)
        else:
            # tile phase on the rank's own row tiles, then join passes against the all-gathered neighbour lists
            total, tile_budget, per_pass = self._budget(nt_all)
            lists, nbytes = eng.stream_knn_begin(ptrs, n_all, nt_all, self.n_anchors, dimp, tile_begin, tile_count,
                                                 self.n_neighbors, tile_budget)
            floor_updates = JOIN_YIELD * n_all * (self.n_neighbors - 1)
            for p in range(self.join_passes + self.join_extra if tile_budget < nt_all else 0):
                if p >= self.join_passes and tile_budget + (p + 1) * max(per_pass, 1) > total:
                    break      # the budget has no room for another pass
                lists_all, owner = comm.allgather_device(eng, lists, nbytes)
                lists, upd = eng.stream_knn_join(lists_all, max(per_pass, 1))
                del owner
                # every rank takes the same decision: the yield of the pass summed over ranks
                if p + 1 >= self.join_passes and comm.allgather_f64((upd,)).sum() <= floor_updates:
                    break
            row_ids, idx, dist, tile_evals = eng.stream_knn_end()
            # rows and neighbours are numbered by their position in the rank-ordered concatenation of the shards:
            # back to global row ids, and every row back to the rank that owns it
            starts = np.concatenate([[0], np.cumsum([n for _, n in self.shards])]).astype(np.int64)
            bases = np.array([b for b, _ in self.shards], dtype=np.int64)

            def to_global(pos):
                r = np.searchsorted(starts, pos, side="right") - 1
                return pos - starts[r] + bases[r]

            real = row_ids >= 0
            pos = row_ids[real]
            dest = (np.searchsorted(starts, pos, side="right") - 1).astype(np.int64)
            gid, gidx, gdist = comm.exchange_rows(dest, [to_global(pos), to_global(idx[real]), dist[real]])
            row_ids, idx, dist = gid, gidx, gdist
        t4 = time.perf_counter()
        # the ordered column arrays (every row of the data set) stay alive: query() runs against them
        self._columns = dict(ptrs=ptrs, n_all=n_all, nt_all=nt_all, dimp=dimp)
        if row_ids is None:   # rows already in this shard's order (emitted on the device)
            ng_idx, ng_dist = idx, dist
        else:                 # tile order + global row ids: reorder on the host
            real = row_ids >= 0
            loc = row_ids[real] - self.base
            k = self.n_neighbors
            ng_idx = np.zeros((self.n_local, k), dtype=np.int64)
            ng_dist = np.zeros((self.n_local, k), dtype=np.float64)
            ng_idx[loc], ng_dist[loc] = idx[real], dist[real]
        self.neighbor_graph = (ng_idx, ng_dist)
        self.tile_evals = int(tile_evals)
        self.evals += self.tile_evals * TILE * TILE
        self.n_tiles_total = nt_all
        self.timings = dict(get_anchors=t1 - t0, order=t2 - t1, exchange=0.0, knn=t4 - t3, total=time.perf_counter() - t0)
        return self

    def query(self, Q, nn=15, p_work=0.1, device=None):
        """The nn nearest data rows of every row of Q (indices [nq, nn] into the global row
        numbering, distances [nq, nn]) -- Annchor.query (annchor.py:643-683) for the streamed
        form: the queries get the fitted anchors' distances, are ordered into tiles like the data,
        and every query tile evaluates its best-ranked ceil(p_work * #tiles) data tiles with the
        same kernel.  Any rank can answer: after fit() each holds all rows in tile order."""
        from . import _native

        if not hasattr(self, "_columns"):
            raise RuntimeError("fit() first")
        Q = np.ascontiguousarray(Q, dtype=np.float32)
        if Q.ndim != 2 or Q.shape[1] != self.dim:
            raise ValueError("queries must be float32 [nq, %d]" % self.dim)
        qe = _native.Engine(self._engine.device if device is None else device)
        try:
            qe.stream_bind(Q, 0)
            for r in range(self.n_anchors):
                qe.stream_anchor_round(self.anchor_vectors[r], r, self.n_anchors)
            _, _, _, dimp = qe.stream_order(0)
            c = self._columns
            idx, dist, tile_evals = qe.stream_query(c["ptrs"], c["n_all"], c["nt_all"], self.n_anchors, dimp, nn, p_work)
        finally:
            qe.close()
        self.evals += int(tile_evals) * TILE * TILE
        return idx, dist

    def gather_graph(self):
        """Full graph (all shards, global row order) on every rank: the final neighbour-graph
        gather -- two tensor all-gathers (indices, distances) of shards padded to the largest."""
        k = self.n_neighbors
        most = max(n for _, n in self.shards)
        idx = np.full((most, k), -1, dtype=np.int64)
        dist = np.full((most, k), np.inf, dtype=np.float64)
        idx[:self.n_local], dist[:self.n_local] = self.neighbor_graph
        all_idx, all_dist = self.comm.allgather_host(idx), self.comm.allgather_host(dist)
        order = sorted(range(len(self.shards)), key=lambda r: self.shards[r][0])
        return (np.concatenate([all_idx[r][:self.shards[r][1]] for r in order]),
                np.concatenate([all_dist[r][:self.shards[r][1]] for r in order]))
