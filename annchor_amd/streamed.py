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
  features      ONE all-gather of the ranks' anchor distances (n_anchors floats per row: the sweeps of the
                max-min rounds left them on the device) -- nothing is recomputed.
  locality      the rows are ordered into 128-row tiles of a k-d order in anchor space with per-anchor
                distance intervals -- the same tile structure whatever the number of ranks; rank r owns a
                contiguous range of the global tile order and sorts, level by level, only the segments
                that hold its range; ONE all-gather of the order's slices (4 bytes per row).
  refine+top-k  each rank evaluates its own row tiles against the column tiles that its
                triangle bound cannot exclude (MFMA tile GEMM + in-LDS top-k), within the
                p_work tile budget; then the join passes, each after an all-gather of the ranks'
                current neighbour lists and of the reverse lists of the columns each rank owns.
  result        the finished rows go back to the ranks that own them (all-to-all); each rank holds
                the graph rows of its shard; `gather_graph()` assembles the full graph on every rank.

With one rank the collectives are no-ops and no torch import happens.
"""
import os

import numpy as np

TILE = 128
MIN_TILE_BUDGET = 64   # tile evaluations per row tile below which p_work is raised (see StreamedAnnchor.__init__)
JOIN_YIELD = 0.01   # extra join passes run while a pass still replaces more than this share of all list entries


# ----------------------------------------------------------------------- comms
# Everything that crosses ranks on the fit path is a numeric buffer in DEVICE memory owned by the engine (the
# arg-max candidates of a max-min round, the raw rows, the neighbour lists, the finished graph rows as records):
# the comm adapters below receive device pointers.  With an `nccl` process group (RCCL over xGMI) the pointers are
# wrapped as tensors and the collective runs on the engine's own stream -- no host staging, no host wait; any other
# backend (gloo: CPU tests, single-GPU rehearsals) runs the SAME collective calls on host copies of the buffers.
# No pickled objects anywhere.
class SingleComm:
    rank, world = 0, 1
    backend = "none"

    def allgather_small(self, values):
        return np.asarray(values, dtype=np.float64)[None, :]

    def allgather_into(self, engine, src, dst, nbytes):
        if src != dst:     # (an in-place gather of one rank's slice is already where it belongs)
            engine.device_copy(dst, src, nbytes, "d2d")

    def alltoall_records(self, engine, send, send_counts, words):
        n = int(send_counts[0])
        recv = engine.stream_route_recv(n)
        if n:
            engine.device_copy(recv, send, n * words * 8, "d2d")
        return recv, n


class TorchComm:
    """torch.distributed adapter over device pointers.  `nccl` groups (RCCL over xGMI) move the engine's buffers
    directly, ordered on the engine's stream; any other backend stages through host memory around the same calls."""

    SMALL = 64   # doubles per rank in a control-plane exchange

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.on_device = self.backend == "nccl"
        dev = "cuda" if self.on_device else "cpu"
        # control plane (shard extents, the join passes' yield, all-to-all counts): preallocated, a few numbers
        self._small_in = torch.zeros(self.SMALL, dtype=torch.float64, device=dev)
        self._small_out = torch.zeros(self.SMALL * self.world, dtype=torch.float64, device=dev)
        self._streams = {}

    # -- plumbing
    def _stream(self, engine):
        """Context manager: torch's current stream = the engine's stream (nccl), nothing otherwise."""
        import contextlib

        if not self.on_device:
            return contextlib.nullcontext()
        key = (engine.device, engine.hip_stream())
        if key not in self._streams:
            self._streams[key] = self.torch.cuda.ExternalStream(key[1], device=self.torch.device("cuda", engine.device))
        return self.torch.cuda.stream(self._streams[key])

    def _tensor_in(self, engine, ptr, nbytes):
        """uint8 tensor holding `nbytes` bytes at device pointer `ptr`: a view (nccl) or a host copy."""
        if nbytes == 0:
            return self.torch.empty(0, dtype=self.torch.uint8, device="cuda" if self.on_device else "cpu")
        if self.on_device:
            return device_tensor_u8(ptr, nbytes, engine.device)
        host = np.empty(nbytes, dtype=np.uint8)
        engine.device_copy(host.ctypes.data, ptr, nbytes, "d2h")
        return self.torch.from_numpy(host)

    def _tensor_out(self, engine, ptr, nbytes):
        if nbytes == 0 or not self.on_device:
            return self.torch.empty(nbytes, dtype=self.torch.uint8, device="cuda" if self.on_device else "cpu")
        return device_tensor_u8(ptr, nbytes, engine.device)

    def _commit(self, engine, tensor, ptr):
        """Host-staged backends: write the collective's output back to device memory."""
        if not self.on_device and tensor.numel():
            host = np.ascontiguousarray(tensor.numpy())
            engine.device_copy(ptr, host.ctypes.data, host.nbytes, "h2d")

    # -- control plane
    def allgather_small(self, values):
        """[world, len(values)] float64: every rank's few numbers."""
        v = np.asarray(values, dtype=np.float64).reshape(-1)
        assert v.size <= self.SMALL
        self._small_in[:v.size].copy_(self.torch.from_numpy(v))
        self.dist.all_gather_into_tensor(self._small_out, self._small_in, group=self.group)
        return self._small_out.view(self.world, self.SMALL)[:, :v.size].cpu().numpy().copy()

    # -- data plane
    def allgather_into(self, engine, src, dst, nbytes):
        """Every rank's `nbytes` bytes at `src`, concatenated in rank order at `dst` (device pointers)."""
        with self._stream(engine):
            inp = self._tensor_in(engine, src, nbytes)
            out = self._tensor_out(engine, dst, nbytes * self.world)
            self.dist.all_gather_into_tensor(out, inp, group=self.group)
            self._commit(engine, out, dst)

    def alltoall_records(self, engine, send, send_counts, words):
        """Records (`words` int64 words each) grouped by destination rank at `send`: one all_to_all_single with split
        sizes; returns (device pointer of the received records in source-rank order, their number)."""
        torch = self.torch
        C = self.allgather_small(send_counts).astype(np.int64)       # C[src, dst]
        recv_counts = [int(v) for v in C[:, self.rank]]
        n_send, n_recv = int(np.sum(send_counts)), int(sum(recv_counts))
        recv = engine.stream_route_recv(n_recv)
        with self._stream(engine):
            inp = self._tensor_in(engine, send, n_send * words * 8).view(torch.int64).view(n_send, words)
            out = self._tensor_out(engine, recv, n_recv * words * 8).view(torch.int64).view(n_recv, words)
            self.dist.all_to_all_single(out, inp, output_split_sizes=recv_counts, input_split_sizes=[int(v) for v in send_counts],
                                        group=self.group)
            self._commit(engine, out.view(-1).view(torch.uint8), recv)
        return recv, n_recv


class RcclComm:
    """Collectives from INSIDE the library (csrc/comm.hip: RCCL loaded with dlopen, enqueued on the engine's stream by the same
    C calls that enqueue the kernels): no torch on the data path, and the anchor rounds of a fit are one C call.  The
    communicator needs a 128-byte id made on one rank (`_native.comm_unique_id()`) and handed to the others --
    `RcclComm.from_torch(engine)` does that over an existing torch.distributed group of any backend (128 bytes, once).

    `side_id`: a second id for a second communicator on a stream of its own -- the rows' all-gather of a fit then runs beside
    the anchor rounds and the k-d order (`allgather_begin`; csrc/comm.hip).  `preflight`: a checked 1 KB all-gather on every
    communicator under its own timeout (a mis-wired job fails here, loudly).  `timeout`: host waits of the engine longer than
    this abort the communicators and raise (a dead peer rank no longer blocks the others for good; default 300 s,
    ANNCHOR_COMM_TIMEOUT_S)."""

    backend = "rccl"

    def __init__(self, engine, world, rank, unique_id, side_id=None, preflight=60.0, timeout=None):
        self.engine, self.world, self.rank = engine, int(world), int(rank)
        self._small = 0
        engine.comm_init(unique_id, world, rank)
        try:
            if timeout is not None:
                engine.comm_set_timeout(timeout)
            self.overlap = False
            if side_id is not None and os.environ.get("ANNCHOR_COMM_OVERLAP", "1") != "0":
                engine.comm_init_side(side_id)
                self.overlap = True
            if preflight:
                engine.comm_preflight(preflight)
            self._small = engine.device_alloc(8 * 64 * (self.world + 1))
        except Exception:
            engine.comm_destroy()
            raise

    @classmethod
    def from_torch(cls, engine, group=None, **kw):
        import torch.distributed as dist
        from . import _native

        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [(_native.comm_unique_id(), _native.comm_unique_id()) if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(engine, world, rank, box[0][0], side_id=box[0][1], **kw)

    def allgather_begin(self, engine, src, dst, nbytes):
        """The all-gather on the side communicator's stream, beside whatever the engine enqueues next; the library waits for it
        where it first needs the result (annchor_stream_rows_end / _order_end)."""
        engine.comm_allgather_begin(src, dst, nbytes)

    def close(self):
        if self._small:
            self.engine.device_free(self._small)
            self._small = 0
        self.engine.comm_destroy()

    def allgather_small(self, values):
        v = np.zeros(64, dtype=np.float64)
        vals = np.asarray(values, dtype=np.float64).reshape(-1)
        assert vals.size <= 64
        v[:vals.size] = vals
        eng = self.engine
        eng.device_copy(self._small, v.ctypes.data, 512, "h2d")
        eng.comm_allgather(self._small, self._small + 512, 512)
        out = np.empty((self.world, 64), dtype=np.float64)
        eng.device_copy(out.ctypes.data, self._small + 512, out.nbytes, "d2h")
        return out[:, :vals.size].copy()

    def allgather_into(self, engine, src, dst, nbytes):
        engine.comm_allgather(src, dst, nbytes)

    def alltoall_records(self, engine, send, send_counts, words):
        C = self.allgather_small(send_counts).astype(np.int64)       # C[src, dst]
        recv_counts = C[:, self.rank].copy()
        n_recv = int(recv_counts.sum())
        recv = engine.stream_route_recv(n_recv)
        engine.comm_alltoall_records(send, np.asarray(send_counts, dtype=np.int64), recv, recv_counts, words)
        return recv, n_recv


def make_comm(engine=None, group=None, prefer=None, verbose=True):
    """The communicator of a multi-rank build for the current torch.distributed job (one rank / no job: SingleComm).

    `nccl` groups (RCCL over xGMI): the collectives run from INSIDE the library on `engine` (RcclComm: two communicators,
    pre-flight, dead-peer timeout) -- the default when an engine is given; if RCCL cannot be loaded, a communicator cannot be
    made or the pre-flight fails ON ANY RANK, every rank falls back to torch.distributed on device pointers (TorchComm) and
    says so.  Other backends (gloo: CPU tests, rehearsals on one GPU) use TorchComm.  `prefer` / ANNCHOR_COMM = "torch" | "rccl"
    overrides.  The torch path's collectives are guarded by torch.distributed's own timeout (init_process_group(timeout=...))."""
    try:
        import torch.distributed as dist
    except ImportError:
        return SingleComm()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return SingleComm()
    prefer = prefer or os.environ.get("ANNCHOR_COMM")
    want_rccl = dist.get_backend(group) == "nccl" and engine is not None and prefer != "torch"
    if prefer == "rccl" and not want_rccl:
        raise RuntimeError("ANNCHOR_COMM=rccl needs an nccl process group and an engine")
    if not want_rccl:
        return TorchComm(group)
    import torch

    comm, err = None, ""
    try:
        comm = RcclComm.from_torch(engine, group)
    except Exception as e:   # dlopen / ncclCommInitRank / pre-flight
        err = "%s: %s" % (type(e).__name__, e)
    ok = torch.tensor([1.0 if comm is not None else 0.0], device="cuda")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if float(ok.item()) >= 1.0:
        return comm
    if comm is not None:
        comm.close()
    if verbose and (err or dist.get_rank(group) == 0):
        print("annchor: in-library RCCL communicator not available on every rank%s; falling back to torch.distributed collectives"
              % ((" (this rank: %s)" % err) if err else ""), flush=True)
    if prefer == "rccl":
        raise RuntimeError("ANNCHOR_COMM=rccl: the in-library communicator could not be made (%s)" % (err or "another rank failed"))
    return TorchComm(group)


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
        ext = self.comm.allgather_small((self.base, self.n_local))
        self.shards = [(int(b), int(n)) for b, n in ext]
        self.n_total = int(sum(n for _, n in self.shards))
        # position in the rank-ordered concatenation of the shards <-> global row id
        self._starts = np.concatenate([[0], np.cumsum([n for _, n in self.shards])]).astype(np.int64)
        self._bases = np.array([b for b, _ in self.shards], dtype=np.int64)
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
        """Max-min rounds (pickers.py:18-52) over the sharded rows.  Per round every rank leaves its local arg-max --
        (value, global row, that row's coordinates): 2 + dim doubles -- in a device buffer; ONE all-gather; the engine then
        picks the winner on the device (largest value, first index) and sweeps its rows with the winner's coordinates,
        which the all-gather already delivered (no broadcast from the owner).  Nothing waits for the host until the
        anchors are downloaded at the end."""
        eng, comm, na = self._engine, self.comm, self.n_anchors
        np.random.seed(self.random_seed)
        ix = int(self._to_global(np.random.randint(self.n_total)))  # identical on every rank
        sharded = comm.world > 1 or self.force_exchange
        cand, gathered, nbytes = eng.stream_anchor_begin(na, ix, comm.world)
        if getattr(comm, "backend", None) == "rccl" or (not sharded and hasattr(eng, "stream_anchor_rounds")):
            eng.stream_anchor_rounds(na)   # every round -- collective, pick, sweep -- enqueued by one C call
            na_done = na
        else:
            na_done = 0
        for r in range(na_done, na):
            if sharded:
                comm.allgather_into(eng, cand, gathered, nbytes)
                eng.stream_anchor_step(gathered, comm.world, r)
            else:
                eng.stream_anchor_step(cand, 1, r)
        self.A, self.anchor_vectors = eng.stream_anchor_end(na)   # anchor_vectors: kept for query()
        self.evals += na * self.n_total

    def _to_global(self, pos):
        """Positions in the rank-ordered concatenation of the shards -> global row ids."""
        r = np.searchsorted(self._starts, pos, side="right") - 1
        return pos - self._starts[r] + self._bases[r]

    def fit(self):
        import time

        t0 = time.perf_counter()
        eng, comm = self._engine, self.comm
        out = None
        if self.n_local * self.n_neighbors >= (1 << 20):
            # the graph's host arrays, allocated and page-faulted on a helper thread while the GPU works (_native.GraphBuffers)
            try:
                from ._native import GraphBuffers

                out = GraphBuffers(self.n_local, self.n_neighbors)
            except ImportError:   # (the CPU protocol tests run against a stand-in engine without the library)
                out = None
        sharded = comm.world > 1 or self.force_exchange
        counts = np.array([n for _, n in self.shards], dtype=np.int64)
        rows_early = sharded and getattr(comm, "overlap", False) and hasattr(eng, "comm_allgather_begin")
        if rows_early:
            # the rows' all-gather (half of a fit's collective bytes) starts NOW, on the side communicator's stream: it runs beside
            # the anchor rounds, the anchor distances' all-gather and the k-d order -- none of them reads the gathered rows
            send, recv, nbytes = eng.stream_rows_begin(counts)
            comm.allgather_begin(eng, send, recv, nbytes)
        self.get_anchors()
        t1 = time.perf_counter()
        if sharded:
            # ONE tile structure for the whole data set, whatever the number of ranks: every rank gets all rows
            # (one all-gather of the resident shards; it needs them as columns anyway), recomputes their anchor
            # distances from the anchors it already knows (one pass, no collective) and orders them itself; the ranks
            # then own contiguous ranges of the GLOBAL tile order.  Ordering each shard separately (round 1) made a
            # tile's cell G times larger -- recall at N = 400 000 fell from 0.990 (1 rank) to 0.968 (2) and 0.942 (4).
            if not rows_early:
                send, recv, nbytes = eng.stream_rows_begin(counts)
                comm.allgather_into(eng, send, recv, nbytes)
            # the anchor distances of the own rows (the max-min sweeps left them on the device): ONE all-gather of
            # n_anchors floats per row instead of every rank recomputing every row's
            send, recv, nbytes = eng.stream_anchor_dists_begin(counts)
            comm.allgather_into(eng, send, recv, nbytes)
            eng.stream_rows_end(counts)
            tiles_per_rank = -(-((self.n_total + TILE - 1) // TILE) // comm.world)
            tile_begin, tile_count = comm.rank * tiles_per_rank, tiles_per_rank
            # the k-d order: a rank sorts, level by level, only the segments that hold its own tile range; the slices of the
            # order (4 bytes per row) are all-gathered and every rank gathers the rows into tile order (it holds every row
            # as a column anyway)
            send, recv, nbytes = eng.stream_order_begin(tiles_per_rank * comm.world, tile_begin, tile_count)
            comm.allgather_into(eng, send, recv, nbytes)
            ptrs, n_pad, nt, dimp = eng.stream_order_end()
        else:
            ptrs, n_pad, nt, dimp = eng.stream_order(0)
            tile_begin, tile_count = 0, nt
        t2 = t3 = time.perf_counter()
        n_all, nt_all = n_pad, nt
        if not sharded:
            _, idx, dist, tile_evals = eng.stream_knn(ptrs, n_all, nt_all, self.n_anchors, dimp, 0, nt, self.n_neighbors,
                                                      self.p_work, n_local=self.n_local, join_passes=self.join_passes,
                                                      join_extra=self.join_extra, **({"out": out} if out is not None else {}))
        else:
            # tile phase on the rank's own row tiles, then join passes against the all-gathered neighbour lists
            total, tile_budget, per_pass = self._budget(nt_all)
            lists, nbytes = eng.stream_knn_begin(ptrs, n_all, nt_all, self.n_anchors, dimp, tile_begin, tile_count,
                                                 self.n_neighbors, tile_budget)
            lists_all = eng.stream_lists_all(comm.world, nbytes)
            floor_updates = JOIN_YIELD * n_all * (self.n_neighbors - 1)
            for p in range(self.join_passes + self.join_extra if tile_budget < nt_all else 0):
                if p >= self.join_passes and tile_budget + (p + 1) * max(per_pass, 1) > total:
                    break      # the budget has no room for another pass
                comm.allgather_into(eng, lists, lists_all, nbytes)
                # reverse neighbour lists: each rank builds those of the columns it owns, the slices are all-gathered
                rev, rev_all, rbytes = eng.stream_join_rev_begin(lists_all)
                comm.allgather_into(eng, rev, rev_all, rbytes)
                lists, upd = eng.stream_knn_join(lists_all, max(per_pass, 1))
                # every rank takes the same decision: the yield of the pass summed over ranks
                if p + 1 >= self.join_passes and comm.allgather_small((upd,)).sum() <= floor_updates:
                    break
            # the finished rows -- this rank computed its range of the global tile order: rows of arbitrary shards -- go
            # back to the ranks that own them: records bucketed by owner on the device, ONE all-to-all, scattered into the
            # shard's own row order, downloaded once
            send, send_counts, words, tile_evals = eng.stream_route_begin(self._starts, self._bases)
            recv, n_recv = comm.alltoall_records(eng, send, send_counts, words)
            idx, dist = eng.stream_route_end(n_recv, int(counts.max()), self.n_local, self.n_neighbors,
                                             **({"out": out} if out is not None else {}))
        t4 = time.perf_counter()
        # the ordered column arrays (every row of the data set) stay alive: query() runs against them
        self._columns = dict(ptrs=ptrs, n_all=n_all, nt_all=nt_all, dimp=dimp)
        self._sharded = sharded
        self.neighbor_graph = (idx, dist)
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
        if self._sharded:   # the columns are numbered by position in the rank-ordered concatenation of the shards
            idx = self._to_global(idx)
        return idx, dist

    def gather_graph(self):
        """Full graph (all shards, global row order) on every rank: the final neighbour-graph gather -- two
        all-gathers (indices, distances) of the shards' device-resident graph rows, padded to the largest shard."""
        k = self.n_neighbors
        if not getattr(self, "_sharded", False):
            return self.neighbor_graph
        eng, comm = self._engine, self.comm
        pi, pd, rows, kk = eng.stream_graph_device()
        assert kk == k and rows == max(n for _, n in self.shards)
        out = []
        for ptr, dtype in ((pi, np.int64), (pd, np.float64)):
            nbytes = rows * k * 8
            dst = eng.device_alloc(nbytes * comm.world)
            try:
                comm.allgather_into(eng, ptr, dst, nbytes)
                host = np.empty((comm.world, rows, k), dtype=dtype)
                eng.device_copy(host.ctypes.data, dst, host.nbytes, "d2h")
            finally:
                eng.device_free(dst)
            out.append(host)
        order = sorted(range(len(self.shards)), key=lambda r: self.shards[r][0])
        return tuple(np.concatenate([a[r][:self.shards[r][1]] for r in order]) for a in out)
