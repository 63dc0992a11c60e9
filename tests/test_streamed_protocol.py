"""CPU tests of the multi-rank (row-sharded) orchestration: two gloo ranks with a NumPy
engine must pick the same anchors and produce the same graph as one rank."""
import os
import socket
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))


def _data(n=700, d=12, seed=5):
    rng = np.random.default_rng(seed)
    Z = rng.standard_normal((n, 4))
    W = rng.standard_normal((4, d))
    return (Z @ W + 0.05 * rng.standard_normal((n, d))).astype(np.float32)


def test_helpers():
    from annchor_amd.streamed import combine_argmax, owner_of

    shards = [(0, 300), (300, 280), (580, 120)]
    assert owner_of(0, shards) == 0 and owner_of(299, shards) == 0 and owner_of(300, shards) == 1 and owner_of(699, shards) == 2
    with pytest.raises(ValueError):
        owner_of(700, shards)
    assert combine_argmax([(1.0, 5), (3.0, 9), (3.0, 7), (2.0, 1)]) == 7   # first index on ties
    assert combine_argmax([(-np.inf, 3)]) == 3


def test_single_rank_matches_bruteforce_and_oracle_anchors():
    from fake_stream_engine import FakeStreamEngine
    from annchor_amd.streamed import StreamedAnnchor
    from oracle import annchor_oracle as O

    X = _data()
    sa = StreamedAnnchor(X, n_anchors=6, n_neighbors=5, p_work=1.0, random_seed=42, engine=FakeStreamEngine()).fit()

    def one_to_all(ix):
        return np.sqrt(((X - X[ix][None, :]) ** 2).sum(axis=1, dtype=np.float32)).astype(np.float32)

    A, _ = O.maxmin_anchors(one_to_all, len(X), 6, 42)   # the reference picker semantics (pickers.py:18-52)
    assert np.array_equal(sa.A, A)
    idx, dist = sa.neighbor_graph
    assert np.array_equal(idx[:, 0], np.arange(len(X))) and np.all(dist[:, 0] == 0)
    D = np.sqrt(((X[:, None, :].astype(np.float64) - X[None, :, :]) ** 2).sum(-1))
    np.testing.assert_allclose(dist, np.sort(D, axis=1)[:, :5], rtol=1e-5, atol=1e-6)


def _worker(rank, world, port, out):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fake_stream_engine import FakeStreamEngine
    from annchor_amd.streamed import StreamedAnnchor, TorchComm

    X = _data()
    cuts = [0, 410, 700]   # ragged shards: 410 and 290 rows (different tile counts -> padding path)
    Xl = X[cuts[rank]:cuts[rank + 1]]
    sa = StreamedAnnchor(Xl, n_anchors=6, n_neighbors=5, p_work=1.0, random_seed=42, base=cuts[rank], comm=TorchComm(),
                         engine=FakeStreamEngine()).fit()
    gi, gd = sa.gather_graph()
    if rank == 0:
        np.savez(out, A=sa.A, idx=gi, dist=gd, evals=sa.evals, joins=getattr(sa._engine, "joins", 0))
    dist.destroy_process_group()


def test_two_gloo_ranks_equal_one_rank(tmp_path):
    import torch.multiprocessing as mp

    from fake_stream_engine import FakeStreamEngine
    from annchor_amd.streamed import StreamedAnnchor

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "w2.npz")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    R = np.load(out)
    one = StreamedAnnchor(_data(), n_anchors=6, n_neighbors=5, p_work=1.0, random_seed=42, engine=FakeStreamEngine()).fit()
    assert np.array_equal(R["A"], one.A)
    assert int(R["joins"]) == 2   # both join passes ran against the all-gathered neighbour lists
    assert np.array_equal(R["idx"], one.neighbor_graph[0])
    np.testing.assert_allclose(R["dist"], one.neighbor_graph[1], rtol=0, atol=0)


def _worker_cuts(rank, world, port, out, n, cuts):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fake_stream_engine import FakeStreamEngine
    from annchor_amd.streamed import StreamedAnnchor, TorchComm

    X = _data(n=n)
    Xl = X[cuts[rank]:cuts[rank + 1]]
    sa = StreamedAnnchor(Xl, n_anchors=6, n_neighbors=5, p_work=1.0, random_seed=42, base=cuts[rank], comm=TorchComm(),
                         engine=FakeStreamEngine()).fit()
    own = sa.neighbor_graph
    assert own[0].shape == (len(Xl), 5) and np.array_equal(own[0][:, 0], np.arange(cuts[rank], cuts[rank + 1]))
    gi, gd = sa.gather_graph()
    if rank == 0:
        np.savez(out, A=sa.A, idx=gi, dist=gd)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,cuts", [
    (3, 700, [0, 300, 580, 700]),                                     # ragged, every shard a different tile count
    (4, 1061, [0, 266, 532, 798, 1061]),                              # C5's ratios: equal shards but a shorter last one
    (8, 1061, [0, 133, 266, 399, 532, 665, 798, 931, 1061]),          # 9 tiles over 8 ranks: 16 with padding, ranks 5-7 own
])                                                                     # (almost) only padding tiles -> empty all-to-all slices
def test_many_gloo_ranks_equal_one_rank(tmp_path, world, n, cuts):
    """4- and 8-rank rehearsal of the row-sharded protocol at BASELINE configs[4]'s shape ratios (N not a multiple of
    128 x world, uneven last shard, tile count padded to a multiple of the rank count): anchors and graph equal to the
    one-rank build."""
    import torch.multiprocessing as mp

    from fake_stream_engine import FakeStreamEngine
    from annchor_amd.streamed import StreamedAnnchor

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "wn.npz")
    mp.spawn(_worker_cuts, args=(world, port, out, n, cuts), nprocs=world, join=True)
    R = np.load(out)
    one = StreamedAnnchor(_data(n=n), n_anchors=6, n_neighbors=5, p_work=1.0, random_seed=42, engine=FakeStreamEngine()).fit()
    assert np.array_equal(R["A"], one.A)
    assert np.array_equal(R["idx"], one.neighbor_graph[0])
    np.testing.assert_allclose(R["dist"], one.neighbor_graph[1], rtol=0, atol=0)


def test_non_contiguous_shard_bases_map_back():
    """Shards need not be contiguous or in rank order in the global numbering: rows, neighbours and query results are
    mapped through the (starts, bases) tables (one rank, exchange forced)."""
    from fake_stream_engine import FakeStreamEngine
    from annchor_amd.streamed import StreamedAnnchor

    X = _data(n=300)
    a = StreamedAnnchor(X, n_anchors=4, n_neighbors=4, p_work=1.0, base=1000, engine=FakeStreamEngine(), force_exchange=True).fit()
    b = StreamedAnnchor(X, n_anchors=4, n_neighbors=4, p_work=1.0, engine=FakeStreamEngine()).fit()
    assert np.array_equal(a.neighbor_graph[0], b.neighbor_graph[0] + 1000)
    assert np.array_equal(a.neighbor_graph[1], b.neighbor_graph[1])
    assert np.array_equal(a.A, b.A + 1000)


def _worker_early(rank, world, port, out):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fake_stream_engine import FakeStreamEngine
    from annchor_amd.streamed import StreamedAnnchor, TorchComm, make_comm

    class EarlyRows(TorchComm):   # the shape of RcclComm's side communicator: the rows' gather is asked for before the anchor rounds
        overlap = True
        begun = 0

        def allgather_begin(self, engine, src, dst, nbytes):
            self.begun += 1
            self.allgather_into(engine, src, dst, nbytes)

    assert isinstance(make_comm(), TorchComm)          # a gloo job: torch.distributed on host copies
    eng = FakeStreamEngine()
    eng.comm_allgather_begin = None                     # (an engine that has the side path)
    cuts = [0, 410, 700]   # (ragged shards: the send buffer is a padded copy)
    X = _data()
    comm = EarlyRows()
    sa = StreamedAnnchor(X[cuts[rank]:cuts[rank + 1]], n_anchors=6, n_neighbors=5, p_work=1.0, random_seed=42, base=cuts[rank], comm=comm,
                         engine=eng).fit()
    assert comm.begun == 1
    gi, gd = sa.gather_graph()
    if rank == 0:
        np.savez(out, A=sa.A, idx=gi, dist=gd)
    dist.destroy_process_group()


def test_rows_gather_started_before_the_anchor_rounds(tmp_path):
    """fit() with a communicator that can run the rows' all-gather beside the anchor rounds (RcclComm's side communicator,
    csrc/comm.hip) asks for it FIRST -- stream_rows_begin + allgather_begin, then the rounds, the anchor distances' gather and the
    ordering -- and gives the graph of the one-rank build; make_comm() picks TorchComm for a gloo job and SingleComm without one."""
    import torch.multiprocessing as mp

    from fake_stream_engine import FakeStreamEngine
    from annchor_amd.streamed import SingleComm, StreamedAnnchor, make_comm

    assert isinstance(make_comm(), SingleComm)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "early.npz")
    mp.spawn(_worker_early, args=(2, port, out), nprocs=2, join=True)
    R = np.load(out)
    one = StreamedAnnchor(_data(), n_anchors=6, n_neighbors=5, p_work=1.0, random_seed=42, engine=FakeStreamEngine()).fit()
    assert np.array_equal(R["A"], one.A)
    assert np.array_equal(R["idx"], one.neighbor_graph[0])
    np.testing.assert_allclose(R["dist"], one.neighbor_graph[1], rtol=0, atol=0)
