"""The row-sharded streamed build of G ranks on ONE GPU with the ranks' GPU work SERIALISED: what each rank's kernels cost
when they have a GPU to themselves -- the per-stage input of the scaling model in DESIGN.md section 7.

G threads of one process, one engine (context) each, all on device 0.  A token (lock) is held by whichever rank is inside
the library; a collective = the rank waits for its own stream, hands the token on, meets the others at a barrier, and
the copies that RCCL would do over xGMI are device-to-device copies between the contexts' buffers (same process, same
device: the pointers are valid everywhere).  So the per-kernel times the library records (ProfScope: HIP events around
every kernel family) are exclusive-GPU times of THAT rank's share of the work -- unlike the gloo rehearsal
(tools/c5_rehearsal.py, tests/test_c5_gpu.py) where eight processes' kernels interleave on the device.  The collectives'
payloads are logged in bytes per rank; their time on 8 x MI355X is modelled, not measured (no multi-GPU box here).

  python tools/serial_ranks.py --n 8000000 --worlds 1,2,4,8 --out gpurun_out/serial_ranks_c5.json
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

D, NA, K, PW = 128, 32, 15, 0.1


def shard_rows(r, n, seed=4321):
    import torch

    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    W = torch.randn(8, D, generator=g, device="cuda", dtype=torch.float32)
    g.manual_seed(seed + 1 + r)
    Z = torch.randn(n, 8, generator=g, device="cuda", dtype=torch.float32)
    X = Z @ W + 0.05 * torch.randn(n, D, generator=g, device="cuda", dtype=torch.float32)
    out = X.cpu().numpy()
    del X, Z
    torch.cuda.empty_cache()
    return out


class World:
    def __init__(self, G):
        self.G = G
        self.barrier = threading.Barrier(G)
        self.token = threading.Lock()
        self.slots = [None] * G
        self.log = []          # (kind, bytes per rank) in call order, rank 0's view


class ThreadComm:
    """The comm adapter protocol of annchor_amd.streamed (allgather_small / allgather_into / alltoall_records) between the
    threads of one process; the caller holds the world's token whenever it is inside the library."""
    backend = "threads"

    def __init__(self, world, rank):
        self.w, self.rank, self.world = world, rank, world.G

    def _meet(self, engine):
        """own stream finished -> token released -> everybody here -> token back"""
        if engine is not None:
            engine.synchronize()
        self.w.token.release()
        self.w.barrier.wait()
        self.w.token.acquire()

    def allgather_small(self, values):
        v = np.asarray(values, dtype=np.float64).reshape(-1).copy()
        self.w.slots[self.rank] = v
        self._meet(None)
        out = np.stack([self.w.slots[r] for r in range(self.world)])
        self._meet(None)
        return out

    def allgather_into(self, engine, src, dst, nbytes):
        self.w.slots[self.rank] = src
        if self.rank == 0:
            self.w.log.append(("all_gather", int(nbytes)))
        self._meet(engine)
        for r in range(self.world):
            if dst + r * nbytes != self.w.slots[r]:
                engine.device_copy(dst + r * nbytes, self.w.slots[r], nbytes, "d2d")
        self._meet(engine)

    # the shape of RcclComm's side communicator (csrc/comm.hip): the rows' all-gather is asked for BEFORE the anchor rounds and runs
    # beside them, the anchor distances' all-gather and the k-d order.  Here the copy happens at once (one GPU); the log marks
    # the collective as overlapped and the model (tools/scaling_model.py) charges only what the overlapped stages do not cover.
    overlap = True

    def allgather_begin(self, engine, src, dst, nbytes):
        n0 = len(self.w.log)
        self.allgather_into(engine, src, dst, nbytes)
        if self.rank == 0 and len(self.w.log) > n0:
            self.w.log[-1] = ("all_gather_overlapped", int(nbytes))

    def alltoall_records(self, engine, send, send_counts, words):
        sc = np.asarray(send_counts, dtype=np.int64)
        self.w.slots[self.rank] = (send, sc)
        self._meet(engine)
        n_recv = int(sum(int(self.w.slots[r][1][self.rank]) for r in range(self.world)))
        recv = engine.stream_route_recv(n_recv)
        at = 0
        for r in range(self.world):
            p, c = self.w.slots[r]
            cnt = int(c[self.rank])
            off = int(c[:self.rank].sum())
            if cnt:
                engine.device_copy(recv + at * words * 8, p + off * words * 8, cnt * words * 8, "d2d")
            at += cnt
        if self.rank == 0:
            self.w.log.append(("all_to_all", int(sc.sum()) * words * 8))
        self._meet(engine)
        return recv, n_recv


def run_world(G, n, shards, bases, fits, results):
    from annchor_amd.streamed import SingleComm, StreamedAnnchor

    W = World(G)
    out = [None] * G

    def rank_main(rank):
        W.token.acquire()
        try:
            comm = ThreadComm(W, rank) if G > 1 else SingleComm()
            res = None
            for it in range(fits):
                if rank == 0:
                    W.log.clear()      # (only rank 0 appends: the log of the last fit survives)
                sa = StreamedAnnchor(shards[rank], n_anchors=NA, n_neighbors=K, p_work=PW, base=int(bases[rank]), comm=comm, device=0)
                sa._engine.prof_enable(True)
                sa.fit()
                prof = sa._engine.prof_get()
                res = dict(prof={k: v["ms"] for k, v in prof.items()}, graph=sa.neighbor_graph, tile_evals=sa.tile_evals,
                           nt=sa.n_tiles_total)
                if it + 1 < fits:
                    sa._engine.close()
                else:
                    res["sa"] = sa
                if G > 1 and it + 1 < fits:
                    comm._meet(None)
            out[rank] = res
        except BaseException as e:      # a rank that dies must not leave the others at a barrier
            out[rank] = e
            W.barrier.abort()
            raise
        finally:
            W.token.release()

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(G)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    wall = time.perf_counter() - t0
    for o in out:
        if isinstance(o, BaseException):
            raise o
    results[G] = dict(out=out, log=list(W.log), wall=wall)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--fits", type=int, default=2)
    ap.add_argument("--recall-rows", type=int, default=1000)
    ap.add_argument("--out", default="gpurun_out/serial_ranks.json")
    args = ap.parse_args()
    import torch

    torch.cuda.set_device(0)
    n = args.n
    worlds = [int(v) for v in args.worlds.split(",")]
    Gmax = max(worlds)
    # the data set is the same for every world: the finest sharding's shards, merged for the coarser ones
    per = -(-n // Gmax)
    fine = [shard_rows(r, min(per, n - r * per)) for r in range(Gmax)]
    Xall = np.concatenate(fine)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from test_c5_gpu import check_graph, truth_f64

    rows = np.sort(np.random.default_rng(78).choice(n, args.recall_rows, replace=False))
    bd = truth_f64(Xall[rows], [Xall], K)
    report = {"workload": "synthetic Euclidean f32 (SURVEY 8d recipe) N=%d d=%d n_anchors=%d k=%d p_work=%.2f, streamed form; G ranks as "
                          "threads of one process on ONE GPU, GPU work serialised by a token (per-rank kernel times are exclusive-GPU times)"
                          % (n, D, NA, K, PW), "worlds": {}}
    results = {}
    for G in worlds:
        step = Gmax // G
        shards = [np.concatenate(fine[r * step:(r + 1) * step]) for r in range(G)]
        bases = np.concatenate([[0], np.cumsum([len(s) for s in shards])]).astype(np.int64)
        run_world(G, n, shards, bases, args.fits, results)
        R = results[G]
        gi = np.concatenate([o["graph"][0] for o in R["out"]])
        gd = np.concatenate([o["graph"][1] for o in R["out"]])
        assert np.array_equal(gi[:, 0], np.arange(n))
        recall = check_graph(rows, Xall[rows], Xall[gi[rows]], gi, gd, bd)
        names = sorted({k for o in R["out"] for k in o["prof"]})
        stages = {k: [round(o["prof"].get(k, 0.0), 3) for o in R["out"]] for k in names}
        report["worlds"][str(G)] = {
            "recall_at_k": recall, "recall_rows": len(rows), "tile_evals_all_ranks": int(sum(o["tile_evals"] for o in R["out"])),
            "n_tiles": int(R["out"][0]["nt"]),
            "kernel_ms_per_rank": stages,
            "kernel_ms_max_over_ranks": {k: max(v) for k, v in stages.items()},
            "kernel_ms_sum_of_stage_maxima": round(sum(max(v) for k, v in stages.items() if k != "stream_tile_two_stage_kernel"), 3),   # (that scope is inside stream_tile_gemm_topk)
            "collectives_rank0": [{"kind": k, "bytes_per_rank": b} for k, b in R["log"]],
            "collective_bytes_per_rank_total": int(sum(b for _, b in R["log"])),
        }
        for o in R["out"]:
            o["sa"]._engine.close()
        del R, gi, gd
        results.pop(G)
        print(G, json.dumps(report["worlds"][str(G)]["kernel_ms_max_over_ranks"]), "recall", recall, flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
