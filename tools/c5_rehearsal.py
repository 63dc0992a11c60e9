#!/usr/bin/env python3
"""
BASELINE configs[4] at its stated size on ONE GPU: synthetic Euclidean float32, N = 8 000 000, d = 128,
n_anchors = 32, k = 15, p_work = 0.1, either as `--world 8` gloo ranks that all use GPU 0 (the
device-pointer protocol, the split sizes and the buffer layouts of an 8-GPU run; the collectives staged through
the host) or as `--world 1` (the N = 1 point of C5's curve).  The shards are uneven (the last one is shorter and
no shard is a multiple of 128 rows).  Two fits per rank; checked on rank 0 and written as JSON:

  * recall@15 on 2000 rows (250 of every shard) against an INDEPENDENT float64 NumPy brute force (columns streamed
    through the host in blocks), reported distances at rtol 1e-5 against the float64 distance of the reported pair;
  * gather_graph(): column 0 is the row's own global id for every row, every rank's own rows sit at their global
    positions;
  * the tile budget (ceil(p_work * #tiles) evaluations per row tile, joins included) is respected;
  * device memory in use after the third fit is not above the second fit's (no growth; the first fit allocates, later
    ones reuse the closed engine's blocks from the library's pool).

    python tools/c5_rehearsal.py --world 8 --out gpurun_out/c5_rehearsal_w8.json
"""
import argparse
import ctypes
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

D, K, NA, PW = 128, 15, 32, 0.1


def shard_sizes(n, world):
    """Uneven on purpose: ranks 0 .. world-2 get ceil(n / world) + 37 rows, the last one the (shorter) rest."""
    if world == 1:
        return [n]
    per = -(-n // world) + 37
    sizes = [per] * (world - 1)
    sizes.append(n - per * (world - 1))
    assert sizes[-1] > 0
    return sizes


def shard(rank, rows):
    from bench import euclid_shard

    return euclid_shard(rank, rows, D)


def brute_f64(Xq, shards, k, block=40000, threads=16):
    """k smallest float64 distances of every row of Xq (float32 rows) to all rows of all shards, NumPy only."""
    from concurrent.futures import ThreadPoolExecutor

    Q = Xq.astype(np.float64)
    qq = (Q * Q).sum(1)
    best = np.full((len(Q), k), np.inf)
    chunks = np.array_split(np.arange(len(Q)), threads)

    def part(args):
        rows, d2 = args
        if d2.shape[1] <= k:
            return np.pad(d2[rows], ((0, 0), (0, k - d2.shape[1])), constant_values=np.inf)
        return np.partition(d2[rows], k - 1, axis=1)[:, :k]

    with ThreadPoolExecutor(threads) as ex:
        for S in shards:
            for c0 in range(0, len(S), block):
                C = S[c0:c0 + block].astype(np.float64)
                d2 = np.maximum(qq[:, None] + (C * C).sum(1)[None, :] - 2.0 * (Q @ C.T), 0.0)
                small = np.concatenate(list(ex.map(part, [(r, d2) for r in chunks])), axis=0)
                best = np.partition(np.concatenate([best, small], axis=1), k - 1, axis=1)[:, :k]
    return np.sqrt(np.sort(best, axis=1))


def worker(rank, world, port, n, out, fits):
    import torch

    from annchor_amd.streamed import SingleComm, StreamedAnnchor, TorchComm

    if world > 1:
        import torch.distributed as dist

        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import datetime

        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=60))
    sizes = shard_sizes(n, world)
    bases = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    X = shard(rank, sizes[rank])
    comm = TorchComm() if world > 1 else SingleComm()
    torch.cuda.set_device(0)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    from annchor_amd import _native

    used, parked, times, stage, last = [], [], [], [], None
    for it in range(fits):
        if last is not None:       # one engine alive at a time: eight ranks of ~25 GB each share the one device
            last._engine.close()
            last = None
        sa = StreamedAnnchor(X, n_anchors=NA, n_neighbors=K, p_work=PW, base=int(bases[rank]), comm=comm, device=0)
        sa._engine.prof_enable(True)
        barrier()
        t0 = time.perf_counter()
        sa.fit()
        barrier()
        times.append(time.perf_counter() - t0)
        stage.append({a: round(b, 3) for a, b in sa.timings.items()})
        last = sa
        barrier()
        free, total = torch.cuda.mem_get_info(0)
        used.append(int(total - free))
        pk = ctypes.c_int64()
        _native.load_library().annchor_parked_bytes(0, ctypes.byref(pk))
        parked.append(int(pk.value))
        barrier()
    sa = last
    ev = float(comm.allgather_small((sa.tile_evals,)).sum())
    own_i, own_d = sa.neighbor_graph
    ok_own = bool(np.array_equal(own_i[:, 0], np.arange(bases[rank], bases[rank + 1])) and np.all(own_d[:, 0] == 0)
                  and np.all(np.diff(own_d, axis=1) >= 0))
    t0 = time.perf_counter()
    gi, gd = sa.gather_graph()
    gather_s = time.perf_counter() - t0
    ok_slice = bool(np.array_equal(gi[bases[rank]:bases[rank + 1]], own_i) and np.array_equal(gd[bases[rank]:bases[rank + 1]], own_d))
    flags = comm.allgather_small((float(ok_own), float(ok_slice)))
    prof = sa._engine.prof_get()
    res = None
    if rank == 0:
        from annchor_amd import compare_neighbor_graphs

        nt = sa.n_tiles_total
        total, tile_budget, per_pass = sa._budget(nt)
        shards = [X] + [shard(r, sizes[r]) for r in range(1, world)]
        rng = np.random.default_rng(77)
        rows = np.sort(np.concatenate([bases[r] + rng.choice(sizes[r], 2000 // world, replace=False) for r in range(world)]))
        owner = np.searchsorted(bases, rows, side="right") - 1
        Xq = np.stack([shards[o][g - bases[o]] for g, o in zip(rows, owner)])
        t0 = time.perf_counter()
        bd = brute_f64(Xq, shards, K + 1)          # column 0 = the row itself
        truth_s = time.perf_counter() - t0
        bd[:, 0] = 0.0
        err = compare_neighbor_graphs((gi[rows], bd[:, :K]), (gi[rows], gd[rows]), K)
        recall = 1.0 - err / float(len(rows) * K)
        # every reported distance is the float64 distance of the reported pair (rtol 1e-5: float32 norms)
        nb = gi[rows]
        ob = np.searchsorted(bases, nb, side="right") - 1
        worst = 0.0
        for t in range(len(rows)):
            Y = np.stack([shards[o][g - bases[o]] for g, o in zip(nb[t], ob[t])]).astype(np.float64)
            dd = np.sqrt(((Y - Xq[t].astype(np.float64)[None, :]) ** 2).sum(1))
            worst = max(worst, float(np.max(np.abs(dd - gd[rows[t]]) / np.maximum(dd, 1e-3))))
        srt = np.sort(nb, axis=1)
        gemm = prof.get("stream_tile_gemm_topk", dict(ms=0.0, launches=1))
        tile_phase_evals, join_chunks = sa._engine.stream_last_counts()
        gemm_s = gemm["ms"] / max(1, gemm["launches"]) * 1e-3
        res = {
            "workload": "synthetic Euclidean f32 (SURVEY 8d recipe) N=%d d=%d n_anchors=%d k=%d p_work=%.2f, streamed form, %d gloo rank(s) on ONE GPU"
                        % (n, D, NA, K, PW, world),
            "shard_rows": sizes, "n_tiles": int(nt), "fits": fits, "fit_time_s": [round(t, 3) for t in times],
            "stage_s_rank0": stage, "gather_graph_s": round(gather_s, 2),
            "budget_tiles_per_row_tile": {"total": total, "tile_phase": tile_budget, "per_join_pass": per_pass},
            "tile_evals_all_ranks": int(ev), "tile_budget_all": int(total) * int(nt), "budget_respected": bool(ev <= total * nt),
            "tile_fraction_of_brute_force": ev / float(nt) / float(nt),
            "recall_at_k": recall, "recall_rows": int(len(rows)), "errors": int(err),
            "recall_truth": "float64 NumPy brute force of %d rows (%d per shard) against all %d rows, %.0f s" % (len(rows), 2000 // world, n, truth_s),
            "max_rel_error_of_reported_distances": worst, "distances_within_rtol_1e-5": bool(worst <= 1e-5),
            "no_duplicate_neighbours": bool(np.all(srt[:, 1:] != srt[:, :-1])),
            "gather_graph_row_order_ok": bool(np.array_equal(gi[:, 0], np.arange(n)) and np.all(gd[:, 0] == 0)),
            "every_rank_rows_ok": bool(np.all(flags[:, 0] == 1)), "every_rank_slice_of_gather_ok": bool(np.all(flags[:, 1] == 1)),
            # fit 1 allocates; from fit 2 on the blocks of the closed engine are reused (the library parks blocks >= 16 MB in a
            # process-wide pool): the device total (all ranks) must stop growing
            "device_bytes_in_use_after_fit": used, "rank0_parked_bytes_after_fit": parked,
            "no_device_memory_growth": bool(len(used) < 3 or used[-1] <= used[-2] + (256 << 20)),
            "kernels_ms_rank0_last_fit": {kk: round(v["ms"], 2) for kk, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:14]},
            "tile_phase_rank0": {"tile_pairs": int(tile_phase_evals), "join_chunks": int(join_chunks), "kernel_s": round(gemm_s, 3),
                                 "tflops_f32": tile_phase_evals * 128.0 * 128 * 2 * 128 / max(gemm_s, 1e-9) / 1e12,
                                 "note": "with several ranks on one GPU the ranks' kernels share the device: kernel_s is wall time of rank 0's launch"},
        }
        res["all_checks_pass"] = bool(res["budget_respected"] and recall >= 0.99 and res["distances_within_rtol_1e-5"]
                                      and res["no_duplicate_neighbours"] and res["gather_graph_row_order_ok"] and res["every_rank_rows_ok"]
                                      and res["every_rank_slice_of_gather_ok"] and res["no_device_memory_growth"])
        os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
        with open(out, "w") as f:
            json.dump(res, f, indent=1)
        print(json.dumps(res))
    sa._engine.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rows", type=int, default=8_000_000)
    ap.add_argument("--fits", type=int, default=3)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "c5_rehearsal.json"))
    a = ap.parse_args()
    if a.world == 1:
        worker(0, 1, 0, a.rows, a.out, a.fits)
        return
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(worker, args=(a.world, port, a.rows, a.out, a.fits), nprocs=a.world, join=True)


if __name__ == "__main__":
    main()
