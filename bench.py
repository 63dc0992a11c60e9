#!/usr/bin/env python3
"""
bench.py -- k-NN graph build benchmark (BASELINE.json metric: k-NN graph build time +
recall@k vs brute force).

A "step" is one `fit()` over the workload's data set, inputs already resident in HBM (the
constructor uploads; only fit() is timed).
  --gpus 1 : BASELINE configs[1]: load_strings Levenshtein, N=1600, n_anchors=15, k=25,
             p_work=0.12 on one MI355X (the metric's quoted configuration).  The line also
             carries C4 (digits Wasserstein), C3 (Euclidean N=10^6) and the pair-list kernels
             at N=16000 as extra blocks.
  --gpus N : (torch.distributed.run, one rank per GPU, RCCL) the row-sharded Euclidean build:
             BASELINE configs[2] -- N=10^6 float32 x 128, n_anchors=32, k=15, p_work=0.1 --
             with the rows sharded N/G per GPU: STRONG scaling (total work fixed), value =
             graphs/s = 1 / fit time.  The strings workload is 1600 points in a chain of
             dependent launches and does not shard; per-rank replicas of it are reported as an
             extra block.  With 8 ranks BASELINE configs[4] (N=8*10^6) is added as a block.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

INT32_VALU_PEAK_TOPS = 256 * 4 * 32 * 2.4e9 / 1e12  # 78.6 T lane-ops/s (256 CUs x 4 SIMD32 x 2.4 GHz)
HBM_PEAK_GBS = 8000.0
BF16_MFMA_PEAK_TFLOPS = 2500.0   # dense bf16 / fp16 MFMA (MI355X_MICROARCH.md; measured 2495)
LEV_OPS_PER_WORD_STEP = 17  # Myers/Hyyro recurrence, 32-bit ops per (pattern word x text symbol)


def strings_workload():
    from annchor_amd.datasets import load_strings

    X = load_strings()["X"]
    cfg = dict(n_anchors=15, n_neighbors=25, p_work=0.12, random_seed=42)
    return X, "levenshtein", None, cfg, "load_strings Levenshtein N=1600 n_anchors=15 k=25 p_work=0.12 niters=2"


def lev_work(ann, X):
    """Exact algorithmic work of the Levenshtein kernel over one fit(): every pair the
    metric was evaluated on = anchor rows + (computed, non-anchor) pairs."""
    lens = np.array([len(s) for s in X], dtype=np.int64)
    IJs, ncm = ann.IJs, ann.not_computed_mask
    anc = ann.features[:, 3] > 0
    ev = IJs[(~ncm) & (~anc)]
    A = ann.A
    ai = np.repeat(A, len(X))
    aj = np.tile(np.arange(len(X)), len(A))
    li = np.concatenate([lens[ev[:, 0]], lens[ai]])
    lj = np.concatenate([lens[ev[:, 1]], lens[aj]])
    m, n = np.minimum(li, lj), np.maximum(li, lj)
    word_steps = int((((m + 31) // 32) * n).sum())
    cells = int((m * n).sum())
    return len(li), word_steps, cells


def euclid_shard(rank, n_per_rank, d=128):
    """SURVEY.md section 8d recipe (8-d latent manifold in 128-d), sharded by rows: shard r is
    generated from its own stream so that no rank ever holds the whole set."""
    W = np.random.default_rng(1234).standard_normal((8, d))
    rng = np.random.default_rng(10_000 + rank)
    Z = rng.standard_normal((n_per_rank, 8))
    return (Z @ W + 0.05 * rng.standard_normal((n_per_rank, d))).astype(np.float32)


def pairlist_at_scale(local, n=16000):
    """The pair-list kernels at a size where HBM traffic, not launch latency, decides: N = 16000
    Euclidean float64 points whose locality keeps every pair (127 M candidate pairs, ~1 GB per
    per-pair column).  Two fits, the second one profiled with HIP events per kernel family.
    Reported per family: average launch time, algorithmic GB/s (DESIGN.md's per-pair bytes) and
    the fraction of the 8 TB/s HBM peak."""
    from annchor_amd import Annchor
    rng = np.random.default_rng(5)
    Z = rng.standard_normal((n, 6))
    X = (Z @ rng.standard_normal((6, 48)) + 0.05 * rng.standard_normal((n, 48))).astype(np.float64)
    cfg = dict(n_anchors=24, n_neighbors=15, p_work=0.05, n_samples=5000)
    Annchor(X, "euclidean", device=local, **cfg).fit()
    ann = Annchor(X, "euclidean", device=local, **cfg)
    ann._engine.prof_enable(1)
    t = time.perf_counter()
    ann.fit()
    dt = time.perf_counter() - t
    fams = {}
    for name, e in sorted(ann._engine.prof_get().items(), key=lambda kv: -kv[1]["ms"]):
        if not e["launches"]:
            continue
        us = e["ms"] / e["launches"] * 1e3
        gbs = e["alg_bytes"] / e["launches"] / us / 1e3
        fams[name] = {"avg_launch_us": round(us, 1), "launches": int(e["launches"]), "alg_GBps": round(gbs, 1),
                      "hbm_frac": round(gbs / HBM_PEAK_GBS, 4)}
    res = _scale_result(ann, n, dt, fams)
    res["pmc_source"] = scale_pmc_fractions(fams)
    res["sampler"] = type(ann.sampler).__name__ + " (the default at this size: sampler=None)"
    ann._engine.close()   # ~10 GB of device arena: release it now, not whenever the collector runs
    # the same fit with the NumPy-stream sampler forced (the default below 4 M candidate pairs): its host-side shuffle of the
    # whole pair list is what the automatic choice avoids
    Annchor(X, "euclidean", device=local, sampler="legacy", **cfg).fit()   # warm (the 10 GB arena is re-created)
    ann = Annchor(X, "euclidean", device=local, sampler="legacy", **cfg)
    t = time.perf_counter()
    ann.fit()
    res["fit_time_s_legacy_sampler"] = time.perf_counter() - t
    res["host_stage_ms_legacy_sampler"] = {k: round(v * 1e3, 1) for k, v in ann.timings.items()}
    ann._engine.close()
    return res


def levenshtein_100k_block(local, n=100000, k=15):
    """A slow metric beyond the size whose complete pair list fits (46 341 points): 100 000 clustered synthetic strings of
    ~120 symbols (annchor_amd.datasets.synthetic_string_clusters: the shape of load_strings), the candidate list thinned by the
    reference's own locality filter (3 of the 5 nearest of 60 anchors in common: ~6 x 10^8 candidates), 2 % of all pairs
    evaluated; recall against the exact rows of 100 points (one-to-all launches of the same metric kernel)."""
    from annchor_amd import Annchor, compare_neighbor_graphs
    from annchor_amd.datasets import synthetic_string_clusters
    X = synthetic_string_clusters(n)
    cfg = dict(n_anchors=60, n_neighbors=k, p_work=0.02, n_samples=5000, locality=5, loc_thresh=3)
    # two fits: the first of the process also pays for ~50 GB of first-time device allocations (0.01-1.3 s on a fresh box,
    # depending on what the device was doing before); the second, what a process that fits repeatedly sees, is the one reported
    first = None
    for rep in range(2):
        ann = Annchor(X, "levenshtein", device=local, **cfg)   # (default sampler: DeviceStratifiedSampler at this size)
        t = time.perf_counter()
        ann.fit()
        dt = time.perf_counter() - t
        if rep == 0:
            first = dt
            ann._engine.close()
    # recall on 1000 rows: exact rows from one-to-all launches of the metric kernel; 20 of those rows re-computed with the CPU
    # oracle's C Levenshtein (oracle/lev.c) so that the truth does not rest on the kernel under test alone
    rows = np.random.default_rng(5).choice(n, 1000, replace=False)
    err = 0
    z = np.zeros((1, k), dtype=np.int64)
    exact_rows = {}
    for r in rows:
        d = ann._engine.metric_pairs(np.stack([np.full(n, r), np.arange(n)], axis=1))
        if len(exact_rows) < 20:
            exact_rows[int(r)] = d.copy()
        d[r] = -1
        want = np.sort(d)[:k]
        want[0] = 0
        err += compare_neighbor_graphs((z, want[None, :]), (z, ann.neighbor_graph[1][r][None, :]), k)
    oracle_check = None
    try:
        from oracle import metrics as om

        P = om.PackedStrings(list(X))
        bad = 0
        for r, d in exact_rows.items():
            bad += int(np.sum(P.pairs(np.stack([np.full(n, r), np.arange(n)], axis=1)) != d))
        oracle_check = {"rows": len(exact_rows), "mismatching_distances": bad}
    except Exception as e:   # never at the cost of the line
        oracle_check = {"error": "%s: %s" % (type(e).__name__, e)}
    res = {"workload": "synthetic clustered strings (length ~120) Levenshtein N=%d n_anchors=60 k=%d p_work=0.02 locality=5 loc_thresh=3, "
                       "default plugins (pair-list form, candidate list thinned by the locality filter)" % (n, k),
           "fit_time_s": dt, "first_fit_time_s": first, "candidate_pairs": int(ann.n_pairs), "evals": int(ann.evals),
           "recall_at_k": 1.0 - err / (len(rows) * k), "recall_rows": int(len(rows)), "truth_rows_checked_against_oracle_lev_c": oracle_check,
           "note": "beyond 46 341 points the complete pair list (2^30 candidates) no longer fits; fit_time_s = second fit of the "
                   "process (device blocks of the first are reused), first_fit_time_s includes the first-time allocations"}
    ann._engine.close()
    return res


# kernel family (ProfScope name) -> kernels of the rocprofv3 PMC passes that belong to it
FAMILY_KERNELS = {
    "update_bounds_intersect": ("k_update_bounds_rows", "k_update_bounds"),
    "radix_select_f64": ("k_sel2_", "k_sel3_"),
    "sampler_select_by_rank": ("k_rb_",),
    "transpose_column_half": ("k_transpose_cols<true>",),
    "transpose_column_half_mask": ("k_transpose_cols<false>",),
    "topk_tie_groups": ("k_tie_",),
    "ecdf_probability": ("k_prob", "k_ecdf_index"),
    "predict_clip_label_merge": ("k_predict_merge",),
    "locality_emit_pairs": ("k_emit_pairs", "k_emit_cols"),
    "topk_split_compact": ("k_cut_",),
    "computed_neighbour_csr": ("k_comp_",),
    "bounds_dad_features": ("k_features", "k_anchor_flags"),
    "row_kth_threshold": ("k_row_thresh",),
    "locality_keep_bitmap": ("k_loc_thresh", "k_keep_bits", "k_row_prefix", "k_min_i32"),
    "guarantee_nmin_lists": ("k_gn_lists",),
    "row_topk_graph": ("k_get_nn",),
    "sampler_bin_counts": ("k_bin_counts",),
    "euclidean_pairs": ("k_euclid",),
}


def scale_pmc_fractions(fams):
    """Second view of the same launches: HBM bytes from the committed rocprofv3 PMC passes over tools/pairlist_scale.py
    (profiles/rNN*_scale_pmc_traffic.json: FETCH_SIZE / WRITE_SIZE in separate runs, KiB, FETCH doubled as the gfx950
    guide prescribes) per fit of that run, against this run's kernel times: `hbm_frac_pmc`.  The first view,
    `hbm_frac`, prices DESIGN.md's algorithmic bytes."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_scale_pmc_traffic.json")))
    if not files:
        return None
    T = json.load(open(files[-1]))
    # fits the PMC run covered: its launches of the once-per-locality kernel against this run's count for ONE fit
    sid_pmc = sum(v["launches"] for k, v in T.items() if k.replace("void ", "").startswith("k_sid"))
    fits_in_pmc_run = max(1.0, sid_pmc / max(1, fams.get("locality_sid", {}).get("launches", 1)))
    for fam, e in fams.items():
        pref = FAMILY_KERNELS.get(fam)
        if not pref:
            continue
        tot = sum(v["hbm_bytes_per_launch_corrected"] * v["launches"] for k, v in T.items()
                  if any(k.replace("void ", "").startswith(p) for p in pref))
        if tot <= 0:
            continue
        per_fit = tot / fits_in_pmc_run
        t_fit = e["avg_launch_us"] * e["launches"] * 1e-6
        e["pmc_GB_per_fit"] = round(per_fit / 1e9, 3)
        e["hbm_frac_pmc"] = round(per_fit / t_fit / 1e9 / HBM_PEAK_GBS, 4)
    return os.path.basename(files[-1])


def _scale_result(ann, n, dt, fams):
    return {"workload": "synthetic Euclidean f64 N=%d d=48 n_anchors=24 k=15 p_work=0.05 (pair-list form)" % n,
            "pairs": int(ann.n_pairs), "evals": int(ann.evals), "fit_time_s_profiled": dt,
            "host_stage_ms": {k: round(v * 1e3, 1) for k, v in ann.timings.items()}, "kernels": fams,
            "note": "default arguments: from 4 M candidate pairs sampler=None is the order-free DeviceStratifiedSampler (GPU draw); "
                    "fit_time_s_legacy_sampler = the same fit with sampler='legacy' (NumPy-stream shuffle of the pair list on one host thread)"}


def euclid_run(world, rank, local, dist_mod, n_per_rank, steps, warmup, torch, recall_rows=10000):
    """BASELINE configs[2] / [4]: synthetic Euclidean float32 (SURVEY.md 8d recipe), d=128, n_anchors=32,
    k=15, p_work=0.1, rows sharded contiguously across the ranks (streamed form).  Timed like the
    headline: barrier + synchronize on both sides of every fit, maximum over ranks."""
    from annchor_amd.streamed import SingleComm, StreamedAnnchor, TorchComm

    X = euclid_shard(rank, n_per_rank)
    # multi-rank: ONE engine kept for all fits and the collectives from inside the library on it (csrc/comm.hip: RCCL on the engine's
    # stream, a second communicator for the rows' all-gather beside the anchor rounds, pre-flight, dead-peer timeout); if that
    # cannot be set up on every rank, torch.distributed on device pointers (make_comm says so).  ANNCHOR_BENCH_COMM=torch forces
    # the torch path.  Neither has run on more than one GPU before the driver's scaling run.
    from annchor_amd.streamed import make_comm

    comm, shared_engine = SingleComm(), None
    if world > 1:
        if dist_mod.get_backend() == "nccl" and os.environ.get("ANNCHOR_BENCH_COMM") != "torch":
            from annchor_amd import _native

            shared_engine = _native.Engine(local)
            comm = make_comm(shared_engine)
            if getattr(comm, "backend", "") != "rccl":
                shared_engine.close()
                shared_engine = None
        else:
            comm = TorchComm()
    k, pw, na = 15, 0.1, 32
    times, last = [], None
    red_dev = "cuda" if (dist_mod is not None and dist_mod.get_backend() == "nccl") else "cpu"
    for it in range(warmup + steps):
        sa = StreamedAnnchor(X, n_anchors=na, n_neighbors=k, p_work=pw, base=rank * n_per_rank, comm=comm, device=local,
                             **({"engine": shared_engine} if shared_engine is not None else {}))
        sa._engine.prof_enable(True)
        torch.cuda.synchronize()
        if dist_mod is not None:
            dist_mod.barrier()
        t0 = time.perf_counter()
        sa.fit()
        torch.cuda.synchronize()
        if dist_mod is not None:
            dist_mod.barrier()
        dt = time.perf_counter() - t0
        if dist_mod is not None:
            tt = torch.tensor([dt], device=red_dev, dtype=torch.float64)
            dist_mod.all_reduce(tt, op=dist_mod.ReduceOp.MAX)
            dt = float(tt.item())
        if it >= warmup:
            times.append(dt)
        if last is not None and shared_engine is None:
            last._engine.close()
        last = sa
    tiles_all = last.tile_evals
    if dist_mod is not None:
        tt = torch.tensor([float(last.tile_evals)], device=red_dev, dtype=torch.float64)
        dist_mod.all_reduce(tt, op=dist_mod.ReduceOp.SUM)
        tiles_all = int(tt.item())
    out = None
    if rank == 0:
        from annchor_amd import compare_neighbor_graphs

        prof = last._engine.prof_get()
        gemm = prof.get("stream_tile_gemm_topk", dict(ms=0.0, launches=1))
        gemm_s = gemm["ms"] / max(1, gemm["launches"]) * 1e-3
        # tile evaluations of the tile phase on this rank = all of this rank's minus its join chunks (the
        # join kernel is timed separately); its flops = evaluations x 128 x 128 x 2 x d
        nt_all = last.n_tiles_total
        total, tile_budget, per_pass = last._budget(nt_all)
        # recall on recall_rows rows of rank 0's shard: exact k-NN over ALL shards from the tile kernel with
        # the full budget (streamed query), i.e. GPU brute force of that subset
        m = min(recall_rows, n_per_rank)
        rows = np.sort(np.random.default_rng(99).choice(n_per_rank, m, replace=False))
        t_q = time.perf_counter()
        ti, td = last.query(X[rows], nn=k, p_work=1.0)
        truth_s = time.perf_counter() - t_q
        err = compare_neighbor_graphs((ti, td), (last.neighbor_graph[0][rows], last.neighbor_graph[1][rows]), k)
        # the truth above comes from the library's own tile kernel (full budget).  The HEADLINE recall is measured against an
        # INDEPENDENT float64 brute force -- plain torch float64 matrix products on the GPU, none of this build's kernels -- on
        # every 4th of those rows (2 500 at the default), all shards regenerated on this rank, up to 2 x 10^6 total rows
        # (N = 8 x 10^6: tests/test_c5_gpu.py does it on 1 000 rows)
        indep = None
        if world * n_per_rank <= 2_000_000:
            t_i = time.perf_counter()
            sel = np.arange(0, len(rows), 4)
            sub = rows[sel]
            shards = [X if r == rank else euclid_shard(r, n_per_rank) for r in range(world)]
            bd = truth_f64_torch(X[sub], shards, k, torch)
            del shards
            bd[:, 0] = 0.0   # the row itself (cancellation noise of the expanded form)
            gi = last.neighbor_graph[0][sub]
            e_np = compare_neighbor_graphs((gi, bd), (gi, last.neighbor_graph[1][sub]), k)
            e_tt = compare_neighbor_graphs((gi, bd), (gi, td[sel]), k)
            indep = {"rows": int(len(sub)), "recall_at_k": 1.0 - e_np / (float(len(sub)) * k),
                     "truth": "float64 brute force of %d rows against all %d rows (torch float64 on the GPU; independent of the build's kernels)"
                              % (len(sub), world * n_per_rank),
                     "tile_kernel_truth_vs_float64_truth_errors": int(e_tt), "seconds": round(time.perf_counter() - t_i, 1)}
        fit_s = float(np.mean(times))
        n_total = world * n_per_rank
        out = {
            "workload": "synthetic Euclidean f32 (8-d latent in 128-d, SURVEY 8d recipe) N=%d d=128 n_anchors=32 k=15 p_work=0.1, "
                        "streamed form, rows sharded %d per GPU over %d GPU(s)" % (n_total, n_per_rank, world),
            "fit_time_s": fit_s, "graphs_per_s": 1.0 / fit_s, "rows_per_s": n_total / fit_s,
            # headline: the independent float64 figure when it was taken; the tile-kernel-truth figure beside it
            "recall_at_k": indep["recall_at_k"] if indep else 1.0 - err / (float(m) * k),
            "recall_rows": indep["rows"] if indep else int(m),
            "recall_truth": indep["truth"] if indep else
                            "exact k-NN of %d fixed rows of rank 0's shard over all shards (tile kernel, full budget, %.2f s)" % (m, truth_s),
            "recall_at_k_tile_kernel_truth": {"recall_at_k": 1.0 - err / (float(m) * k), "rows": int(m),
                                              "truth": "exact k-NN of %d fixed rows over all shards from the build's own tile kernel with the full budget (%.2f s)"
                                                       % (m, truth_s)},
            "recall_independent_float64": indep,
            "budget_tiles_per_row_tile": {"total": total, "tile_phase": tile_budget, "per_join_pass": per_pass},
            "tile_evals_all_ranks": int(tiles_all),
            "tile_fraction_of_brute_force": tiles_all / float(nt_all) / float(nt_all),
            "stage_s_rank0": {a: round(b, 4) for a, b in last.timings.items()},
            "kernels_ms_rank0": {kk: round(v["ms"], 3) for kk, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
        }
        if gemm_s > 0:
            tile_phase_evals, join_chunks = last._engine.stream_last_counts()
            out["join_chunks_rank0"] = int(join_chunks)
            flops = tile_phase_evals * 128.0 * 128.0 * 2.0 * 128.0
            kind, guard_rows = last._engine.stream_last_kernel(with_guard=True)
            out["split_kernel_guard_rows_rank0"] = int(guard_rows)   # rows whose list boundary is within the split products' error (> 0.5 %: exact-f32 rerun)
            if kind == 0:
                out["roofline"] = {"kernel": "stream_tile_gemm_topk (k_st_knn, v_mfma_f32_32x32x2_f32)", "bound": "mfma",
                                   "achieved": flops / gemm_s / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                                   "frac": flops / gemm_s / 1e12 / 157.3, "traffic": pmc_traffic("k_st_knn<"),
                                   "tile_pairs": int(tile_phase_evals),
                                   "note": "algorithmic flops = tile pairs of the tile phase (<= its budget x row tiles of rank 0) "
                                           "x 128 x 128 x 2 x d; peak = dense f32 MFMA"}
            elif hasattr(last._engine, "stream_last_tile_kernels") and last._engine.stream_last_tile_kernels()[0] and "stream_tile_two_stage_kernel" in prof:
                # the two-stage tile phase (csrc/knnh.hip): k_st_knnh behind k_st_knnbf's warm-up of 33 tiles per row tile.  The dominant
                # kernel streams the fp16 hi halves of every evaluated column tile ONCE: 128 columns x 256 B + 128 norms = 33 280 B per tile
                # pair -- its algorithmic bytes -- and 8 MFMAs per 32 x 32 x 128 block (one product per pair of hi halves)
                tk = prof["stream_tile_two_stage_kernel"]
                t_h = tk["ms"] / max(1, tk["launches"]) * 1e-3
                warm_pairs = min(int(tile_phase_evals), int(last.n_tiles_total if world == 1 else -(-last.n_tiles_total // world)) * 33)
                pairs_h = max(0, int(tile_phase_evals) - warm_pairs)
                bytes_h = pairs_h * 33280.0
                out["roofline"] = {"kernel": "stream_tile_gemm_topk / k_st_knnh (two-stage: fp16 hi-only MFMA filter with a rigorous bound + exact float32 differences for the survivors)",
                                   "bound": "hbm", "achieved": bytes_h / t_h / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_h / t_h / 1e9 / HBM_PEAK_GBS,
                                   "traffic": pmc_traffic("k_st_knnh", largest=True), "tile_pairs": pairs_h, "kernel_ms": round(t_h * 1e3, 3),
                                   "warm_up": {"kernel": "k_st_knnbf<128, 16> with a budget of 33 tiles per row tile", "tile_pairs_at_most": warm_pairs,
                                               "ms": round((gemm_s - t_h) * 1e3, 3), "traffic": pmc_traffic("k_st_knnbf<128, 16, false", largest=True)},
                                   "mfma_view": {"bound": "mfma", "achieved": pairs_h * 128.0 * 128.0 * 2.0 * 128.0 / t_h / 1e12, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                                 "frac": pairs_h * 128.0 * 128.0 * 2.0 * 128.0 / t_h / 1e12 / BF16_MFMA_PEAK_TFLOPS,
                                                 "note": "issued = algorithmic here: one fp16 MFMA product per coordinate pair (the split-fp16 kernel issued three)"},
                                   "note": "algorithmic bytes = tile pairs of k_st_knnh x 33 280 B (the hi halves of 128 columns + their norms, fetched once each); the "
                                           "kernel is a per-workgroup latency chain (barrier -> 16 MFMAs on 16 LDS operand reads -> choice of the next tile -> requests, per 64 "
                                           "columns: ~4200 ticks where the matrix pipe needs 830 and HBM 2220 -- tools/microbench/stream_seq.hip, DESIGN.md section 7), on "
                                           "neither roof; traffic = the PMC passes' bytes per launch"}
            else:
                name = "k_st_knnbf"
                out["roofline"] = {"kernel": "stream_tile_gemm_topk (%s: split-fp16 tile GEMMs, 3 x v_mfma_f32_32x32x16_f16 per 16 dimensions)" % name,
                                   "bound": "mfma", "achieved": 3.0 * flops / gemm_s / 1e12, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": 3.0 * flops / gemm_s / 1e12 / BF16_MFMA_PEAK_TFLOPS, "traffic": pmc_traffic(name, largest=True),
                                   "tile_pairs": int(tile_phase_evals),
                                   "hbm_view": None,
                                   "f32_equivalent": {"achieved": flops / gemm_s / 1e12, "unit": "TFLOP/s",
                                                      "of_the_f32_mfma_peak_157.3": flops / gemm_s / 1e12 / 157.3},
                                   "note": "achieved = MFMA flops issued: every float is split into fp16 hi + lo (centred, scaled by a power of two) and a dot product is "
                                           "hi.hi + hi.lo + lo.hi, i.e. 3 x (tile pairs x 128 x 128 x 2 x d); peak = dense fp16 / bf16 MFMA "
                                           "(MI355X_MICROARCH.md: ~2.5 PFLOP/s).  The split products are as accurate as the f32 MFMA stream (2^-22 |x||y|, tools/microbench/f16_split.hip) and select K + 2 columns per row; their "
                                           "exact float32 distances decide the K that are kept, so the graph is the exact-f32 kernel's "
                                           "(f32_equivalent = the algorithmic f32 flops of the same tile pairs / the same time)"}
    if out and out.get("roofline") and out["roofline"].get("traffic") and "hbm_view" in out["roofline"]:
        r = out["roofline"]
        # the same launch against the HBM roof: bytes the PMC passes saw leave the fabric per launch / this run's kernel time
        r["hbm_view"] = {"bound": "hbm", "traffic_GB": round(r["traffic"] / 1e9, 1), "GBps": round(r["traffic"] / gemm_s / 1e9, 1),
                         "peak": HBM_PEAK_GBS, "frac": r["traffic"] / gemm_s / 1e9 / HBM_PEAK_GBS,
                         "streamed_operand_GB": round(tile_phase_evals * 65536.0 / 1e9, 1),
                         "note": "64 KB of split-fp16 column operands per tile pair, every one fetched from beyond L2: the kernel is "
                                 "closer to the HBM roof than to the MFMA roof; traffic from the committed PMC pass"}
    last._engine.close()
    return out


def truth_f64_torch(Xq, shards, k, torch, block=250_000):
    """k smallest float64 distances of every row of Xq to all rows of `shards` (host float32 arrays): torch float64 on the GPU
    (expanded form with float64 norms), column blocks merged by torch.topk -- a brute force that shares nothing with the build."""
    Q = torch.from_numpy(Xq).cuda().double()
    qq = (Q * Q).sum(1)
    best = torch.full((len(Xq), k), float("inf"), dtype=torch.float64, device="cuda")
    for S in shards:
        for c0 in range(0, len(S), block):
            C = torch.from_numpy(S[c0:c0 + block]).cuda().double()
            d2 = (qq[:, None] + (C * C).sum(1)[None, :] - 2.0 * (Q @ C.T)).clamp_(min=0.0)
            small = torch.topk(d2, min(k, d2.shape[1]), dim=1, largest=False).values
            best = torch.topk(torch.cat([best, small], dim=1), k, dim=1, largest=False).values
            del C, d2
    out = torch.sqrt(torch.sort(best, dim=1).values).cpu().numpy()
    del Q, best
    torch.cuda.empty_cache()
    return out


def _latest_profile(suffix):
    """Newest committed profiles/rNN*_<suffix> (rounds sort lexicographically)."""
    import glob

    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_" + suffix))
                   if "_scale_" not in os.path.basename(f))   # (the *_scale_* files belong to tools/pairlist_scale.py)
    return files[-1] if files else None


def pmc_traffic(kernel_prefix, largest=False):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/rNN_pmc_traffic.json: separate
    --pmc FETCH_SIZE / WRITE_SIZE runs of this bench, KiB units, FETCH doubled as the gfx950 guide prescribes).
    largest=True: of the matching instantiations the one with the most bytes per launch (the streamed tile kernel is
    also launched, with another list capacity, by the recall check's full-budget query)."""
    try:
        T = json.load(open(_latest_profile("pmc_traffic.json")))
        tot = n = 0
        best = 0
        for name, v in T.items():  # launch-weighted over every instantiation of the kernel
            if name.replace("void ", "").startswith(kernel_prefix):
                tot += v["hbm_bytes_per_launch_corrected"] * v["launches"]
                n += v["launches"]
                best = max(best, int(v["hbm_bytes_per_launch_corrected"]))
        if n:
            return best if largest else int(tot / n)
    except Exception:
        pass
    return None


def pmc_valu_busy():
    """VALU-busy share of the Levenshtein launches from the committed PMC pass (profiles/rNN_pmc_lev.json,
    tools/pmc_lev2.sh / pmc_lev.sh): SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / (kernel cycles x 1024 SIMDs)."""
    try:
        T = json.load(open(_latest_profile("pmc_lev.json")))
        return {k: round(v["SQ_ACTIVE_INST_VALU"] * 4 / (v["GRBM_GUI_ACTIVE"] / 8 * 1024), 3)
                for k, v in T.items() if "k_lev" in k and "classify" not in k and "SQ_ACTIVE_INST_VALU" in v and v.get("GRBM_GUI_ACTIVE")}
    except Exception:
        return None


def cpu_baseline(X, cfg):
    """The oracle (CPU restatement) timed on this box: one full fit() of the same
    workload (~10 s): C Levenshtein (Myers, OpenMP over all cores) + NumPy pipeline."""
    from oracle import annchor_oracle as O
    from oracle import metrics as om

    P = om.PackedStrings(list(X))
    P.pairs(np.array([[0, 1]]))  # build / warm the C library
    t = time.perf_counter()
    ora = O.OracleAnnchor(len(X), P.pairs, **cfg).fit()
    dt = time.perf_counter() - t
    return dict(value=1.0 / dt, unit="graphs/s", fit_time_s=dt, cores=int(getattr(P, "threads", os.cpu_count())),
                kind="port", port_of="the oracle: NumPy pipeline + C/OpenMP metric (a NumPy-pipeline port, not a C++ restatement of the pipeline)",
                sample="1 full fit() of the same workload (metric in C/OpenMP, pipeline in NumPy)",
                evals=int(ora.evals))


def cpu_baseline_euclid(n, k=15, budget_s=15.0):
    """CPU baseline for the Euclidean workload: the exact k-NN rows of a bounded sample of rows against all N columns
    (what the reference's BruteForce does per row, annchor.py:1004-1023, here as blocked float32 NumPy/BLAS GEMMs +
    argpartition on all cores), extrapolated to N rows.  The reference's Annchor.fit() cannot run this size at all
    (its pair list alone is ~24 TB at N = 10^6)."""
    X = euclid_shard(0, n)
    sq = (X * X).sum(axis=1)
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s and done < n:
        R = X[done:done + 256]
        d2 = sq[done:done + 256, None] + sq[None, :] - 2.0 * (R @ X.T)
        part = np.argpartition(d2, k, axis=1)[:, :k]
        np.take_along_axis(d2, part, axis=1).sort(axis=1)
        done += len(R)
    dt = time.perf_counter() - t0
    rows_per_s = done / dt
    return dict(value=rows_per_s / n, unit="graphs/s", rows_per_s=rows_per_s, cores=int(os.cpu_count()), kind="port",
                sample="exact k-NN of %d rows against all %d columns in %.1f s (blocked f32 NumPy GEMM + argpartition), "
                       "extrapolated to all rows" % (done, n, dt))


def strings_run(args, steps, warmup, world, rank, local, dist, torch, all_cpus, affinity):
    """BASELINE configs[1] (the metric's quoted configuration) on every rank; rank 0 returns the line."""
    from annchor_amd import Annchor, compare_neighbor_graphs
    from annchor_amd import _native as _nat

    class _A:   # the old body reads args.steps / args.warmup
        pass
    a_ = _A()
    a_.__dict__.update(vars(args))
    a_.steps, a_.warmup = steps, warmup
    args = a_
    X, metric, kwargs, cfg, workload = strings_workload()
    # constructors (engine creation, upload, plumbing smoke test) are outside the timed region
    anns, ctor_s = [], []
    for _ in range(args.warmup + args.steps):
        t_c = time.perf_counter()
        anns.append(Annchor(X, metric, func_kwargs=kwargs, device=local, **cfg))
        ctor_s.append(time.perf_counter() - t_c)
    for a in anns[:args.warmup]:
        a.fit()
    timed = anns[args.warmup:]
    if not args.no_kernel_events:
        # inside the timed region only the metric kernels (the roofline kernel) carry HIP events
        # (10 events per fit: one pair around the anchor rounds, one per pair-list launch); the table of all kernel families comes from untimed fits below
        for a in timed:
            a._engine.prof_enable(2)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    import gc

    gc.collect()
    gc.disable()   # no cyclic-GC pause inside the timed region (the fits create no reference cycles to speak of)
    sync()
    t0 = time.perf_counter()
    for a in timed:
        a.fit()  # ends with the D2H of the graph on the engine's stream => complete
    for a in timed:
        a._engine.synchronize()
    sync()
    elapsed = time.perf_counter() - t0
    gc.enable()
    red_dev = "cuda" if args.backend == "nccl" else "cpu"
    if dist is not None:
        tt = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    out = None
    if rank == 0:
        ann = timed[-1]
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            "metric": "knn_graph_builds_per_s (1 / Annchor.fit() wall-clock; BASELINE: k-NN graph build time + recall@k)",
            "value": world * args.steps / elapsed,
            "unit": "graphs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "fit_time_s": ms_per_step / 1e3,
            "higher_is_better": True,
            "scaling": "n/a" if world == 1 else "weak",   # (one GPU: nothing scales; N > 1 replicas: per-GPU work fixed)
            "vs_baseline": None,
            "dtype": "int32 bit-vectors (Levenshtein) + f64 (bounds/regression/selection)",
            "data": "reference fixture (annchor/data/edit_data.npz: 1600 synthetic strings, length 378-594)",
            "config": {"workload": workload, "graphs_per_step_per_gpu": 1,
                       "parallelism": "one GPU" if world == 1 else "independent graph build per GPU (replicas)"},
            "rng_stream_cache": "off (ANNCHOR_RNG_NO_CACHE=1: every fit regenerates its MT19937 streams)",
            "device": ann._engine.device_name(),
            "cpu_affinity": affinity or "unbound",
            # outside the timed region (the contract times fit() with the inputs resident): string encoding,
            # context creation and the 1 MB upload.  The first constructions of a process create the context
            # shells the later ones reuse (they are all alive at once here), hence the minimum as well.
            "constructor_ms": {"median": round(float(np.median(ctor_s)) * 1e3, 3), "min": round(float(np.min(ctor_s)) * 1e3, 3)},
        }
        # ---- recall vs brute force (golden truth regenerated with the oracle metric)
        G = np.load(os.path.join(ROOT, "tests", "golden", "strings_full.npz"))
        truth = (G["truth_idx"].astype(np.int64), G["truth_dist"].astype(np.float64))
        k = cfg["n_neighbors"]
        err = compare_neighbor_graphs(truth, ann.neighbor_graph, k)
        out["errors_vs_bruteforce"] = int(err)
        out["recall_at_k"] = 1.0 - err / (k * len(X))
        out["evals"] = int(ann.evals)
        out["host_stage_ms"] = {s: round(v * 1e3, 3) for s, v in ann.timings.items()}
        # ---- per-kernel device time: the roofline kernel from the timed region (HIP events on the
        # engine stream), every other kernel family from untimed fits of the same workload
        if not args.no_kernel_events:
            agg = {}
            for a in timed:
                for name, e in a._engine.prof_get().items():
                    g = agg.setdefault(name, dict(ms=0.0, launches=0, alg_bytes=0.0, fits=args.steps, timed=True))
                    g["ms"] += e["ms"]; g["launches"] += e["launches"]; g["alg_bytes"] += e["alg_bytes"]
            n_extra = 3
            for _ in range(n_extra):
                extra = Annchor(X, metric, func_kwargs=kwargs, device=local, **cfg)
                extra._engine.prof_enable(1)
                extra.fit()
                for name, e in extra._engine.prof_get().items():
                    if name in agg and agg[name]["timed"]:
                        continue   # measured in the timed region
                    g = agg.setdefault(name, dict(ms=0.0, launches=0, alg_bytes=0.0, fits=n_extra, timed=False))
                    g["ms"] += e["ms"]; g["launches"] += e["launches"]; g["alg_bytes"] += e["alg_bytes"]
            kernels = {}
            for name, g in sorted(agg.items(), key=lambda kv: -kv[1]["ms"] / max(1, kv[1]["fits"])):
                if g["launches"] == 0:
                    continue
                avg_ms = g["ms"] / g["launches"]
                gbs = g["alg_bytes"] / g["launches"] / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
                kernels[name] = dict(ms_per_fit=round(g["ms"] / g["fits"], 4), launches_per_fit=g["launches"] / g["fits"],
                                     timed_region=g["timed"],
                                     avg_launch_us=round(avg_ms * 1e3, 2), alg_GBps=round(gbs, 1),
                                     hbm_frac=round(gbs / HBM_PEAK_GBS, 4))
                if gbs > HBM_PEAK_GBS:
                    # more algorithmic bytes per second than HBM delivers: the operands (the computed-neighbour lists at this
                    # size: a few MB) are re-read from L2 / MALL, so the byte model is not HBM traffic -- no HBM fraction
                    kernels[name]["hbm_frac"] = None
                    kernels[name]["cache_resident"] = True
            out["kernels"] = kernels
            out["device_ms_per_fit"] = round(sum(v["ms_per_fit"] for v in kernels.values()), 3)
            dom = next(iter(kernels))
            if dom == "levenshtein_pairs":
                npairs, word_steps, cells = lev_work(ann, X)
                lev_s = agg[dom]["ms"] / args.steps * 1e-3
                ach = word_steps * LEV_OPS_PER_WORD_STEP / lev_s / 1e12
                out["roofline"] = {
                    "kernel": dom, "bound": "valu_int32", "achieved": ach, "peak": INT32_VALU_PEAK_TOPS,
                    "unit": "Tops/s", "frac": ach / INT32_VALU_PEAK_TOPS, "traffic": pmc_traffic("k_lev"),
                    "traffic_source": "committed rocprofv3 PMC pass of this command (%s), not measured in this run"
                                      % os.path.basename(_latest_profile("pmc_traffic.json") or "none"),
                    "gcups": cells / lev_s / 1e9, "pairs_per_fit": npairs, "word_steps_per_fit": word_steps,
                    # the same kernel priced against the HBM roof, for readers who want that view: algorithmic bytes
                    # (both strings of every pair, once) / kernel time -- tiny by construction, see `note`
                    "hbm_view": {"bound": "hbm", "achieved": kernels[dom]["alg_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": kernels[dom]["hbm_frac"]},
                    "valu_busy_pmc": pmc_valu_busy(),
                    "note": "integer-ALU bound (string pool is 1 MB, cache resident): HBM fraction is not meaningful "
                            "for this kernel; the HBM-bound pair-list kernels are listed under `kernels`.  `peak` is the "
                            "nominal 2-cycle SIMD-32 rate; these integer ops issue at ~4 cycles per wave instruction "
                            "(tools/microbench/valu_peak.hip); `valu_busy_pmc` = the committed PMC pass over isolated launches (pair lists at the issue ceiling, the one-wave-deep anchor rounds ~20 %)",
                }
            else:
                g = kernels[dom]
                out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": g["alg_GBps"], "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": g["hbm_frac"], "traffic": None}
        if not args.no_cpu_baseline and world == 1:
            os.sched_setaffinity(0, all_cpus)   # the CPU baselines get every core of the host, not one NUMA node
            out["cpu_baseline"] = cpu_baseline(X, cfg)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
            try:
                out["cpu_baseline_python_metric"] = cpu_baseline_python_metric(local)
            except Exception as e:   # never at the cost of the line
                out["cpu_baseline_python_metric"] = {"error": "%s: %s" % (type(e).__name__, e)}
            if affinity:
                _nat.bind_to_device_numa(local)

    return out


def device_sampler_block(local):
    """The headline workload with the DeviceStratifiedSampler plugin (order-free hashed choice on the GPU)
    in place of the default sampler's NumPy-stream shuffle: a different random sample, so a different
    draw of the error count; not the headline (the default sampler is the reference's)."""
    from annchor_amd import Annchor, compare_neighbor_graphs
    from annchor_amd.samplers import DeviceStratifiedSampler

    X, metric, kwargs, cfg, workload = strings_workload()
    anns = [Annchor(X, metric, func_kwargs=kwargs, device=local, sampler=DeviceStratifiedSampler(), **cfg) for _ in range(23)]
    ts = []
    for a in anns:
        t = time.perf_counter()
        a.fit()
        ts.append(time.perf_counter() - t)
    G = np.load(os.path.join(ROOT, "tests", "golden", "strings_full.npz"))
    truth = (G["truth_idx"].astype(np.int64), G["truth_dist"].astype(np.float64))
    err = compare_neighbor_graphs(truth, anns[-1].neighbor_graph, cfg["n_neighbors"])
    res = {"workload": workload + ", sampler=DeviceStratifiedSampler()", "fit_time_s": float(np.median(ts[3:])),
           "pinning": "self-pinned: the oracle this sampler is tested against (oracle.hashed_stratified_sample) restates the build's own "
                      "plugin -- the reference draws with numba's RNG and has no counterpart; its acceptance test is statistical "
                      "(24 seeds, error counts against the legacy sampler's: tests/test_gpu_plugins.py)",
           "errors_vs_bruteforce": int(err), "host_stage_ms": {k: round(v * 1e3, 3) for k, v in anns[-1].timings.items()}}
    for a in anns:
        a._engine.close()
    return res


def c4_block(local):
    """BASELINE configs[3]: load_digits Wasserstein (exact EMD), N=1797, n_anchors=20, k=25, p_work=0.16."""
    from annchor_amd import Annchor, compare_neighbor_graphs
    from annchor_amd.datasets import load_digits

    d = load_digits()
    cfg = dict(n_anchors=20, n_neighbors=25, n_samples=5000, p_work=0.16, random_seed=42)
    mk = lambda: Annchor(d["X"], "wasserstein", func_kwargs={"cost_matrix": d["cost_matrix"]}, device=local, **cfg)  # noqa: E731
    mk().fit()
    ts = []
    for _ in range(3):
        ann = mk()
        ann._engine.prof_enable(1)
        t = time.perf_counter()
        ann.fit()
        ts.append(time.perf_counter() - t)
    prof = ann._engine.prof_get()
    emd = prof.get("wasserstein_pairs", prof.get("emd_pairs", None))
    err = compare_neighbor_graphs(d["neighbor_graph"], ann.neighbor_graph, 25)
    res = {"workload": "load_digits Wasserstein (exact EMD, 8x8 images) N=1797 n_anchors=20 k=25 n_samples=5000 p_work=0.16 niters=2",
           "fit_time_s": float(np.median(ts)), "evals": int(ann.evals),
           "errors_vs_stored_exact_graph": int(err), "recall_at_k": 1.0 - err / (25.0 * 1797),
           "reference_published": {"fit_time_s": 21.3, "errors": 8, "us_per_metric_call": 203,
                                   "source": "doc/user_guide.rst:104,189-209 (other hardware)"},
           "kernels_ms": {kk: round(v["ms"], 3) for kk, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:6]}}
    if emd and emd["ms"] > 0:
        res["metric_kernel_us_per_pair"] = emd["ms"] * 1e3 / max(1, ann.evals)
    else:
        top = max(prof.items(), key=lambda kv: kv[1]["ms"])
        res["metric_kernel_us_per_pair"] = top[1]["ms"] * 1e3 / max(1, ann.evals)
        res["metric_kernel"] = top[0]
    # ---- the exact-OT kernel against its ceiling: it is neither HBM nor MFMA work (operands in LDS, wave-uniform control
    # flow, DPP reductions) -- the bound is instruction issue.  Instruction counts and VALU-busy cycles come from the
    # committed rocprofv3 PMC pass over one fit of this workload (tools/pmc_emd.sh -> profiles/rNN*_pmc_emd.json).
    try:
        f = _latest_profile("pmc_emd.json")
        P = json.load(open(f))
        solves = float(ann.evals)
        cyc = P["GRBM_GUI_ACTIVE"] / 8.0                      # (summed over the 8 XCDs)
        valu, salu, lds = P["SQ_INSTS_VALU"] / solves, P["SQ_INSTS_SALU"] / solves, P["SQ_INSTS_LDS"] / solves
        kernel_s = (emd["ms"] if emd else 0.0) * 1e-3
        peak_wave_instr = 256 * 4 * 2.4e9 / 2.0             # nominal: one wave64 VALU instruction per SIMD per 2 cycles
        res["roofline_issue"] = {
            "kernel": "wasserstein_pairs (k_emd_ns: transportation simplex on a spanning-tree basis, one lane per node, one wave per solve)",
            "bound": "valu_issue",
            "valu_instr_per_solve": round(valu), "salu_instr_per_solve": round(salu), "lds_instr_per_solve": round(lds),
            "valu_busy_pmc": round(P["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * 1024), 3),
            "any_inst_busy_pmc": round(P["SQ_ACTIVE_INST_ANY"] * 4 / (cyc * 1024), 3),
            "achieved": solves / kernel_s / 1e6 if kernel_s > 0 else None, "unit": "M solves/s",
            "peak": peak_wave_instr / valu / 1e6, "frac": (solves / kernel_s) / (peak_wave_instr / valu) if kernel_s > 0 else None,
            "source": os.path.basename(f),
            "note": "peak = nominal VALU issue rate (1024 SIMDs x 2.4 GHz / 2 cycles) / VALU instructions per solve (PMC pass of this "
                    "workload); the 20 anchor rounds are launches of 1797 solves that last as long as their slowest solve (latency, "
                    "not issue), the two refinement launches (~110 000 solves each) are the issue-bound part"}
    except Exception as e:   # no PMC file committed yet
        res["roofline_issue"] = {"error": "%s: %s" % (type(e).__name__, e)}
    ann._engine.close()
    return res


def py_myers_levenshtein(a, b):
    """A user-style Python metric: Myers / Hyyro bit-vector edit distance on Python's big integers (pattern a as one
    len(a)-bit integer, one step per symbol of b; ~0.5 ms per pair of 500-symbol strings -- the textbook two-row DP in
    pure Python costs ~60 ms per pair, which would make configs[0] at N = 1600 a 40 s leg on 256 cores)."""
    m = len(a)
    if m == 0:
        return float(len(b))
    peq = {}
    for i, ch in enumerate(a):
        peq[ch] = peq.get(ch, 0) | (1 << i)
    mask = (1 << m) - 1
    top = 1 << (m - 1)
    pv, mv, score = mask, 0, m
    for ch in b:
        eq = peq.get(ch, 0)
        xv = eq | mv
        xh = (((eq & pv) + pv) ^ pv) | eq
        ph = mv | (~(xh | pv) & mask)
        mh = pv & xh
        if ph & top:
            score += 1
        elif mh & top:
            score -= 1
        ph = ((ph << 1) | 1) & mask
        mh = (mh << 1) & mask
        pv = mh | (~(xv | ph) & mask)
        mv = ph & xv
    return float(score)


def cpu_baseline_python_metric_leg(X, local, n=None):
    """CPU baseline #2 (BASELINE.md section 3 / configs[0] "plumbing"): Annchor(X, python_callable) at configs[0] AS WRITTEN
    (load_strings, N = 1600, n_anchors = 15, k = 25, p_work = 0.12: 158 626 metric evaluations) -- the metric is an arbitrary
    Python function evaluated on the HOST through the chunked joblib get_exact_ijs (annchor_amd/utils.py; the reference's
    per-pair form: utils.py:152-175), everything downstream of it on the GPU."""
    from annchor_amd import Annchor

    Xs = np.array(list(X if n is None else X[:: max(1, len(X) // n)][:n]))
    cfg = dict(n_anchors=15, n_neighbors=25, n_samples=5000, p_work=0.12, random_seed=42) if n is None else \
        dict(n_anchors=6, n_neighbors=8, n_samples=300, p_work=0.3, random_seed=42)
    t = time.perf_counter()
    ann = Annchor(Xs, py_myers_levenshtein, device=local, **cfg)
    t_ctor = time.perf_counter() - t
    ann.fit()
    dt = time.perf_counter() - t
    dev = Annchor(Xs, "levenshtein", device=local, **cfg).fit()
    same = bool(np.array_equal(ann.neighbor_graph[1], dev.neighbor_graph[1]) and np.array_equal(ann.neighbor_graph[0], dev.neighbor_graph[0]))
    st = getattr(ann.get_exact_ijs, "state", {})
    return {"value": ann.evals / dt, "unit": "metric evaluations/s", "fit_time_s": dt, "constructor_s": t_ctor, "evals": int(ann.evals),
            "cores": int(os.cpu_count()), "workers_used": int(st.get("workers", 0)), "metric_s_per_pair": st.get("t_pair"),
            "kind": "port", "sample": "Annchor(load_strings X[:%d], python big-integer Myers Levenshtein callable, n_anchors=%d k=%d p_work=%g) "
                                      "incl. constructor and pool start: host metric via chunked joblib (loky), pipeline on the GPU"
                                      % (len(Xs), cfg["n_anchors"], cfg["n_neighbors"], cfg["p_work"]),
            "graph_equals_device_metric_run": same}


def cpu_baseline_python_metric(local, budget_s=120):
    """The python-metric leg in a child process with a hard time limit: joblib's loky start-up (one worker per core) normally takes
    ~15 s of the leg's ~18 s, but on one box of the pool it did not come up at all (TimeoutError after the constructor's 30 s probe,
    minutes of wall-clock before that) -- the bench line must not wait for it."""
    import subprocess

    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--python-metric-leg", str(int(local))], capture_output=True, text=True,
                           timeout=budget_s)
    except subprocess.TimeoutExpired:
        return {"error": "abandoned after %d s (joblib / loky workers did not come up on this host)" % budget_s}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": "child exited %d: %s" % (r.returncode, r.stderr.strip().splitlines()[-1][:200] if r.stderr.strip() else "")}
    return json.loads(lines[-1])


_REAL_STDOUT = None


def emit(obj):
    """The ONE JSON line, on the process's original stdout (see main: everything else is sent to stderr)."""
    out = _REAL_STDOUT or sys.stdout
    out.write(json.dumps(obj) + "\n")
    out.flush()


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--python-metric-leg":
        from annchor_amd.datasets import load_strings

        res = cpu_baseline_python_metric_leg(load_strings()["X"], int(sys.argv[2]))
        print(json.dumps(res), flush=True)
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not record per-kernel HIP events")
    ap.add_argument("--no-euclid", action="store_true", help="skip the secondary row-sharded Euclidean workload")
    ap.add_argument("--no-scale", action="store_true", help="skip the pair-list kernel table at N=16000 (127 M pairs)")
    ap.add_argument("--euclid-rows", type=int, default=1_000_000, help="TOTAL rows of the Euclidean workload (sharded over the GPUs)")
    ap.add_argument("--no-c5", action="store_true", help="8 ranks: skip the BASELINE configs[4] block (N = 8 000 000)")
    ap.add_argument("--euclid-timeout", type=int, default=600, help="seconds before the secondary workload is abandoned")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 (nccl = RCCL; gloo only to rehearse the multi-rank flow)")
    ap.add_argument("--no-numa-bind", action="store_true", help="do not bind the process to the CPUs of the GPU's NUMA node")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses GPU 0 (implies --backend gloo)")
    ap.add_argument("--workload", choices=["strings", "euclid"], default=None,
                    help="headline workload.  strings = BASELINE configs[1] (load_strings Levenshtein, the configuration the metric "
                         "is quoted on; does not shard: N > 1 runs per-GPU replicas); euclid = BASELINE configs[2] (N = 10^6 rows "
                         "sharded over the GPUs, strong scaling).  Default: strings at --gpus 1, euclid at --gpus N > 1; "
                         "`--gpus 1 --workload euclid` prints the N > 1 headline fields for one GPU, the consistent N = 1 point of a "
                         "1 -> 8 curve")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a plain `python bench.py --gpus N`: this process becomes the launcher of its N ranks (one per GPU, the command line the
        # driver uses) and passes their ONE JSON line through
        import socket
        import subprocess

        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)
    # stdout carries the JSON line and nothing else: the library mirrors the reference's print() notices ("Increasing p_work ...",
    # the note about point sets beyond the complete pair list), so file descriptor 1 is pointed at stderr for the rest of the run
    global _REAL_STDOUT
    _REAL_STDOUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    sys.stdout = sys.stderr

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        if args.share_gpu:
            args.backend, local = "gloo", 0
            # (rehearsal only: several processes on one GPU cannot all keep the persistent anchor launch's waves resident -- each
            # fit would sit out its 20 ms time limit and take the one-workgroup rescue form; one process per GPU is the real layout)
            os.environ.setdefault("ANNCHOR_LEV_PERSIST", "0")
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    assert args.gpus == world, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)

    from annchor_amd import _native as _nat

    os.environ["ANNCHOR_RNG_NO_CACHE"] = "1"   # no per-seed MT19937 stream cache: every fit pays for its own streams

    # one process per GPU, bound to the CPUs next to it (the launcher's numactl, done here so that
    # the driver's plain `python bench.py` / torch.distributed.run command lines get it too)
    all_cpus = os.sched_getaffinity(0)
    affinity = None if args.no_numa_bind else _nat.bind_to_device_numa(local)


    workload = args.workload or ("strings" if world == 1 else "euclid")
    quoted = ("BASELINE.json's metric (k-NN graph build time + recall@k) is quoted on configs[1] = --workload strings "
              "(the default at --gpus 1); --workload euclid (the default at --gpus N > 1) is configs[2], the workload that shards")

    if workload == "strings" and world == 1:
        out = strings_run(args, args.steps, args.warmup, world, rank, local, dist, torch, all_cpus, affinity)
        out["config"]["baseline_quoted_on"] = quoted
        try:
            out["c2_device_sampler_plugin"] = device_sampler_block(local)
        except Exception as e:
            out["c2_device_sampler_plugin"] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:
            out["c4_digits_wasserstein"] = c4_block(local)
        except Exception as e:
            out["c4_digits_wasserstein"] = {"error": "%s: %s" % (type(e).__name__, e)}
        # ---- the pair-list kernels where HBM decides (the line is complete by now)
        if not args.no_scale:
            try:
                out["pairlist_kernels_at_scale"] = pairlist_at_scale(local)
                ks = out["pairlist_kernels_at_scale"]["kernels"]
                pick = [n for n in ("row_kth_threshold", "row_topk_graph", "guarantee_nmin_lists", "bounds_dad_features",
                                    "predict_clip_label_merge", "topk_split_compact", "ecdf_probability") if n in ks]
                out["roofline_pairlist_kernels_at_scale"] = {
                    "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "workload": out["pairlist_kernels_at_scale"]["workload"],
                    "kernels": {n: {"achieved": ks[n]["alg_GBps"], "frac": ks[n]["hbm_frac"], "frac_pmc": ks[n].get("hbm_frac_pmc")}
                                for n in pick},
                    "note": "frac = algorithmic bytes (SURVEY 8d per-pair figures) / time / 8 TB/s; frac_pmc = HBM bytes of the "
                            "committed rocprofv3 PMC passes over the same workload / this run's time / 8 TB/s"}
            except Exception as e:
                out["pairlist_kernels_at_scale"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if not args.no_scale:
            try:
                out["levenshtein_100k_thinned_pairlist"] = levenshtein_100k_block(local)
            except Exception as e:
                out["levenshtein_100k_thinned_pairlist"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if not args.no_euclid:
            try:
                out["c3_euclid_streamed"] = euclid_run(1, 0, local, None, args.euclid_rows, 2, 1, torch)
            except Exception as e:
                out["c3_euclid_streamed"] = {"error": "%s: %s" % (type(e).__name__, e)}
        emit(out)
        return

    if workload == "strings":
        # the strings workload (1600 points, a chain of dependent launches) does not shard: one independent graph
        # build per GPU, value = graphs/s over all ranks ("replicas only")
        st = strings_run(args, args.steps, args.warmup, world, rank, local, dist, torch, all_cpus, affinity)
        if rank == 0:
            st["config"]["baseline_quoted_on"] = quoted
            emit(st)
        dist.destroy_process_group()
        return

    # ---------------------------------------------------------------- the row-sharded build (any number of GPUs)
    import threading

    def bail():
        if rank == 0:
            emit({"metric": "knn_graph_builds_per_s", "value": 0.0, "unit": "graphs/s", "n_gpus": world,
                              "error": "timed out after %d s" % args.euclid_timeout})
        os._exit(1)

    dog = threading.Timer(args.euclid_timeout, bail)   # a failed rank would leave the others inside a collective forever
    dog.daemon = True
    dog.start()
    n_per_rank = args.euclid_rows // world
    res = euclid_run(world, rank, local, dist, n_per_rank, args.steps, args.warmup, torch)
    out = None
    if rank == 0:
        out = {
            "metric": "knn_graph_builds_per_s (1 / fit() wall-clock; BASELINE: k-NN graph build time + recall@k)",
            "value": res["graphs_per_s"], "unit": "graphs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["fit_time_s"] * 1e3, "fit_time_s": res["fit_time_s"], "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 rows; tile phase in two stages: fp16 hi-half products with f32 accumulation (v_mfma_f32_32x32x16_f16) and a proven error bound "
                     "decide which columns MAY enter a row's list, exact f32 sums of (x - y)^2 decide which do (the first 32 tiles per row tile and the "
                     "join passes: split fp16 hi + lo products to 2^-22 |x||y|, kept columns re-ranked by the same exact f32 sums); reported distances "
                     "exact f32, widened to f64",
            "data": "synthetic (SURVEY.md 8d recipe: 8-d latent manifold in 128-d, float32), generated per shard",
            "config": {"workload": res["workload"] + ("" if world == 1 and args.workload == "euclid" else
                                                      " -- NOT the --gpus 1 default workload (configs[1] strings, which does not shard): the "
                                                      "one-GPU point of THIS curve is value_single_gpu_same_workload, or `--gpus 1 --workload euclid`"),
                       "total_rows": n_per_rank * world, "baseline_quoted_on": quoted,
                       "parallelism": "one GPU" if world == 1 else
                                      "rows sharded N/G per GPU; every exchange on device buffers over RCCL: per anchor round one "
                                      "all-gather of (value, row id, coordinates); one all-gather of the raw rows (every rank then builds "
                                      "the same global tile order), all-gather of the neighbour lists before each join pass, all-to-all "
                                      "of the finished rows to their owners; full-graph gather not timed (each rank keeps its rows)"},
            "recall_at_k": res["recall_at_k"], "rows_per_s": res["rows_per_s"],
            "roofline": res.get("roofline"), "detail": res,
            "cpu_affinity": affinity or "unbound",
        }
    if world == 1:
        dog.cancel()
        if not args.no_cpu_baseline:
            os.sched_setaffinity(0, all_cpus)
            out["cpu_baseline"] = cpu_baseline_euclid(n_per_rank)
        emit(out)
        return
    # the same workload on ONE GPU, measured in this run by rank 0 while the others wait (strong-scaling reference)
    try:
        if rank == 0:
            Xall = np.concatenate([euclid_shard(r, n_per_rank) for r in range(world)])
            from annchor_amd.streamed import StreamedAnnchor

            ts = []
            for it in range(3):
                sa = StreamedAnnchor(Xall, n_anchors=32, n_neighbors=15, p_work=0.1, device=local)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                sa.fit()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
                sa._engine.close()
            one = float(np.mean(ts[1:]))
            out["single_gpu_same_workload"] = {"fit_time_s": one, "value": 1.0 / one}
            # ONE curve for a reader of SCALE_rNN.json: the same workload's one-GPU point, measured in this run, at top level (the plain
            # `--gpus 1` line is another workload -- configs[1], which does not shard -- and must not be divided into this one)
            out["value_single_gpu_same_workload"] = 1.0 / one
            out["speedup_vs_single_gpu"] = one / res["fit_time_s"]
            out["scaling_efficiency_vs_single_gpu"] = one / res["fit_time_s"] / world
            del Xall
        dist.barrier()
    except Exception as e:
        if rank == 0:
            out["single_gpu_same_workload"] = {"error": "%s: %s" % (type(e).__name__, e)}
    # BASELINE configs[4]: N = 8 000 000 over 8 GPUs
    if world == 8 and not args.no_c5:
        try:
            c5 = euclid_run(world, rank, local, dist, 1_000_000, 1, 1, torch)
            if rank == 0:
                out["c5_euclid_8M_rows"] = c5
        except Exception as e:
            if rank == 0:
                out["c5_euclid_8M_rows"] = {"error": "%s: %s" % (type(e).__name__, e)}
    # per-rank replicas of the strings workload (BASELINE configs[1]); it does not shard
    try:
        st = strings_run(args, 10, 2, world, rank, local, dist, torch, all_cpus, affinity)
        if rank == 0:
            out["strings_replicas"] = {kk: st[kk] for kk in ("value", "unit", "ms_per_step", "errors_vs_bruteforce", "evals", "config")
                                       if kk in st}
    except Exception as e:
        if rank == 0:
            out["strings_replicas"] = {"error": "%s: %s" % (type(e).__name__, e)}
    dog.cancel()
    if rank == 0:
        emit(out)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
