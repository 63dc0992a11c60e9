"""Per-stage scaling model of the row-sharded streamed build from tools/serial_ranks.py's measurements.

Inputs that are MEASURED (one MI355X, each rank's kernels alone on the GPU): per-stage kernel time of every rank at G = 1, 2, 4,
8 ranks; the payload of every collective (bytes per rank).  What is MODELLED (no multi-GPU box here): the collectives' time
over xGMI -- 7 links per GPU at ~153 GB/s (both directions) = 76 GB/s per direction peak, 70 % of it assumed (54 GB/s);
a fully connected mesh, so an all-gather among G ranks moves every rank's payload over G - 1 links in parallel: time =
LAT + payload / 54 GB/s, an all-to-all LAT + payload / G / 54 GB/s per peer, in parallel; LAT = 25 us per collective call.
The rows' all-gather (kind "all_gather_overlapped": half of a fit's collective bytes) runs on a second communicator's stream beside
the anchor sweeps and the k-d order (csrc/comm.hip, round 6): only what those stages do not cover is charged; everything else is
issued on the compute stream and charged in full.

  python tools/scaling_model.py profiles/r05_serial_ranks_c5.json [--md]
"""
import json
import sys

LINK_GBS = 54.0
LAT_US = 25.0

KIND = {   # stage -> how it is partitioned
    "stream_tile_gemm_topk": "rows sharded",
    "stream_join_candidates": "rows sharded",
    "stream_rank_tile_pairs": "rows sharded",
    "stream_join_gemm_topk": "rows sharded",
    "stream_join_reverse_lists": "columns sharded (+ one read of every list entry)",
    "stream_order_tiles": "own tile range per level (+ the top log2 G levels in full)",
    "stream_order_gather_rows": "replicated: every rank holds every row as a column",
    "stream_anchor_one_to_all": "rows sharded",
    "stream_anchor_dists_assemble": "replicated copy of the gathered distances",
    "stream_finalize": "rows sharded",
    "stream_route_rows": "rows sharded",
    "stream_route_scatter": "rows sharded",
    "exclusive_scan": "columns sharded",
    "stream_tile_two_stage_kernel": "(inside `stream_tile_gemm_topk`: `k_st_knnh` without its warm-up; not added again)",
    "stream_tile_exact_repair": "one workgroup per flagged row (a handful at C5, none at C3)",
}
NESTED = {"stream_tile_two_stage_kernel"}   # ProfScopes inside another stage's scope: shown, not summed


def stage_sum(w):
    return sum(v for k, v in w["kernel_ms_max_over_ranks"].items() if k not in NESTED)


def model(path):
    R = json.load(open(path))
    W = R["worlds"]
    t1 = stage_sum(W["1"])
    out = {"workload": R["workload"], "assumptions": {"link_GBps_per_direction": LINK_GBS, "collective_latency_us": LAT_US,
                                                      "overlap": "the rows' all-gather beside the anchor sweeps and the k-d order; nothing else"}, "worlds": {}}
    for G in sorted(W, key=int):
        w = W[G]
        g = int(G)
        comm_ms, comm_bytes, ov_ms, hidden_ms = 0.0, 0, 0.0, 0.0
        for c in w["collectives_rank0"]:
            b = c["bytes_per_rank"]
            comm_bytes += b
            per_link = b / g if c["kind"] == "all_to_all" else b   # all-to-all: a rank's payload is split over its peers
            t = LAT_US * 1e-3 + per_link / (LINK_GBS * 1e9) * 1e3
            if c["kind"] == "all_gather_overlapped":
                ov_ms += t       # runs on the side communicator's stream beside the stages below
            else:
                comm_ms += t
        if ov_ms:
            # what the rows' all-gather overlaps (streamed.py: started before get_anchors(), joined in annchor_stream_order_end): the
            # anchor sweeps, the k-d order's level sorts, the anchor rounds' and anchor distances' collectives (charged above)
            km = w["kernel_ms_max_over_ranks"]
            cover = km.get("stream_anchor_one_to_all", 0.0) + km.get("stream_order_tiles", 0.0) + km.get("stream_anchor_dists_assemble", 0.0)
            hidden_ms = min(ov_ms, cover)
            comm_ms += ov_ms - hidden_ms
        if g == 1:
            comm_ms = 0.0
        kern = stage_sum(w)
        total = kern + comm_ms
        out["worlds"][G] = {
            "kernel_ms_max_over_ranks": w["kernel_ms_max_over_ranks"], "kernel_ms": round(kern, 2),
            "collective_calls": len(w["collectives_rank0"]) if g > 1 else 0, "collective_MB_per_rank": round(comm_bytes / 1e6, 1) if g > 1 else 0.0,
            "collective_ms_modelled": round(comm_ms, 2), "rows_allgather_ms": round(ov_ms, 2), "rows_allgather_hidden_ms": round(hidden_ms, 2), "total_ms": round(total, 2), "speedup": round(t1 / total, 2),
            "efficiency": round(t1 / total / g, 3), "recall_at_k": w["recall_at_k"]}
    return out


def markdown(M):
    Gs = sorted(M["worlds"], key=int)
    stages = sorted({k for g in Gs for k in M["worlds"][g]["kernel_ms_max_over_ranks"]},
                    key=lambda k: -M["worlds"]["1"]["kernel_ms_max_over_ranks"].get(k, 0.0))
    lines = ["| stage (kernel ms, max over ranks) | how it is partitioned | " + " | ".join("G = %s" % g for g in Gs) + " |",
             "|---|---|" + "---|" * len(Gs)]
    for s in stages:
        lines.append("| `%s` | %s | " % (s, KIND.get(s, "")) + " | ".join("%.2f" % M["worlds"][g]["kernel_ms_max_over_ranks"].get(s, 0.0) for g in Gs) + " |")
    lines.append("| **kernels** | | " + " | ".join("**%.1f**" % M["worlds"][g]["kernel_ms"] for g in Gs) + " |")
    lines.append("| collectives: calls / MB per rank | | " + " | ".join("%d / %.0f" % (M["worlds"][g]["collective_calls"], M["worlds"][g]["collective_MB_per_rank"]) for g in Gs) + " |")
    lines.append("| collectives, modelled ms | | " + " | ".join("%.1f" % M["worlds"][g]["collective_ms_modelled"] for g in Gs) + " |")
    lines.append("| **total ms -> speedup (efficiency)** | | " + " | ".join("**%.1f** -> %.2fx (%.0f %%)" % (M["worlds"][g]["total_ms"], M["worlds"][g]["speedup"],
                                                                                                 100 * M["worlds"][g]["efficiency"]) for g in Gs) + " |")
    return "\n".join(lines)


if __name__ == "__main__":
    M = model(sys.argv[1])
    if "--md" in sys.argv:
        print(markdown(M))
    else:
        print(json.dumps(M, indent=1))
