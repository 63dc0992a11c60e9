"""
ctypes binding of libannchor_hip.so (include/annchor_hip.h).

The product path is the HIP library; there is no CPU implementation behind this
module.  Importing works without a GPU (so that the host logic can be tested), but
creating an `Engine` fails loudly when the library or a device is missing.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ANNCHOR_HIP_LIB", os.path.join(_HERE, "libannchor_hip.so"))   # override: profiling builds

# field ids (include/annchor_hip.h)
F_D, F_A, F_SID, F_IJS, F_I_PTR, F_I_IDX, F_FEATURES, F_NCM, F_RA, F_LABELS, F_THRESH, F_PROB, F_CAND, F_NEXT, F_DAD = range(1, 16)
_FIELD_DTYPE = {
    F_D: np.float64, F_A: np.int64, F_SID: np.uint64, F_IJS: np.int64, F_I_PTR: np.int64, F_I_IDX: np.int64,
    F_FEATURES: np.float64, F_NCM: np.uint8, F_RA: np.float64, F_LABELS: np.int64, F_THRESH: np.float64,
    F_PROB: np.float64, F_CAND: np.int64, F_NEXT: np.int64, F_DAD: np.float64,
}

METRIC_LEVENSHTEIN, METRIC_EUCLIDEAN_F32, METRIC_EUCLIDEAN_F64, METRIC_WASSERSTEIN = 1, 2, 3, 4
METRIC_COSINE_F32, METRIC_COSINE_F64 = 5, 6

# every entry point declared in include/annchor_hip.h: name -> (restype, argtypes)
_vp, _i32, _i64, _dbl = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_double
_SIGNATURES = {
    "annchor_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_vp)]),
    "annchor_destroy": (None, [_vp]),
    "annchor_release_parked": (ctypes.c_int, []),
    "annchor_parked_bytes": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_i64)]),
    "annchor_last_error": (ctypes.c_char_p, [_vp]),
    "annchor_create_error": (ctypes.c_char_p, []),
    "annchor_device_name": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_int]),
    "annchor_device_pci_bus_id": (ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, ctypes.c_int]),
    "annchor_device_mem_info": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "annchor_synchronize": (ctypes.c_int, [_vp]),
    "annchor_lev_persist_state": (ctypes.c_int, [ctypes.c_int]),
    "annchor_last_kernel_ms": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_float)]),
    "annchor_set_strings": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i32]),
    "annchor_set_strings_u16": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i32]),
    "annchor_set_points_f32": (ctypes.c_int, [_vp, _vp, _i64, _i32]),
    "annchor_set_points_f64": (ctypes.c_int, [_vp, _vp, _i64, _i32]),
    "annchor_set_points_cosine_f32": (ctypes.c_int, [_vp, _vp, _i64, _i32]),
    "annchor_set_points_cosine_f64": (ctypes.c_int, [_vp, _vp, _i64, _i32]),
    "annchor_set_histograms": (ctypes.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "annchor_set_opaque": (ctypes.c_int, [_vp, _i64]),
    "annchor_metric_pairs": (ctypes.c_int, [_vp, _vp, _i64, _vp]),
    "annchor_brute_force": (ctypes.c_int, [_vp, _i32, _vp, _vp]),
    "annchor_pick_anchors_maxmin": (ctypes.c_int, [_vp, _i32, _i64]),
    "annchor_pick_anchors_selected": (ctypes.c_int, [_vp, _vp, _i32]),
    "annchor_set_anchor_distances": (ctypes.c_int, [_vp, _vp, _i32, _vp, _i32]),
    "annchor_build_locality": (ctypes.c_int, [_vp, _i32, _i32, _i32, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "annchor_build_query_locality": (ctypes.c_int, [_vp, _i64, _i32, _i32, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "annchor_compute_features": (ctypes.c_int, [_vp]),
    "annchor_count_uncomputed": (ctypes.c_int, [_vp, ctypes.POINTER(_i64)]),
    "annchor_kth_uncomputed_dad": (ctypes.c_int, [_vp, _vp, _i32, _vp]),
    "annchor_bin_counts": (ctypes.c_int, [_vp, _vp, _i32, _vp]),
    "annchor_sampler_stats": (ctypes.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "annchor_select_by_rank": (ctypes.c_int, [_vp, _vp, _i32, _vp, _vp, _i64, _vp]),
    "annchor_sample_pairs": (ctypes.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "annchor_sample_pairs_device": (ctypes.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _i64]),
    "annchor_hash_sample_pairs_device": (ctypes.c_int, [_vp, _vp, _i32, _vp, _vp, ctypes.c_uint64, _vp]),
    "annchor_download_samples": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "annchor_fit_regression_device": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32]),
    "annchor_fit_errors_device": (ctypes.c_int, [_vp]),
    "annchor_model_download": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "annchor_errors_download": (ctypes.c_int, [_vp, _vp, _i64]),
    "annchor_model_download_with_errors": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, ctypes.POINTER(_i64)]),
    "annchor_sample_pairs_device_draw": (ctypes.c_int, [_vp, _vp, _i32, _vp, _vp, ctypes.c_uint32, ctypes.POINTER(_i64), ctypes.POINTER(_i32)]),
    "annchor_legacy_choice_ranks_device": (ctypes.c_int, [_vp, ctypes.c_uint32, _vp, _vp, _i32, _vp, ctypes.POINTER(_i32)]),
    "annchor_hash_sample": (ctypes.c_int, [_vp, _vp, _i32, _vp, _vp, ctypes.c_uint64, _vp, ctypes.POINTER(_i64)]),
    "annchor_hash_sample_pairs": (ctypes.c_int, [_vp, _vp, _i32, _vp, _vp, ctypes.c_uint64, _vp, _vp, _vp, ctypes.POINTER(_i64)]),
    "annchor_legacy_prefetch": (ctypes.c_int, [ctypes.c_uint32, _i64]),
    "annchor_legacy_generate": (ctypes.c_int, [ctypes.c_uint32, _i64]),
    "annchor_legacy_generate_at_next_wait": (ctypes.c_int, [_vp, ctypes.c_uint32, _i64, _i64]),
    "annchor_legacy_choice_ranks": (ctypes.c_int, [ctypes.c_uint32, _vp, _vp, _i32, _vp, _vp]),
    "annchor_legacy_choice_begin": (ctypes.c_int, [ctypes.c_uint32, _vp, _vp, _i32, ctypes.POINTER(_vp)]),
    "annchor_legacy_choice_end": (ctypes.c_int, [_vp, _vp, _vp]),
    "annchor_ols_bins": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _vp]),
    "annchor_gather_features": (ctypes.c_int, [_vp, _vp, _i64, _vp]),
    "annchor_evaluate_samples": (ctypes.c_int, [_vp, _vp, _i64, _vp]),
    "annchor_set_samples": (ctypes.c_int, [_vp, _vp, _i64, _vp]),
    "annchor_predict_merge": (ctypes.c_int, [_vp, _vp, _i32, _vp, _vp, _i32, _i32, _vp]),
    "annchor_merge_host_prediction": (ctypes.c_int, [_vp, _vp, _i32, _i32]),
    "annchor_set_labels": (ctypes.c_int, [_vp, _vp]),
    "annchor_select_candidates": (ctypes.c_int, [_vp, _i32, _i32, _vp, _vp, _i32, _i64, _i32,
                                                 ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "annchor_select_prepare": (ctypes.c_int, [_vp, _i32, _i32]),
    "annchor_mark_candidates": (ctypes.c_int, [_vp]),
    "annchor_refine_candidates": (ctypes.c_int, [_vp]),
    "annchor_park_refine": (ctypes.c_int, [_vp, _i32]),
    "annchor_set_refined": (ctypes.c_int, [_vp, _vp, _i64]),
    "annchor_update_bounds": (ctypes.c_int, [_vp]),
    "annchor_neighbor_graph": (ctypes.c_int, [_vp, _i32, _vp, _vp]),
    "annchor_stream_bind": (ctypes.c_int, [_vp, _vp, _i64, _i32, _i64, _i32]),
    "annchor_stream_anchor_round": (ctypes.c_int, [_vp, _vp, _i32, _i32, ctypes.POINTER(_dbl), ctypes.POINTER(_i64)]),
    "annchor_stream_get_row": (ctypes.c_int, [_vp, _i64, _vp]),
    "annchor_stream_order": (ctypes.c_int, [_vp, _i32] + [ctypes.POINTER(_vp)] * 6 + [ctypes.POINTER(_i64), ctypes.POINTER(_i32),
                                                                                   ctypes.POINTER(_i32)]),
    "annchor_stream_knn": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _dbl, _i32, _i32, _vp,
                                          _vp, _vp, ctypes.POINTER(_i64)]),
    "annchor_stream_knn_begin": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                                ctypes.POINTER(_vp), ctypes.POINTER(_i64)]),
    "annchor_stream_knn_join": (ctypes.c_int, [_vp, _vp, _i32, ctypes.POINTER(_vp), ctypes.POINTER(_i64)]),
    "annchor_stream_join_rev_begin": (ctypes.c_int, [_vp, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_i64)]),
    "annchor_stream_order_begin": (ctypes.c_int, [_vp, _i32, _i32, _i32, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_i64)]),
    "annchor_stream_order_end": (ctypes.c_int, [_vp] + [ctypes.POINTER(_vp)] * 6 + [ctypes.POINTER(_i64), ctypes.POINTER(_i32),
                                                                                       ctypes.POINTER(_i32)]),
    "annchor_stream_last_counts": (ctypes.c_int, [_vp, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "annchor_stream_last_kernel": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(_i64)]),
    "annchor_stream_last_tile_kernels": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "annchor_stream_budget": (ctypes.c_int, [_i32, _dbl, _i32, ctypes.POINTER(_i32), ctypes.POINTER(_i32), ctypes.POINTER(_i32)]),
    "annchor_stream_knn_end": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.POINTER(_i64)]),
    "annchor_stream_knn_run": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _dbl, _i32, _i32,
                                               ctypes.POINTER(_i64)]),
    "annchor_stream_knn_fetch": (ctypes.c_int, [_vp, _vp, _vp, _vp]),
    "annchor_stream_query": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _dbl, _vp, _vp,
                                            ctypes.POINTER(_i64)]),
    "annchor_stream_join_tables": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "annchor_stream_hip_stream": (ctypes.c_int, [_vp, ctypes.POINTER(_vp)]),
    "annchor_stream_anchor_begin": (ctypes.c_int, [_vp, _i32, _i64, _i32, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_i64)]),
    "annchor_stream_anchor_step": (ctypes.c_int, [_vp, _vp, _i32, _i32]),
    "annchor_stream_anchor_end": (ctypes.c_int, [_vp, _vp, _vp]),
    "annchor_stream_rows_begin": (ctypes.c_int, [_vp, _i32, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_i64)]),
    "annchor_stream_rows_end": (ctypes.c_int, [_vp, _i32, _vp]),
    "annchor_stream_anchor_dists_begin": (ctypes.c_int, [_vp, _i32, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_i64)]),
    "annchor_stream_lists_all": (ctypes.c_int, [_vp, _i32, _i64, ctypes.POINTER(_vp)]),
    "annchor_stream_route_begin": (ctypes.c_int, [_vp, _i32, _vp, _vp, ctypes.POINTER(_vp), _vp, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "annchor_stream_route_recv": (ctypes.c_int, [_vp, _i64, ctypes.POINTER(_vp)]),
    "annchor_comm_unique_id": (ctypes.c_int, [_vp]),
    "annchor_comm_init": (ctypes.c_int, [_vp, _vp, _i32, _i32]),
    "annchor_comm_destroy": (ctypes.c_int, [_vp]),
    "annchor_comm_set_timeout": (ctypes.c_int, [_vp, ctypes.c_double]),
    "annchor_comm_preflight": (ctypes.c_int, [_vp, ctypes.c_double]),
    "annchor_comm_init_side": (ctypes.c_int, [_vp, _vp]),
    "annchor_comm_allgather_begin": (ctypes.c_int, [_vp, _vp, _vp, _i64]),
    "annchor_comm_side_join": (ctypes.c_int, [_vp]),
    "annchor_comm_allgather": (ctypes.c_int, [_vp, _vp, _vp, _i64]),
    "annchor_comm_alltoall_records": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i32]),
    "annchor_stream_anchor_rounds": (ctypes.c_int, [_vp, _i32]),
    "annchor_stream_route_end": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp]),
    "annchor_stream_graph_device": (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_i64), ctypes.POINTER(_i32)]),
    "annchor_device_alloc": (ctypes.c_int, [_vp, _i64, ctypes.POINTER(_vp)]),
    "annchor_device_free": (ctypes.c_int, [_vp, _vp]),
    "annchor_device_copy": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i32]),
    "annchor_enemies_candidates": (ctypes.c_int, [_vp, _vp, _i32, _i32, ctypes.POINTER(_i64)]),
    "annchor_enemies_predict": (ctypes.c_int, [_vp, _vp, _i32, _vp, _vp, _vp]),
    "annchor_enemies_first": (ctypes.c_int, [_vp, _i32, _i32, _i32, _vp, _i64, ctypes.POINTER(_i64)]),
    "annchor_enemies_set_exact": (ctypes.c_int, [_vp, _vp, _i64]),
    "annchor_enemies_graph": (ctypes.c_int, [_vp, _i32, _vp, _vp]),
    "annchor_enemies_download": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "annchor_graph_to_coo": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, ctypes.POINTER(_i64)]),
    "annchor_field_size": (ctypes.c_int, [_vp, _i32, ctypes.POINTER(_i64)]),
    "annchor_download": (ctypes.c_int, [_vp, _i32, _vp, _i64]),
    "annchor_upload": (ctypes.c_int, [_vp, _i32, _vp, _i64]),
    "annchor_prof_enable": (ctypes.c_int, [_vp, _i32]),
    "annchor_prof_reset": (ctypes.c_int, [_vp]),
    "annchor_prof_get": (ctypes.c_int, [_vp, _i32, _vp, _vp, _vp, _vp]),
}

_lib = None


class NativeError(RuntimeError):
    pass


def exported_symbols():
    return sorted(_SIGNATURES)


def load_library():
    """Load libannchor_hip.so and bind every declared entry point (no GPU needed)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                "libannchor_hip.so is not built (%s).  Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `python annchor_amd/build.py`; there is no CPU fallback." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _ptr(a):
    return a.ctypes.data if a is not None else None


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def release_parked_contexts():
    """Free the runtime shells (stream, pinned staging, device slab) that destroyed engines left
    parked for the next one; returns how many there were."""
    return int(load_library().annchor_release_parked())


PAIR_BYTES = 130            # device memory per candidate pair of the pair-list form (with scratch)
PAIR_LIST_MAX = (1 << 30) - 1   # int32 positions into the per-point index (two entries per pair)


def pairlist_point_limit(device=0):
    """Largest data set whose complete pair list (nx (nx - 1) / 2 candidates, the worst case of the locality filter) the
    pair-list form will materialise on `device`: bounded by the int32 pair positions (2^30 pairs: 46 341 points) and by
    80 % of the device's free memory at PAIR_BYTES per pair.  None when no device can be asked.  Blocks this process keeps
    parked for reuse count as free -- they are NOT handed back to the driver here: a gigabyte request that follows the release
    of tens of gigabytes waited ~1.5 s on this stack (every second fit of 100 000 points, tools/lev100k_profile.py)."""
    f, t, parked = _i64(), _i64(), _i64()
    try:
        if load_library().annchor_device_mem_info(int(device), ctypes.byref(f), ctypes.byref(t)) != 0:
            return None
        if load_library().annchor_parked_bytes(int(device), ctypes.byref(parked)) != 0:
            return None
    except NativeError:
        return None
    pairs = min(PAIR_LIST_MAX, int(0.8 * (f.value + parked.value) / PAIR_BYTES))
    return int((1 + (1 + 8 * pairs) ** 0.5) // 2)


def bind_to_device_numa(device=0):
    """Restrict this process (and the threads it starts from now on) to the CPUs of the NUMA node
    the GPU hangs off -- what `numactl --cpunodebind` does for a launcher.  Call it before the first
    Annchor / Engine is created.  On the MI355X box the C2 fit takes 5.25 ms from the GPU's node and
    5.57 ms from the other one.  Returns a description, or None when the topology cannot be read."""
    import os

    buf = ctypes.create_string_buffer(64)
    try:
        if load_library().annchor_device_pci_bus_id(int(device), buf, 64) != 0:
            return None
        bus = buf.value.decode().lower()
        with open("/sys/bus/pci/devices/%s/numa_node" % bus) as fh:
            node = int(fh.read())
        if node < 0:
            return None
        with open("/sys/devices/system/node/node%d/cpulist" % node) as fh:
            cpus = set()
            for part in fh.read().strip().split(","):
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return "NUMA node %d of GPU %s (%d CPUs)" % (node, bus, len(cpus))
    except (OSError, ValueError, AttributeError):
        return None


def stream_budget(n_tiles, p_work, join_passes):
    """(total, tile_phase, per_pass): the streamed form's per-row-tile work budget and its split
    between the tile phase and the join passes (include/annchor_hip.h: annchor_stream_budget)."""
    T, tp, pp = _i32(), _i32(), _i32()
    rc = load_library().annchor_stream_budget(int(n_tiles), float(p_work), int(join_passes), ctypes.byref(T), ctypes.byref(tp),
                                              ctypes.byref(pp))
    if rc != 0:
        raise NativeError("annchor_stream_budget failed (%d)" % rc)
    return T.value, tp.value, pp.value


# A stream is produced AHEAD of its draw only up to this many 32-bit words (1 GB): the ahead-of-time size is a bound from the
# point count (1.5 x nx (nx - 1) / 2 -- 7.5 x 10^9 words for the thinned lists of 10^5 points, most of which the draw would never
# read); beyond it the draw itself generates what it consumes, sized from the real bin populations (annchor_legacy_choice_ranks).
LEGACY_EAGER_MAX_DRAWS = 1 << 28


def comm_unique_id():
    """128 bytes from which the ranks of a job build their in-library RCCL communicator (call on ONE rank, hand the bytes
    to the others by any means: Engine.comm_init)."""
    buf = (ctypes.c_uint8 * 128)()
    rc = load_library().annchor_comm_unique_id(buf)
    if rc != 0:
        raise NativeError("annchor_comm_unique_id failed (%d): RCCL not loadable?" % rc)
    return bytes(buf)


def legacy_prefetch(seed, ndraws):
    """Begin producing the legacy MT19937 stream of `seed` on a background thread."""
    if 0 <= seed < 2 ** 32 and 0 < ndraws <= LEGACY_EAGER_MAX_DRAWS:
        rc = load_library().annchor_legacy_prefetch(int(seed), int(ndraws))
        if rc != 0:
            raise NativeError("annchor_legacy_prefetch failed (%d)" % rc)


def legacy_generate(seed, ndraws):
    """Produce the legacy MT19937 stream of `seed` on the calling thread (returns when it is there)."""
    if 0 <= seed < 2 ** 32 and 0 < ndraws <= LEGACY_EAGER_MAX_DRAWS:
        rc = load_library().annchor_legacy_generate(int(seed), int(ndraws))
        if rc != 0:
            raise NativeError("annchor_legacy_generate failed (%d)" % rc)


def lev_persist_state(set=-1):
    """The picker's persistent anchor launch: 1 armed, 0 switched off (the library does that itself after two launches that gave
    up -- a GPU shared with another process); set=1 re-arms, set=0 switches it off."""
    return int(load_library().annchor_lev_persist_state(int(set)))


def legacy_choice_ranks(seed, counts, want):
    """np.random.seed(seed); per bin np.random.permutation(counts[b])[:want[b]] (or the whole
    bin when it is smaller) -- NumPy's legacy stream, generated by the library's host code."""
    lib = load_library()
    counts, want = _c(counts, np.int64), _c(want, np.int64)
    out = np.zeros(int(np.minimum(counts, want).sum()), dtype=np.int64)
    n_out = np.zeros(len(counts), dtype=np.int64)
    rc = lib.annchor_legacy_choice_ranks(int(seed) & 0xFFFFFFFF, _ptr(counts), _ptr(want), len(counts), _ptr(out), _ptr(n_out))
    if rc != 0:
        raise NativeError("annchor_legacy_choice_ranks failed (%d)" % rc)
    return np.split(out, np.cumsum(n_out)[:-1])


_DGELSD = None


def _dgelsd_pointer():
    """Address of scipy's Fortran dgelsd (the routine scipy.linalg.lapack.dgelsd wraps), or 0."""
    global _DGELSD
    if _DGELSD is None:
        try:
            from scipy.linalg import cython_lapack
            cap = cython_lapack.__pyx_capi__["dgelsd"]
            api = ctypes.pythonapi
            api.PyCapsule_GetName.restype, api.PyCapsule_GetName.argtypes = ctypes.c_char_p, [ctypes.py_object]
            api.PyCapsule_GetPointer.restype, api.PyCapsule_GetPointer.argtypes = ctypes.c_void_p, [ctypes.py_object, ctypes.c_char_p]
            _DGELSD = api.PyCapsule_GetPointer(cap, api.PyCapsule_GetName(cap)) or 0
        except Exception:  # noqa: BLE001 -- no scipy.linalg.cython_lapack: the Python path does the work
            _DGELSD = 0
    return _DGELSD


def ols_bins(Xs, ys, cuts):
    """Per-partition OLS pieces (coef, xmean, ymean, status) for samples grouped by partition; Xs is
    the (n, nf) design matrix in column-major order, ys the targets, cuts the partition boundaries.
    None when scipy's LAPACK pointer is not available."""
    ptr = _dgelsd_pointer()
    if not ptr:
        return None
    n, nf = Xs.shape
    Xf = np.asfortranarray(Xs, dtype=np.float64)
    ys = _c(ys, np.float64)
    cuts = _c(cuts, np.int64)
    nb = len(cuts) - 1
    coef, xm = np.zeros((nb, nf)), np.zeros((nb, nf))
    ym, status = np.zeros(nb), np.zeros(nb, dtype=np.int32)
    rc = load_library().annchor_ols_bins(ptr, _ptr(Xf), _ptr(ys), n, nf, _ptr(cuts), nb, _ptr(coef), _ptr(xm), _ptr(ym), _ptr(status))
    if rc != 0:
        raise NativeError("annchor_ols_bins failed (%d)" % rc)
    return coef, xm, ym, status


def legacy_choice_begin(seed, counts, want):
    """legacy_choice_ranks on the library's persistent worker thread: returns a ticket at once."""
    lib = load_library()
    counts, want = _c(counts, np.int64), _c(want, np.int64)
    t = _vp()
    rc = lib.annchor_legacy_choice_begin(int(seed) & 0xFFFFFFFF, _ptr(counts), _ptr(want), len(counts), ctypes.byref(t))
    if rc != 0:
        raise NativeError("annchor_legacy_choice_begin failed (%d)" % rc)
    return (t, int(np.minimum(counts, want).sum()), len(counts))


def legacy_choice_end(ticket):
    """Wait for legacy_choice_begin's draw; same return value as legacy_choice_ranks."""
    t, total, nbins = ticket
    out = np.zeros(total, dtype=np.int64)
    n_out = np.zeros(nbins, dtype=np.int64)
    rc = load_library().annchor_legacy_choice_end(t, _ptr(out), _ptr(n_out))
    if rc != 0:
        raise NativeError("annchor_legacy_choice_ranks failed (%d)" % rc)
    return np.split(out, np.cumsum(n_out)[:-1])


class GraphBuffers:
    """The result arrays of a large graph build -- int64 [rows, k] and float64 [rows, k] -- allocated AND touched on a
    helper thread while the GPU works.  A fresh NumPy array is untouched virtual memory: the device-to-host copy that
    fills it first has to fault every page in (240 MB at N = 10^6: 14 ms of page faults around a 4 ms copy at the 56 GB/s
    this host's PCIe link delivers).  The sizes are known when fit() starts, the host has nothing to do during the tile phase,
    and ctypes releases the GIL around the library calls: the faults move off the critical path."""

    def __init__(self, rows, k):
        import threading

        self.rows, self.k = int(rows), int(k)
        self.idx = self.dist = None
        self._t = threading.Thread(target=self._make, daemon=True)
        self._t.start()

    def _make(self):
        idx = np.empty((self.rows, self.k), dtype=np.int64)
        dist = np.empty((self.rows, self.k), dtype=np.float64)
        idx.fill(0)
        dist.fill(0.0)
        self.idx, self.dist = idx, dist

    def take(self, rows, k):
        """(idx, dist) if the shape matches, else fresh arrays."""
        self._t.join()
        if self.idx is not None and (self.rows, self.k) == (int(rows), int(k)):
            return self.idx, self.dist
        return np.empty((int(rows), int(k)), dtype=np.int64), np.empty((int(rows), int(k)), dtype=np.float64)


class Engine:
    """One device context (one GPU).  All pipeline state lives in HBM inside it."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = _vp()
        rc = self.lib.annchor_create(int(device), ctypes.byref(h))
        if rc != 0:
            raise NativeError("annchor_create(device=%d) failed (%d): %s -- the HIP path is the only path; "
                              "no CPU fallback exists." % (device, rc, self.lib.annchor_create_error().decode()))
        self.h = h
        self.device = device
        self.nx = 0
        self.metric = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.annchor_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise NativeError("libannchor_hip error %d: %s" % (rc, self.lib.annchor_last_error(self.h).decode()))

    def device_name(self):
        buf = ctypes.create_string_buffer(256)
        self._chk(self.lib.annchor_device_name(self.h, buf, 256))
        return buf.value.decode()

    def synchronize(self):
        self._chk(self.lib.annchor_synchronize(self.h))

    def last_kernel_ms(self):
        ms = ctypes.c_float()
        self._chk(self.lib.annchor_last_kernel_ms(self.h, ctypes.byref(ms)))
        return ms.value

    # ------------------------------------------------------------ data set
    def set_strings(self, codes, offs, lens, alphabet):
        offs, lens = _c(offs, np.int64), _c(lens, np.int32)
        if np.asarray(codes).dtype == np.uint16:   # more than 256 distinct symbols: 16-bit codes
            codes = _c(codes, np.uint16)
            self._chk(self.lib.annchor_set_strings_u16(self.h, _ptr(codes), _ptr(offs), _ptr(lens), len(lens), int(alphabet)))
        else:
            codes = _c(codes, np.uint8)
            self._chk(self.lib.annchor_set_strings(self.h, _ptr(codes), _ptr(offs), _ptr(lens), len(lens), int(alphabet)))
        self.nx, self.metric = len(lens), METRIC_LEVENSHTEIN

    def set_points(self, X, cosine=False):
        X = np.asarray(X)
        if X.dtype == np.float32:
            X = _c(X, np.float32)
            fn = self.lib.annchor_set_points_cosine_f32 if cosine else self.lib.annchor_set_points_f32
            self._chk(fn(self.h, _ptr(X), X.shape[0], X.shape[1]))
            self.metric = METRIC_COSINE_F32 if cosine else METRIC_EUCLIDEAN_F32
        else:
            X = _c(X, np.float64)
            fn = self.lib.annchor_set_points_cosine_f64 if cosine else self.lib.annchor_set_points_f64
            self._chk(fn(self.h, _ptr(X), X.shape[0], X.shape[1]))
            self.metric = METRIC_COSINE_F64 if cosine else METRIC_EUCLIDEAN_F64
        self.nx = X.shape[0]

    def set_histograms(self, X, cost):
        X, cost = _c(X, np.float64), _c(cost, np.float64)
        assert cost.shape == (X.shape[1], X.shape[1]), "cost_matrix must be [nbins, nbins]"
        self._chk(self.lib.annchor_set_histograms(self.h, _ptr(X), X.shape[0], X.shape[1], _ptr(cost)))
        self.nx, self.metric = X.shape[0], METRIC_WASSERSTEIN

    def set_opaque(self, nx):
        self._chk(self.lib.annchor_set_opaque(self.h, int(nx)))
        self.nx, self.metric = int(nx), 0

    # ----------------------------------------------------------- metric (a2)
    def metric_pairs(self, IJ):
        IJ = _c(IJ, np.int64).reshape(-1, 2)
        out = np.zeros(IJ.shape[0], dtype=np.float64)
        self._chk(self.lib.annchor_metric_pairs(self.h, _ptr(IJ), IJ.shape[0], _ptr(out)))
        return out

    def brute_force(self, k):
        idx = np.zeros((self.nx, k), dtype=np.int64)
        dist = np.zeros((self.nx, k), dtype=np.float64)
        self._chk(self.lib.annchor_brute_force(self.h, int(k), _ptr(idx), _ptr(dist)))
        return idx, dist

    # -------------------------------------------------------------- anchors
    def pick_anchors_maxmin(self, na, first):
        self._chk(self.lib.annchor_pick_anchors_maxmin(self.h, int(na), int(first)))

    def pick_anchors_selected(self, A):
        A = _c(A, np.int64)
        self._chk(self.lib.annchor_pick_anchors_selected(self.h, _ptr(A), len(A)))

    def set_anchor_distances(self, D, A):
        D = _c(D, np.float64)
        A = _c(A, np.int64).reshape(-1)
        if D.ndim != 2 or D.shape[0] != self.nx:
            raise ValueError("anchor distances must be [nx=%d, n_anchors], got %r" % (self.nx, D.shape))
        self._chk(self.lib.annchor_set_anchor_distances(self.h, _ptr(D), D.shape[1], _ptr(A) if len(A) else None, len(A)))

    # ------------------------------------------------------------- pipeline
    def build_locality(self, locality, loc_thresh, loc_min):
        n, m = _i64(), _i64()
        self._chk(self.lib.annchor_build_locality(self.h, int(locality), int(loc_thresh), int(loc_min),
                                                  ctypes.byref(n), ctypes.byref(m)))
        return n.value, m.value

    def build_query_locality(self, nx_base, locality, loc_thresh):
        n, m = _i64(), _i64()
        self._chk(self.lib.annchor_build_query_locality(self.h, int(nx_base), int(locality), int(loc_thresh),
                                                        ctypes.byref(n), ctypes.byref(m)))
        return n.value, m.value

    def compute_features(self):
        self._chk(self.lib.annchor_compute_features(self.h))

    def count_uncomputed(self):
        n = _i64()
        self._chk(self.lib.annchor_count_uncomputed(self.h, ctypes.byref(n)))
        return n.value

    def sampler_stats(self, iq1, iq3, n_partitions):
        """(q1, q3, edges or None, counts or None): the two quantiles and -- when the library could chain them on the device --
        the bin edges and populations of a sampling step, in one host wait."""
        ks = np.array([iq1, iq3], dtype=np.int64)
        q = np.empty(2, dtype=np.float64)
        edges = np.empty(n_partitions + 1, dtype=np.float64)
        counts = np.empty(n_partitions, dtype=np.int64)
        fused = ctypes.c_int32(0)
        self._chk(self.lib.annchor_sampler_stats(self.h, _ptr(ks), int(n_partitions), _ptr(q), _ptr(edges), _ptr(counts),
                                                 ctypes.byref(fused)))
        if fused.value:
            return q[0], q[1], edges, counts
        return q[0], q[1], None, None

    def kth_uncomputed_dad(self, ks):
        ks = _c(ks, np.int64)
        out = np.zeros(len(ks), dtype=np.float64)
        self._chk(self.lib.annchor_kth_uncomputed_dad(self.h, _ptr(ks), len(ks), _ptr(out)))
        return out

    def bin_counts(self, bins):
        bins = _c(bins, np.float64)
        out = np.zeros(len(bins) - 1, dtype=np.int64)
        self._chk(self.lib.annchor_bin_counts(self.h, _ptr(bins), len(bins) - 1, _ptr(out)))
        return out

    def select_by_rank(self, bins, bin_of, ranks):
        bins, bin_of, ranks = _c(bins, np.float64), _c(bin_of, np.int32), _c(ranks, np.int64)
        out = np.zeros(len(ranks), dtype=np.int64)
        self._chk(self.lib.annchor_select_by_rank(self.h, _ptr(bins), len(bins) - 1, _ptr(bin_of), _ptr(ranks),
                                                  len(ranks), _ptr(out)))
        return out

    def sample_pairs(self, bins, counts, bin_of, ranks):
        """select_by_rank + gather_features + evaluate_samples in one call (device metric only)."""
        bins, counts = _c(bins, np.float64), _c(counts, np.int64)
        bin_of, ranks = _c(bin_of, np.int32), _c(ranks, np.int64)
        m = len(ranks)
        pos, feats, y = np.empty(m, dtype=np.int64), np.empty((m, 4), dtype=np.float64), np.empty(m, dtype=np.float64)
        self._chk(self.lib.annchor_sample_pairs(self.h, _ptr(bins), len(bins) - 1, _ptr(counts), _ptr(bin_of), _ptr(ranks), m,
                                                _ptr(pos), _ptr(feats), _ptr(y)))
        return pos, feats, y

    # ---- the iteration's models fitted on the device (csrc/model.hip): no host round trip between the stages
    def sample_pairs_device(self, bins, counts, bin_of, ranks):
        bins, counts = _c(bins, np.float64), _c(counts, np.int64)
        bin_of, ranks = _c(bin_of, np.int32), _c(ranks, np.int64)
        self._chk(self.lib.annchor_sample_pairs_device(self.h, _ptr(bins), len(bins) - 1, _ptr(counts), _ptr(bin_of), _ptr(ranks), len(ranks)))
        return len(ranks)

    def hash_sample_pairs_device(self, bins, counts, want, seed_key):
        bins, counts, want = _c(bins, np.float64), _c(counts, np.int64), _c(want, np.int64)
        m = _i64()
        self._chk(self.lib.annchor_hash_sample_pairs_device(self.h, _ptr(bins), len(bins) - 1, _ptr(counts), _ptr(want),
                                                            ctypes.c_uint64(int(seed_key)), ctypes.byref(m)))
        return int(m.value)

    def download_samples(self, m, predict=True):
        """(positions, feature rows, distances, unclipped predictions) of the device-resident sample."""
        pos, feats, y = np.empty(m, dtype=np.int64), np.empty((m, 4), dtype=np.float64), np.empty(m, dtype=np.float64)
        sp = np.empty(m, dtype=np.float64) if predict else None
        self._chk(self.lib.annchor_download_samples(self.h, _ptr(pos), _ptr(feats), _ptr(y), _ptr(sp)))
        return pos, feats, y, sp

    def fit_regression_device(self, bins, first, is_metric):
        bins = _c(bins, np.float64)
        self._chk(self.lib.annchor_fit_regression_device(self.h, _ptr(bins), len(bins) - 1, int(first), int(is_metric)))

    def fit_errors_device(self):
        self._chk(self.lib.annchor_fit_errors_device(self.h))

    def model_download(self, nb, with_errors=True):
        """(W [nb, 3], c [nb], status [nb], err_ptr [nb + 1] or None, flags [3]); waits; clears the sticky flags."""
        W, c = np.zeros((nb, 3)), np.zeros(nb)
        status, flags = np.zeros(nb, dtype=np.int32), np.zeros(3, dtype=np.int32)
        ep = np.zeros(nb + 1, dtype=np.int64) if with_errors else None
        self._chk(self.lib.annchor_model_download(self.h, _ptr(W), _ptr(c), _ptr(status), _ptr(ep), _ptr(flags)))
        return W, c, status, ep, flags

    def model_download_with_errors(self, nb, cap):
        """model_download + errors_download behind one wait: (W, c, status, err_ptr, flags, errs or None)."""
        W, c = np.zeros((nb, 3)), np.zeros(nb)
        status, flags = np.zeros(nb, dtype=np.int32), np.zeros(3, dtype=np.int32)
        ep = np.zeros(nb + 1, dtype=np.int64)
        errs = np.empty(int(cap), dtype=np.float64)
        n = _i64()
        self._chk(self.lib.annchor_model_download_with_errors(self.h, _ptr(W), _ptr(c), _ptr(status), _ptr(ep), _ptr(flags), _ptr(errs),
                                                             int(cap), ctypes.byref(n)))
        return W, c, status, ep, flags, (errs[:n.value] if n.value >= 0 else None)

    def errors_download(self, n):
        out = np.empty(int(n), dtype=np.float64)
        self._chk(self.lib.annchor_errors_download(self.h, _ptr(out), int(n)))
        return out

    def sample_pairs_device_draw(self, bins, counts, want, seed):
        """The legacy draw + the sampling step in one call, the draw's trace on the device: (taken, number of samples)."""
        bins, counts, want = _c(bins, np.float64), _c(counts, np.int64), _c(want, np.int64)
        n, taken = _i64(), _i32()
        self._chk(self.lib.annchor_sample_pairs_device_draw(self.h, _ptr(bins), len(bins) - 1, _ptr(counts), _ptr(want), int(seed),
                                                           ctypes.byref(n), ctypes.byref(taken)))
        return bool(taken.value), int(n.value)

    def legacy_choice_ranks_device(self, seed, counts, want):
        """legacy_choice_ranks through the device-side trace (tests): list of per-bin rank arrays, or None when not applicable."""
        counts, want = _c(counts, np.int64), _c(want, np.int64)
        n_out = np.minimum(counts, want)
        out = np.full(int(n_out.sum()), -1, dtype=np.int64)
        taken = _i32()
        self._chk(self.lib.annchor_legacy_choice_ranks_device(self.h, int(seed), _ptr(counts), _ptr(want), len(counts), _ptr(out),
                                                             ctypes.byref(taken)))
        if not taken.value:
            return None
        offs = np.concatenate([[0], np.cumsum(n_out)])
        return [out[offs[b]:offs[b + 1]] for b in range(len(counts))]

    def hash_sample(self, bins, counts, want, seed_key):
        """Hashed stratified choice (annchor_hash_sample): positions of the samples, partition by partition."""
        bins, counts, want = _c(bins, np.float64), _c(counts, np.int64), _c(want, np.int64)
        out = np.empty(int(np.minimum(counts, want).sum()), dtype=np.int64)
        n = _i64()
        self._chk(self.lib.annchor_hash_sample(self.h, _ptr(bins), len(bins) - 1, _ptr(counts), _ptr(want), int(seed_key) & (2 ** 64 - 1),
                                               _ptr(out), ctypes.byref(n)))
        return out[:n.value]

    def hash_sample_pairs(self, bins, counts, want, seed_key):
        """hash_sample + gather_features + evaluate_samples in one call (device metric only)."""
        bins, counts, want = _c(bins, np.float64), _c(counts, np.int64), _c(want, np.int64)
        m = int(np.minimum(counts, want).sum())
        pos, feats, y = np.empty(m, dtype=np.int64), np.empty((m, 4), dtype=np.float64), np.empty(m, dtype=np.float64)
        n = _i64()
        self._chk(self.lib.annchor_hash_sample_pairs(self.h, _ptr(bins), len(bins) - 1, _ptr(counts), _ptr(want), int(seed_key) & (2 ** 64 - 1),
                                                     _ptr(pos), _ptr(feats), _ptr(y), ctypes.byref(n)))
        return pos[:n.value], feats[:n.value], y[:n.value]

    def gather_features(self, pos):
        pos = _c(pos, np.int64)
        out = np.zeros((len(pos), 4), dtype=np.float64)
        self._chk(self.lib.annchor_gather_features(self.h, _ptr(pos), len(pos), _ptr(out)))
        return out

    def evaluate_samples(self, pos):
        pos = _c(pos, np.int64)
        out = np.zeros(len(pos), dtype=np.float64)
        self._chk(self.lib.annchor_evaluate_samples(self.h, _ptr(pos), len(pos), _ptr(out)))
        return out

    def set_samples(self, pos, y):
        pos, y = _c(pos, np.int64), _c(y, np.float64)
        self._chk(self.lib.annchor_set_samples(self.h, _ptr(pos), len(pos), _ptr(y)))

    def predict_merge(self, bins, W, c, first, is_metric, n_samples):
        bins, W, c = _c(bins, np.float64), _c(W, np.float64), _c(c, np.float64)
        sp = np.zeros(n_samples, dtype=np.float64)
        self._chk(self.lib.annchor_predict_merge(self.h, _ptr(bins), len(bins) - 1, _ptr(W), _ptr(c), int(first),
                                                 int(is_metric), _ptr(sp)))
        return sp

    def n_pairs(self):
        return self.field_size(F_NCM)

    def merge_host_prediction(self, pred, first, is_metric):
        pred = _c(pred, np.float64).reshape(-1)
        if pred.shape[0] != self.n_pairs():
            raise ValueError("regression.predict returned %d values for %d candidate pairs" % (pred.shape[0], self.n_pairs()))
        self._chk(self.lib.annchor_merge_host_prediction(self.h, _ptr(pred), int(first), int(is_metric)))

    def set_labels(self, labels):
        labels = _c(labels, np.int64).reshape(-1)
        if labels.shape[0] != self.n_pairs():
            raise ValueError("error_predictor.predict returned %d labels for %d candidate pairs" % (labels.shape[0], self.n_pairs()))
        if labels.size and (labels.min() < 0 or labels.max() >= 255):
            raise ValueError("error labels must lie in 0..254 (got %d..%d)" % (labels.min(), labels.max()))
        self._chk(self.lib.annchor_set_labels(self.h, _ptr(labels)))

    def select_candidates(self, n_neighbors, nmin, errs_list, n_refine, lookahead, n_labels=None):
        """errs_list None: the residual lists fit_errors_device left on the device (n_labels of them)."""
        nc, nn = _i64(), _i64()
        if errs_list is None:
            self._chk(self.lib.annchor_select_candidates(self.h, int(n_neighbors), int(nmin), None, None, int(n_labels), int(n_refine),
                                                         int(lookahead), ctypes.byref(nc), ctypes.byref(nn)))
            return nc.value, nn.value
        ptr = np.zeros(len(errs_list) + 1, dtype=np.int64)
        np.cumsum([len(e) for e in errs_list], out=ptr[1:])
        errs = _c(np.concatenate(errs_list) if len(errs_list) else np.zeros(0), np.float64)
        self._chk(self.lib.annchor_select_candidates(self.h, int(n_neighbors), int(nmin), _ptr(errs), _ptr(ptr),
                                                     len(errs_list), int(n_refine), int(lookahead),
                                                     ctypes.byref(nc), ctypes.byref(nn)))
        return nc.value, nn.value

    def select_prepare(self, n_neighbors, nmin):
        """Thresholds + guarantee_nmin of the next select_candidates, launched ahead (no host wait)."""
        self._chk(self.lib.annchor_select_prepare(self.h, int(n_neighbors), int(nmin)))

    def mark_candidates(self):
        self._chk(self.lib.annchor_mark_candidates(self.h))

    def refine_candidates(self):
        self._chk(self.lib.annchor_refine_candidates(self.h))

    def park_refine(self, action):
        """1: the next sampler_stats call queues the refinement launch behind its download; 2: launch it now if still parked."""
        self._chk(self.lib.annchor_park_refine(self.h, int(action)))

    def set_refined(self, exact):
        exact = _c(exact, np.float64)
        self._chk(self.lib.annchor_set_refined(self.h, _ptr(exact), len(exact)))

    def update_bounds(self):
        self._chk(self.lib.annchor_update_bounds(self.h))

    def neighbor_graph(self, k):
        idx = np.zeros((self.nx, k), dtype=np.int64)
        dist = np.zeros((self.nx, k), dtype=np.float64)
        self._chk(self.lib.annchor_neighbor_graph(self.h, int(k), _ptr(idx), _ptr(dist)))
        return idx, dist

    # ------------------------------------------------------- streamed form
    def stream_bind(self, X, global_base=0, device_ptr=None, shape=None):
        """Bind this rank's rows: a float32 NumPy array, or a raw device pointer + shape."""
        if device_ptr is not None:
            n, d = shape
            self._chk(self.lib.annchor_stream_bind(self.h, device_ptr, int(n), int(d), int(global_base), 1))
        else:
            X = _c(X, np.float32)
            n, d = X.shape
            self._chk(self.lib.annchor_stream_bind(self.h, _ptr(X), n, d, int(global_base), 0))
        self.nx, self.metric = int(n), METRIC_EUCLIDEAN_F32
        self._stream_dim = int(d)

    def stream_anchor_round(self, anchor_vec, rnd, n_anchors):
        v = _c(anchor_vec, np.float32)
        mx, arg = _dbl(), _i64()
        self._chk(self.lib.annchor_stream_anchor_round(self.h, _ptr(v), int(rnd), int(n_anchors), ctypes.byref(mx), ctypes.byref(arg)))
        return mx.value, arg.value

    def stream_get_row(self, local_idx):
        out = np.zeros(self._stream_dim, dtype=np.float32)
        self._chk(self.lib.annchor_stream_get_row(self.h, int(local_idx), _ptr(out)))
        return out

    def stream_order(self, min_tiles=0):
        ptrs = [_vp() for _ in range(6)]
        n_pad, nt, dimp = _i64(), _i32(), _i32()
        self._chk(self.lib.annchor_stream_order(self.h, int(min_tiles), *[ctypes.byref(p) for p in ptrs], ctypes.byref(n_pad),
                                                ctypes.byref(nt), ctypes.byref(dimp)))
        names = ("Xs", "rs", "perm", "lo", "hi", "mid")
        return {k: p.value for k, p in zip(names, ptrs)}, n_pad.value, nt.value, dimp.value

    def stream_order_begin(self, min_tiles, tile_begin, tile_count):
        """The level sorts of the k-d order restricted to the caller's tile range; (device pointer of its slice of the order,
        the all-gather target, bytes per rank)."""
        ol, oa, nb = _vp(), _vp(), _i64()
        self._chk(self.lib.annchor_stream_order_begin(self.h, int(min_tiles), int(tile_begin), int(tile_count), ctypes.byref(ol), ctypes.byref(oa),
                                                      ctypes.byref(nb)))
        return ol.value, oa.value, nb.value

    def stream_order_end(self):
        ptrs = [_vp() for _ in range(6)]
        n_pad, nt, dimp = _i64(), _i32(), _i32()
        self._chk(self.lib.annchor_stream_order_end(self.h, *[ctypes.byref(p) for p in ptrs], ctypes.byref(n_pad), ctypes.byref(nt),
                                                    ctypes.byref(dimp)))
        names = ("Xs", "rs", "perm", "lo", "hi", "mid")
        return {k: p.value for k, p in zip(names, ptrs)}, n_pad.value, nt.value, dimp.value

    def stream_knn(self, ptrs, n_all, nt_all, n_anchors, dim_padded, tile_begin, tile_count, k, p_work, n_local=None, join_passes=0, join_extra=0,
                   out=None):
        """n_local given: graph rows come back in the bound shard's own row order ([n_local, k],
        row_ids is None); otherwise in tile order with row_ids (global id per row, -1 = padding).  out: GraphBuffers."""
        rows = tile_count * 128 if n_local is None else int(n_local)
        row_ids = np.zeros(rows, dtype=np.int64) if n_local is None else None
        ev = _i64()
        self._chk(self.lib.annchor_stream_knn_run(self.h, ptrs["Xs"], ptrs["rs"], ptrs["perm"], ptrs["lo"], ptrs["hi"], ptrs["mid"], int(n_all),
                                                  int(nt_all), int(n_anchors), int(dim_padded), int(tile_begin), int(tile_count),
                                                  int(k), float(p_work), int(join_passes), int(join_extra), ctypes.byref(ev)))
        # (the result arrays only now: at N = 8 x 10^6 the helper thread needs longer for its 1.9 GB than the anchor rounds and
        # the ordering take)
        if out is not None:
            idx, dist = out.take(rows, k)
        else:
            idx = np.empty((rows, k), dtype=np.int64)
            dist = np.empty((rows, k), dtype=np.float64)
        self._chk(self.lib.annchor_stream_knn_fetch(self.h, _ptr(row_ids) if row_ids is not None else None, _ptr(idx), _ptr(dist)))
        return row_ids, idx, dist, ev.value

    def stream_knn_begin(self, ptrs, n_all, nt_all, n_anchors, dim_padded, tile_begin, tile_count, k, tile_budget):
        """Tile phase of a row-sharded build (tile_budget column tiles per row tile: stream_budget);
        returns (device pointer, bytes) of this rank's lists."""
        lp, nb = _vp(), _i64()
        self._chk(self.lib.annchor_stream_knn_begin(self.h, ptrs["Xs"], ptrs["rs"], ptrs["perm"], ptrs["lo"], ptrs["hi"], ptrs["mid"],
                                                    int(n_all), int(nt_all), int(n_anchors), int(dim_padded), int(tile_begin),
                                                    int(tile_count), int(k), int(tile_budget), ctypes.byref(lp), ctypes.byref(nb)))
        self._knn_shape = (int(tile_count), int(k))
        return lp.value, nb.value

    def stream_knn_join(self, lists_all, per_pass):
        """One join pass against the all-gathered lists; returns (device pointer of the new local lists,
        list entries the pass replaced)."""
        lp, upd = _vp(), _i64()
        self._chk(self.lib.annchor_stream_knn_join(self.h, lists_all, int(per_pass), ctypes.byref(lp), ctypes.byref(upd)))
        return lp.value, upd.value

    def stream_join_rev_begin(self, lists_all):
        """Reverse neighbour lists of this rank's columns from the all-gathered lists; (device pointer of the slice, the gather target, bytes per rank)."""
        rl, ra, nb = _vp(), _vp(), _i64()
        self._chk(self.lib.annchor_stream_join_rev_begin(self.h, lists_all, ctypes.byref(rl), ctypes.byref(ra), ctypes.byref(nb)))
        return rl.value, ra.value, nb.value

    def stream_knn_end(self, n_local=None):
        tile_count, k = self._knn_shape
        rows = tile_count * 128 if n_local is None else int(n_local)
        row_ids = np.zeros(rows, dtype=np.int64) if n_local is None else None
        idx = np.empty((rows, k), dtype=np.int64)
        dist = np.empty((rows, k), dtype=np.float64)
        ev = _i64()
        self._chk(self.lib.annchor_stream_knn_end(self.h, _ptr(row_ids) if row_ids is not None else None, _ptr(idx), _ptr(dist),
                                                  ctypes.byref(ev)))
        return row_ids, idx, dist, ev.value

    # ---- row-sharded builds: device-resident exchange buffers (include/annchor_hip.h, csrc/sharded.hip)
    def hip_stream(self):
        """The context's hipStream_t (an integer address): collectives are ordered on it."""
        p = _vp()
        self._chk(self.lib.annchor_stream_hip_stream(self.h, ctypes.byref(p)))
        return p.value or 0

    def stream_anchor_begin(self, n_anchors, first_global, world=1):
        """(candidate pointer, gathered pointer, bytes per candidate): this rank's candidate record for the first anchor
        and the all-gather target of the rounds (device memory)."""
        p, g, nb = _vp(), _vp(), _i64()
        self._chk(self.lib.annchor_stream_anchor_begin(self.h, int(n_anchors), int(first_global), int(world), ctypes.byref(p), ctypes.byref(g),
                                                       ctypes.byref(nb)))
        return p.value, g.value, nb.value

    def stream_anchor_step(self, gathered, world, rnd):
        self._chk(self.lib.annchor_stream_anchor_step(self.h, gathered, int(world), int(rnd)))

    def legacy_generate_at_next_wait(self, seed, ndraws, chunk=0):
        """The legacy MT19937 stream of `seed` on this thread at the context's next host waits, `chunk` words per wait (returns at once)."""
        if 0 <= seed < 2 ** 32 and 0 < ndraws <= LEGACY_EAGER_MAX_DRAWS:
            self._chk(self.lib.annchor_legacy_generate_at_next_wait(self.h, int(seed), int(ndraws), int(chunk)))

    def stream_anchor_rounds(self, n_anchors):
        """All max-min rounds in one call (collectives from inside the library when the context has a communicator)."""
        self._chk(self.lib.annchor_stream_anchor_rounds(self.h, int(n_anchors)))

    # ---- in-library RCCL (csrc/comm.hip)
    def comm_init(self, unique_id, world, rank):
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        self._chk(self.lib.annchor_comm_init(self.h, buf, int(world), int(rank)))

    def comm_destroy(self):
        self._chk(self.lib.annchor_comm_destroy(self.h))

    def comm_set_timeout(self, seconds):
        self._chk(self.lib.annchor_comm_set_timeout(self.h, float(seconds)))

    def comm_preflight(self, seconds=60.0):
        self._chk(self.lib.annchor_comm_preflight(self.h, float(seconds)))

    def comm_init_side(self, unique_id):
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        self._chk(self.lib.annchor_comm_init_side(self.h, buf))

    def comm_allgather_begin(self, send, recv, nbytes):
        self._chk(self.lib.annchor_comm_allgather_begin(self.h, send, recv, int(nbytes)))

    def comm_side_join(self):
        self._chk(self.lib.annchor_comm_side_join(self.h))

    def comm_allgather(self, send, recv, nbytes):
        self._chk(self.lib.annchor_comm_allgather(self.h, send, recv, int(nbytes)))

    def comm_alltoall_records(self, send, send_counts, recv, recv_counts, words):
        sc, rc = _c(send_counts, np.int64), _c(recv_counts, np.int64)
        self._chk(self.lib.annchor_comm_alltoall_records(self.h, send, _ptr(sc), recv, _ptr(rc), int(words)))

    def stream_anchor_end(self, n_anchors):
        A = np.empty(int(n_anchors), dtype=np.int64)
        V = np.empty((int(n_anchors), self._stream_dim), dtype=np.float32)
        self._chk(self.lib.annchor_stream_anchor_end(self.h, _ptr(A), _ptr(V)))
        return A, V

    def stream_rows_begin(self, counts):
        counts = _c(counts, np.int64)
        s, r, nb = _vp(), _vp(), _i64()
        self._chk(self.lib.annchor_stream_rows_begin(self.h, len(counts), _ptr(counts), ctypes.byref(s), ctypes.byref(r), ctypes.byref(nb)))
        return s.value, r.value, nb.value

    def stream_anchor_dists_begin(self, counts):
        """Send / receive buffers and bytes per rank of the all-gather of the ranks' own anchor distances."""
        counts = _c(counts, np.int64)
        s, r, nb = _vp(), _vp(), _i64()
        self._chk(self.lib.annchor_stream_anchor_dists_begin(self.h, len(counts), _ptr(counts), ctypes.byref(s), ctypes.byref(r),
                                                             ctypes.byref(nb)))
        return s.value, r.value, nb.value

    def stream_rows_end(self, counts):
        counts = _c(counts, np.int64)
        self._chk(self.lib.annchor_stream_rows_end(self.h, len(counts), _ptr(counts)))
        self.nx = int(counts.sum())

    def stream_lists_all(self, world, bytes_per_rank):
        p = _vp()
        self._chk(self.lib.annchor_stream_lists_all(self.h, int(world), int(bytes_per_rank), ctypes.byref(p)))
        return p.value

    def stream_route_begin(self, starts, bases):
        """Finish the build begun with stream_knn_begin; (send pointer, records per destination rank, int64 words per
        record, tile evaluations)."""
        starts, bases = _c(starts, np.int64), _c(bases, np.int64)
        world = len(bases)
        send, words, ev = _vp(), _i64(), _i64()
        counts = np.zeros(world, dtype=np.int64)
        self._chk(self.lib.annchor_stream_route_begin(self.h, world, _ptr(starts), _ptr(bases), ctypes.byref(send), _ptr(counts),
                                                      ctypes.byref(words), ctypes.byref(ev)))
        return send.value, counts, words.value, ev.value

    def stream_route_recv(self, n_recv):
        p = _vp()
        self._chk(self.lib.annchor_stream_route_recv(self.h, int(n_recv), ctypes.byref(p)))
        return p.value

    def stream_route_end(self, n_recv, rows_padded, n_own, k, out=None):
        if out is not None:
            idx, dist = out.take(n_own, k)
        else:
            idx = np.empty((int(n_own), int(k)), dtype=np.int64)
            dist = np.empty((int(n_own), int(k)), dtype=np.float64)
        self._chk(self.lib.annchor_stream_route_end(self.h, int(n_recv), int(rows_padded), _ptr(idx), _ptr(dist)))
        return idx, dist

    def stream_graph_device(self):
        """(idx pointer, dist pointer, rows_padded, k) of the graph rows stream_route_end left on the device."""
        pi, pd, rows, k = _vp(), _vp(), _i64(), _i32()
        self._chk(self.lib.annchor_stream_graph_device(self.h, ctypes.byref(pi), ctypes.byref(pd), ctypes.byref(rows), ctypes.byref(k)))
        return pi.value, pd.value, rows.value, k.value

    def stream_query(self, cols, n_all, nt_all, n_anchors, dim_padded, nn, p_work):
        """nn nearest data rows of every (bound + ordered) query row; cols = the data set's column arrays."""
        idx = np.empty((self.nx, nn), dtype=np.int64)
        dist = np.empty((self.nx, nn), dtype=np.float64)
        ev = _i64()
        self._chk(self.lib.annchor_stream_query(self.h, cols["Xs"], cols["rs"], cols["perm"], cols["lo"], cols["hi"], cols["mid"],
                                                int(n_all), int(nt_all), int(n_anchors), int(dim_padded), int(nn), float(p_work),
                                                _ptr(idx), _ptr(dist), ctypes.byref(ev)))
        return idx, dist, ev.value

    def stream_last_counts(self):
        """(tile evaluations of the tile phase, 128-column runs of the join passes) of the last build."""
        a, b = _i64(), _i64()
        self._chk(self.lib.annchor_stream_last_counts(self.h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def stream_last_kernel(self, with_guard=False):
        """Kernel of the last build's tile phase: 0 exact float32 tile GEMMs, 1 split-fp16; with_guard: also the number of
        rows the split kernel flagged as ill-conditioned (more than 1 in 200 repeats the tile phase on the exact kernel)."""
        k, g = ctypes.c_int32(), _i64()
        self._chk(self.lib.annchor_stream_last_kernel(self.h, ctypes.byref(k), ctypes.byref(g)))
        return (int(k.value), int(g.value)) if with_guard else int(k.value)

    def stream_last_tile_kernels(self):
        """(two_stage, repaired) of the last build's tile phase: k_st_knnh ran behind the warm-up; flagged rows were repaired exactly."""
        a, b = ctypes.c_int32(), ctypes.c_int32()
        self._chk(self.lib.annchor_stream_last_tile_kernels(self.h, ctypes.byref(a), ctypes.byref(b)))
        return bool(a.value), bool(b.value)

    def stream_join_tables(self, gathered, world, n_anchors, n_tiles, joined):
        self._chk(self.lib.annchor_stream_join_tables(self.h, gathered, int(world), int(n_anchors), int(n_tiles), joined))

    def device_alloc(self, nbytes):
        p = _vp()
        self._chk(self.lib.annchor_device_alloc(self.h, int(nbytes), ctypes.byref(p)))
        return p.value

    def device_free(self, dptr):
        self._chk(self.lib.annchor_device_free(self.h, dptr))

    def device_copy(self, dst, src, nbytes, kind):
        self._chk(self.lib.annchor_device_copy(self.h, dst, src, int(nbytes), {"h2d": 1, "d2h": 2, "d2d": 3}[kind]))

    # ---- nearest enemies (csrc/enemies.hip)
    def enemies_candidates(self, codes, loc_thresh, loc_min):
        codes = _c(codes, np.int32)
        n = _i64()
        self._chk(self.lib.annchor_enemies_candidates(self.h, _ptr(codes), int(loc_thresh), int(loc_min), ctypes.byref(n)))
        self._n_enemy_pairs = n.value
        return n.value

    def enemies_predict(self, bins=None, W=None, c=None, pred=None):
        if pred is not None:
            pred = _c(pred, np.float64).reshape(-1)
            if pred.shape[0] != self._n_enemy_pairs:
                raise ValueError("regression.predict returned %d values for %d enemy pairs" % (pred.shape[0], self._n_enemy_pairs))
            self._chk(self.lib.annchor_enemies_predict(self.h, None, 0, None, None, _ptr(pred)))
            return
        bins, W, c = _c(bins, np.float64), _c(W, np.float64), _c(c, np.float64)
        self._chk(self.lib.annchor_enemies_predict(self.h, _ptr(bins), len(bins) - 1, _ptr(W), _ptr(c), None))

    def enemies_first(self, first, nn, evaluate):
        """evaluate: the todo pairs are evaluated on the device, returns their number; otherwise returns them (int64 [m, 2])."""
        m = _i64()
        if evaluate:
            self._chk(self.lib.annchor_enemies_first(self.h, int(first), int(nn), 1, None, 0, ctypes.byref(m)))
            return m.value
        cap = self.nx * int(first)
        todo = np.empty((cap, 2), dtype=np.int64)
        self._chk(self.lib.annchor_enemies_first(self.h, int(first), int(nn), 0, _ptr(todo), cap, ctypes.byref(m)))
        return todo[:m.value]

    def enemies_set_exact(self, exact):
        exact = _c(exact, np.float64)
        self._chk(self.lib.annchor_enemies_set_exact(self.h, _ptr(exact), len(exact)))

    def enemies_graph(self, nn):
        idx = np.empty((self.nx, nn), dtype=np.int64)
        dist = np.empty((self.nx, nn), dtype=np.float64)
        self._chk(self.lib.annchor_enemies_graph(self.h, int(nn), _ptr(idx), _ptr(dist)))
        return idx, dist

    def enemies_download(self):
        """(IJs_new [n, 2], features [n, 4], RA [n], ncm bool [n], I_ptr [nx + 1], I_idx [2 n]) of the enemy pairs."""
        n = self._n_enemy_pairs
        ij, feats = np.empty((n, 2), dtype=np.int64), np.empty((n, 4), dtype=np.float64)
        RA, ncm = np.empty(n, dtype=np.float64), np.empty(n, dtype=np.uint8)
        ptr, idx = np.empty(self.nx + 1, dtype=np.int64), np.empty(2 * n, dtype=np.int64)
        self._chk(self.lib.annchor_enemies_download(self.h, _ptr(ij), _ptr(feats), _ptr(RA), _ptr(ncm), _ptr(ptr), _ptr(idx)))
        return ij, feats, RA, ncm.astype(bool), ptr, idx

    def graph_to_coo(self, idx, dist):
        """Symmetric COO (rows, cols, vals) of a k-NN graph, every cell once, vals = distance + eps."""
        idx, dist = _c(idx, np.int64), _c(dist, np.float64)
        nx, k = idx.shape
        rows, cols = np.empty(2 * nx * k, dtype=np.int64), np.empty(2 * nx * k, dtype=np.int64)
        vals = np.empty(2 * nx * k, dtype=np.float64)
        nnz = _i64()
        self._chk(self.lib.annchor_graph_to_coo(self.h, _ptr(idx), _ptr(dist), nx, k, _ptr(rows), _ptr(cols), _ptr(vals), ctypes.byref(nnz)))
        n = nnz.value
        return rows[:n], cols[:n], vals[:n]

    # ---------------------------------------------------------- state access
    def field_size(self, field):
        n = _i64()
        self._chk(self.lib.annchor_field_size(self.h, int(field), ctypes.byref(n)))
        return n.value

    def download(self, field):
        n = self.field_size(field)
        out = np.zeros(n, dtype=_FIELD_DTYPE[field])
        if n:
            self._chk(self.lib.annchor_download(self.h, int(field), _ptr(out), n))
        return out

    def upload(self, field, arr):
        arr = _c(arr, _FIELD_DTYPE[field]).reshape(-1)
        self._chk(self.lib.annchor_upload(self.h, int(field), _ptr(arr), arr.size))

    # ------------------------------------------------------------ profiling
    def prof_enable(self, on=True):
        """on: False/0 off, True/1 every kernel family, 2 the metric kernels only."""
        self._chk(self.lib.annchor_prof_enable(self.h, int(on)))

    def prof_reset(self):
        self._chk(self.lib.annchor_prof_reset(self.h))

    def prof_get(self):
        cap = 64
        names = (ctypes.c_char_p * cap)()
        ms = np.zeros(cap, dtype=np.float64)
        launches = np.zeros(cap, dtype=np.int64)
        alg = np.zeros(cap, dtype=np.float64)
        n = self.lib.annchor_prof_get(self.h, cap, names, _ptr(ms), _ptr(launches), _ptr(alg))
        if n < 0:
            self._chk(n)
        return {names[i].decode(): dict(ms=float(ms[i]), launches=int(launches[i]), alg_bytes=float(alg[i]))
                for i in range(n)}
