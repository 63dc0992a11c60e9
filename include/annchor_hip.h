/*
 * annchor_hip.h -- C-ABI of libannchor_hip.so, the MI355X (gfx950) implementation of
 * the ANNchor `Annchor.fit()` -> `neighbor_graph` hot path.
 *
 * The reference (gchq/annchor) is pure Python + numba and has no FFI of its own; the
 * entry points below are exactly what a binding of its hot-path functions would
 * call.  Each one cites the reference function it replaces (paths relative to the
 * reference repository root).  INTEGRATION.md shows the ctypes stubs.
 *
 * Conventions
 *   - every function returns 0 on success or a negative ANNCHOR_E* code;
 *     annchor_last_error(ctx) gives a message for the last failure on that ctx;
 *   - all pointer arguments are HOST pointers owned by the caller; the library
 *     copies in/out and keeps no host pointer after returning;
 *   - all pipeline state lives in HBM inside the opaque context between calls;
 *   - one in-flight call per context (no internal locking); contexts are independent;
 *   - "pair position" = row index into the candidate pair list IJs built by
 *     annchor_build_locality (sorted by (i, j), i < j).
 */
#ifndef ANNCHOR_HIP_H
#define ANNCHOR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct annchor_ctx annchor_ctx;

enum {
    ANNCHOR_OK = 0,
    ANNCHOR_EINVAL = -1,   /* bad argument / call out of order            */
    ANNCHOR_EHIP = -2,     /* HIP runtime error (message has the details) */
    ANNCHOR_ENODEV = -3,   /* no usable GPU                               */
    ANNCHOR_ELIMIT = -4,   /* size outside what this build supports       */
    ANNCHOR_ESTATE = -5    /* algorithmic failure the reference raises on */
};

enum { ANNCHOR_METRIC_NONE = 0, ANNCHOR_METRIC_LEVENSHTEIN = 1, ANNCHOR_METRIC_EUCLIDEAN_F32 = 2,
       ANNCHOR_METRIC_EUCLIDEAN_F64 = 3, ANNCHOR_METRIC_WASSERSTEIN = 4, ANNCHOR_METRIC_COSINE_F32 = 5,
       ANNCHOR_METRIC_COSINE_F64 = 6 };

/* fields for annchor_download / annchor_upload */
enum {
    ANNCHOR_F_D = 1,        /* float64 [nx, na]  anchor distances (row-major as the reference's D) */
    ANNCHOR_F_A = 2,        /* int64   [nA]      anchor indices                                    */
    ANNCHOR_F_SID = 3,      /* uint64  [nx][w]   nearest-anchor bitmask (bit a = anchor a in sid); w = 1 / 2 / 4 words for <= 64 / 128 / 256 anchors */
    ANNCHOR_F_IJS = 4,      /* int64   [n, 2]    candidate pairs                                   */
    ANNCHOR_F_I_PTR = 5,    /* int64   [nx+1]    CSR offsets of I                                  */
    ANNCHOR_F_I_IDX = 6,    /* int64   [2n]      CSR pair positions of I                           */
    ANNCHOR_F_FEATURES = 7, /* float64 [n, 4]    lb, ub, dad, is_anchor                            */
    ANNCHOR_F_NCM = 8,      /* uint8   [n]       not_computed_mask                                 */
    ANNCHOR_F_RA = 9,       /* float64 [n]       RefineApprox                                      */
    ANNCHOR_F_LABELS = 10,  /* int64   [n]       error-bin label per pair                          */
    ANNCHOR_F_THRESH = 11,  /* float64 [nx]      per-row threshold                                 */
    ANNCHOR_F_PROB = 12,    /* float64 [n]       ECDF probability (-1 on computed pairs)           */
    ANNCHOR_F_CAND = 13,    /* int64   [ncand]   pair positions selected for refinement (ascending) */
    ANNCHOR_F_NEXT = 14,    /* int64   [nnext]   lookahead pair positions (ascending)              */
    ANNCHOR_F_DAD = 15      /* float64 [n]       double anchor distance only                       */
};

/* ---------------------------------------------------------------- lifecycle */
int annchor_create(int device, annchor_ctx **out);
void annchor_destroy(annchor_ctx *ctx);
/* A destroyed context parks its stream, pinned staging, events and device slab (up to 2 GB) for the
 * next context on the same device (at most four such shells per process).  This frees them; returns
 * how many there were. */
int annchor_release_parked(void);
/* Device memory this process holds for reuse on `device` (parked slabs + cached blocks of closed contexts): memory the next context
 * gets without asking the driver, so it counts as free when a data set is sized (annchor_amd/_native.py: pairlist_point_limit). */
int annchor_parked_bytes(int device, int64_t *bytes);
const char *annchor_last_error(annchor_ctx *ctx);
/* Non-ctx error text for failures of annchor_create itself. */
const char *annchor_create_error(void);
int annchor_device_name(annchor_ctx *ctx, char *buf, int buflen);
/* PCI bus id of a device ("0000:c1:00.0", buflen >= 16): for binding the process to the GPU's NUMA node. */
int annchor_device_pci_bus_id(int device, char *buf, int buflen);
/* Free / total memory of a device in bytes (the host derives the largest pair list it will materialise from it). */
int annchor_device_mem_info(int device, int64_t *free_bytes, int64_t *total_bytes);
int annchor_synchronize(annchor_ctx *ctx);
/* Device-side elapsed time (ms) of the work enqueued by the most recent call,
 * measured with HIP events on the context's stream. */
int annchor_last_kernel_ms(annchor_ctx *ctx, float *ms);

/* ------------------------------------------------------------------ data set
 * Replaces the `X` operand of get_exact(f, X, IJ) (annchor/utils.py:110-177).
 * Strings: symbols[offs[s] .. offs[s]+lens[s]) are dense symbol codes
 * 0..alphabet-1 (the host maps characters to codes). */
int annchor_set_strings(annchor_ctx *ctx, const uint8_t *symbols, const int64_t *offs,
                        const int32_t *lens, int64_t nx, int32_t alphabet);
/* The same for alphabets of 257 .. 65 535 distinct symbols: 16-bit dense codes (a Unicode corpus).  Levenshtein then runs the
 * kernel that computes its match words per column (about six times slower than the table-driven ones, exact for any alphabet). */
int annchor_set_strings_u16(annchor_ctx *ctx, const uint16_t *symbols, const int64_t *offs, const int32_t *lens, int64_t nx,
                            int32_t alphabet);
int annchor_set_points_f32(annchor_ctx *ctx, const float *X, int64_t nx, int32_t dim);
int annchor_set_points_f64(annchor_ctx *ctx, const double *X, int64_t nx, int32_t dim);
/* Same data, metric = cosine distance 1 - u.v / (|u| |v|), clipped to [0, 2]
 * (annchor/utils.py:14,67 -> scipy.spatial.distance.cosine). */
int annchor_set_points_cosine_f32(annchor_ctx *ctx, const float *X, int64_t nx, int32_t dim);
int annchor_set_points_cosine_f64(annchor_ctx *ctx, const double *X, int64_t nx, int32_t dim);
/* Wasserstein: hist float64 [nx, nbins], cost float64 [nbins, nbins]
 * (annchor/utils.py:75-86, func_kwargs['cost_matrix']).  Up to 64 bins: any histograms, any cost matrix.  65 .. 1024 bins:
 * histograms with at most 32 non-zero entries each under a metric ground cost (zero diagonal, triangle inequality) -- kept as
 * (bin, mass) lists; anything else returns ANNCHOR_ELIMIT. */
int annchor_set_histograms(annchor_ctx *ctx, const double *hist, int64_t nx, int32_t nbins,
                           const double *cost);
/* Data set without a device metric (user metric evaluated on the host). */
int annchor_set_opaque(annchor_ctx *ctx, int64_t nx);

/* --------------------------------------------------------- metric boundary a2
 * get_exact(f, X, IJ): out[t] = f(X[IJ[t,0]], X[IJ[t,1]]) as float64
 * (annchor/utils.py:110-177).  ij: int64 [n, 2]; may contain i == j and repeats. */
int annchor_metric_pairs(annchor_ctx *ctx, const int64_t *ij, int64_t n, double *out);
/* BruteForce.fit (annchor/annchor.py:1004-1023): all-pairs metric + per-row stable
 * sort; writes the first k columns of (argsort(D), sort(D)). */
int annchor_brute_force(annchor_ctx *ctx, int32_t k, int64_t *ng_idx, double *ng_dist);

/* ------------------------------------------------------------------ anchors a6
 * MaxMinAnchorPicker.get_anchors (annchor/pickers.py:18-52).  `first` is the
 * host-drawn np.random.randint(nx).  Fills A and D inside the context. */
int annchor_pick_anchors_maxmin(annchor_ctx *ctx, int32_t n_anchors, int64_t first);
/* SelectedAnchorPicker / RandomAnchorPicker (pickers.py:86-128): given indices. */
int annchor_pick_anchors_selected(annchor_ctx *ctx, const int64_t *A, int32_t n_anchors);
/* Any other picker (ExternalAnchorPicker, user classes): the host hands over
 * (A, D, evals) as AnchorPicker.get_anchors returns them.  D float64 [nx, na]. */
int annchor_set_anchor_distances(annchor_ctx *ctx, const double *D, int32_t n_anchors,
                                 const int64_t *A, int32_t nA);

/* ----------------------------------------------------------------- locality a7
 * Annchor.get_locality + get_check/adjust_check/get_IJs_from_check
 * (annchor/annchor.py:208-256, annchor/utils.py:437-540).  Returns the number of
 * candidate pairs and the smallest |I[i]| (annchor.py:252-256 raises when it is
 * below n_neighbors -- the host does that). */
int annchor_build_locality(annchor_ctx *ctx, int32_t locality, int32_t loc_thresh, int32_t loc_min,
                           int64_t *n_pairs, int64_t *min_row_len);

/* Query form (Annchor.query, annchor/query_functions.py:18-62): the bound data set is X
 * (rows [0, nx_base)) followed by the queries; candidate pairs are (i, nx_base + j) with
 * |sid[i] & sid[nx_base + j]| >= loc_thresh, sorted by (j, i); rows of X are empty. */
int annchor_build_query_locality(annchor_ctx *ctx, int64_t nx_base, int32_t locality, int32_t loc_thresh,
                                 int64_t *n_pairs, int64_t *min_row_len);

/* ------------------------------------------------------------- features a8-a10
 * get_bounds_njit_ijs, get_dad_ijs, get_features_IJ (utils.py:274-301,355-380;
 * annchor.py:258-303): lb, ub, dad, is_anchor, not_computed_mask for every pair. */
int annchor_compute_features(annchor_ctx *ctx);

/* ------------------------------------------------------------------ sample a11
 * Helpers for Sampler.sample (annchor/samplers.py:75-140) on the device-resident
 * state.  kth_uncomputed_dad: np.partition(dad[ncm], k)[k] for each requested k. */
int annchor_count_uncomputed(annchor_ctx *ctx, int64_t *n_unc);
int annchor_kth_uncomputed_dad(annchor_ctx *ctx, const int64_t *ks, int32_t nk, double *out);
/* counts[b] = #{uncomputed pairs with bins[b] <= dad < bins[b+1]} (utils.py:547-549). */
int annchor_bin_counts(annchor_ctx *ctx, const double *bins, int32_t nbins, int64_t *counts);
/* Sampler.get_partition + the partition populations in ONE host round trip (samplers.py:75-105): q[0..1] = the ks[0]-th / ks[1]-th
 * smallest dad among the uncomputed pairs; *fused = 1: edges[0..n_partitions] = {-inf, np.linspace(q[0], q[1], n_partitions - 1),
 * +inf} computed on the device (the caller compares them with its own np.linspace) and counts[b] as annchor_bin_counts would
 * give for them; *fused = 0 (very long lists, degenerate keys): only q is set, the caller continues with annchor_bin_counts. */
int annchor_sampler_stats(annchor_ctx *ctx, const int64_t *ks, int32_t n_partitions, double *q, double *edges, int64_t *counts,
                          int32_t *fused);
/* For each request t: the pair position of the ranks[t]-th (0-based, position
 * order) uncomputed pair inside bin bin_of[t]. */
int annchor_select_by_rank(annchor_ctx *ctx, const double *bins, int32_t nbins, const int32_t *bin_of,
                           const int64_t *ranks, int64_t nreq, int64_t *positions);
/* Host-only helper (no device work, no context): the draws of
 * `np.random.seed(seed)` followed, per bin, by
 * `np.random.choice(ixmask, want[b], replace=False)` (annchor/utils.py:543-578),
 * returned as ranks into the bin (= positions of the legacy permutation prefix);
 * a bin with counts[b] < want[b] yields 0..counts[b]-1 without consuming the stream.
 * ranks_out must hold sum(min(counts, want)); n_out[b] = entries written for bin b. */
/* Start generating the raw MT19937 stream of `seed` on a background host thread (it
 * depends on the seed only); a later annchor_legacy_choice_ranks(seed, ...) consumes it.
 * Purely an overlap device: results are identical with or without it. */
int annchor_legacy_prefetch(uint32_t seed, int64_t ndraws);
/* The same stream generated on the CALLING thread before the call returns (a warm core: ~0.4 ms for the 1.9 M words of
 * a C2 sampling step, against ~2 ms on a freshly woken producer thread); fit() calls it while the GPU runs a stage the
 * host would otherwise only wait for. */
int annchor_legacy_generate(uint32_t seed, int64_t ndraws);
/* ... on the calling thread at the context's NEXT host waits, `chunk` words per wait (<= 0: all at the first): after the caller's
 * next enqueues, while the GPU works on them */
int annchor_legacy_generate_at_next_wait(annchor_ctx *ctx, uint32_t seed, int64_t ndraws, int64_t chunk);
int annchor_legacy_choice_ranks(uint32_t seed, const int64_t *counts, const int64_t *want, int32_t nbins,
                                int64_t *ranks_out, int64_t *n_out);
/* The same draw on the library's persistent worker thread (warm core, warm caches), so that it
 * overlaps device work the caller enqueues meanwhile: begin() copies the inputs and returns at
 * once; end() waits, writes ranks_out / n_out as annchor_legacy_choice_ranks would, returns its
 * status and releases the ticket. */
int annchor_legacy_choice_begin(uint32_t seed, const int64_t *counts, const int64_t *want, int32_t nbins, void **ticket);
int annchor_legacy_choice_end(void *ticket, int64_t *ranks_out, int64_t *n_out);

/* Host-only helper (no device work, no context): the per-partition ordinary least squares of
 * SimpleStratifiedLinearRegression.fit (annchor/regressors.py:60-84; sklearn LinearRegression =
 * centre, LAPACK dgelsd, intercept) for all partitions in one call.  dgelsd_ptr: the Fortran
 * dgelsd of the caller's LAPACK (Python: scipy.linalg.cython_lapack.__pyx_capi__["dgelsd"]).
 * X: nf columns of ld doubles (column k at X + k * ld), y: ld doubles, samples grouped by
 * partition: partition b = rows cuts[b] .. cuts[b+1].  Out: coef [nbins, nf], xmean [nbins, nf],
 * ymean [nbins] (the intercept is ymean - xmean . coef), status [nbins] (0 = solved, 1 = fewer
 * rows than features, > 1 = LAPACK failure: solve that partition on the caller's general path). */
int annchor_ols_bins(void *dgelsd_ptr, const double *X, const double *y, int64_t ld, int32_t nf, const int64_t *cuts,
                     int32_t nbins, double *coef, double *xmean, double *ymean, int32_t *status);
/* Gather features [m, 4] at the given pair positions (self.features[sample_ixs]). */
int annchor_gather_features(annchor_ctx *ctx, const int64_t *pos, int64_t m, double *feats);
/* get_sample (annchor.py:336-343): evaluate the metric on the sample pairs, clear
 * their not_computed_mask bit, remember (positions, y) for the merge. */
int annchor_evaluate_samples(annchor_ctx *ctx, const int64_t *pos, int64_t m, double *sample_y);
/* The built-in sampling step in one call (get_sample, annchor.py:313-343): (bin, rank) -> pair
 * positions [nreq], their feature rows [nreq, 4], their exact distances [nreq]; clears
 * not_computed_mask for them.  counts = annchor_bin_counts for the same edges.  Equivalent to
 * annchor_select_by_rank + annchor_gather_features + annchor_evaluate_samples with one host
 * wait and no host round trip of the positions. */
int annchor_sample_pairs(annchor_ctx *ctx, const double *bins, int32_t nbins, const int64_t *counts, const int32_t *bin_of,
                         const int64_t *ranks, int64_t nreq, int64_t *positions, double *feats, double *sample_y);
/* Same, when the metric was evaluated by the host. */
int annchor_set_samples(annchor_ctx *ctx, const int64_t *pos, int64_t m, const double *sample_y);

/* ----------------------------------------------------- regression/errors a12-13
 * SimpleStratifiedLinearRegression.predict + clip + RefineApprox merge
 * (regressors.py:71-103, annchor.py:356-380) and
 * SimpleStratifiedErrorRegression.predict (error_predictors.py:56-67), fused.
 * bins float64 [nb+1], W float64 [nb,3], c float64 [nb].  sample_predict receives
 * the UNCLIPPED prediction at the current sample positions (annchor.py:357). */
int annchor_predict_merge(annchor_ctx *ctx, const double *bins, int32_t nb, const double *W,
                          const double *c, int32_t first_iteration, int32_t is_metric,
                          double *sample_predict);
/* Custom Regression / ErrorPredictor plugins: host-computed arrays are merged. */
int annchor_merge_host_prediction(annchor_ctx *ctx, const double *pred, int32_t first_iteration,
                                  int32_t is_metric);
int annchor_set_labels(annchor_ctx *ctx, const int64_t *labels);

/* ------------------------------------------- a11-a13 with the iteration's models fitted on the device
 * The same three stages -- get_sample (annchor.py:313-343), fit_predict_regression (annchor.py:345-380,
 * regressors.py:39-103), fit_predict_errors (annchor.py:382-393, error_predictors.py:26-53) -- without a host round
 * trip between them, for the reference's default plugins and a device metric:
 *   annchor_sample_pairs_device   = annchor_sample_pairs, results left in device memory;
 *   annchor_fit_regression_device = per-partition OLS with intercept on (lb, ub, dad) -> y (Householder QR on the centred
 *     samples, float64; coefficients agree with the reference's LAPACK dgelsd solution to ~1e-13 relative, not bit for
 *     bit -- annchor_predict_merge with host-fitted coefficients remains for that), then the fused predict / clip /
 *     merge / label pass from the coefficients in place;
 *   annchor_fit_errors_device     = per-partition sorted residuals into the context (annchor_select_candidates with
 *     errs == NULL reads them there).
 * Nothing waits for the host.  Problems a kernel finds (a sampled (bin, rank) that does not exist, a partition the QR
 * does not take -- rank deficient or fewer rows than columns --, an empty residual list) raise sticky flags that
 * annchor_model_download returns (and clears) together with the fitted coefficients; the caller redoes the step on
 * the host path then.  annchor_download_samples / annchor_errors_download materialise the sample arrays and the
 * residual lists for the host (plugin-visible attributes; not on the fit path). */
int annchor_sample_pairs_device(annchor_ctx *ctx, const double *bins, int32_t nbins, const int64_t *counts, const int32_t *bin_of,
                                const int64_t *ranks, int64_t nreq);
/* The same for DeviceStratifiedSampler's order-free choice (annchor_hash_sample_pairs with the results left in device memory
 * and no host wait: a partition whose key list came out short or overflowed raises the sticky sample-step flag, read with
 * annchor_model_download).  *n_out = sum over the partitions of min(want, counts). */
int annchor_hash_sample_pairs_device(annchor_ctx *ctx, const double *bins, int32_t nbins, const int64_t *counts, const int64_t *want,
                                     uint64_t seed_key, int64_t *n_out);
/* The built-in sampling step with the LEGACY draw (SimpleStratifiedSampler: NumPy's stream, utils.py:543-578) in one call, the
 * draw's backward trace on the device: the host walks the stream (the rejection scan), every bin's swap partners are uploaded on a
 * side stream as they complete, three kernels undo the swaps for the kept entries and write the chosen ranks into the slot map
 * the rank -> position kernels read.  Same samples, same order as annchor_legacy_choice_ranks + annchor_sample_pairs_device.
 * *taken = 0: not applicable here (a partition keeps more than 8192 entries): draw on the host.  The second form returns
 * the ranks alone (tests). */
int annchor_sample_pairs_device_draw(annchor_ctx *ctx, const double *bins, int32_t nbins, const int64_t *counts, const int64_t *want,
                                     uint32_t seed, int64_t *n_out, int32_t *taken);
int annchor_legacy_choice_ranks_device(annchor_ctx *ctx, uint32_t seed, const int64_t *counts, const int64_t *want, int32_t nbins,
                                       int64_t *ranks_out, int32_t *taken);
int annchor_download_samples(annchor_ctx *ctx, int64_t *positions, double *feats, double *sample_y, double *sample_predict);
int annchor_fit_regression_device(annchor_ctx *ctx, const double *bins, int32_t nbins, int32_t first_iteration, int32_t is_metric);
int annchor_fit_errors_device(annchor_ctx *ctx);
int annchor_model_download(annchor_ctx *ctx, double *W, double *c, int32_t *status, int64_t *err_ptr, int32_t *flags /*[3]*/);
int annchor_errors_download(annchor_ctx *ctx, double *errs, int64_t n_errs);
/* Both behind one host wait (what fit() calls after its last iteration): errs has room for errs_cap doubles (2 x n_samples
 * always suffices); *n_errs = err_ptr[nbins], or -1 when the flags report a failed step. */
int annchor_model_download_with_errors(annchor_ctx *ctx, double *W, double *c, int32_t *status, int64_t *err_ptr, int32_t *flags /*[3]*/,
                                       double *errs, int64_t errs_cap, int64_t *n_errs);

/* --------------------------------------------------------------- selection a14
 * select_refine_candidate_pairs (annchor.py:395-473) up to, not including, the
 * metric call: thresh, guarantee_nmin (when nmin > 0), p, ECDF prob (errs:
 * concatenated sorted residuals, err_ptr int64 [nlabels+1]; errs == NULL: the lists
 * annchor_fit_errors_device left on the device), top-n_refine and
 * lookahead selection.  Ties at a cut (np.argpartition is arbitrary there): probability descending, then
 * the fixed pseudo-random order (position * 0x9E3779B97F4A7C15 mod 2^64) >> 11 ascending, then position
 * ascending. */
int annchor_select_candidates(annchor_ctx *ctx, int32_t n_neighbors, int32_t nmin, const double *errs,
                              const int64_t *err_ptr, int32_t nlabels, int64_t n_refine,
                              int32_t lookahead, int64_t *n_cand, int64_t *n_next);
/* The part of annchor_select_candidates that depends on RefineApprox and the mask only -- row thresholds
 * (annchor.py:399-404) and guarantee_nmin (utils.py:600-621) -- launched ahead, without a host wait: the caller
 * fits its error model (error_predictors.py:26-53, host) while it runs, then calls annchor_select_candidates
 * with the same n_neighbors / nmin, which picks up from there.  Any call that changes RefineApprox or the mask
 * in between voids the preparation (the selection then starts from the beginning). */
int annchor_select_prepare(annchor_ctx *ctx, int32_t n_neighbors, int32_t nmin);
/* Clear not_computed_mask for the selected candidates ahead of their refinement
 * (annchor.py:473 does it after the metric calls).  Lets the next iteration's sampling
 * statistics (which depend on the mask and dad only, samplers.py:119-140) be taken, and the
 * host-side draw be made, while the refinement kernel runs.  Idempotent with
 * annchor_refine_candidates / annchor_set_refined. */
int annchor_mark_candidates(annchor_ctx *ctx);
/* Evaluate the metric on the selected candidates and write back
 * (annchor.py:467-473). */
int annchor_refine_candidates(annchor_ctx *ctx);
/* The max-min picker's Levenshtein rounds run as ONE persistent launch when every wave of it can be resident (reference: the
 * loop of annchor/pickers.py:44-50).  A launch that gives up (GPU shared with another process) falls back to a one-workgroup form
 * after its time limit; after two such launches the library switches the persistent form off for the process.
 * set = 0: switch it off, 1: re-arm it, other: query.  Returns 1 when it is armed. */
int annchor_lev_persist_state(int set);

/* fit() pipelining: action 1 parks the refinement launch so that the next annchor_sampler_stats (the NEXT iteration's sampling
 * statistics, which depend on the candidate marks only) queues it behind its download and waits for the statistics alone;
 * action 2 launches it now if it is still parked. */
int annchor_park_refine(annchor_ctx *ctx, int32_t action);
/* Same, host-evaluated metric: exact float64 [n_cand] in ANNCHOR_F_CAND order. */
int annchor_set_refined(annchor_ctx *ctx, const double *exact, int64_t n_cand);

/* ----------------------------------------------------------- bound update a15
 * update_anchor_points + update_bounds/get_bounds_alt (annchor.py:475-512,
 * utils.py:304-352) over the lookahead pairs, all chunks (no wall-clock cut). */
int annchor_update_bounds(annchor_ctx *ctx);

/* ------------------------------------------------------------------ graph a16
 * get_ann + get_nn (annchor.py:514-530, utils.py:383-429).  ng_idx int64 [nx, k],
 * ng_dist float64 [nx, k], column 0 = self. */
int annchor_neighbor_graph(annchor_ctx *ctx, int32_t n_neighbors, int64_t *ng_idx, double *ng_dist);

/* ------------------------------------------------ streamed form (large-N Euclidean)
 * Tile-granular form of the same path for float32 points when the pair list cannot
 * exist (N >> 10^4).  One context = one GPU = one
 * contiguous shard of rows [global_base, global_base + n_local).  See
 * annchor_amd/csrc/streamed.hip for how each stage of Annchor.fit()
 * (annchor/annchor.py:532-623) maps onto tiles.
 *
 * annchor_stream_bind: copy the shard (host or device pointer) into the context.
 * annchor_stream_anchor_round: one max-min round (annchor/pickers.py:44-50) on the shard:
 *   distances of every local row to `anchor_vec` (host float32 [dim]), running-min update,
 *   local arg-max (local index, first index on ties).  The host combines ranks.
 * annchor_stream_get_row: fetch one local row (the next anchor's coordinates).
 * annchor_stream_order: order the bound rows into 128-row tiles (balanced k-d splits in
 *   anchor-distance space) with their anchor-distance intervals; returns DEVICE pointers (owned by
 *   the context): Xs float32 [n_pad, dim_padded], rs float32 [n_pad], perm int64 [n_pad] (global
 *   ids, -1 on padding), lo/hi/mid float32 [n_anchors, n_tiles] (per-tile min / max / mean distance
 *   to each anchor).  min_tiles pads the tile count (e.g. to a multiple of the rank count).
 *   Row-sharded runs (annchor_amd/streamed.py): after the sharded anchor rounds every rank binds ALL
 *   rows (all-gathered raw shards, device pointer), replays the anchors through
 *   annchor_stream_anchor_round and orders them itself -- one tile structure whatever the rank
 *   count -- and owns a contiguous range of the global tile order.
 * annchor_stream_knn: k-NN rows of tiles [tile_begin, tile_begin+tile_count) against all
 *   column tiles (DEVICE pointers, concatenated over ranks); get_ann-shaped HOST outputs
 *   (annchor/annchor.py:514-530): column 0 = self.  Two phases: the budgeted tile phase
 *   (ceil(p_work * n_tiles) column tiles per row tile, the work budget of annchor.py:438-442),
 *   then `join_passes` passes of the streamed update_anchor_points (annchor.py:475-512,
 *   utils.py:304-352): pairs that share a computed neighbour -- j in list(c), c in list(i) --
 *   are evaluated exactly, per row tile, as gathered tile GEMMs; up to `join_extra` further
 *   passes run while a pass still replaces more than 1 % of the list entries (small data sets,
 *   whose tile budget is coarse).  Join passes need the launch to cover every row tile (one rank).
 * annchor_stream_knn_begin / _join / _end: the same build in steps for row-sharded runs: the
 *   ranks all-gather their list buffers (*lists_local: int32 [tile_count*128][k-1] ordered column
 *   indices, device memory; lists_bytes its size) after _begin and after every _join into
 *   lists_all [n_all][k-1], the input of the next _join (*updates: list entries the pass replaced
 *   on this rank); _end = annchor_stream_knn's outputs. */
int annchor_stream_bind(annchor_ctx *ctx, const float *X, int64_t n_local, int32_t dim, int64_t global_base,
                        int32_t x_on_device);
int annchor_stream_anchor_round(annchor_ctx *ctx, const float *anchor_vec, int32_t round, int32_t n_anchors,
                                double *local_max, int64_t *local_arg);
int annchor_stream_get_row(annchor_ctx *ctx, int64_t local_idx, float *out);
int annchor_stream_order(annchor_ctx *ctx, int32_t min_tiles, void **Xs, void **rs, void **perm, void **lo, void **hi,
                         void **mid, int64_t *n_pad, int32_t *n_tiles, int32_t *dim_padded);
/* The ordering in two steps, for row-sharded runs (no reference counterpart: annchor/utils.py:494-540 materialises every
 * pair; this is the streamed form's locality stage).  _order_begin runs the k-d level sorts restricted at
 * every level to the segments that hold the caller's tile range [tile_begin, tile_begin + tile_count) (tile_count <= 0:
 * all tiles) -- a rank that owns 1 / G of the tiles sorts ~(levels / G + 2) n keys instead of levels x n -- and returns
 * its slice of the order (*order_local: uint32 [tile_count x 128], device), the all-gather target (*order_all: uint32
 * [n_tiles x 128]; rank r's slice at r x *order_bytes) and the slice's size.  _order_end (after the all-gather; at once
 * when the range was everything) gathers the rows into tile order and returns annchor_stream_order's outputs. */
int annchor_stream_order_begin(annchor_ctx *ctx, int32_t min_tiles, int32_t tile_begin, int32_t tile_count, void **order_local,
                               void **order_all, int64_t *order_bytes);
int annchor_stream_order_end(annchor_ctx *ctx, void **Xs, void **rs, void **perm, void **lo, void **hi, void **mid, int64_t *n_pad,
                             int32_t *n_tiles, int32_t *dim_padded);
int annchor_stream_knn(annchor_ctx *ctx, const void *Xs_all, const void *rs_all, const void *perm_all, const void *lo_all,
                       const void *hi_all, const void *mid_all, int64_t n_all, int32_t nt_all, int32_t n_anchors, int32_t dim_padded,
                       int32_t tile_begin, int32_t tile_count, int32_t k, double p_work, int32_t join_passes, int32_t join_extra,
                       int64_t *row_ids, int64_t *ng_idx, double *ng_dist, int64_t *tile_evals);
/* annchor_stream_knn in two halves: _run leaves the finished graph on the device, _fetch downloads it (same outputs as
 * annchor_stream_knn) -- for hosts that prepare the result arrays on another thread while the GPU works. */
int annchor_stream_knn_run(annchor_ctx *ctx, const void *Xs_all, const void *rs_all, const void *perm_all, const void *lo_all,
                           const void *hi_all, const void *mid_all, int64_t n_all, int32_t nt_all, int32_t n_anchors,
                           int32_t dim_padded, int32_t tile_begin, int32_t tile_count, int32_t k, double p_work, int32_t join_passes,
                           int32_t join_extra, int64_t *tile_evals);
int annchor_stream_knn_fetch(annchor_ctx *ctx, int64_t *row_ids, int64_t *ng_idx, double *ng_dist);
int annchor_stream_knn_begin(annchor_ctx *ctx, const void *Xs_all, const void *rs_all, const void *perm_all, const void *lo_all,
                             const void *hi_all, const void *mid_all, int64_t n_all, int32_t nt_all, int32_t n_anchors,
                             int32_t dim_padded, int32_t tile_begin, int32_t tile_count, int32_t k, int32_t tile_budget,
                             void **lists_local, int64_t *lists_bytes);
int annchor_stream_knn_join(annchor_ctx *ctx, const void *lists_all, int32_t per_pass, void **lists_local, int64_t *updates);
/* Row-sharded join pass, first half (the streamed analogue of update_anchor_points' computed-neighbour lists,
 * annchor/annchor.py:475-512, restricted to the columns a rank owns): the reverse neighbour lists of this rank's tile range
 * from the all-gathered lists.  *rev_local: int32 [tile_count x 128][15] (device: the rank's slice), *rev_all: int32 [n_all][15], *rev_bytes: bytes per rank.  The host all-gathers the
 * slices and calls annchor_stream_knn_join with the same lists_all; a host that skips this call gets every column's
 * reverse list built by the join pass itself (one rank). */
int annchor_stream_join_rev_begin(annchor_ctx *ctx, const void *lists_all, void **rev_local, void **rev_all, int64_t *rev_bytes);
/* The work budget of the streamed form.  p_work is the share of THIS form's brute force -- n_tiles
 * tile evaluations per row tile -- that one row tile may spend, all phases included:
 * total = ceil(p_work * n_tiles) = tile_phase + join_passes * per_pass (per_pass = runs of 128
 * gathered columns one join pass may evaluate; when a tile has more candidates, the ones reached
 * over the most two-hop paths are kept). */
int annchor_stream_budget(int32_t n_tiles, double p_work, int32_t join_passes, int32_t *total, int32_t *tile_phase,
                          int32_t *per_pass);
int annchor_stream_knn_end(annchor_ctx *ctx, int64_t *row_ids, int64_t *ng_idx, double *ng_dist, int64_t *tile_evals);
/* Split of the last build's tile_evals: tile evaluations of the tile phase, 128-column runs of the join passes. */
int annchor_stream_last_counts(annchor_ctx *ctx, int64_t *tile_phase_evals, int64_t *join_chunks);
/* Which kernel evaluated the last build's tile phase: 0 = exact float32 tile GEMMs (v_mfma_f32_32x32x2_f32; padded
 * dim 256, more than 30 neighbours, or the fallback below), 1 = split-fp16 tile GEMMs (three v_mfma_f32_32x32x16_f16 per
 * 16 dimensions on centred, power-of-two scaled rows: products to ~2^-22 |x||y|, the accuracy of the f32 MFMA stream; K + 2 columns kept per row and re-ranked by their exact float32 distances;
 * csrc/knnbf.hip).  *guard_rows = rows the split kernel flagged: its K-th exact distance came within twice the measured
 * error of the products of the list's last approximate entry, i.e. a neighbour may have stayed outside the list; when
 * more than 1 row in 200 is flagged the tile phase is repeated on the exact kernel (kind 0 is reported then).  The
 * reported neighbour distances are exact float32 in every case (the metric of the reference on float32 rows:
 * distances.py:8-13). */
int annchor_stream_last_kernel(annchor_ctx *ctx, int32_t *kind, int64_t *guard_rows);
/* Round 6.  *two_stage = 1: the tile phase of the last build ran k_st_knnh (csrc/knnh.hip) behind a short k_st_knnbf warm-up -- fp16
 * hi-only products with a rigorous error bound decide which columns MAY enter a row's list, float32 differences of the original rows
 * (the reference's arithmetic, distances.py:8-13) decide which do: graph builds at padded dimension 128, <= 14 neighbours kept.
 * *repaired = 1: rows flagged by the split kernels' guard (see above) were evaluated again with float32 differences over the column
 * tiles their row tile evaluated (csrc/repair.hip: any dimension) -- this replaces both the 1-in-200 allowance and the repetition on
 * the exact-f32 kernel (ANNCHOR_ST_FALLBACK=rerun keeps those). */
int annchor_stream_last_tile_kernels(annchor_ctx *ctx, int32_t *two_stage, int32_t *repaired);
/* Queries against a fitted data set in the streamed form (Annchor.query, annchor.py:643-683 ->
 * query_functions.py:183-212, for data sets beyond the pair-list form).  The context holds
 * the QUERY rows: bound with annchor_stream_bind (global_base 0), given the data set's anchor
 * vectors through annchor_stream_anchor_round, ordered with annchor_stream_order.  The six
 * column arrays are the data set's.  out_idx int64 [nq, nn] (global ids), out_dist float64
 * [nq, nn], in the queries' own order. */
int annchor_stream_query(annchor_ctx *ctx, const void *Xs_all, const void *rs_all, const void *perm_all, const void *lo_all,
                         const void *hi_all, const void *mid_all, int64_t n_all, int32_t nt_all, int32_t n_anchors, int32_t dim_padded,
                         int32_t nn, double p_work, int64_t *out_idx, double *out_dist, int64_t *tile_evals);
/* Interval tables after a rank-major all-gather ([world][n_anchors][n_tiles], device) joined along
 * the tile axis ([n_anchors][world * n_tiles], device): the layout the column arguments above use.
 * (For hosts that order every shard on its own rank and all-gather the ordered shards; the bundled
 * host no longer does -- see annchor_stream_order.) */
int annchor_stream_join_tables(annchor_ctx *ctx, const void *gathered, int32_t world, int32_t n_anchors, int32_t n_tiles,
                               void *joined);
/* ---- Row-sharded builds with device-resident exchange buffers (annchor_amd/csrc/sharded.hip; SURVEY.md section 8(e);
 * no reference counterpart: the reference has no collectives).  One process per GPU; the host hands the DEVICE buffers
 * below to its collective library (RCCL over xGMI) on the context's stream (annchor_stream_hip_stream) -- no host
 * round trip per max-min round, no host staging of rows, lists or results.
 *
 * Anchors (annchor/pickers.py:44-50 across ranks): _anchor_begin leaves this rank's candidate for the first anchor
 *   ((0, first_global or -1, that row's coordinates): 2 + dim doubles) in *cand; per round the host all-gathers the
 *   candidates (rank order) into *gathered (room for world of them; a single rank may pass *cand itself) and calls _anchor_step(gathered, world, round): the winner (largest value, then smallest global
 *   row) becomes anchor `round`, the local rows are swept with its coordinates (running minimum reset for rounds 0 and 1,
 *   the reference's D[1:] quirk) and this rank's next candidate replaces *cand.  Nothing waits for the host;
 *   _anchor_end downloads the anchors' global rows and coordinates.
 * Rows: _rows_begin gives the all-gather's send buffer (the shard padded with zero rows to the largest shard) and
 *   receive buffer ([world][most][dim]); _rows_end compacts what arrived, rebinds the context to ALL rows (global_base 0:
 *   rows are numbered by position in the rank-ordered concatenation).  Anchor distances: _anchor_dists_begin (between the
 *   anchor rounds and _rows_end) gives the send / receive buffers of the all-gather of the ranks' own anchor distances
 *   (float [n_anchors][most] per rank -- the "all-gather of anchor feature vectors"); _rows_end assembles D [n_anchors][total]
 *   from them (a host that skips the exchange gets them recomputed from the anchors, every row on every rank).
 *   annchor_stream_order_begin / _end then build the one global tile order, each rank ordering its own tile range.
 * Lists: annchor_stream_lists_all = the all-gather target for annchor_stream_knn_join.
 * Result: _route_begin replaces annchor_stream_knn_end: finished rows become records [global id, k-1 neighbour ids,
 *   k-1 float64 distances] (int64 words) grouped by owner rank (starts / bases: first position / first global row of
 *   every rank's shard); the host exchanges them with one all-to-all (send_counts -> receive counts) into the buffer of
 *   _route_recv; _route_end scatters them into this rank's own row order, keeps the graph rows on the device
 *   (annchor_stream_graph_device: int64 / float64 [rows_padded][k], padding rows (-1, inf) -- the source of a final
 *   graph all-gather) and downloads them (get_ann-shaped, annchor.py:514-530). */
int annchor_stream_hip_stream(annchor_ctx *ctx, void **stream);
int annchor_stream_anchor_begin(annchor_ctx *ctx, int32_t n_anchors, int64_t first_global, int32_t world, void **cand, void **gathered,
                                int64_t *cand_bytes);
int annchor_stream_anchor_step(annchor_ctx *ctx, const void *gathered, int32_t world, int32_t round);

/* ------------------------------------------------------------ in-library collectives (csrc/comm.hip)
 * RCCL called from inside the library, on the context's stream, so that the row-sharded build's rounds never return to
 * the host language: rank 0 makes a 128-byte id (annchor_comm_unique_id), the host hands it to every rank by whatever it
 * has (torch.distributed / MPI broadcast, a file), every rank calls annchor_comm_init on its context.  RCCL is loaded
 * with dlopen at first use (ANNCHOR_RCCL_LIB overrides the name); nothing here is touched by single-GPU use.
 *   annchor_comm_allgather          every rank's nbytes at send -> rank order at recv (device pointers)
 *   annchor_comm_alltoall_records   records of `words` 8-byte words grouped by destination -> source-rank order
 *   annchor_stream_anchor_rounds    ALL max-min rounds after annchor_stream_anchor_begin: per round the all-gather of the
 *                                   candidates, the winner's pick, the sweep (no communicator: one rank, no collective) */
int annchor_comm_unique_id(uint8_t *id128);
int annchor_comm_init(annchor_ctx *ctx, const uint8_t *id128, int32_t world, int32_t rank);
int annchor_comm_destroy(annchor_ctx *ctx);
/* Robustness of a multi-rank job (no reference counterpart: the reference has no collectives):
 *   annchor_comm_set_timeout   host waits of a context with a communicator run under a watchdog: one that lasts longer than
 *                              `seconds` (default 300; ANNCHOR_COMM_TIMEOUT_S; <= 0: off) aborts the communicators
 *                              (ncclCommAbort) and the call fails with "collective timed out" -- a dead peer no longer blocks
 *                              the other ranks for good
 *   annchor_comm_preflight     1 KB all-gather of (rank, position) bytes on every communicator of the context, checked, under
 *                              a timeout of its own: a mis-wired job fails here, loudly
 *   annchor_comm_init_side     a second communicator (its own 128-byte id) on a stream of its own
 *   annchor_comm_allgather_begin   ONE large all-gather beside the engine stream's work (the raw rows of the row-sharded
 *                              build beside the anchor rounds and the k-d order); the library waits for it where it first needs
 *                              the rows (annchor_stream_rows_end / _order_end); annchor_comm_side_join for other users */
int annchor_comm_set_timeout(annchor_ctx *ctx, double seconds);
int annchor_comm_preflight(annchor_ctx *ctx, double seconds);
int annchor_comm_init_side(annchor_ctx *ctx, const uint8_t *id128);
int annchor_comm_allgather_begin(annchor_ctx *ctx, const void *send, void *recv, int64_t nbytes);
int annchor_comm_side_join(annchor_ctx *ctx);
int annchor_comm_allgather(annchor_ctx *ctx, const void *send, void *recv, int64_t nbytes);
int annchor_comm_alltoall_records(annchor_ctx *ctx, const void *send, const int64_t *send_counts, void *recv,
                                  const int64_t *recv_counts, int32_t words);
int annchor_stream_anchor_rounds(annchor_ctx *ctx, int32_t n_anchors);
int annchor_stream_anchor_end(annchor_ctx *ctx, int64_t *A, float *anchor_vectors);
int annchor_stream_rows_begin(annchor_ctx *ctx, int32_t world, const int64_t *counts, void **send, void **recv, int64_t *bytes_per_rank);
int annchor_stream_anchor_dists_begin(annchor_ctx *ctx, int32_t world, const int64_t *counts, void **send, void **recv,
                                      int64_t *bytes_per_rank);
int annchor_stream_rows_end(annchor_ctx *ctx, int32_t world, const int64_t *counts);
int annchor_stream_lists_all(annchor_ctx *ctx, int32_t world, int64_t bytes_per_rank, void **all);
int annchor_stream_route_begin(annchor_ctx *ctx, int32_t world, const int64_t *starts, const int64_t *bases, void **send,
                               int64_t *send_counts, int64_t *record_words, int64_t *tile_evals);
int annchor_stream_route_recv(annchor_ctx *ctx, int64_t n_recv, void **recv);
int annchor_stream_route_end(annchor_ctx *ctx, int64_t n_recv, int64_t rows_padded, int64_t *ng_idx, double *ng_dist);
int annchor_stream_graph_device(annchor_ctx *ctx, void **idx, void **dist, int64_t *rows_padded, int32_t *k);
/* Raw device copies for hosts that stage the exchanges through host memory (process groups without device
 * collectives, e.g. gloo). */
int annchor_device_alloc(annchor_ctx *ctx, int64_t bytes, void **dptr);
int annchor_device_free(annchor_ctx *ctx, void *dptr);
int annchor_device_copy(annchor_ctx *ctx, void *dst, const void *src, int64_t bytes, int32_t kind /*1 H2D, 2 D2H, 3 D2D*/);

/* ------------------------------------------------------------- nearest enemies (f4)
 * Annchor.get_nearest_enemies (annchor/annchor.py:685-782; get_check with the label filter utils.py:454-491,
 * adjust_check utils.py:437-451, get_IJs_from_check utils.py:502-540) on a fitted pair list, in stages:
 *   annchor_enemies_candidates  y = dense label codes (HOST int32 [nx]): the enemy pairs sharing nearest anchors
 *     (per-point threshold among enemies, symmetrised when one was lowered) that fit() does not hold, as a sorted pair
 *     list + per-point index, with their bounds / dad / anchor flag; *n_new = their number;
 *   annchor_enemies_predict     their predicted distances, clipped to [lb, ub] (annchor.py:724-728): the fitted
 *     stratified regression (HOST bins / W / c), or -- pred != NULL -- a custom regression's host predictions;
 *   annchor_enemies_first       per point the `first` closest-looking enemies among its fitted + new entries; the
 *     not-computed ones are evaluated exactly (device metric: here; otherwise the pairs are returned in todo_ij and
 *     annchor_enemies_set_exact takes the values); RefineApprox / not_computed_mask are updated in place (:744-761);
 *   annchor_enemies_graph       the nn nearest computed enemies per point (:763-781): idx int64 [nx, nn] (the other
 *     endpoint), dist float64 [nx, nn];
 *   annchor_enemies_download    the new pairs for the host's views (the reference appends them to IJs, features,
 *     RefineApprox, not_computed_mask and I, :729-740).
 * Ties by list order (fitted entries by other endpoint, then new entries by other endpoint). */
int annchor_enemies_candidates(annchor_ctx *ctx, const int32_t *y, int32_t loc_thresh, int32_t loc_min, int64_t *n_new);
int annchor_enemies_predict(annchor_ctx *ctx, const double *bins, int32_t nbins, const double *W, const double *c, const double *pred);
int annchor_enemies_first(annchor_ctx *ctx, int32_t first, int32_t nn, int32_t evaluate, int64_t *todo_ij, int64_t cap, int64_t *n_todo);
int annchor_enemies_set_exact(annchor_ctx *ctx, const double *exact, int64_t n_todo);
int annchor_enemies_graph(annchor_ctx *ctx, int32_t nn, int64_t *idx, double *dist);
int annchor_enemies_download(annchor_ctx *ctx, int64_t *ij, double *feats, double *RA, uint8_t *ncm, int64_t *I_ptr, int64_t *I_idx);

/* Annchor.to_sparse_matrix (annchor/annchor.py:625-641): the symmetric sparse distance matrix of a
 * k-NN graph (HOST arrays ng_idx int64 [nx, k], ng_dist float64 [nx, k]) in COO form: every cell once,
 * value = distance + nextafter(0, 1), later assignments of the reference's loop order win.  rows /
 * cols / vals are HOST arrays with room for 2 * nx * k entries; *nnz entries are written. */
int annchor_graph_to_coo(annchor_ctx *ctx, const int64_t *ng_idx, const double *ng_dist, int64_t nx, int32_t k,
                         int64_t *rows, int64_t *cols, double *vals, int64_t *nnz);

/* DeviceStratifiedSampler (annchor_amd/samplers.py; protocol of annchor/samplers.py:75-110): the
 * stratified draw with an order-free random choice.  Partition b (bins[b] <= dad < bins[b+1], not
 * computed; counts[b] members, from annchor_bin_counts) keeps the min(want[b], counts[b]) members with
 * the smallest key = splitmix64(seed_key ^ position), ties to the smaller position.  positions (HOST,
 * room for sum of want): the samples, partition by partition, ascending position inside a partition. */
int annchor_hash_sample(annchor_ctx *ctx, const double *bins, int32_t nbins, const int64_t *counts, const int64_t *want,
                        uint64_t seed_key, int64_t *positions, int64_t *n_out);
/* The same choice followed by the samples' feature rows (annchor.py:325-338) and exact distances
 * (get_exact on the sampled pairs, annchor.py:340) in one call: feats float64 [m, 4], sample_y
 * float64 [m]; the sampled pairs become computed.  Device metric only. */
int annchor_hash_sample_pairs(annchor_ctx *ctx, const double *bins, int32_t nbins, const int64_t *counts, const int64_t *want,
                              uint64_t seed_key, int64_t *positions, double *feats, double *sample_y, int64_t *n_out);

/* -------------------------------------------------------------- state access */
int annchor_field_size(annchor_ctx *ctx, int32_t field, int64_t *n_elems);
int annchor_download(annchor_ctx *ctx, int32_t field, void *dst, int64_t n_elems);
int annchor_upload(annchor_ctx *ctx, int32_t field, const void *src, int64_t n_elems);

/* ------------------------------------------------------------------ profiling
 * Per-kernel-family accumulated device time since the last reset (HIP events on
 * the stream): names[i] is a static string; returns the number of entries.
 * on: 0 = off, 1 = every kernel family, 2 = the metric kernels only ("*_pairs"; cheap enough
 * to leave on inside a timed region). */
int annchor_prof_enable(annchor_ctx *ctx, int32_t on);
int annchor_prof_reset(annchor_ctx *ctx);
int annchor_prof_get(annchor_ctx *ctx, int32_t max_entries, const char **names, double *ms,
                     int64_t *launches, double *alg_bytes);

#ifdef __cplusplus
}
#endif
#endif /* ANNCHOR_HIP_H */
