"""
annchor_amd.annchor -- host orchestration of the ANNchor k-NN graph build on MI355X.

Mirrors the reference's user API (annchor/annchor.py): `Annchor(X, func, ...)`,
`.fit()`, `.neighbor_graph`, the per-stage methods `fit()` calls, `BruteForce`, and
`compare_neighbor_graphs`; same constructor arguments, same budget arithmetic, same
plugin protocols (AnchorPicker / Sampler / Regression / ErrorPredictor /
get_exact_ijs).  What differs is where the work happens: every stage is a call into
libannchor_hip.so (include/annchor_hip.h) operating on state that stays in HBM.
NumPy views of that state (`D`, `IJs`, `I`, `features`, `RefineApprox`, ...) are
materialised only when a user plugin or a test asks for them.

There is no CPU implementation of the pipeline in this package: without the HIP
library and a GPU, constructing an `Annchor` raises.
"""
import time
from collections import Counter

import os
import sys

import numpy as np

from . import _native
from .distances import DeviceMetric
from .error_predictors import SimpleStratifiedErrorRegression
from .pickers import MaxMinAnchorPicker
from .regressors import SimpleStratifiedLinearRegression
from .samplers import DeviceStratifiedSampler, NothingToSample, SimpleStratifiedSampler
from .utils import get_exact_ijs_, get_function_from_input, test_parallelisation

from .distances import euclidean as distances_euclidean  # noqa: E402
from .distances import cosine as distances_cosine  # noqa: E402

# Where the NEXT sampling step's draw runs once its statistics are known (inside fit()): "defer" = in get_sample, on the calling
# thread, after the refinement and update_bounds launches are enqueued (they execute meanwhile); "worker" = at once, on the
# library's persistent worker thread.  Measured in one process (tools/draw_ab.py, C2): 3.94-4.16 ms per fit against 4.38-4.68 -- the
# second thread's wake-up and its traffic cost the calling thread more than the extra overlap buys.
_DRAW_OVERLAP = {"worker": True, "defer": "defer"}[os.environ.get("ANNCHOR_DRAW", "defer")]
PAIRLIST_MAX_POINTS = 30000   # float32 Euclidean / cosine data above this size takes the streamed (tile-granular) form
PAIRLIST_HARD_MAX = None      # override (tests); None: from the device -- 2^30 candidate pairs (int32 positions) and 80 % of its free
                              # memory at ~130 B per pair: 46 341 points on a 288 GB MI355X (_native.pairlist_point_limit)


PAIRLIST_BITMAP_MAX_POINTS = 400000   # keep bitmap + rank table: 12 B per 64 pairs = 30 GB here
# sampler=None: from this many candidate pairs (nx (nx - 1) / 2: 2829 points) the order-free DeviceStratifiedSampler draws on the GPU;
# below it the NumPy-stream SimpleStratifiedSampler (graphs bit-identical to the CPU restatement's in tests/).  The reference's own draw
# (samplers.py:75-140, utils.py:543-578) runs numba's RNG inside njit -- a stream nobody reproduces -- so beyond the sizes whose
# graphs the tests pin to the NumPy stream there is nothing to be bit-equal to, and walking a stream as long as the pair list
# on one host thread dominates the fit (N = 16 000: 178 of 193 ms).  sampler="legacy" / "device" force either.
DEVICE_SAMPLER_MIN_PAIRS = 4_000_000
STREAM_MAX_DIM = 1024   # streamed form: rows of up to 1024 dimensions (beyond 128 the k-blocked tile kernel, csrc/knnbk.hip)
DEVICE_MODEL_MAX_PER_BIN = 6144   # samples per partition up to which the iteration's models are fitted on the device (model.hip: ERR_CAP 8192)


def pairlist_hard_max(device=0):
    if PAIRLIST_HARD_MAX is not None:
        return PAIRLIST_HARD_MAX
    lim = _native.pairlist_point_limit(device)
    return 30000 if lim is None else lim

FEATURE_NAMES = ["lower bound", "upper bound", "double anchor distance", "is anchor"]


def budget(nx, n_anchors, n_neighbors, n_samples, p_work, loc_min=None, quiet=True):
    """Constructor arithmetic of the reference (annchor.py:117-148,167-168)."""
    N = (nx * (nx - 1)) // 2
    na = int(np.sum([nx - j for j in range(1, n_anchors + 1)]))
    say = (lambda *a: None) if quiet else print
    if p_work > 1:
        say("Warning: p_work should not exceed 1.  Setting it to 1.")
        p_work = 1.0
    min_p_work = (2 * (na + n_samples) + 1) / N
    min_p_work = 1 if min_p_work > 1 else min_p_work
    if p_work < min_p_work:
        say("Warning: Too many anchors/samples for specified p_work.")
        say("Increasing p_work to %5.3f." % min_p_work)
        p_work = min_p_work
    if p_work > 0.75:
        say("Warning: High Value of p_work.")
        say("Think about decreasing n_anchors or n_samples," + " or using BruteForce.")
    lm = 10 * n_neighbors if loc_min is None else loc_min
    lm = np.clip(lm, 0, nx - 1)
    return dict(N=N, na=na, p_work=p_work, loc_min=lm)


class _DeviceModelRefused(Exception):
    """A device-fitted iteration met a partition its solver does not take; fit() starts over on the host solver."""


class _IndexCSR:
    """Read-only stand-in for the reference's typed dict `I` (utils.py:533-540):
    I[i] -> int64 array of positions in IJs that contain i."""

    def __init__(self, ptr, idx):
        self.ptr, self.idx = ptr, idx

    def __getitem__(self, i):
        return self.idx[self.ptr[i]:self.ptr[i + 1]]

    def __len__(self):
        return len(self.ptr) - 1

    def keys(self):
        return range(len(self))


class _DeviceExact:
    """get_exact(f, X, IJ) (utils.py:110-177).  The data set bound at construction is resident
    on the device: pairs of THAT data set under THAT metric are one kernel call.  The protocol
    lets a plugin pass any f and any X, though -- those are honoured: another data set under a
    bundled metric is evaluated on the device through a scratch context, any other callable on
    the host."""

    def __init__(self, engine, f, X):
        self.engine, self._f, self._X = engine, f, X

    def __call__(self, f, X, IJ):
        IJ = np.asarray(IJ, dtype=np.int64).reshape(-1, 2)
        if f is self._f and X is self._X:
            return self.engine.metric_pairs(IJ)
        if isinstance(f, DeviceMetric):
            return np.asarray(f.many([X[i] for i in IJ[:, 0]], [X[j] for j in IJ[:, 1]]), dtype=np.float64)
        return np.array([f(X[i], X[j]) for i, j in IJ], dtype=np.float64)


class Annchor:
    """Quickly computes the approximate k-NN graph for slow metrics (annchor.py:21-115).

    Parameters are those of the reference, plus `device` (GPU ordinal, default 0), `streamed`
    (None: float32 Euclidean / cosine data above PAIRLIST_MAX_POINTS points takes the streamed
    tile-granular form, announced on stdout; True / False force either form) and `ols`: inside fit(),
    with the default plugins and a device metric, the per-partition least-squares fits of the regression
    (regressors.py:39-69) run on the GPU ('device': Householder QR in float64, coefficients equal to the
    reference's LAPACK dgelsd solution to ~1e-13, no host round trip inside an iteration) or on the host
    with scipy's own dgelsd ('lapack': the reference's coefficients bit for bit on the same LAPACK build).
    """

    def __init__(self, X, func, func_kwargs=None, n_anchors=20, n_neighbors=15, n_samples=5000, p_work=0.1,
                 anchor_picker=None, sampler=None, regression=None, error_predictor=None, random_seed=42,
                 locality=5, loc_thresh=1, loc_min=None, verbose=False, is_metric=True, get_exact_ijs=None,
                 backend="loky", niters=2, lookahead=5, device=0, streamed=None, ols="device"):
        self.X = X
        self.nx = len(X)
        if ols not in ("device", "lapack"):
            raise ValueError("ols must be 'device' or 'lapack'")
        self.ols = ols
        self.f = get_function_from_input(func, func_kwargs)
        self.evals = 0
        self.n_anchors = n_anchors
        self.n_neighbors = n_neighbors
        self.n_samples = n_samples
        b = budget(self.nx, n_anchors, n_neighbors, n_samples, p_work, loc_min, quiet=False)
        self.N, self.na, self.p_work = b["N"], b["na"], b["p_work"]

        self.anchor_picker = MaxMinAnchorPicker() if anchor_picker is None else anchor_picker
        if isinstance(sampler, str) and sampler not in ("auto", "legacy", "device"):
            raise ValueError("sampler must be a Sampler object, None / 'auto', 'legacy' (SimpleStratifiedSampler) or 'device' "
                             "(DeviceStratifiedSampler)")
        sampler_auto = sampler is None or (isinstance(sampler, str) and sampler == "auto")
        if sampler_auto or sampler == "legacy":
            sampler_obj = SimpleStratifiedSampler()   # (auto: replaced below for large pair lists)
        elif isinstance(sampler, str):
            sampler_obj = DeviceStratifiedSampler()
        else:
            sampler_obj = sampler
        self.sampler = sampler_obj
        if isinstance(sampler, str):
            sampler = None   # the string forms are the built-in samplers: "default plugins" below
        self.regression = SimpleStratifiedLinearRegression() if regression is None else regression
        self.error_predictor = SimpleStratifiedErrorRegression() if error_predictor is None else error_predictor

        self.random_seed = random_seed
        self.verbose = verbose
        self.locality = locality
        self.loc_thresh = loc_thresh
        self.loc_min = b["loc_min"]
        self.is_metric = is_metric
        self.niters = niters
        self.lookahead = lookahead
        self.feature_names = list(FEATURE_NAMES)
        assert backend in ["loky", "multiprocessing"]
        self.backend = backend

        # ---- large float32 Euclidean / cosine data: the streamed (tile-granular) form.  The pair list
        # of the reference would hold ~nx^2/2 entries (hard part 3).  `streamed`:
        # None = by size, True / False = forced.  The streamed form computes in float32 and has no
        # sampler / regression / error model, so it is only taken for float32 input with the
        # default plugins -- never by narrowing float64 data behind the caller's back.
        self._streamed = None
        self._cosine_streamed = False
        self._anchors_on_device = False
        self._first_merge = True
        self._sample_ticket, self._pipelined, self._fit_it = None, False, None
        Xa = X if isinstance(X, np.ndarray) else None
        bundled = self.f is distances_euclidean or self.f is distances_cosine
        defaults = (anchor_picker is None and sampler is None and regression is None and error_predictor is None
                    and get_exact_ijs is None)
        can_stream = (bundled and defaults and Xa is not None and Xa.ndim == 2 and Xa.dtype == np.float32 and Xa.shape[1] <= STREAM_MAX_DIM)
        if streamed == "cast":
            # explicit opt-in: float64 (or integer) rows are narrowed to float32 and take the streamed form
            if not (bundled and defaults and Xa is not None and Xa.ndim == 2 and Xa.shape[1] <= STREAM_MAX_DIM):
                raise ValueError("streamed='cast' needs a numeric [n, dim <= 1024] array, the 'euclidean' or 'cosine' metric and "
                                 "the default plugins")
            Xa = np.ascontiguousarray(Xa, dtype=np.float32)
            can_stream, streamed = True, True
        if streamed is True and not can_stream:
            raise ValueError("streamed=True needs a float32 [n, dim <= 1024] array, the 'euclidean' or 'cosine' metric and the "
                             "default plugins (float64 data is not narrowed behind the caller's back: streamed='cast' narrows it)")
        want_stream = can_stream and (streamed is True or (streamed is None and self.nx > PAIRLIST_MAX_POINTS))
        hard_max = PAIRLIST_MAX_POINTS if (want_stream or self.nx <= PAIRLIST_MAX_POINTS) else pairlist_hard_max(device)
        self._pairlist_hard_max = hard_max
        if not want_stream and self.nx > hard_max:
            # Beyond the size whose COMPLETE pair list fits, the pair-list form still runs when the locality filter
            # (locality / loc_thresh / loc_min, annchor.py:74-80) thins the candidates to what the device holds: the count is
            # known after the anchors, and `annchor_build_locality` refuses there (2^30 pairs, device memory).  The keep bitmap
            # itself (12 B per 64 pairs) bounds the point count.
            if self.nx > PAIRLIST_BITMAP_MAX_POINTS or loc_thresh <= 1:
                raise ValueError("%d points: the complete candidate pair list of the reference form (~nx^2/2 entries at ~130 B) is "
                                 "materialised up to %d points on this device (2^30 pairs / 80 %% of its free memory).  Up to %d points "
                                 "the pair-list form runs when the locality filter keeps fewer candidates than that (loc_thresh >= 2 "
                                 "of `locality` nearest anchors in common; refused after the anchors if it does not).  Larger sets "
                                 "need the streamed form: float32 [n, dim <= 1024] data ('cast' narrows float64), 'euclidean' or "
                                 "'cosine', default plugins%s."
                                 % (self.nx, hard_max, PAIRLIST_BITMAP_MAX_POINTS, "" if streamed is not False else " (and streamed != False)"))
            print("Note: %d points is beyond the %d whose complete pair list fits this device; the fit goes on only if the locality "
                  "filter (locality=%d, loc_thresh=%d) keeps fewer than 2^30 candidate pairs." % (self.nx, hard_max, locality, loc_thresh),
                  file=sys.stderr)
            self._pairlist_hard_max = hard_max = self.nx
        if sampler_auto and not want_stream and self.N >= DEVICE_SAMPLER_MIN_PAIRS:
            self.sampler = DeviceStratifiedSampler()
            print("Note: %d candidate pairs: the samples are drawn by the order-free DeviceStratifiedSampler on the GPU (the same "
                  "stratified draw as samplers.py:75-140 with a hashed choice instead of the host's sequential NumPy-stream "
                  "shuffle of the whole pair list; sampler='legacy' keeps that one).  The choice is made here, on nx (nx - 1) / 2 "
                  ">= %d (nx >= 2829), before the locality filter fixes the real candidate count." % (self.N, DEVICE_SAMPLER_MIN_PAIRS),
                  file=sys.stderr)
        if want_stream:
            from .streamed import StreamedAnnchor

            if streamed is None:
                print("Note: %d float32 points under '%s': using the streamed (tile-granular) form -- anchors, a p_work tile budget "
                      "and %d neighbour-join passes (niters); n_samples, locality, loc_thresh, loc_min, lookahead and is_metric "
                      "do not apply to it (streamed=False forces the pair-list form)."
                      % (self.nx, self.f.name, niters))
            Xs = np.ascontiguousarray(Xa)
            if self.f is distances_cosine:
                # on the unit sphere |u - v|^2 = 2 - 2 cos(u, v): cosine distance = (Euclidean distance)^2 / 2
                # of the normalised rows, the same neighbours in the same order
                norms = np.linalg.norm(Xs.astype(np.float64), axis=1)
                if not np.all(norms > 0):
                    raise ValueError("cosine distance is undefined for zero rows")
                Xs = (Xs / norms[:, None]).astype(np.float32)
                self._cosine_streamed = True
            self._streamed = StreamedAnnchor(Xs, n_anchors=n_anchors, n_neighbors=n_neighbors, p_work=self.p_work,
                                             random_seed=random_seed, device=device, join_passes=max(int(niters), 0))
            self._engine = self._streamed._engine
            self._device_metric = True
            self.get_exact_ijs = _DeviceExact(self._engine, self.f, self.X)
            self.get_exact_query_ijs = None
            self._cache, self.timings = {}, {}
            return

        # ---- the engine: fails loudly when the HIP library or a GPU is missing
        self._engine = _native.Engine(device)
        self._device_metric = isinstance(self.f, DeviceMetric) and get_exact_ijs is None
        if self._device_metric:
            self.f.bind(self._engine, X)
            self.get_exact_ijs = _DeviceExact(self._engine, self.f, self.X)
        else:
            self._engine.set_opaque(self.nx)
            if get_exact_ijs is None:
                self.get_exact_ijs = get_exact_ijs_(self.f, verbose=self.verbose, backend=backend)
            else:
                self.get_exact_ijs = get_exact_ijs
        if hasattr(self.get_exact_ijs, "state"):   # the built-in host evaluator sizes its pool by the evaluations ahead of it
            self.get_exact_ijs.state["expected_pairs"] = int(self.p_work * self.N)
        test_parallelisation(self.get_exact_ijs, self.f, self.X, self.nx, backend, s=20)
        self.get_exact_query_ijs = None
        self._anchors_on_device = False
        self._cache = {}
        self._first_merge = True
        self._sample_ticket, self._pipelined, self._fit_it = None, False, None
        self.timings = {}

    # ------------------------------------------------------------ metric boundary
    def _device_get_exact_ijs(self, f, X, IJ):
        """get_exact(f, X, IJ) (utils.py:110-177) against the uploaded data set."""
        return self._engine.metric_pairs(np.asarray(IJ, dtype=np.int64))
    # (the `get_exact_ijs` attribute is a _DeviceExact holding the engine only: a bound method of
    # self stored on self would be a reference cycle, and the engine -- a stream, pinned memory and
    # the device arena -- would then live until the cyclic collector happens to run)

    # ------------------------------------------------------------- lazy NumPy views
    def _view(self, key, loader):
        if key not in self._cache:
            if getattr(self, "_enemy_extended", False):
                if key in ("labels", "thresh", "cand", "next"):
                    raise RuntimeError("get_nearest_enemies() extended IJs / RefineApprox / not_computed_mask / features / I by the "
                                       "enemy pairs; '%s' of the fitted pair list is no longer aligned with them (read it before, "
                                       "or refit)" % key)
                if key in ("IJs", "RA", "ncm", "features", "I"):   # fitted list + enemy pairs, from the device state
                    from . import enemies

                    self._cache.update(enemies.extended_views(self))
                    return self._cache[key]
            self._cache[key] = loader()
        return self._cache[key]

    def _invalidate(self, *keys):
        for k in keys:
            self._cache.pop(k, None)

    @property
    def D(self):
        return self._view("D", lambda: self._engine.download(_native.F_D).reshape(self.nx, self.n_anchors))

    @property
    def A(self):
        return self._view("A", lambda: self._engine.download(_native.F_A))

    @property
    def IJs(self):
        return self._view("IJs", lambda: self._engine.download(_native.F_IJS).reshape(-1, 2))

    @property
    def I(self):
        return self._view("I", lambda: _IndexCSR(self._engine.download(_native.F_I_PTR),
                                                 self._engine.download(_native.F_I_IDX)))

    @property
    def sid(self):
        def load():
            m = self._engine.download(_native.F_SID).reshape(self.nx, -1)   # 1 / 2 / 4 mask words per point (<= 64 / 128 / 256 anchors)
            return np.array([[a for a in range(self.n_anchors) if (int(w[a >> 6]) >> (a & 63)) & 1] for w in m])
        return self._view("sid", load)

    @property
    def features(self):
        return self._view("features", lambda: self._engine.download(_native.F_FEATURES).reshape(-1, 4))

    @property
    def not_computed_mask(self):
        return self._view("ncm", lambda: self._engine.download(_native.F_NCM).astype(bool))

    @property
    def RefineApprox(self):
        return self._view("RA", lambda: self._engine.download(_native.F_RA))

    @property
    def errors(self):
        return self._view("labels", lambda: self._engine.download(_native.F_LABELS))

    @property
    def thresh(self):
        return self._view("thresh", lambda: self._engine.download(_native.F_THRESH))

    # ---- the last sampling step's arrays: inside fit() they stay in device memory with the default plugins (the models
    # are fitted there, csrc/model.hip) and are downloaded when somebody reads them
    def _materialise_samples(self):
        if self.__dict__.get("_samples_on_device"):
            self.__dict__["_samples_on_device"] = False
            m = self.__dict__["_samples_m"]
            pos, feats, y, sp = self._engine.download_samples(m, predict=self.__dict__.get("_predict_on_device", False))
            d = self.__dict__
            d["_sample_ixs"], d["_sample_features"], d["_sample_y"] = pos, feats, y
            if sp is not None:
                d["_sample_predict"] = sp

    def _sample_attr(name):   # noqa: N805 -- class-body helper
        def get(self):
            self._materialise_samples()
            return self.__dict__.get("_" + name)

        def put(self, value):
            self.__dict__["_" + name] = value

        return property(get, put)

    sample_ixs = _sample_attr("sample_ixs")
    sample_features = _sample_attr("sample_features")
    sample_y = _sample_attr("sample_y")
    sample_predict = _sample_attr("sample_predict")
    del _sample_attr

    def _models_on_device(self):
        """fit() keeps an iteration on the device from the sampling step to the candidate selection when everything
        in between is the reference's default: device metric, the built-in stratified sampler / regression / error
        model on the double anchor distance, and ols='device'."""
        return (self._pipelined and self._device_metric and self.ols == "device" and self._sampler_on_device()
                and not self.__dict__.get("_device_models_off", False)
                and type(self.sampler) in (SimpleStratifiedSampler, DeviceStratifiedSampler)
                and type(self.regression) is SimpleStratifiedLinearRegression
                and list(self.regression.reg_feature_names) == ["lower bound", "upper bound", "double anchor distance"]
                and self.regression.partition_feature_name == "double anchor distance"
                and type(self.error_predictor) is SimpleStratifiedErrorRegression
                and self.error_predictor.partition_feature_name == "double anchor distance")

    def _device_model_takes(self, ticket):
        """Limits of csrc/model.hip, checked on the sampling step's own numbers (bin populations and quotas are on the
        host by now): at least 3 samples in every partition (fewer rows than columns: dgelsd's minimum-norm answer is the
        host's) and at most DEVICE_MODEL_MAX_PER_BIN -- the residual sorter holds ERR_CAP = 8192 per partition, and a sample
        on a bin edge belongs to two error partitions (error_predictors.py:50-52: closed on both sides)."""
        if ticket.get("error") is not None or "counts" not in ticket:
            return True   # (finish_device raises / nothing to judge)
        P = self.sampler.n_partitions
        n = int(ticket["n_samples"])
        want = ticket.get("want")
        if want is None:
            want = n // P + (np.arange(P) < n % P)
        per_bin = np.minimum(np.asarray(ticket["counts"], dtype=np.int64), np.asarray(want, dtype=np.int64))
        return bool(per_bin.min() >= 3 and per_bin.max() <= DEVICE_MODEL_MAX_PER_BIN)

    # --------------------------------------------------------------------- stages
    def _pair_list_stage(self, what):
        if getattr(self, "_enemy_extended", False):
            raise RuntimeError("%s: get_nearest_enemies() extended this object's pair list by the enemy pairs (and refined "
                               "entries of the fitted one) -- build a new Annchor to run the stages again" % what)
        if self._streamed is not None:
            raise NotImplementedError("%s is a stage of the pair-list form; this object runs the streamed form "
                                      "(fit() does everything; pass streamed=False for the staged pipeline)" % what)

    def get_anchors(self):
        """annchor.py:191-206."""
        self._pair_list_stage("get_anchors")
        self._anchors_on_device = False
        A, D, evals = self.anchor_picker.get_anchors(self)
        if not self._anchors_on_device:
            D = np.ascontiguousarray(D, dtype=np.float64)
            self._engine.set_anchor_distances(D, np.asarray(A, dtype=np.int64))
        self._invalidate("D", "A")
        self.evals += evals

    def get_locality(self):
        """annchor.py:208-256."""
        self._pair_list_stage("get_locality")
        n, min_len = self._engine.build_locality(self.locality, self.loc_thresh, int(self.loc_min))
        self.n_pairs = n
        self._invalidate("IJs", "I", "sid", "features", "ncm", "RA", "labels", "thresh")
        if min_len < self.n_neighbors:
            raise Exception("Error: Not enough candidates in pool for all indices.\n"
                            + "Try again with higher locality.")

    def get_features(self):
        """annchor.py:258-311."""
        self._pair_list_stage("get_features")
        self._engine.compute_features()
        self._first_merge = True
        self._invalidate("features", "ncm", "RA", "labels", "thresh")

    def _sampler_on_device(self):
        """The built-in stratified sampler partitioning on the double anchor distance runs against
        the resident state; any other sampler (or partition feature) gets the NumPy arrays of the
        protocol."""
        return (type(self.sampler) in (SimpleStratifiedSampler, DeviceStratifiedSampler)
                and self.sampler.partition_feature_name == "double anchor distance")

    def get_sample(self):
        """annchor.py:313-343."""
        self._pair_list_stage("get_sample")
        eng = self._engine
        if self._sampler_on_device():
            ticket, self._sample_ticket = self._sample_ticket, None
            if ticket is None:
                ticket = self.sampler.begin_device(eng, self.n_samples, self.random_seed, overlap=False,
                                                   **({"device_trace": True} if self._models_on_device() else {}))
            if self._models_on_device() and not self._device_model_takes(ticket):
                # decided BEFORE the iteration: the models of this fit are fitted on the host (the device kernels would only raise
                # their flags at the end and the whole fit would start over)
                self._device_models_off = True
            if self._models_on_device():   # ... and everything stays there: the models are fitted on the device
                _, self.n_samples, self.sample_bins = self.sampler.finish_device(ticket, evaluate="device")
                self._samples_on_device, self._samples_m, self._predict_on_device = True, self.n_samples, False
                self._invalidate("ncm")
                self.evals += self.n_samples
                return
            self._samples_on_device = False
            if self._device_metric:   # positions, feature rows and distances in one device pass
                (self.sample_ixs, self.n_samples, self.sample_bins, self.sample_features,
                 self.sample_y) = self.sampler.finish_device(ticket, evaluate=True)
                self._invalidate("ncm")
                self.evals += self.sample_y.shape[0]
                return
            self.sample_ixs, self.n_samples, self.sample_bins = self.sampler.finish_device(ticket)
        else:
            self.sample_ixs, self.n_samples, self.sample_bins = self.sampler.sample(
                self.features, self.feature_names, self.n_samples, self.not_computed_mask, self.random_seed)
        self.sample_ixs = np.asarray(self.sample_ixs, dtype=np.int64)
        self.sample_features = eng.gather_features(self.sample_ixs)
        if self._device_metric:
            self.sample_y = eng.evaluate_samples(self.sample_ixs)
        else:
            self.sample_ijs = self.IJs[self.sample_ixs]
            self.sample_y = np.asarray(self.get_exact_ijs(self.f, self.X, self.sample_ijs), dtype=np.float64)
            eng.set_samples(self.sample_ixs, self.sample_y)
        self._invalidate("ncm")
        self.evals += self.sample_y.shape[0]

    def fit_predict_regression(self):
        """annchor.py:345-380."""
        self._pair_list_stage("fit_predict_regression")
        self._model_on_device = False
        if self.__dict__.get("_samples_on_device") and self._models_on_device():
            # per-partition OLS + predict / clip / merge / label on the device, no host wait (csrc/model.hip)
            self._engine.fit_regression_device(self.sample_bins, self._first_merge, self.is_metric)
            self._model_on_device, self._predict_on_device, self._fused_labels = True, True, True
            self._first_merge = False
            self._invalidate("RA", "labels")
            return
        self.regression.fit(self.sample_features, self.feature_names, self.sample_y, sample_bins=self.sample_bins)
        model = self.regression.coefficients() if type(self.regression) is SimpleStratifiedLinearRegression else None
        self._fused_labels = False
        if model is not None:
            bins, W, c = model
            self.sample_predict = self._engine.predict_merge(bins, W, c, self._first_merge, self.is_metric,
                                                             len(self.sample_ixs))
            # the fused kernel labels pairs with the same bin edges; valid for the
            # built-in error predictor when it is handed these edges (it is, below)
            self._fused_labels = (type(self.error_predictor) is SimpleStratifiedErrorRegression
                                  and self.error_predictor.partition_feature_name == "double anchor distance"
                                  and np.array_equal(bins, self.sample_bins))
        else:
            pred = np.asarray(self.regression.predict(self.features, self.feature_names), dtype=np.float64)
            self.sample_predict = pred[self.sample_ixs]
            self._engine.merge_host_prediction(pred, self._first_merge, self.is_metric)
        self._first_merge = False
        self._invalidate("RA", "labels")

    def fit_predict_errors(self):
        """annchor.py:382-393."""
        self._pair_list_stage("fit_predict_errors")
        if self._pipelined and self._fit_it is not None:
            # inside fit(): the row thresholds and guarantee_nmin of the coming selection depend on RefineApprox and
            # the mask only -- launch them now, they run while the error model is fitted on the host
            nn = self.n_neighbors
            self._engine.select_prepare(nn, 3 * nn // 2 if self._fit_it == 0 else 0)
        if self.__dict__.get("_model_on_device"):
            self._engine.fit_errors_device()   # sorted residuals per partition, on the device
            self._invalidate("labels")
            if self._fit_it is not None and self.__dict__.get("_make_stream"):
                self._make_stream(self._fit_it + 1)   # the GPU is busy with this iteration's models: produce the next draw's stream
            return
        self.error_predictor.fit(self.sample_features, self.feature_names, self.sample_y - self.sample_predict,
                                 sample_bins=self.sample_bins)
        if not self._fused_labels:
            labels = self._label_slots(self.error_predictor.predict(self.features, self.feature_names))
            self._engine.set_labels(labels)
        self._invalidate("labels")

    def _label_slots(self, labels):
        """error_predictor.predict's labels as POSITIONS in error_predictor.labels -- arbitrary integer keys
        (error_predictors.py:47 takes any) -- which is how the residual lists are handed to the device."""
        labels = np.asarray(labels)
        keys = list(self.error_predictor.labels)
        if keys == list(range(len(keys))):
            return labels
        order = np.argsort(np.asarray(keys), kind="stable")
        sk = np.asarray(keys)[order]
        at = np.minimum(np.searchsorted(sk, labels), len(sk) - 1)
        if not np.array_equal(sk[at], labels):
            raise ValueError("error_predictor.predict returned a label that is not in error_predictor.labels")
        return order[at]

    def select_refine_candidate_pairs(self, w=0.5, it=0):
        """annchor.py:395-473."""
        self._pair_list_stage("select_refine_candidate_pairs")
        nn = self.n_neighbors
        n_refine = int((self.p_work * self.N - self.na - self.n_samples) * w) + 1  # annchor.py:440
        n_refine = 0 if n_refine < 0 else n_refine
        nmin = 3 * nn // 2 if it == 0 else 0
        if self.__dict__.get("_model_on_device"):
            nb = len(self.sample_bins) - 1
            ncand, nnext = self._engine.select_candidates(nn, nmin, None, n_refine, self.lookahead, n_labels=nb)
            # (the coefficients stay on the device; what a kernel could not do -- a partition the QR does not take, a
            # failed sample step -- raises sticky flags that fit() reads once, after the last iteration: _adopt_device_model)
            self._device_model_nb = nb
        else:
            labels = list(self.error_predictor.labels)   # (any integer keys: the device labels are positions in this list)
            errs = [np.asarray(self.error_predictor.errs[lab], dtype=np.float64) for lab in labels]
            if labels != list(range(len(labels))) and self._fused_labels:
                raise NotImplementedError("the fused label pass numbers the partitions 0..L-1; got labels %r" % (labels[:8],))
            ncand, nnext = self._engine.select_candidates(nn, nmin, errs, n_refine, self.lookahead)
        self.n_refine = n_refine
        self._invalidate("RA", "thresh", "cand", "next")
        if self._pipelined and self._device_metric and ncand and it < self.niters - 1 and self._sampler_on_device():
            # inside fit(): the next sampling step's statistics depend on the mask and dad only,
            # so take them now and let the host draw overlap the refinement kernel
            self._engine.mark_candidates()
            # (the refinement launch rides behind the statistics' download: the host waits for the statistics alone)
            self._engine.park_refine(1)
            try:
                self._sample_ticket = self.sampler.begin_device(self._engine, self.n_samples, self.random_seed, overlap=_DRAW_OVERLAP,
                                                                **({"device_trace": True} if self._models_on_device() else {}))
            finally:
                self._engine.park_refine(2)
        elif self._device_metric:
            self._engine.refine_candidates()
        elif ncand:
            mapback = self._engine.download(_native.F_CAND)
            exact = np.asarray(self.get_exact_ijs(self.f, self.X, self.IJs[mapback]), dtype=np.float64)
            self._engine.set_refined(exact)
        self.evals += ncand
        self._invalidate("RA", "ncm")

    def _adopt_device_model(self, nb):
        """After a device-fitted iteration: the coefficients into the regression object (what its fit() would have
        left there) and the sticky flags; False when the host has to redo the models."""
        # (coefficients, flags and the sorted residual lists behind ONE wait: a later predict_merge / select_candidates with
        # host lists, or closing the engine, would overwrite the device copy of the lists)
        W, c, status, ep, flags, flat = self._engine.model_download_with_errors(nb, 2 * int(self.n_samples) + 16)
        if flags[0] == 3:
            raise _native.NativeError("sample step: the draw's trace kernel gave up waiting for the host's partner stream "
                                      "(ANNCHOR_DRAW_STREAM_TIMEOUT_MS; ANNCHOR_DRAW_STREAM=0 uploads the partners instead)")
        if flags[0] == 1:
            raise _native.NativeError("sample step: a (bin, rank) entry does not exist (stale counts?)")
        if flags[0] or flags[1] or flags[2] or status.any() or flat is None:   # (flags[0] == 2: a hashed key list came out short -- retry the waiting way)
            self._device_model_refused = (status.copy(), flags.copy())   # (kept for diagnostics / tests)
            return False
        reg = self.regression
        reg.n_partitions, reg.sample_bins, reg.coef_, reg.intercept_ = nb, self.sample_bins, W, c
        self._device_err_ptr = ep
        ep_ = self.error_predictor
        ep_.partition_bins, ep_.n_partitions, ep_.labels = self.sample_bins, nb, range(nb)
        ep_.errs = {b: flat[ep[b]:ep[b + 1]].copy() for b in range(nb)}
        return True

    @property
    def nextback(self):
        return self._view("next", lambda: self._engine.download(_native.F_NEXT))

    @property
    def mapback(self):
        return self._view("cand", lambda: self._engine.download(_native.F_CAND))

    def update_anchor_points(self, timeout=None, chunk_size=None):
        """annchor.py:475-512 (no wall-clock cut: all lookahead pairs are processed)."""
        self._pair_list_stage("update_anchor_points")
        self._engine.update_bounds()
        self._invalidate("features")

    def get_ann(self):
        """annchor.py:514-530."""
        self._pair_list_stage("get_ann")
        self.neighbor_graph = self._engine.neighbor_graph(self.n_neighbors)

    def fit(self):
        """Computes the approximate nearest-neighbour graph (annchor.py:532-623)."""
        if self._streamed is not None:
            st = self._streamed.fit()
            self.neighbor_graph = st.neighbor_graph
            if self._cosine_streamed:
                self.neighbor_graph = (st.neighbor_graph[0], st.neighbor_graph[1] ** 2 / 2.0)
            self._cache["A"] = st.A
            self.evals, self.timings = st.evals, st.timings
            return self
        origin = time.perf_counter()
        t = self.timings = {}
        legacy_rng = self._sampler_on_device() and type(self.sampler) is SimpleStratifiedSampler
        ndraws = self.N + self.N // 2 + 4096
        seed0 = self.random_seed + getattr(self.sampler, "loop_num", 0)

        def make_stream(it):
            """The sampler's MT19937 stream of iteration `it` depends on the seed only: produced on THIS thread (warm
            core, ~0.4 ms) at a point where the host would otherwise just wait for the GPU."""
            if legacy_rng and it < self.niters:
                s = time.perf_counter()
                _native.legacy_generate(seed0 + it, ndraws)
                t["rng_stream"] = t.get("rng_stream", 0.0) + time.perf_counter() - s   # (overlaps GPU work; not a stage)

        self._make_stream = make_stream

        def stage(name, fn, *a, **k):
            s = time.perf_counter()
            fn(*a, **k)
            t[name] = t.get(name, 0.0) + time.perf_counter() - s
            if self.verbose:
                print("%40s: %6.3f | %6.3f" % (name, time.perf_counter() - s, time.perf_counter() - origin))

        evals0, loop0, n_samples0 = self.evals, getattr(self.sampler, "loop_num", None), self.n_samples
        self._pipelined = True
        self._device_models_off = False
        try:
            try:
                return self._fit_stages(stage, make_stream, t, origin)
            except _DeviceModelRefused:
                # a partition the device's solver does not take (fewer rows than columns; a rank-deficient one gets its
                # minimum-norm solution on the device): the whole fit again with scipy's dgelsd
                self.evals, self.n_samples = evals0, n_samples0   # (a sampling step may have lowered n_samples)
                if loop0 is not None:
                    self.sampler.loop_num = loop0
                self._first_merge, self._sample_ticket, self._model_on_device = True, None, False
                t.clear()
                ols, self.ols = self.ols, "lapack"
                try:
                    return self._fit_stages(stage, make_stream, t, origin)
                finally:
                    self.ols = ols
        finally:   # (an exception inside a stage -- NothingToSample, a plugin error -- must not leave the object in pipelined mode)
            self._pipelined, self._fit_it, self._sample_ticket = False, None, None
            self._make_stream = None

    def _fit_stages(self, stage, make_stream, t, origin):
        stage("get_anchors", self.get_anchors)
        legacy_rng = self._sampler_on_device() and type(self.sampler) is SimpleStratifiedSampler
        ndraws = self.N + self.N // 2 + 4096
        seed0 = self.random_seed + getattr(self.sampler, "loop_num", 0)
        fused = self._models_on_device()
        if self._anchors_on_device:
            # the anchor rounds are running: nothing to wait for yet.  The first draw's stream is produced at the engine's next
            # host wait -- inside get_locality, AFTER its kernels are queued behind the rounds (producing it here left the GPU idle
            # from the end of the rounds to the end of the generation once the rounds had become one persistent launch)
            if legacy_rng:
                self._engine.legacy_generate_at_next_wait(seed0, ndraws)   # (in one piece: two pieces over the next two waits measured slower)
        elif legacy_rng:
            _native.legacy_prefetch(seed0, ndraws)
        stage("get_locality", self.get_locality)
        stage("get_features", self.get_features)
        niters = self.niters
        for it in range(niters):
            if legacy_rng and not fused and it + 1 < niters:
                _native.legacy_prefetch(seed0 + it + 1, ndraws)   # host-fitted models keep this thread busy: a producer thread
            try:
                stage("get_sample", self.get_sample)
            except NothingToSample as err:
                if it == 0:
                    raise ValueError("Sampler raised NothingToSample on first iteration.") from err
                print("Warning: main loop terminated early with nothing " + "left to sample.")
                break
            stage("fit_predict_regression", self.fit_predict_regression)
            self._fit_it = it
            stage("fit_predict_errors", self.fit_predict_errors)
            self._fit_it = None
            stage("select_refine_candidate_pairs", self.select_refine_candidate_pairs, w=1 / niters, it=it)
            if it < niters - 1:
                stage("update_anchor_points", self.update_anchor_points)
        stage("get_ann", self.get_ann)
        if self.__dict__.get("_model_on_device") and not self._adopt_device_model(self._device_model_nb):
            raise _DeviceModelRefused()
        t["total"] = time.perf_counter() - origin
        return self

    def query(self, Q, nn=15, p_work=0.3, get_exact_query_ijs=None):
        """Query new data against the fitted index (annchor.py:643-683 ->
        query_functions.py:183-212): anchor distances of the queries, shared-nearest-anchor
        candidates, bounds / dad features, the FITTED regression + error model, one
        select-and-refine pass with the work budget p_work * len(Q) * nx, then the nn nearest
        per query.  Returns (indices [nq, nn], distances [nq, nn]) into X.

        Runs on a second engine holding X followed by Q; the fitted state is untouched.
        `get_exact_query_ijs(f, X, Z, IJ)` (pairs index (X[i], Z[j])) replaces the metric
        evaluator as in the reference."""
        if self._streamed is not None:   # large float32 Euclidean data: tile-granular query, same kernel as fit()
            Qs = np.asarray(Q, dtype=np.float32)
            if self._cosine_streamed:
                Qs = (Qs / np.linalg.norm(Qs.astype(np.float64), axis=1)[:, None]).astype(np.float32)
            idx, dist = self._streamed.query(Qs, nn=nn, p_work=p_work)
            return (idx, dist ** 2 / 2.0) if self._cosine_streamed else (idx, dist)
        if self.p_work > 1:
            print("Warning: p_work should not exceed 1.  Setting it to 1.")
            self.p_work = 1.0
        nq, nx = len(Q), self.nx
        na = self.n_anchors * nq
        nbf = nq * nx
        limit = ((nq * nn * 3) // 2 - 1 + na) / nbf   # annchor.py:668-675
        if p_work < limit:
            print("Warning: p_work too low")
            print("Increasing p_work to %5.3f" % limit)
            p_work = limit
        if get_exact_query_ijs is not None:
            self.get_exact_query_ijs = get_exact_query_ijs
        device_q = isinstance(self.f, DeviceMetric) and self._device_metric and self.get_exact_query_ijs is None
        eng = _native.Engine(self._engine.device)
        A = np.asarray(self.A, dtype=np.int64)
        assert len(A) == self.n_anchors, "query() needs anchors that are data-set members (annchor.py uses X[A])"
        if device_q:
            both = list(self.X) + list(Q) if not isinstance(self.X, np.ndarray) else np.concatenate([self.X, np.asarray(Q)])
            self.f.bind(eng, both)
            IJa = np.stack([np.repeat(A, nq), np.tile(nx + np.arange(nq), len(A))], axis=1)
            QD = eng.metric_pairs(IJa).reshape(len(A), nq).T            # get_query_anchor_dists (:10-15)
        else:
            eng.set_opaque(nx + nq)
            if self.get_exact_query_ijs is None:
                def default_q(f, X, Z, IJ):
                    from joblib import Parallel, delayed
                    from .utils import CPU_COUNT
                    return np.array(Parallel(n_jobs=CPU_COUNT, backend=self.backend, timeout=30)(
                        delayed(f)(X[i], Z[j]) for i, j in IJ), dtype=np.float64)
                self.get_exact_query_ijs = default_q
            XA = [self.X[a] for a in A]
            IJa = np.array([[i, j] for j in range(nq) for i in range(len(A))])
            QD = np.asarray(self.get_exact_query_ijs(self.f, XA, Q, IJa), dtype=np.float64).reshape(nq, len(A))
        eng.set_anchor_distances(np.vstack([self.D, QD]), A)
        n_pairs, min_len = eng.build_query_locality(nx, self.locality, self.loc_thresh)   # get_query_locality (:18-37)
        if min_len <= nn:
            raise Exception("Error: Not enough candidates in pool for all queries.\nTry again with higher locality.")
        eng.compute_features()                                                            # get_query_features (:40-67)
        model = self.regression.coefficients() if type(self.regression) is SimpleStratifiedLinearRegression else None
        fused = (model is not None and type(self.error_predictor) is SimpleStratifiedErrorRegression
                 and np.array_equal(model[0], self.error_predictor.partition_bins))
        if fused:
            eng.predict_merge(model[0], model[1], model[2], True, True, 0)                # predict + clip (:199-202)
        else:
            feats = eng.download(_native.F_FEATURES).reshape(-1, 4)
            eng.merge_host_prediction(np.asarray(self.regression.predict(feats, self.feature_names), dtype=np.float64), True, True)
            eng.set_labels(self._label_slots(self.error_predictor.predict(feats, self.feature_names[:-1])))
        labels = list(self.error_predictor.labels)
        errs = [np.asarray(self.error_predictor.errs[lab], dtype=np.float64) for lab in labels]
        n_refine = int((p_work * nbf - na)) + 1                                           # query_functions.py:163-168
        ncand, _ = eng.select_candidates(nn, 3 * nn // 2, errs, max(n_refine, 0), 1)
        if device_q:
            eng.refine_candidates()
        elif ncand:
            IJc = eng.download(_native.F_IJS).reshape(-1, 2)[eng.download(_native.F_CAND)]
            IJc[:, 1] -= nx
            eng.set_refined(np.asarray(self.get_exact_query_ijs(self.f, self.X, Q, IJc), dtype=np.float64))
        self.query_evals = na + ncand
        idx, dist = eng.neighbor_graph(nn + 1)                                            # get_nn(nq, nn + 1, ...) (:210)
        eng.close()
        return idx[nx:, 1:], dist[nx:, 1:]

    # ---------------------------------------------- nearest enemies / selective subset
    def _require_pair_list(self, what):
        if self._streamed is not None:
            raise NotImplementedError(what + " needs the pair-list form (streamed=False)")

    def get_nearest_enemies(self, y, nn=3, loc_min=100):
        """annchor.py:685-782: the nn nearest points of a different label for every point;
        result in self.nearest_enemy_graph = (indices [nx, nn], distances [nx, nn])."""
        from . import enemies

        self._require_pair_list("get_nearest_enemies")
        enemies.nearest_enemies(self, y, nn=nn, loc_min=loc_min)

    def annchor_selective_subset(self, y, dne=None, alpha=0):
        """annchor.py:784-901: indices of a selective subset of X for labels y."""
        from . import enemies

        self._require_pair_list("annchor_selective_subset")
        return enemies.selective_subset(self, y, dne=dne, alpha=alpha)

    def alpha_rss(self, y, dne=None, alpha=0):
        """annchor.py:903-927."""
        from . import enemies

        self._require_pair_list("alpha_rss")
        return enemies.alpha_rss(self, y, dne=dne, alpha=alpha)

    def to_sparse_matrix(self):
        """annchor.py:625-641: DOK sparse distance matrix of the k-NN graph (symmetric; every
        stored entry carries eps = nextafter(0, 1) so that explicit zeros survive; where a pair is
        listed from both sides the later row's value wins, as in the reference's loop order).
        The symmetric COO form is emitted on the device (annchor_graph_to_coo), each cell once."""
        from scipy.sparse import coo_matrix

        idx, dist = self.neighbor_graph
        nx = idx.shape[0]
        rows, cols, vals = self._engine.graph_to_coo(idx, dist)
        return coo_matrix((vals, (rows, cols)), shape=(nx, nx), dtype=np.float64).todok()

    @staticmethod
    def _sparse_from_graph_host(idx, dist):
        """The same matrix from flat NumPy arrays (test cross-check of the device emission)."""
        from scipy.sparse import coo_matrix

        nx, k = idx.shape
        eps = np.nextafter(0, 1, dtype=np.float64)
        i = np.repeat(np.arange(nx, dtype=np.int64), k)
        j = idx.ravel().astype(np.int64)
        v = dist.ravel() + eps
        rows = np.stack([i, j], axis=1).ravel()
        cols = np.stack([j, i], axis=1).ravel()
        vals = np.repeat(v, 2)
        key = rows * nx + cols
        order = np.argsort(key, kind="stable")
        ks = key[order]
        last = np.ones(ks.shape[0], dtype=bool)
        last[:-1] = ks[1:] != ks[:-1]
        sel = order[last]
        return coo_matrix((vals[sel], (rows[sel], cols[sel])), shape=(nx, nx), dtype=np.float64).todok()


class BruteForce:
    """All-pairs k-NN graph (annchor.py:943-1023)."""

    def __init__(self, X, func, func_kwargs=None, verbose=False, get_exact_ijs=None, backend="loky", device=0):
        self.X = X
        self.nx = len(X)
        self.f = get_function_from_input(func, func_kwargs)
        self.verbose = verbose
        assert backend in ["loky", "multiprocessing"]
        self._device_metric = isinstance(self.f, DeviceMetric) and get_exact_ijs is None
        if self._device_metric:
            self._engine = _native.Engine(device)
            self.f.bind(self._engine, X)
            self.get_exact_ijs = _DeviceExact(self._engine, self.f, self.X)
        elif get_exact_ijs is None:
            self.get_exact_ijs = get_exact_ijs_(self.f, verbose=self.verbose, backend=backend)
        else:
            self.get_exact_ijs = get_exact_ijs
        if hasattr(self.get_exact_ijs, "state"):
            self.get_exact_ijs.state["expected_pairs"] = self.nx * (self.nx - 1) // 2
        test_parallelisation(self.get_exact_ijs, self.f, self.X, self.nx, backend, s=20)

    def fit(self, n_neighbors=None):
        """Full rows like the reference (annchor.py:1020-1023) by default; `n_neighbors` keeps
        only the first columns (needed above ~8000 points)."""
        if self._device_metric:
            self.neighbor_graph = self._engine.brute_force(self.nx if n_neighbors is None else min(n_neighbors, self.nx))
            return self
        iu = np.triu_indices(self.nx, k=1)
        IJs = np.stack(iu, axis=1)
        dists = self.get_exact_ijs(self.f, self.X, IJs)
        self.D = np.zeros(shape=(self.nx, self.nx))
        self.D[iu] = dists
        self.D = self.D + self.D.T
        order = np.argsort(self.D, axis=1, kind="stable")
        self.neighbor_graph = (order, np.take_along_axis(self.D, order, axis=1))
        return self


def compare_neighbor_graphs(nng_1, nng_2, n_neighbors):
    """Number of incorrect NN pairs of nng_2 w.r.t. nng_1, counting equal distances
    as interchangeable (annchor.py:1026-1066)."""
    nx = nng_1[0].shape[0]
    h = []
    for ix in range(nx):
        a = Counter(np.round(nng_1[1][ix][:n_neighbors], 3).astype(np.float32))
        b = Counter(np.round(nng_2[1][ix][:n_neighbors], 3).astype(np.float32))
        h.append(len(a - b))
    return int(np.sum(h))
