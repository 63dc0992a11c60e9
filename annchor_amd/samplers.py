"""
Samplers (reference annchor/samplers.py:18-175).  Protocol, unchanged:
    sampler.sample(features[n,4], feature_names, n_samples, not_computed_mask, random_seed)
        -> (sample_ixs, n_samples, sample_bins);  raises NothingToSample

`SimpleStratifiedSampler` additionally implements `sample_device(ann, ...)`: the same
draw, bit for bit (same order statistics, same bins, same NumPy legacy RNG stream --
`np.random.seed(seed + loop_num)` then one `np.random.choice(..., replace=False)` per
bin), but with the O(n) work -- quantiles of the feature over not-computed pairs,
bin populations, rank -> pair position -- done by device kernels on the resident
state instead of on a materialised features array.
"""
from abc import ABC, abstractmethod

import os

import numpy as np


_WORKER_ALWAYS = os.environ.get("ANNCHOR_DRAW_WORKER_ALWAYS", "0") == "1"
_TRACE_ON_HOST = os.environ.get("ANNCHOR_DRAW_TRACE", "device") == "host"   # the draw's backward trace on helper threads, as before round 4


class NothingToSample(Exception):
    pass


class SamplingError(Exception):
    def __init__(self, message):
        super().__init__(message)


class Sampler(ABC):
    """Base class (samplers.py:22-110): descendants implement get_partition and may
    override sample_partition."""

    def __init__(self, partition_feature_name, n_partitions):
        self.partition_feature_name = partition_feature_name
        self.n_partitions = n_partitions
        self.loop_num = 0

    @abstractmethod
    def get_partition(self, sample_feature, new_samples):
        pass

    def sample_partition(self, indices, n_samples, sample_feature, sample_bins, random_seed):
        """Draw without replacement from every partition (utils.py:543-578): partition b is
        lo_b <= x < hi_b and gets n_samples // P (+1 for the first n_samples % P partitions), or all
        of its members when it has fewer.  The reference seeds numba's RNG inside njit; without
        numba the NumPy legacy stream is the reproducible equivalent: one
        `permutation(len(members))[:want]` per partition, in partition order, after
        `seed(random_seed + loop_num)` -- the draws `np.random.choice(members, want, replace=False)`
        makes."""
        P = self.n_partitions
        edges = np.asarray(sample_bins, dtype=np.float64)
        part = np.searchsorted(edges, sample_feature, side="right") - 1      # last edge <= x
        grouped = np.argsort(part, kind="stable")                            # members of a partition keep their order
        start = np.searchsorted(part[grouped], np.arange(P + 1), side="left")
        quota = n_samples // P + (np.arange(P) < n_samples % P)
        np.random.seed(random_seed + self.loop_num)
        picked = []
        for b in range(P):
            members = indices[grouped[start[b]:start[b + 1]]]
            if len(members) >= quota[b]:
                members = members[np.random.permutation(len(members))[:quota[b]]]
            picked.append(members)
        self.loop_num += 1
        if min(len(m) for m in picked) < 2:
            raise Exception("Some sampler bins contain too few samples")
        return np.concatenate(picked)

    def sample(self, features, feature_names, n_samples, not_computed_mask, random_seed):
        """samplers.py:75-110: partition the not-computed pairs on one feature and draw from each
        partition; returns (positions in the pair list, how many, partition edges)."""
        open_pairs = np.flatnonzero(not_computed_mask)
        if open_pairs.size == 0:
            raise NothingToSample()
        values = np.asarray(features)[open_pairs, feature_names.index(self.partition_feature_name)]
        edges, granted = self.get_partition(values, n_samples)
        if granted != n_samples:
            print("Warning: n_samples has changed from %d to %d." % (n_samples, granted))
        if granted == 0:
            raise NothingToSample()
        drawn = self.sample_partition(open_pairs, granted, values, edges, random_seed)
        if len(drawn) != granted:
            print("Warning: Some bins contained fewer samples than requested")
        return drawn, len(drawn), edges


class SimpleStratifiedSampler(Sampler):
    """samplers.py:113-140: 1 % / 99 % (fallback 10 % / 90 %) order statistics of the
    feature, 6 linspace edges + +-inf => 7 bins."""

    def __init__(self, partition_feature_name="double anchor distance", n_partitions=7):
        super().__init__(partition_feature_name, n_partitions)

    @staticmethod
    def _quantile_ranks(n, n_samples, n_partitions):
        iq1, iq3 = int(n / 100), int(99 * n / 100)
        if (iq1 * n_partitions) < n_samples:
            iq1, iq3 = int(n / 10), int(9 * n / 10)
        if (iq1 * n_partitions) < n_samples:
            n_samples = iq1 * n_partitions
            print("Warning: n_samples too large for data set size.\n" + "Reducing n_samples to %d." % n_samples)
        return iq1, iq3, n_samples

    def _device_partition(self, engine, iq1, iq3, n_samples):
        """get_partition's bins and their populations from the device-resident state: one host wait when the library can chain
        quantiles -> edges -> counts on the device (its edges must be NumPy's, bit for bit), three otherwise."""
        stats = getattr(engine, "sampler_stats", None)
        if stats is not None and self.n_partitions >= 2:
            q1, q3, edges, counts = stats(iq1, iq3, self.n_partitions)
        else:
            (q1, q3), edges, counts = engine.kth_uncomputed_dad([iq1, iq3]), None, None
        sample_bins = np.hstack([-np.inf, np.linspace(q1, q3, self.n_partitions - 1), np.inf])
        if n_samples == 0:
            raise NothingToSample()
        if edges is None or not np.array_equal(edges, sample_bins):
            counts = engine.bin_counts(sample_bins)
        return sample_bins, counts

    def get_partition(self, sample_feature, n_samples):
        iq1, iq3, n_samples = self._quantile_ranks(sample_feature.shape[0], n_samples, self.n_partitions)
        q1 = np.partition(sample_feature, iq1)[iq1]
        q3 = np.partition(sample_feature, iq3)[iq3]
        sample_bins = np.linspace(q1, q3, self.n_partitions - 1)
        return np.hstack([-np.inf, sample_bins, np.inf]), n_samples

    def sample_device(self, engine, n_samples, random_seed):
        """Same result as sample(), computed against the device-resident state."""
        return self.finish_device(self.begin_device(engine, n_samples, random_seed, overlap=False))

    def begin_device(self, engine, n_samples, random_seed, overlap=True, device_trace=False):
        """First half of sample_device: the statistics the draw depends on (number of
        not-computed pairs, dad quantiles, bin counts -- functions of not_computed_mask and dad
        only), then the draw itself.  Annchor.fit() calls this as soon as the refinement candidates
        are known, so that the draw overlaps the refinement kernel (overlap=True: the draw runs on
        the library's persistent worker thread; overlap=False: nothing is enqueued in between, so
        it runs here, on the calling thread's warm core).
        Errors are kept in the ticket and raised by finish_device, where sample() would raise."""
        from . import _native

        ticket = {"engine": engine, "error": None, "draw": None, "per_bin": None}
        try:
            if self.partition_feature_name != "double anchor distance":
                raise NotImplementedError
            n_unc = engine.count_uncomputed()
            if n_unc == 0:
                raise NothingToSample()
            iq1, iq3, new_n = self._quantile_ranks(n_unc, n_samples, self.n_partitions)
            if new_n != n_samples:
                print("Warning: n_samples has changed from %d to %d." % (n_samples, new_n))
            n_samples = new_n
            sample_bins, counts = self._device_partition(engine, iq1, iq3, n_samples)
            bin_size, remainder = n_samples // self.n_partitions, n_samples % self.n_partitions
            want = np.array([bin_size + (nbin < remainder) for nbin in range(self.n_partitions)], dtype=np.int64)
            seed = random_seed + self.loop_num
            ticket.update(n_samples=n_samples, sample_bins=sample_bins, counts=counts)

            if not 0 <= seed < 2 ** 32:  # outside the legacy int-seed range NumPy raises; keep its behaviour
                np.random.seed(seed)
                ticket["per_bin"] = [np.arange(c) if c < w else np.random.permutation(int(c))[:w]
                                     for c, w in zip(counts, want)]
            elif device_trace and hasattr(engine, "sample_pairs_device_draw") and not _TRACE_ON_HOST:
                # fit() with the models on the device: finish_device makes the draw and the sampling step one library call -- the
                # host walks the stream, the backward trace runs on the GPU (annchor_sample_pairs_device_draw)
                ticket["deferred"] = (seed, counts, want)
                ticket["device_trace"] = True
            elif overlap == "defer" and not _WORKER_ALWAYS:
                # the caller still has device work to enqueue (refinement, update_bounds): the draw runs in finish_device, on the
                # calling thread's warm core, while that work executes
                ticket["deferred"] = (seed, counts, want)
            elif overlap or _WORKER_ALWAYS:
                ticket["draw"] = _native.legacy_choice_begin(seed, counts, want)
            else:
                ticket["per_bin"] = _native.legacy_choice_ranks(seed, counts, want)
        except BaseException as err:  # noqa: BLE001
            ticket["error"] = err
        return ticket

    def finish_device(self, ticket, evaluate=False):
        """evaluate=True (device metric): also returns the samples' feature rows and exact
        distances, from the fused annchor_sample_pairs call."""
        if ticket["draw"] is not None:
            from . import _native

            draw, ticket["draw"] = ticket["draw"], None
            try:
                ticket["per_bin"] = _native.legacy_choice_end(draw)
            except BaseException as err:  # noqa: BLE001
                if ticket["error"] is None:
                    ticket["error"] = err
        if (ticket["error"] is None and evaluate == "device" and ticket.get("device_trace") and ticket.get("deferred") is not None):
            seed, counts, want = ticket["deferred"]
            engine, n_samples, sample_bins = ticket["engine"], ticket["n_samples"], ticket["sample_bins"]
            if int(np.minimum(counts, want).min()) < 2:
                self.loop_num += 1
                raise Exception("Some sampler bins contain too few samples")
            taken, m = engine.sample_pairs_device_draw(sample_bins, counts, want, seed)
            if taken:
                ticket.pop("deferred")
                self.loop_num += 1
                if n_samples != m:
                    print("Warning: Some bins contained fewer samples than requested")
                return None, m, sample_bins
        if ticket["error"] is None and ticket.get("deferred") is not None:
            from . import _native

            seed, counts, want = ticket.pop("deferred")
            try:
                ticket["per_bin"] = _native.legacy_choice_ranks(seed, counts, want)
            except BaseException as err:  # noqa: BLE001
                ticket["error"] = err
        if ticket["error"] is not None:
            raise ticket["error"]
        engine, n_samples, sample_bins = ticket["engine"], ticket["n_samples"], ticket["sample_bins"]
        bin_of, ranks = [], []
        for nbin, r in enumerate(ticket["per_bin"]):
            if len(r) < 2:
                self.loop_num += 1
                raise Exception("Some sampler bins contain too few samples")
            bin_of.append(np.full(len(r), nbin, dtype=np.int32))
            ranks.append(np.asarray(r, dtype=np.int64))
        self.loop_num += 1
        bin_of, ranks = np.concatenate(bin_of), np.concatenate(ranks)
        extra = ()
        if evaluate == "device":   # positions, feature rows and distances stay in device memory (no host wait)
            m = engine.sample_pairs_device(sample_bins, ticket["counts"], bin_of, ranks)
            if n_samples != m:
                print("Warning: Some bins contained fewer samples than requested")
            return None, m, sample_bins
        if evaluate:
            sample_ixs, feats, y = engine.sample_pairs(sample_bins, ticket["counts"], bin_of, ranks)
            extra = (feats, y)
        else:
            sample_ixs = engine.select_by_rank(sample_bins, bin_of, ranks)
        if n_samples != sample_ixs.shape[0]:
            print("Warning: Some bins contained fewer samples than requested")
        return (sample_ixs, sample_ixs.shape[0], sample_bins) + extra


_M64 = (1 << 64) - 1


def splitmix64(x):
    """splitmix64 finaliser on uint64 arrays (the key function of DeviceStratifiedSampler)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


class DeviceStratifiedSampler(SimpleStratifiedSampler):
    """Stratified sampler with an ORDER-FREE random choice: same partitions, same quotas and the same
    protocol as SimpleStratifiedSampler (samplers.py:75-140), but a partition's members are chosen by
    key = splitmix64(splitmix64(random_seed + loop_num) ^ pair position) -- the `want` smallest keys -- instead of
    through NumPy's sequential shuffle of the partition (an MT19937 stream as long as the pair list,
    walked by one thread: 2 of the 5.8 ms of the strings fit, 0.3 of 0.33 s at 127 M pairs).  A uniform
    random subset either way; NOT the same subset, so graphs differ from the default sampler's by
    sampling noise.  With a device metric the choice runs on the GPU (annchor_hash_sample); through the
    plain `sample()` protocol the same choice is made in NumPy."""

    def seed_key(self, random_seed):
        """The hash key of one sampling step: splitmix64(random_seed + loop_num).  (Round 2 used random_seed + loop_num
        itself: consecutive steps' keys then differ in the lowest bit only, so step t + 1 drew exactly the list
        neighbours -- position ^ 1 -- of step t's members; hashing the seed decorrelates the steps.)"""
        return int(splitmix64(np.uint64((int(random_seed) + self.loop_num) & _M64)))

    def sample_partition(self, indices, n_samples, sample_feature, sample_bins, random_seed):
        P = self.n_partitions
        edges = np.asarray(sample_bins, dtype=np.float64)
        part = np.searchsorted(edges, sample_feature, side="right") - 1
        quota = n_samples // P + (np.arange(P) < n_samples % P)
        keys = splitmix64(np.uint64(self.seed_key(random_seed)) ^ np.asarray(indices, dtype=np.uint64))
        self.loop_num += 1
        picked = []
        for b in range(P):
            members = np.flatnonzero(part == b)
            if len(members) > quota[b]:
                members = members[np.lexsort((indices[members], keys[members]))[:quota[b]]]
            picked.append(np.sort(indices[members]))
        if min(len(m) for m in picked) < 2:
            raise Exception("Some sampler bins contain too few samples")
        return np.concatenate(picked)

    def begin_device(self, engine, n_samples, random_seed, overlap=True, device_trace=False):   # (device_trace: the legacy sampler's switch)
        ticket = {"engine": engine, "error": None}
        try:
            if self.partition_feature_name != "double anchor distance":
                raise NotImplementedError
            n_unc = engine.count_uncomputed()
            if n_unc == 0:
                raise NothingToSample()
            iq1, iq3, new_n = self._quantile_ranks(n_unc, n_samples, self.n_partitions)
            if new_n != n_samples:
                print("Warning: n_samples has changed from %d to %d." % (n_samples, new_n))
            n_samples = new_n
            sample_bins, counts = self._device_partition(engine, iq1, iq3, n_samples)
            want = n_samples // self.n_partitions + (np.arange(self.n_partitions) < n_samples % self.n_partitions)
            ticket.update(n_samples=n_samples, sample_bins=sample_bins, counts=counts, want=want.astype(np.int64),
                          key=self.seed_key(random_seed))
        except BaseException as err:  # noqa: BLE001
            ticket["error"] = err
        return ticket

    def finish_device(self, ticket, evaluate=False):
        if ticket["error"] is not None:
            raise ticket["error"]
        engine, sample_bins = ticket["engine"], ticket["sample_bins"]
        self.loop_num += 1
        if np.minimum(ticket["counts"], ticket["want"]).min() < 2:
            raise Exception("Some sampler bins contain too few samples")
        extra = ()
        if evaluate == "device":   # positions, feature rows and distances stay in device memory (no host wait)
            m = engine.hash_sample_pairs_device(sample_bins, ticket["counts"], ticket["want"], ticket["key"])
            if ticket["n_samples"] != m:
                print("Warning: Some bins contained fewer samples than requested")
            return None, m, sample_bins
        if evaluate:   # device metric: positions, feature rows and distances in one device pass
            sample_ixs, feats, y = engine.hash_sample_pairs(sample_bins, ticket["counts"], ticket["want"], ticket["key"])
            extra = (feats, y)
        else:
            sample_ixs = engine.hash_sample(sample_bins, ticket["counts"], ticket["want"], ticket["key"])
        if ticket["n_samples"] != sample_ixs.shape[0]:
            print("Warning: Some bins contained fewer samples than requested")
        return (sample_ixs, sample_ixs.shape[0], sample_bins) + extra


class ClusterSampler(Sampler):
    """samplers.py:143-170 (host only: k-means on the feature)."""

    def __init__(self, partition_feature_name="double anchor distance", n_partitions=5):
        super().__init__(partition_feature_name, n_partitions)

    def get_partition(self, sample_feature, n_samples):
        from sklearn.cluster import KMeans

        labels = KMeans(n_clusters=self.n_partitions).fit_predict(sample_feature.reshape(-1, 1))
        partitions = np.array([[np.min(sample_feature[labels == i]), np.max(sample_feature[labels == i])]
                               for i in range(self.n_partitions)])
        partitions = np.sort(partitions.flatten())
        return np.hstack([-np.inf, partitions[1:-1:2], np.inf]), n_samples
