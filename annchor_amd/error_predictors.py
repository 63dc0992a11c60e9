"""
Error predictors (reference annchor/error_predictors.py:18-67).  Protocol, unchanged:
    fit(sample_features, feature_names, sample_error, sample_bins=None)
    predict(features, feature_names) -> int labels[n]
    .errs: dict[label -> sorted float64[]],  .labels: iterable

Written around one sort and binary searches instead of a mask per partition; the
behaviour at shared partition edges is the reference's: a sample ON an edge counts for
both neighbouring partitions when the residual lists are fitted (closed intervals on both
sides), and a pair on an edge is labelled with the later partition.
"""
import numpy as np


class SimpleStratifiedErrorRegression:
    def __init__(self, partition_feature_name="double anchor distance", n_partitions=7):
        self.partition_feature_name = partition_feature_name
        self.n_partitions = n_partitions
        self.labels = range(n_partitions)
        self.partition_bins = None
        self.errs = {}

    def _column(self, features, feature_names):
        return np.asarray(features)[:, feature_names.index(self.partition_feature_name)]

    def _own_edges(self, values):
        """Edges when the caller hands none over: 1 % / 99 % order statistics, evenly spaced between."""
        ordered = np.sort(values)
        n = ordered.shape[0]
        inner = np.linspace(ordered[n // 100], ordered[(99 * n) // 100], self.n_partitions - 1)
        return np.concatenate(([-np.inf], inner, [np.inf]))

    def fit(self, sample_features, feature_names, sample_error, sample_bins=None):
        values = self._column(sample_features, feature_names)
        if sample_bins is None:
            self.partition_bins = self._own_edges(values)
        else:
            self.partition_bins = sample_bins
            self.n_partitions = len(sample_bins) - 1
        self.labels = range(self.n_partitions)
        edges = np.asarray(self.partition_bins, dtype=np.float64)
        residual = np.asarray(sample_error, dtype=np.float64)
        # first / last partition a sample belongs to (a sample ON an inner edge belongs to two)
        lo = np.searchsorted(edges[1:], values, side="left")            # first b with values <= upper edge of b
        hi = np.searchsorted(edges[:-1], values, side="right") - 1      # last b with lower edge of b <= values
        P = self.n_partitions
        lo, hi = np.clip(lo, 0, P - 1), np.clip(hi, 0, P - 1)
        # group by first partition (stable integer sort: radix), sort each group's residuals; the few
        # samples sitting ON an inner edge are added to the later partitions they also count for
        key = lo.astype(np.int16)
        order = np.argsort(key, kind="stable")
        grouped = residual[order]
        cut = np.searchsorted(key[order], np.arange(P + 1), side="left")
        extra = {}
        for t in np.flatnonzero(hi > lo):
            for b in range(int(lo[t]) + 1, int(hi[t]) + 1):
                extra.setdefault(b, []).append(residual[t])
        self.errs = {}
        for b in range(P):
            members = grouped[cut[b]:cut[b + 1]]
            if b in extra:
                members = np.concatenate((members, np.asarray(extra[b], dtype=np.float64)))
            self.errs[b] = np.sort(members)

    def predict(self, features, feature_names):
        values = self._column(features, feature_names)
        edges = np.asarray(self.partition_bins, dtype=np.float64)
        # the last partition whose lower edge is <= value (shared edges go to the later partition)
        lab = np.searchsorted(edges, values, side="right") - 1
        return np.clip(lab, 0, self.n_partitions - 1).astype(int)
