"""
Error predictors (reference annchor/error_predictors.py:18-67).  Protocol, unchanged:
    fit(sample_features, feature_names, sample_error, sample_bins=None)
    predict(features, feature_names) -> int labels[n]
    .errs: dict[label -> sorted float64[]],  .labels: iterable
"""
import numpy as np


class SimpleStratifiedErrorRegression:
    def __init__(self, partition_feature_name="double anchor distance", n_partitions=7):
        self.n_partitions = n_partitions
        self.partition_feature_name = partition_feature_name
        self.labels = range(n_partitions)

    def fit(self, sample_features, feature_names, sample_error, sample_bins=None):
        sample_feature = sample_features[:, feature_names.index(self.partition_feature_name)]
        if sample_bins is None:
            n = sample_feature.shape[0]
            iq1, iq3 = int(n / 100), int(99 * n / 100)
            q1, q3 = np.partition(sample_feature, iq1)[iq1], np.partition(sample_feature, iq3)[iq3]
            self.partition_bins = np.hstack([-np.inf, np.linspace(q1, q3, self.n_partitions - 1), np.inf])
        else:
            self.n_partitions = sample_bins.shape[0] - 1
            self.partition_bins = sample_bins
        self.labels = range(self.n_partitions)
        self.errs = {}
        for nbin in range(self.n_partitions):
            mask = (sample_feature >= self.partition_bins[nbin]) * (sample_feature <= self.partition_bins[nbin + 1])
            self.errs[nbin] = np.sort(sample_error[mask])

    def predict(self, features, feature_names):
        labels = np.empty(shape=features.shape[0]).astype(int)
        feature = features[:, feature_names.index(self.partition_feature_name)]
        for nbin in range(self.n_partitions):
            mask = (feature >= self.partition_bins[nbin]) * (feature <= self.partition_bins[nbin + 1])
            labels[mask] = nbin
        return labels
