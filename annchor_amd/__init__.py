"""annchor_amd -- MI355X-native ANNchor k-NN graph engine (drop-in for `annchor`)."""
from .annchor import Annchor, BruteForce, compare_neighbor_graphs  # noqa: F401

__version__ = "0.1.0"
