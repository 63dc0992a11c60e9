"""MT19937 stream generation rate: a draw over one huge bin right after the prefetch starts waits for
the producer thread, so its duration is the generation time of the words it consumes."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from annchor_amd import _native
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60_000_000
for seed in (101, 102, 103):
    t = time.perf_counter()
    _native.legacy_prefetch(seed, int(n * 1.5))
    _native.legacy_choice_ranks(seed, [n], [10])
    dt = time.perf_counter() - t
    print("seed %d: %.1f ms for ~%.0f M words (%.2f ns/word incl. the scan)" % (seed, dt * 1e3, n * 1.33 / 1e6, dt / (n * 1.33) * 1e9))
