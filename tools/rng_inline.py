import os, sys, time
os.environ["ANNCHOR_RNG_NO_CACHE"] = "1"
sys.path.insert(0, "/root/repo")
import numpy as np
from annchor_amd import _native
nd = 1279200 + 1279200 // 2 + 4096
for r in range(8):
    t = time.perf_counter(); _native.legacy_generate(1000 + r, nd); dt = time.perf_counter() - t
    print("generate %d words: %.3f ms" % (nd, dt * 1e3))
