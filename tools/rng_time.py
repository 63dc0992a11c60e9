import sys, time, numpy as np
sys.path.insert(0,'.')
from annchor_amd import _native
counts = np.array([12000, 150000, 250000, 300000, 280000, 200000, 60000], dtype=np.int64)
want = np.array([715,715,714,714,714,714,714], dtype=np.int64)
for rep in range(5):
    _native.legacy_prefetch(42+rep, int(counts.sum()*1.5)+4096)
    time.sleep(0.05)
    t=time.perf_counter(); r=_native.legacy_choice_ranks(42+rep, counts, want); dt=time.perf_counter()-t
    print("choice_ranks %.3f ms" % (dt*1e3))
t=time.perf_counter(); _native.legacy_prefetch(99, 1900000); r=_native.legacy_choice_ranks(99, counts, want); print("prefetch+ranks no wait %.3f ms"%((time.perf_counter()-t)*1e3))
np.random.seed(42); ref=[np.random.permutation(c)[:w] for c,w in zip(counts,want)]
_native.legacy_prefetch(42, 1900000); r=_native.legacy_choice_ranks(42, counts, want)
print(all(np.array_equal(a,b) for a,b in zip(r,ref)))
for rep in range(4):
    t=time.perf_counter(); r=_native.legacy_choice_ranks(1000+rep, counts, want); print("no prefetch (inline generation) %.3f ms"%((time.perf_counter()-t)*1e3))
