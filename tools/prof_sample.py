import time, numpy as np, sys
sys.path.insert(0,'.')
from annchor_amd import Annchor, _native
from annchor_amd.datasets import load_strings
X=load_strings()["X"]
def T(f,*a):
    t=time.perf_counter(); r=f(*a); return r,(time.perf_counter()-t)*1e3
for rep in range(3):
    ann=Annchor(X,"levenshtein",n_anchors=15,n_neighbors=25,p_work=0.12)
    ann.get_anchors(); ann.get_locality(); ann.get_features()
    e=ann._engine
    n_unc,t1=T(e.count_uncomputed)
    q,t2=T(e.kth_uncomputed_dad,[12000,1200000])
    bins=np.hstack([-np.inf,np.linspace(q[0],q[1],6),np.inf])
    counts,t3=T(e.bin_counts,bins)
    want=np.array([715,715,714,714,714,714,714])
    r,t4=T(_native.legacy_choice_ranks,42,counts,want)
    bin_of=np.repeat(np.arange(7),[len(x) for x in r]).astype(np.int32); ranks=np.concatenate(r)
    pos,t5=T(e.select_by_rank,bins,bin_of,ranks)
    f,t6=T(e.gather_features,pos)
    y,t7=T(e.evaluate_samples,pos)
    print("count %.2f kth %.2f bins %.2f rng %.2f select %.2f gather %.2f eval %.2f"%(t1,t2,t3,t4,t5,t6,t7))
    _,t8=T(ann.get_ann) if False else (0,0)
