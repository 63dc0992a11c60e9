#!/bin/bash
# Device idle gaps of a one-rank streamed fit at C3 size: rocprofv3 kernel + memory-copy trace of three fits, then
# tools/trace_gaps.py on the LAST fit (markers: the constructor's upload ends a fit's predecessor).
N=${1:-1000000}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c3_gaps; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -o c3 -- python -c "
import sys, time; sys.path.insert(0, '$R')
from bench import euclid_shard
from annchor_amd.streamed import StreamedAnnchor
X = euclid_shard(0, $N)
for _ in range(3):
    sa = StreamedAnnchor(X, n_anchors=32, n_neighbors=15, p_work=0.1)
    t0 = time.perf_counter(); sa.fit(); print('FIT', time.perf_counter() - t0, sa.timings); sa._engine.close()
" > $O/run.log 2>&1
grep FIT $O/run.log
python $R/tools/trace_gaps.py $O 0.2 > $O/gaps.txt
tail -40 $O/gaps.txt
find $O -name "*_trace.csv" -delete
