#!/bin/bash
# Device idle gaps of a one-rank streamed fit at N rows (default 8 000 000): rocprofv3 kernel + memory-copy trace of two fits,
# then tools/trace_gaps.py lists every interval > 0.5 ms in which neither a kernel nor a copy ran during the LAST fit.
N=${1:-8000000}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c5_gaps; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$R timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -o c5 -- python -c "
import sys, time; sys.path.insert(0, '$R/tests')
from test_c5_gpu import shard_rows
from annchor_amd.streamed import StreamedAnnchor
X = shard_rows(0, $N)
for _ in range(2):
    sa = StreamedAnnchor(X, n_anchors=32, n_neighbors=15, p_work=0.1)
    t0 = time.time_ns(); sa.fit(); t1 = time.time_ns(); print('FIT', t0, t1, sa.timings); sa._engine.close()
" > $O/run.log 2>&1
tail -2 $O/run.log
python $R/tools/trace_gaps.py $O > $O/gaps.txt
cat $O/gaps.txt
find $O -name "*_trace.csv" -delete
