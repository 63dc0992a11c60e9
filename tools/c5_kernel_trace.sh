#!/bin/bash
# rocprofv3 kernel stats of ONE streamed fit at N rows (default 8 000 000) on one rank: which kernels the ordering, the
# reverse lists and the routing are made of.  Output: gpurun_out/c5_trace/ (copy the *_kernel_stats.csv into profiles/).
N=${1:-8000000}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c5_trace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$R timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o c5 -- python -c "
import sys; sys.path.insert(0, '$R/tests')
from test_c5_gpu import shard_rows
from annchor_amd.streamed import StreamedAnnchor
X = shard_rows(0, $N)
for _ in range(2):
    sa = StreamedAnnchor(X, n_anchors=32, n_neighbors=15, p_work=0.1).fit(); print(sa.timings); sa._engine.close()
" > $O/run.log 2>&1
tail -2 $O/run.log
find $O -name "*kernel_trace.csv" -delete
find $O -name "*kernel_stats.csv" | head
