#!/bin/bash
# rocprofv3 kernel trace of a few C2 fits, then the idle-gap census of the last one (tools/gap_census.py).
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gap; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python $R/tools/fit_times.py 9 > $O/fit_times.log 2> $O/err.log
f=$(find $O -name "*kernel_trace.csv" | head -1)
python $R/tools/gap_census.py $f; python $R/tools/fit_sequence.py $f > $O/sequence.txt
rm -f $f
