set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gap; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o r -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-scale --no-euclid --no-kernel-events > $O/out.json 2> $O/err.log
f=$(find $O -name "*kernel_trace.csv" | head -1)
python $R/tools/gap_census.py $f
