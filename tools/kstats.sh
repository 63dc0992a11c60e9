#!/bin/bash
# rocprofv3 kernel stats of one command, filtered: kstats.sh "<regex>" <python script + args>
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kstats; rm -rf $O; mkdir -p $O
pat=$1; shift
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$R timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o r -- python "$@" > $O/out.log 2>&1
find $O -name "*kernel_trace.csv" -delete
python - "$O" "$pat" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/r_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if re.search(sys.argv[2], r["Name"]):
        print("%-48s calls %4s avg %10.1f us  min %9.1f max %9.1f" % (r["Name"].split("(")[0][:48], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
