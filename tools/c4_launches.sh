#!/bin/bash
# Per-launch durations of the exact-OT kernel in one C4 fit (rocprofv3 kernel trace of tools/c4_time.py).
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c4l; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o r -- python $R/tools/c4_time.py > $O/out.txt 2> $O/err.log
tail -1 $O/out.txt
python - $O <<'P'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40], r["Grid_Size_X"], r["Workgroup_Size_X"]) for r in rows)
em = [e for e in ev if "k_emd" in e[2]][-25:]
d = [(e[1] - e[0]) / 1e3 for e in em]
print("anchor rounds: %d launches, %.0f us total (min %.0f max %.0f), grid %s wg %s" % (len(d[:20]), sum(d[:20]), min(d[:20]), max(d[:20]), em[1][3], em[1][4]))
print("other launches (us):", " ".join("%.0f" % x for x in d[20:]), "| all 25: %.2f ms" % (sum(d) / 1e3))
P
