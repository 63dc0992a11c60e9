#!/bin/bash
# round 6: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the streamed tile kernels at C3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r6traffic}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o pmc -- env PYTHONPATH=$R python $R/tools/st_prof_run.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o pmc -- env PYTHONPATH=$R python $R/tools/st_prof_run.py > /dev/null 2>&1
cd $R && python tools/pmc_summary.py $O/f $O/w $O/traffic.json && python - "$O/traffic.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if k.startswith("k_st_knn") or k.startswith("k_st_join"):
        print("%-40s launches %d  %.1f GB per launch" % (k[:40], v["launches"], v["hbm_bytes_per_launch_corrected"] / 1e9))
PY
rm -rf $O/f $O/w
