#!/bin/bash
# PMC passes over the thinned-pair-list kernels of the N = 100 000 Levenshtein fit (tools/lev100k_profile.py, one fit per pass):
# HBM bytes (FETCH_SIZE / WRITE_SIZE, KiB; FETCH doubled on gfx950 as the guide prescribes), LDS conflicts, L2 hit counts.
# Output: gpurun_out/pmc_ub/pmc_ub.json
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_ub; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  LEV100K_REPS=1 timeout 600 rocprofv3 --pmc $set --output-format csv -d $O/p$i -o p -- env PYTHONPATH=$R python $R/tools/lev100k_profile.py ${1:-100000} > $O/p$i.log 2>&1
done
python - "$O" <<'PY'
import csv, glob, sys, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if not (k.startswith("k_update_bounds") or k.startswith("k_features") or k.startswith("k_transpose") or k.startswith("k_keep") or k.startswith("k_emit")): continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
out = {k: dict(d) | {"launches": max(n[k].values())} for k, d in agg.items()}
for k, d in out.items():
    if "FETCH_SIZE" in d: d["hbm_read_GB"] = d["FETCH_SIZE"] * 2 * 1024 / 1e9
    if "WRITE_SIZE" in d: d["hbm_write_GB"] = d["WRITE_SIZE"] * 1024 / 1e9
json.dump(out, open(sys.argv[1] + "/pmc_ub.json", "w"), indent=1)
for k, d in sorted(out.items()):
    print(k, {c: ("%.4g" % v) for c, v in sorted(d.items())})
PY
rm -rf $O/p[0-9]
