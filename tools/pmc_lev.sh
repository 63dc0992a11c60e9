#!/bin/bash
# PMC pass over the C2 fit: VALU / LDS activity of the Levenshtein kernels.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_lev; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {
  timeout 300 rocprofv3 --pmc $2 --output-format csv -d $O/$1 -o p -- env PYTHONPATH=$R python $R/bench.py --steps 3 --warmup 1 --no-euclid --no-cpu-baseline --no-kernel-events --no-scale > $O/$1.log 2>&1
}
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU"
run b "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
python - "$O" <<'PY'
import csv, glob, sys, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if k.startswith("k_lev"):   # one-to-all anchor rounds (1600 pairs: a few hundred waves) vs pair lists
            g = int(float(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0))
            k += " [anchor rounds]" if 0 < g <= 64 * 4096 else " [pair lists]"
        if k.startswith("k_lev") or k.startswith("k_gn_sweep") or k.startswith("k_row_thresh") or k.startswith("k_features"):
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
out = {k: {c: v / n[k][c] for c, v in d.items()} | {"launches": max(n[k].values())} for k, d in agg.items()}
json.dump(out, open(sys.argv[1] + "/pmc_lev.json", "w"), indent=1)
for k, d in out.items(): print(k, {c: ("%.4g" % v) for c, v in d.items()})
PY
rm -rf $O/a $O/b
