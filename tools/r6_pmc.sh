#!/bin/bash
# round 6: PMC passes over the tile kernel selected by ANNCHOR_ST_KERNEL (tools/r6_pmc.sh <tag> <kernel>)
TAG=$1; export ANNCHOR_ST_KERNEL=$2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/p$i -o p -- env PYTHONPATH=$R python $R/tools/st_prof_run.py > $O/p$i.log 2>&1
done
python - "$O" <<'PY'
import csv, glob, sys, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if not (k.startswith("k_st_knnbf<128, 16, false") or k.startswith("k_st_knnh")): continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
        if "Start_Timestamp" in r: pass
out = {k: {c: v / n[k][c] for c, v in d.items()} | {"launches": max(n[k].values())} for k, d in agg.items()}
json.dump(out, open(sys.argv[1] + "/pmc.json", "w"), indent=1)
SW = 4040934 * 16.0   # slab-waves per launch (approx.)
for k, d in sorted(out.items()):
    print(k)
    for c, v in sorted(d.items()): print("   %-28s %.4g   per slab-wave %.1f" % (c, v, v / SW))
PY
rm -rf $O/p[0-9]
