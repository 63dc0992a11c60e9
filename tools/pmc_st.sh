#!/bin/bash
# PMC passes over the streamed form's tile kernel (one C3-size fit per pass, tools/st_prof_run.py): matrix-pipe
# busy cycles, VALU / LDS activity, waits.  Output: gpurun_out/pmc_st/pmc_st.json (per-launch averages of the
# k_st_knn / k_st_join launches).
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_st; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $O/p$i -o p -- env PYTHONPATH=$R python $R/tools/st_prof_run.py > $O/p$i.log 2>&1
done
python - "$O" <<'PY'
import csv, glob, sys, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if not (k.startswith("k_st_knn") or k.startswith("k_st_join<")): continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
out = {k: {c: v / n[k][c] for c, v in d.items()} | {"launches": max(n[k].values())} for k, d in agg.items()}
for k, d in out.items():
    if d.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs (one matrix pipe each)
        d["mfma_busy_fraction"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8 * 1024)
json.dump(out, open(sys.argv[1] + "/pmc_st.json", "w"), indent=1)
for k, d in sorted(out.items()):
    print(k, {c: ("%.4g" % v) for c, v in sorted(d.items())})
PY
rm -rf $O/p[0-9]
