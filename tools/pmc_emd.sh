#!/bin/bash
# PMC passes over the exact-OT kernel of one C4 fit: VALU / LDS instruction counts, VALU-busy cycles, waves.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_emd; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/p$i -o p -- env PYTHONPATH=$R python $R/tools/c4_one.py > $O/p$i.log 2>&1
done
python - "$O" <<'PY'
import csv, glob, sys, collections, json
tot = collections.defaultdict(float); launches = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if not k.startswith("k_emd"): continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); launches[r["Counter_Name"]] += 1
out = dict(tot); out["launches"] = max(launches.values()) if launches else 0
json.dump(out, open(sys.argv[1] + "/pmc_emd.json", "w"), indent=1)
print(out)
PY
rm -rf $O/p[0-9]
