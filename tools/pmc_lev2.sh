#!/bin/bash
# PMC passes over isolated Levenshtein launches (5 x 65536 pairs, 5 x one-to-all), per kernel variant.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_lev2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in ${LEV_VARIANTS:-1 9 auto}; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS" \
             "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
    i=$((i+1))
    if [ "$v" = auto ]; then   # no forced variant: the picker's rounds run as the persistent launch k_lev_ap
      timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/v${v}_$i -o p -- env PYTHONPATH=$R python $R/tools/lev_launch.py > $O/v${v}_$i.log 2>&1
    else
      ANNCHOR_LEV_R=$v timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/v${v}_$i -o p -- env PYTHONPATH=$R ANNCHOR_LEV_R=$v python $R/tools/lev_launch.py > $O/v${v}_$i.log 2>&1
    fi
  done
done
python - "$O" <<'PY'
import csv, glob, sys, collections, json, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    v = re.search(r"/v(\w+)_\d+/", f).group(1)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if not k.startswith("k_lev"): continue
        if v == "auto" and not k.startswith("k_lev_ap<false>"): continue   # (that pass is there for the persistent kernel only)
        g = int(float(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0))
        k = "R=%s %s %s" % (v, k, "[anchor-like]" if 0 < g <= 64 * 4096 else "[65536 pairs]")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
out = {k: {c: v / n[k][c] for c, v in d.items()} | {"launches": max(n[k].values())} for k, d in agg.items()}
json.dump(out, open(sys.argv[1] + "/pmc_lev2.json", "w"), indent=1)
for k, d in sorted(out.items()):
    cyc = d.get("GRBM_GUI_ACTIVE", 0)
    print(k, {c: ("%.4g" % v) for c, v in sorted(d.items())})
PY
rm -rf $O/v*_[0-9]
