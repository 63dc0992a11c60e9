#!/bin/bash
# round 6: instruction-cache counters of the tile kernels (tools/r6_icache.sh <tag> [lib.so])
TAG=$1; [ -n "${2:-}" ] && export ANNCHOR_HIP_LIB=$2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_INSTS_MFMA SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/p$i -o p -- env PYTHONPATH=$R python $R/tools/st_prof_run.py > $O/p$i.log 2>&1 || tail -3 $O/p$i.log
done
python - "$O" <<'PY'
import csv, glob, sys, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if not (k.startswith("k_st_knnbf<128, 16") or k.startswith("k_st_knnh") or k.startswith("k_st_join")): continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
out = {k: {c: v / n[k][c] for c, v in d.items()} | {"launches": max(n[k].values())} for k, d in agg.items()}
json.dump(out, open(sys.argv[1] + "/pmc.json", "w"), indent=1)
for k, d in sorted(out.items()):
    print(k)
    for c, v in sorted(d.items()): print("   %-28s %.4g" % (c, v))
PY
rm -rf $O/p[0-9]
