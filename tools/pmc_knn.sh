#!/bin/bash
# PMC passes over the streamed k-NN kernel (N = 262144, p_work 0.1): where do its cycles go?
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_knn; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { # name, counters
  timeout 300 rocprofv3 --pmc $2 --output-format csv -d $O/$1 -o p -- env PYTHONPATH=$R python $R/tools/stream_probe.py 262144 0.1 > $O/$1.log 2>&1
  python - "$O/$1" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_st_knn" in r["Kernel_Name"]:
            agg["knn"][r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k, v in agg["knn"].items(): print("  %-28s %.4g  (per launch, %d launches)" % (k, v / n[k], n[k]))
PY
}
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA"
run b "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVES"
run c "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES SQ_INSTS_WAVE32_LDS"
run d "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"
rm -rf $O/a $O/b $O/c $O/d
tail -2 $O/a.log
