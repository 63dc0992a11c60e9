#!/bin/bash
# One GPU-box pass: tests, bench, rocprofv3 kernel stats, PMC traffic (bench workloads and the pair-list
# kernels at N = 16000), PMC activity of the Levenshtein launches.  Outputs under gpurun_out/refresh/.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/refresh; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
timeout 900 python bench.py --workload euclid --steps 5 --warmup 1 > $O/bench_euclid.json 2> $O/bench_euclid.err; tail -c 300 $O/bench_euclid.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-scale > $O/bench_under_rocprof.json 2> $O/rocprof.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-scale > /dev/null 2> $O/pmc_fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-scale > /dev/null 2> $O/pmc_write.err
PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/scale -o r -- python $R/tools/pairlist_scale.py 16000 > $O/pairlist_scale.log 2>&1
PYTHONPATH=$R timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_scale -o pmc -- python $R/tools/pairlist_scale.py 16000 > /dev/null 2> $O/pmc_fetch_scale.err
PYTHONPATH=$R timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_scale -o pmc -- python $R/tools/pairlist_scale.py 16000 > /dev/null 2> $O/pmc_write_scale.err
cd $R
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json
python tools/pmc_summary.py $O/pmc_fetch_scale $O/pmc_write_scale $O/pmc_traffic_scale.json
bash tools/pmc_lev2.sh > $O/pmc_lev2.log 2>&1; cp gpurun_out/pmc_lev2/pmc_lev2.json $O/pmc_lev.json 2>/dev/null
bash tools/pmc_st.sh > $O/pmc_st.log 2>&1; cp gpurun_out/pmc_st/pmc_st.json $O/pmc_st.json 2>/dev/null
bash tools/pmc_emd.sh > $O/pmc_emd.log 2>&1; cp gpurun_out/pmc_emd/pmc_emd.json $O/pmc_emd.json 2>/dev/null
# row-sharded build: per-rank stage times with the ranks' GPU work serialised (the scaling model's input), C3 and C5
timeout 600 python tools/serial_ranks.py --n 1000000 --worlds 1,2,4,8 --out $O/serial_ranks_c3.json > $O/serial_ranks_c3.log 2>&1
timeout 900 python tools/serial_ranks.py --n 8000000 --worlds 1,2,4,8 --out $O/serial_ranks_c5.json > $O/serial_ranks_c5.log 2>&1
python tools/scaling_model.py $O/serial_ranks_c3.json > $O/scaling_model_c3.json; python tools/scaling_model.py $O/serial_ranks_c5.json > $O/scaling_model_c5.json
python tools/scaling_model.py $O/serial_ranks_c5.json --md > $O/scaling_model_c5.md; python tools/scaling_model.py $O/serial_ranks_c3.json --md > $O/scaling_model_c3.md
# round 6: instruction counts per wave and 32-column slab of the tile kernels (two-stage default, round 5's for comparison)
( bash tools/r6_pmc.sh refresh/pmc_tile_h h; bash tools/r6_pmc.sh refresh/pmc_tile_bf4 bf4 ) > $O/pmc_tile.log 2>&1
python - "$O" <<'PY'
import json, sys, os
o = sys.argv[1]; out = {}
for k in ("h", "bf4"):
    p = os.path.join(o, "pmc_tile_" + k, "pmc.json")
    if os.path.exists(p): out["ANNCHOR_ST_KERNEL=" + k] = json.load(open(p))
json.dump(out, open(os.path.join(o, "pmc_tile_kernels.json"), "w"), indent=1)
PY
# rows of more than 128 dimensions (k-blocked kernel)
for d in 256 768; do timeout 300 python tools/dim_probe.py 1000000 $d 2>/dev/null | tail -1 > $O/dim_probe_$d.json; done
find $O -name "*kernel_stats.csv" | head; find $O -name "*counter_collection.csv" -size +30M -delete
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_fetch_scale $O/pmc_write_scale 2>/dev/null
find $O/stats $O/scale -name "*kernel_trace.csv" -delete
du -sh $O
