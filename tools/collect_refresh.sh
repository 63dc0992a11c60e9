#!/bin/bash
# copies one tools/gpu_refresh.sh pass (gpurun_out/refresh/) into profiles/ under the round's prefix:  bash tools/collect_refresh.sh r06
set -eu
P=${1:?prefix, e.g. r06}; O=gpurun_out/refresh; D=profiles
cp $O/stats/r_kernel_stats.csv $D/${P}_kernel_stats.csv
cp $O/scale/r_kernel_stats.csv $D/${P}_scale_kernel_stats.csv
for f in bench bench_euclid bench_under_rocprof pmc_traffic pmc_lev pmc_st pmc_emd pmc_tile_kernels dim_probe_256 dim_probe_768 \
         serial_ranks_c3 serial_ranks_c5 scaling_model_c3 scaling_model_c5; do
  [ -s $O/$f.json ] && cp $O/$f.json $D/${P}_$f.json
done
cp $O/pmc_traffic_scale.json $D/${P}_scale_pmc_traffic.json
cp $O/scaling_model_c3.md $D/${P}_scaling_model_c3.md; cp $O/scaling_model_c5.md $D/${P}_scaling_model_c5.md
cp $O/pairlist_scale.log $D/${P}_pairlist_scale.log
grep -E "passed|failed" $O/pytest_gpu.log > $D/${P}_pytest_gpu.txt
git rev-parse HEAD >> $D/${P}_pytest_gpu.txt
ls -la $D/${P}_*
