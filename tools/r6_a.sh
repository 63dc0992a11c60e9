#!/bin/bash
# round 6, call a: the ADVICE fixes under test + today's baseline of the tile kernel (A/B + phase split)
mkdir -p gpurun_out/r6a
python -m pytest tests/test_streamed_gpu.py -x -q -m gpu -k "near_ties or at_100_neighbours or at_80_neighbours or exact_when_budget" > gpurun_out/r6a/tests.log 2>&1
tail -3 gpurun_out/r6a/tests.log
python tools/st_ab.py 1000000 bf4 > gpurun_out/r6a/st_ab.log 2>&1
cat gpurun_out/r6a/st_ab.log
ANNCHOR_HIP_LIB=$PWD/variants/libprof.so python tools/st_prof_run.py > gpurun_out/r6a/prof.log 2>&1
tail -40 gpurun_out/r6a/prof.log
