#!/bin/bash
# round 6, call b: the three-slot tile kernel (ANNCHOR_ST_KERNEL=bf3) -- smoke, A/B, tests
mkdir -p gpurun_out/r6b
timeout 120 python tools/st_ab.py 200000 bf3 > gpurun_out/r6b/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r6b/smoke.log
timeout 300 python tools/st_ab.py 1000000 bf4 bf3 > gpurun_out/r6b/st_ab.log 2>&1; echo "ab rc=$?"; cat gpurun_out/r6b/st_ab.log
ANNCHOR_ST_KERNEL=bf3 timeout 900 python -m pytest tests/test_streamed_gpu.py -x -q -m gpu > gpurun_out/r6b/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r6b/tests.log
