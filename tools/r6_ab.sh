#!/bin/bash
# round 6: A/B of tile kernels at C3: tools/r6_ab.sh <tag> <kernels...>
TAG=$1; shift
mkdir -p gpurun_out/$TAG
timeout 120 python tools/st_ab.py 200000 ${@: -1} > gpurun_out/$TAG/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/$TAG/smoke.log | cut -c1-250
timeout 400 python tools/st_ab.py 1000000 "$@" > gpurun_out/$TAG/st_ab.log 2>&1; echo "ab rc=$?"; cut -c1-420 gpurun_out/$TAG/st_ab.log
