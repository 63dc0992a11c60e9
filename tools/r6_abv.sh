#!/bin/bash
# round 6: tile-kernel A/B at C3 with a variant library: tools/r6_abv.sh <tag> <lib.so> <kernels...>
TAG=$1; LIB=$2; shift; shift
mkdir -p gpurun_out/$TAG
ANNCHOR_HIP_LIB=$PWD/$LIB timeout 400 python tools/st_ab.py 1000000 "$@" > gpurun_out/$TAG/st_ab.log 2>&1; echo "ab rc=$?"; cut -c1-420 gpurun_out/$TAG/st_ab.log
