#!/bin/bash
# Builds a variant of libannchor_hip.so with extra compiler flags:
#   tools/build_variant.sh <out.so> <extra flags...>      (load it with ANNCHOR_HIP_LIB=<out.so>)
set -e
OUT=$1; shift
HERE=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$(mktemp -d)
for s in $(python3 -c "import sys; sys.path.insert(0, '$HERE'); from annchor_amd.build import SOURCES; print(' '.join(SOURCES))"); do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -ffp-contract=off "$@" -c $HERE/annchor_amd/csrc/$s.hip -o $OBJ/$s.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJ/*.o
rm -rf $OBJ
echo built $OUT
