#!/bin/bash
# One-source variant of the library: tools/variant1.sh <out.so> <source name (no .hip)> [source2 ...] -- <extra flags...>
# compiles the named sources with the extra flags, links them with the library's other objects (annchor_amd/csrc/_obj).
# Load with ANNCHOR_HIP_LIB=<out.so>.  variants/ is git-ignored and travels to the GPU box.
set -e
HERE=$(cd "$(dirname "$0")/.." && pwd)
OUT=$1; shift
SRCS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do SRCS+=("$1"); shift; done
[ "$1" == "--" ] && shift
TMP=$(mktemp -d)
OBJS=()
for o in $HERE/annchor_amd/csrc/_obj/*.o; do
  b=$(basename $o .o); skip=0
  for s in "${SRCS[@]}"; do [ "$s" == "$b" ] && skip=1; done
  [ $skip == 0 ] && OBJS+=("$o")
done
for s in "${SRCS[@]}"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -ffp-contract=off "$@" -c $HERE/annchor_amd/csrc/$s.hip -o $TMP/$s.o &
done
wait
for s in "${SRCS[@]}"; do OBJS+=("$TMP/$s.o"); done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT "${OBJS[@]}"
rm -rf $TMP
echo built $OUT
